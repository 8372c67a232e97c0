#!/bin/bash
# GPU box: the rocprofv3 evidence behind bench.py's roofline object (profiles/README.md says what each file is).
#   1. --kernel-trace --stats of the default bench command            -> gpurun_out/r02/stats/
#   2. separate --pmc passes of the predictor-only eager bench        -> gpurun_out/r02/pmc_{fetch,write,sq,sq2,tcc}/
# Reduced on the build host by scripts/pmc_conv_summary.py into profiles/r02_*.  usage: gpu_collect_profiles.sh [tag]
tag=${1:-r05}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
set -o pipefail
CACHE=/tmp/deepemia_tiles_cache
echo "[collect] tile cache (unprofiled)"
timeout -k 10 300 python3 $R/bench.py --forward-only --eager --steps 1 --warmup 0 --no-cpu-baseline --no-h2d-leg --no-cli-leg --lanes 1 --tiles-cache $CACHE > $O/tiles_cache.json 2> $O/tiles_cache.err || { tail -5 $O/tiles_cache.err; exit 1; }
if [ -z "$PMC_ONLY" ]; then
echo "[collect] kernel trace of the default bench command (tiles from the cache: every kernel in the trace belongs to the path)"
# (two lanes = the rank and its lane child, two processes: no -o, rocprofv3 then writes <hostname>/<pid>_kernel_stats.csv per process)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-h2d-leg --no-cli-leg --tiles-cache $CACHE > $O/bench_under_rocprofv3.json 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
tail -c 600 $O/bench_under_rocprofv3.json; echo
# ... and with ONE lane: the conv launches of one pipeline alone, the figure roofline.avg_launch_us (HIP events around every launch of
# two single-lane eager steps) has to agree with; with two lanes in flight every launch shares the CUs with another grid and lasts longer
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_l1 -o bench -- python3 $R/bench.py --lanes 1 --steps 6 --warmup 2 --no-cpu-baseline --no-h2d-leg --no-cli-leg --tiles-cache $CACHE > $O/bench_l1_under_rocprofv3.json 2> $O/stats_l1.err || { tail -5 $O/stats_l1.err; exit 1; }
fi
# The counter passes run the bench's REAL batches (numpy tiles + device-generated ones, two distinct batches): the generated
# tiles are made once by an UNPROFILED process and cached as .npy under /tmp, so the profiled processes start from an H2D
# copy -- no torch generator kernel runs under --pmc (a round-3 pass died inside one; it was then dodged with --repeat-tiles).
pmc() { # name, counters...
    n=$1; shift
    echo "[collect] pmc pass $n: $*"
    timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$n -o p -- python3 $R/bench.py --forward-only --eager --steps 2 --warmup 1 --no-cpu-baseline --no-h2d-leg --no-cli-leg --lanes 1 --tiles-cache $CACHE > $O/pmc_$n.json 2> $O/pmc_$n.err || { tail -20 $O/pmc_$n.err; return 1; }
}
pmc fetch FETCH_SIZE && pmc write WRITE_SIZE && pmc sq SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    && pmc tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE \
    && pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
rc=$?
echo "[collect] per-layer conv timings"
[ -z "$PMC_ONLY" ] && ( cd $R && timeout -k 10 200 python3 scripts/gpu_conv_layers.py f16x2 48 $O/conv_layers.csv > $O/conv_layers.err 2>&1 ) || tail -3 $O/conv_layers.err
# keep what travels back small: the per-dispatch counter tables and the stats tables only
find $O -name "*.csv" -size +30M -delete
find $O -type f ! -name "*.csv" ! -name "*.json" ! -name "*.err" -delete
du -sh $O
exit $rc
