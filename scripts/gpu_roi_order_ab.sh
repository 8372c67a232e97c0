#!/bin/bash
# GPU box: ROIAlign launch order A/B (DEEPEMIA_ROI_ORDER=0 / 1) -- one-lane kernel traces of the bench (kernel times per step)
# and one TCC counter pass each (L2 hits / misses of roi_align_kernel).  usage: gpu_roi_order_ab.sh [tag]
tag=${1:-r05_roi}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CACHE=/tmp/deepemia_tiles_cache
timeout -k 10 300 python3 $R/bench.py --forward-only --eager --steps 1 --warmup 0 --no-cpu-baseline --no-h2d-leg --no-cli-leg --lanes 1 --tiles-cache $CACHE > $O/tiles_cache.json 2> $O/tiles_cache.err || { tail -5 $O/tiles_cache.err; exit 1; }
for ord in 0 1; do
    export DEEPEMIA_ROI_ORDER=$ord
    echo "[roi A/B] order=$ord kernel trace"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$ord -o bench -- python3 $R/bench.py --lanes 1 --steps 6 --warmup 2 --no-cpu-baseline --no-h2d-leg --no-cli-leg --tiles-cache $CACHE > $O/bench_$ord.json 2> $O/stats_$ord.err || { tail -5 $O/stats_$ord.err; exit 1; }
    echo "[roi A/B] order=$ord TCC pass"
    timeout -k 10 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc_$ord -o p -- python3 $R/bench.py --forward-only --eager --steps 2 --warmup 1 --no-cpu-baseline --no-h2d-leg --no-cli-leg --lanes 1 --tiles-cache $CACHE > $O/tcc_$ord.json 2> $O/tcc_$ord.err || { tail -20 $O/tcc_$ord.err; exit 1; }
done
find $O -name "*.csv" -size +30M -delete
find $O -type f ! -name "*.csv" ! -name "*.json" ! -name "*.err" -delete
du -sh $O
