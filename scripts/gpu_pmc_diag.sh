#!/bin/bash
# Dev helper (GPU): memory-path counters (TLB, L1<->L2 latency, L2<->fabric stalls) of one conv layer, at most four
# counters of a block per pass (more is refused by the hardware and rocprofv3 then hangs).  usage: gpu_pmc_diag.sh <layer> <hint> <tag>
layer=$1; hint=$2; tag=$3
R=$GRAFT_REPO_ROOT
run() { # name, counters...
    n=$1; shift
    ( cd /tmp; export TMPDIR=/tmp; timeout -k 5 90 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/diag_${tag}_$n -o p -- python3 $R/scripts/gpu_profile_layer.py $layer $hint 6 > $R/gpurun_out/diag_${tag}_$n.log 2>&1 || { grep -m2 "rror" $R/gpurun_out/diag_${tag}_$n.log; exit 1; } ) || exit 1
}
run a TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_GUI_ACTIVE
run b TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
run c TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum
run d TCC_TOO_MANY_EA_WRREQS_STALL_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
python3 $R/scripts/pmc_reduce.py conv_p32 $(find $R/gpurun_out -name "*counter_collection.csv" -path "*diag_${tag}_*") | tee $R/gpurun_out/diag_${tag}.json
