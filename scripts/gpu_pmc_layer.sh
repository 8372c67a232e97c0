#!/bin/bash
# Dev helper (GPU): SQ / TCC counter passes over one conv layer shape.  usage: gpu_pmc_layer.sh <layer> <hint> <tag>
layer=$1; hint=$2; tag=$3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
    n=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${tag}_$n -o p -- python3 $R/scripts/gpu_profile_layer.py $layer $hint 6 > $R/gpurun_out/pmc_${tag}_$n.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${tag}_$n.log; }
}
cd $R
( cd /tmp; run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM )
( cd /tmp; run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT )
( cd /tmp; run c SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_BRANCH )
( cd /tmp; run d TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE )
( cd /tmp; run e FETCH_SIZE )
( cd /tmp; run f WRITE_SIZE )
find gpurun_out -name "*counter_collection.csv" -path "*pmc_${tag}_*" > /tmp/csvs.txt
python3 scripts/pmc_reduce.py conv_p32 $(cat /tmp/csvs.txt) | tee gpurun_out/pmc_${tag}.json
