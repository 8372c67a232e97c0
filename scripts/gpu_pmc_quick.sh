#!/bin/bash
# Dev helper (GPU): one SQ pass (wave cycles / waits / busy) for a layer at several tile hints.  usage: gpu_pmc_quick.sh <layer> "<hints>"
layer=$1; hints=$2
R=$GRAFT_REPO_ROOT
for h in $hints; do
    ( cd /tmp; export TMPDIR=/tmp; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcq_${layer}_$h -o p -- python3 $R/scripts/gpu_profile_layer.py $layer $h 6 > $R/gpurun_out/pmcq_${layer}_$h.log 2>&1 )
    echo "== $layer hint $h"; python3 $R/scripts/pmc_reduce.py conv_p32 $(find $R/gpurun_out/pmcq_${layer}_$h -name "*counter_collection.csv")
done
