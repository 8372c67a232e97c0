#!/bin/bash
# Dev helper (GPU): time the key conv layers with the shipped library and with every timing-only variant under build/ab/
# usage: gpu_ablate.sh "<hints>" <mfma|hbm>
hints=${1:-1}; which=${2:-mfma}
for h in $hints; do python scripts/gpu_conv_p32_check.py key $h $which; done
for lib in build/ab/*.so; do
    for h in $hints; do AB_LIB=$lib python scripts/gpu_conv_p32_check.py key $h $which; done
done
