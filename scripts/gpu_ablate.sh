#!/bin/bash
# Dev helper (GPU): time the key conv layers with the shipped library and with every timing-only variant under build/ab/
hint=${1:-1}
python scripts/gpu_conv_p32_check.py key $hint
for lib in build/ab/*.so; do
    AB_LIB=$lib python scripts/gpu_conv_p32_check.py key $hint
done
python scripts/gpu_conv_p32_check.py key $hint
