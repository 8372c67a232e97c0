#!/bin/bash
for us in 0 6 10 15 22 30; do
  for h in 1 7 9; do
    echo -n "stagger ${us}us: "; DEMIA_P32_STAGGER_US=$us python scripts/gpu_conv_p32_check.py key $h hbm 2>&1 | grep -v amdgpu
  done
done
