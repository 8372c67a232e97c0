#!/bin/bash
# GPU box: does a second, independent pipeline on the same GPU add throughput?  single, two concurrent, single
cd $GRAFT_REPO_ROOT
one() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-h2d-leg --steps 12 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
echo "single: $(one)"
one > gpurun_out/two_a.txt & pa=$!
one > gpurun_out/two_b.txt & pb=$!
wait $pa $pb
echo "two concurrent: $(cat gpurun_out/two_a.txt) + $(cat gpurun_out/two_b.txt)"
echo "single: $(one)"
