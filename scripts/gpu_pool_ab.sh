cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_headline_parity.py -x -q -m gpu 2>&1 | tail -5 &&
for r in 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-h2d-leg --steps 12 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pool', d['value'], d['ms_per_step'])" &&
timeout -k 10 200 python bench.py --no-cpu-baseline --no-h2d-leg --steps 12 --warmup 3 --no-plane-pools 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fresh', d['value'], d['ms_per_step'])"
done
