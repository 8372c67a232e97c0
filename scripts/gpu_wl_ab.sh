cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity_maskops.py -x -q -m gpu 2>&1 | tail -3 &&
for r in 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-h2d-leg --steps 12 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('worklist', d['value'], d['ms_per_step'], d['config'].get('post_after_forward_ms_per_step'))" &&
DEEPEMIA_MASK_WORKLISTS=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-h2d-leg --steps 12 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('all-masks', d['value'], d['ms_per_step'], d['config'].get('post_after_forward_ms_per_step'))"
done
