#!/bin/bash
# GPU box: the bench lines committed under profiles/ (tag = $1)
tag=${1:-r03}
cd $GRAFT_REPO_ROOT
O=gpurun_out/final_$tag
mkdir -p $O
t0=$(date +%s)
timeout -k 10 600 python bench.py > $O/bench_f16x2_b48_whole_path.json 2> $O/default.err || { tail -5 $O/default.err; exit 1; }
echo "default bench: $(( $(date +%s) - t0 )) s"; tail -c 400 $O/bench_f16x2_b48_whole_path.json; echo
timeout -k 10 300 python bench.py --forward-only --no-cpu-baseline > $O/bench_f16x2_b48_predictor_only.json 2> $O/fwd.err || { tail -5 $O/fwd.err; exit 1; }
timeout -k 10 600 python bench.py --total-tiles 256 --batch 32 --warmup 1 --parity-only > $O/bench_f16x2_b32_total256_tiles.json 2> $O/t256.err || { tail -5 $O/t256.err; exit 1; }
timeout -k 10 300 python bench.py --precision f16 --no-cpu-baseline --no-h2d-leg > $O/bench_f16_single_plane_b48_whole_path.json 2> $O/f16.err || { tail -5 $O/f16.err; exit 1; }
for f in $O/*.json; do python - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('achieved'), (d.get('parity') or {}).get('ok'))
PY
done
