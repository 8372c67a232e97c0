#!/bin/bash
# GPU box: rocprofv3 kernel trace of a short bench run -> gpurun_out/kb/<tag>_kernel_stats.csv (+ top of the table on stdout)
# usage: gpu_kernel_breakdown.sh <tag> [bench flags...]
tag=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/kb
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o t -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-h2d-leg "$@" > $O/$tag.json 2> $O/$tag.err || { tail -5 $O/$tag.err; exit 1; }
f=$(find $O/$tag -name "*kernel_stats.csv" | head -1)
cp $f $O/${tag}_kernel_stats.csv
find $O/$tag -type f -delete
python3 - $O/${tag}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:45]:
    print(f"{r['Name'][:90]:90} {r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):5.2f}%")
PY
