#!/usr/bin/env python3
"""Reduce the two rocprofv3 PMC passes of the bench (FETCH_SIZE, WRITE_SIZE) to HBM bytes per launch of the dominant
conv kernel -> profiles/r01_conv_<precision>_b<batch>_pmc_traffic.json (read by bench.py for roofline.traffic).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o f --output-format csv -- python3 $R/bench.py --forward-only --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o w --output-format csv -- python3 $R/bench.py --forward-only --steps 2 --warmup 1 --no-cpu-baseline
    python scripts/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv conv_igemm_split_kernel f32x3 16

Units and corrections as MI355X_MICROARCH.md (HBM / rocprofv3) prescribes: both counters are reported in KiB; on
gfx950 FETCH_SIZE under-reports 16-byte-per-lane coalesced reads by 2x (counted x2 here), WRITE_SIZE is exact."""
import csv, json, sys
from pathlib import Path

fetch_csv, write_csv, kernel, prec, batch = sys.argv[1:6]


def total(path, counter):
    s, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            s += float(r["Counter_Value"])
            n += 1
    return s, n


f, nf = total(fetch_csv, "FETCH_SIZE")
w, nw = total(write_csv, "WRITE_SIZE")
assert nf == nw and nf > 0, (nf, nw)
out = {"kernel": kernel, "launches": nf, "FETCH_SIZE_KiB_sum": f, "WRITE_SIZE_KiB_sum": w,
       "command": f"rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --forward-only --steps 2 --warmup 1 "
                  f"--no-cpu-baseline ({prec}, B={batch}, R101, 2048^2)",
       "correction": "gfx950: FETCH_SIZE x2 for 16-B/lane coalesced reads (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; KiB -> bytes x1024",
       "fetch_bytes_per_launch": 2 * f * 1024 / nf, "write_bytes_per_launch": w * 1024 / nw,
       "hbm_bytes_per_launch": (2 * f + w) * 1024 / nf}
dst = Path(__file__).resolve().parent.parent / "profiles" / f"r01_conv_{prec}_b{batch}_pmc_traffic.json"
dst.write_text(json.dumps(out, indent=1))
print(dst, json.dumps(out))
