#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per dispatch of the kernels whose name contains a substring.
usage: pmc_reduce.py <kernel substring> <counter_collection.csv> [more csv ...]  -> JSON on stdout"""
import csv, json, sys
from collections import defaultdict

kernel = sys.argv[1]
acc, cnt = defaultdict(float), defaultdict(int)
for path in sys.argv[2:]:
    disp = defaultdict(dict)
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            disp[r["Dispatch_Id"]][r["Counter_Name"]] = disp[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for d in disp.values():
        for k, v in d.items():
            acc[k] += v
            cnt[k] += 1
print(json.dumps({k: acc[k] / cnt[k] for k in sorted(acc)} | {"dispatches": max(cnt.values()) if cnt else 0}, indent=1))
