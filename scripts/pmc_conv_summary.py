#!/usr/bin/env python3
"""Reduce what scripts/gpu_collect_profiles.sh brought back (gpurun_out/<tag>/) into the committed evidence under profiles/:

    profiles/<tag>_bench_<prec>_b<B>_whole_path_kernel_stats.csv       rocprofv3 --kernel-trace --stats table of the default bench command
    profiles/<tag>_bench_<prec>_b<B>_whole_path_under_rocprofv3.json   the bench line printed by that profiled run
    profiles/<tag>_conv_<prec>_b<B>_pmc_traffic.json                   HBM bytes per conv launch (read by bench.py -> roofline.traffic)
    profiles/<tag>_conv_<prec>_sq.json                                 MFMA busy fraction etc. per conv launch (bench.py -> roofline.mfma_busy_frac)
    profiles/<tag>_conv_<prec>_by_kernel_pmc.csv                       the same counters per kernel instantiation / grid

Units and corrections follow MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE are KiB; gfx950 FETCH_SIZE
under-reports 16-byte-per-lane reads by 2x (doubled here; the LDS-DMA loads of conv_p32 are 16 bytes per lane); GRBM_GUI_ACTIVE
is summed over the 8 XCDs, so cycles = GRBM_GUI_ACTIVE / 8; SQ counters are summed over all SIMDs (256 CUs x 4), so the busy
fraction of the matrix pipes = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024).

usage: pmc_conv_summary.py <tag> <kernel substring> <precision> <batch> <git head>"""
import csv
import json
import shutil
import sys
from collections import defaultdict
from pathlib import Path

tag, kernel, prec, batch, head = sys.argv[1:6]
ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "gpurun_out" / tag
DST = ROOT / "profiles"


def table(name):
    """pass directory -> {dispatch id: {"kernel": name, "grid": g, counters...}} for the conv launches"""
    out = {}
    for path in (SRC / f"pmc_{name}").rglob("*counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            if kernel not in r["Kernel_Name"]:
                continue
            d = out.setdefault(r["Dispatch_Id"], {"kernel": r["Kernel_Name"], "grid": r.get("Grid_Size", ""), "lds": r.get("LDS_Block_Size", "")})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return out


def short(name):
    i = name.find("<")
    return name[:i].split("::")[-1] + (name[i:name.find(">") + 1] if i >= 0 else "")


fetch, write, sq, tcc = table("fetch"), table("write"), table("sq"), table("tcc")
assert len(fetch) == len(write) == len(sq) > 0, (len(fetch), len(write), len(sq))
n = len(fetch)
F = sum(d["FETCH_SIZE"] for d in fetch.values())
W = sum(d["WRITE_SIZE"] for d in write.values())
cmd = "python3 bench.py --forward-only --eager --steps 2 --warmup 1 --no-cpu-baseline --no-h2d-leg --tiles-cache /tmp/deepemia_tiles_cache (the bench's own two distinct 48-tile batches; generated tiles cached by an unprofiled run)"
traffic = {"kernel": kernel, "head": head, "launches": n, "FETCH_SIZE_KiB_sum": F, "WRITE_SIZE_KiB_sum": W,
           "command": f"rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- {cmd}  ({prec}, B={batch}, R101, 2048^2)",
           "correction": "gfx950: FETCH_SIZE x2 for 16-B/lane reads (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; KiB -> bytes x1024",
           "fetch_bytes_per_launch": 2 * F * 1024 / n, "write_bytes_per_launch": W * 1024 / n, "hbm_bytes_per_launch": (2 * F + W) * 1024 / n}
(DST / f"{tag}_conv_{prec}_b{batch}_pmc_traffic.json").write_text(json.dumps(traffic, indent=1) + "\n")

cyc = sum(d["GRBM_GUI_ACTIVE"] for d in sq.values()) / 8.0
busy = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"] for d in sq.values())
tot = {k: sum(d.get(k, 0.0) for d in sq.values()) for k in ("SQ_INSTS_MFMA", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY")}
rec = {"kernel": kernel, "head": head, "launches": n,
       "command": f"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- {cmd}",
       "normalisation": "cycles = GRBM_GUI_ACTIVE / 8 (one XCD's clock); SQ counters are sums over 1024 SIMDs; time-weighted over all conv launches",
       "gpu_cycles_per_launch": cyc / n, "mfma_busy_frac": busy / (cyc * 1024.0), "mfma_insts_per_launch": tot["SQ_INSTS_MFMA"] / n,
       "wave_wait_frac": tot["SQ_WAIT_ANY"] / max(tot["SQ_WAVE_CYCLES"], 1.0), "wave_wait_inst_frac": tot["SQ_WAIT_INST_ANY"] / max(tot["SQ_WAVE_CYCLES"], 1.0),
       "cu_busy_frac": tot["SQ_BUSY_CU_CYCLES"] / (cyc * 256.0 * 4.0) if tot["SQ_BUSY_CU_CYCLES"] else None}
if tcc:
    h, m = sum(d.get("TCC_HIT_sum", 0.0) for d in tcc.values()), sum(d.get("TCC_MISS_sum", 0.0) for d in tcc.values())
    rec["l2_hit_frac"] = h / max(h + m, 1.0)
lds = table("lds")
if lds:
    t = {k: sum(d.get(k, 0.0) for d in lds.values()) for k in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS",
                                                              "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")}
    rec["lds_bank_conflict_frac"] = t["SQ_LDS_BANK_CONFLICT"] / max(t["SQ_LDS_IDX_ACTIVE"], 1.0)
    rec["lds_idx_active_frac"] = t["SQ_LDS_IDX_ACTIVE"] / (cyc * 256.0)        # LDS pipe busy cycles per CU and GPU cycle
    rec["insts_per_launch"] = {k[9:].lower(): v / len(lds) for k, v in t.items() if k.startswith("SQ_INSTS_")}
    rec["command_lds"] = "--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR (same command)"
(DST / f"{tag}_conv_{prec}_sq.json").write_text(json.dumps(rec, indent=1) + "\n")
if (SRC / "conv_layers.csv").exists():
    shutil.copy(SRC / "conv_layers.csv", DST / f"{tag}_conv_layers_{prec}.csv")

# per kernel instantiation and grid: dispatch ids are the same sequence in every pass (same command, same order of launches)
groups = defaultdict(lambda: defaultdict(float))
for did, d in sq.items():
    g = groups[(short(d["kernel"]), d["grid"])]
    g["n"] += 1
    g["cyc"] += d["GRBM_GUI_ACTIVE"] / 8.0
    g["busy"] += d["SQ_VALU_MFMA_BUSY_CYCLES"]
    g["mfma"] += d.get("SQ_INSTS_MFMA", 0.0)
    g["wait"] += d.get("SQ_WAIT_ANY", 0.0)
    g["wave"] += d.get("SQ_WAVE_CYCLES", 0.0)
    if did in fetch and did in write:
        g["bytes"] += (2 * fetch[did]["FETCH_SIZE"] + write[did]["WRITE_SIZE"]) * 1024
with open(DST / f"{tag}_conv_{prec}_by_kernel_pmc.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid_threads", "launches", "cycles_per_launch", "share_of_conv_cycles", "mfma_busy_frac", "mfma_insts_per_launch",
                "wave_wait_frac", "hbm_bytes_per_launch", "hbm_bytes_per_cycle"])
    for (k, grid), g in sorted(groups.items(), key=lambda kv: -kv[1]["cyc"]):
        w.writerow([k, grid, int(g["n"]), round(g["cyc"] / g["n"]), round(g["cyc"] / cyc, 4), round(g["busy"] / (g["cyc"] * 1024.0), 4),
                    round(g["mfma"] / g["n"]), round(g["wait"] / max(g["wave"], 1.0), 4), round(g["bytes"] / g["n"]), round(g["bytes"] / g["cyc"], 1)])

# the default command: one table per PROCESS on the GPU (round 5: lanes 1.. are child processes of the rank) -- the table with the
# most conv launches is the rank's (it also runs the warm-up solo pass and the side legs)
tables = sorted((SRC / "stats").rglob("*kernel_stats.csv"),
                key=lambda p_: -sum(int(r["Calls"]) for r in csv.DictReader(open(p_)) if kernel in r["Name"]))
for k, path in enumerate(tables):
    shutil.copy(path, DST / f"{tag}_bench_{prec}_b{batch}_whole_path_kernel_stats{'' if k == 0 else f'_lane{k}'}.csv")
    rows = [r for r in csv.DictReader(open(path)) if kernel in r["Name"]]
    calls, tot = sum(int(r["Calls"]) for r in rows), sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"default command, process {k}: {calls} {kernel} launches, {tot / max(calls, 1) / 1e3:.1f} us per launch")
line = [l for l in (SRC / "bench_under_rocprofv3.json").read_text().splitlines() if l.startswith("{")]
if line:
    (DST / f"{tag}_bench_{prec}_b{batch}_whole_path_under_rocprofv3.json").write_text(line[-1] + "\n")
# the same command with ONE lane (the figure roofline.avg_launch_us has to agree with)
for path in (SRC / "stats_l1").rglob("*kernel_stats.csv"):
    shutil.copy(path, DST / f"{tag}_bench_{prec}_b{batch}_one_lane_kernel_stats.csv")
    rows = [r for r in csv.DictReader(open(path)) if kernel in r["Name"]]
    calls = sum(int(r["Calls"]) for r in rows)
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"one lane: {calls} {kernel} launches, {tot / 1e6:.2f} ms, {tot / max(calls, 1) / 1e3:.1f} us per launch")
p1 = SRC / "bench_l1_under_rocprofv3.json"
if p1.exists():
    line = [l for l in p1.read_text().splitlines() if l.startswith("{")]
    if line:
        (DST / f"{tag}_bench_{prec}_b{batch}_one_lane_under_rocprofv3.json").write_text(line[-1] + "\n")
        print("HIP events of that run:", json.loads(line[-1])["roofline"]["avg_launch_us"], "us per launch")
print(json.dumps(traffic, indent=1))
print(json.dumps(rec, indent=1))
