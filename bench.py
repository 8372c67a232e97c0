#!/usr/bin/env python3
"""Headline benchmark: EM tiles/s (2048x2048, R101-FPN) on N MI355X.

One "step" = one pass of the hot path (resize -> R101-FPN -> RPN -> ROI heads -> mask paste
to bit-packed full-resolution masks) over one batch of synthetic 2048^2 tiles that is already
resident in HBM when the timed region starts.  One process per GPU; tiles are independent
units sharded over ranks (weak scaling).  See DESIGN.md "Measurement".

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np
import torch

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}  # dense MFMA peaks, MI355X_MICROARCH.md


def cpu_baseline(depth: int, size: int, thr: float, sd) -> dict:
    """Time the CPU oracle (a port of the reference's Detectron2 CPU path) on one tile."""
    from deepemia_amd import synth
    from oracle import maskrcnn_ref

    img = synth.em_tile(0, size)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, ncpu)))  # the GPU box gives one GPU a 16-core share
    t0 = time.perf_counter()
    out = maskrcnn_ref.predict(img, sd, depth, thr)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "tiles/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": f"1 synthetic {size}x{size} tile, R{depth}-FPN fp32 torch-CPU restatement of "
                      f"DefaultPredictor (oracle/maskrcnn_ref.py), {int(out['scores'].shape[0])} instances, {dt:.1f} s"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="tiles per GPU per step")
    ap.add_argument("--precision", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--depth", type=int, default=101)
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--threshold", type=float, default=0.3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = f"cuda:{local_rank if world > 1 else 0}"

    from deepemia_amd import synth
    from deepemia_amd.engine import MaskRCNNEngine

    sd = synth.random_d2_state_dict(args.depth, 2, seed=0)
    eng = MaskRCNNEngine(sd, args.depth, 2, args.threshold, dev, args.precision)
    tiles = np.stack([synth.em_tile(rank * args.batch + i, args.size) for i in range(args.batch)])
    x = torch.from_numpy(tiles).to(dev)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    det_total = 0
    for _ in range(args.warmup):
        raw = eng.forward(x)
    sync_all()
    eng.conv_events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        raw = eng.forward(x)
    sync_all()
    dt = time.perf_counter() - t0
    events, eng.conv_events = eng.conv_events, None
    det_total = int(raw.count.sum().item())
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        conv_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in events)
        conv_flops = sum(f for _, _, f in events)
        launches = len(events)
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        peak = PEAK_TFLOPS[args.precision]
        line = {
            "metric": "EM tiles/s (2048x2048, R101-FPN)", "value": world * args.batch * args.steps / dt,
            "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"configs[1]: R{args.depth}-FPN, {args.size}x{args.size} synthetic EM tiles, "
                                   f"{args.batch} tiles per GPU per step, resize 800 -> backbone/FPN/RPN/ROI heads -> "
                                   f"mask paste to bit-packed {args.size}^2 masks; random-init Detectron2-layout weights, "
                                   f"K=2, threshold {args.threshold}",
                       "tiles_per_step_per_gpu": args.batch, "detections_last_step_rank0": det_total},
            "roofline": {"bound": "mfma", "kernel": "conv_igemm_kernel (implicit-GEMM conv, all tile configs)",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "launches_per_step": launches // max(args.steps, 1),
                         "avg_launch_us": conv_ms * 1e3 / max(launches, 1),
                         "algorithmic_gflop_per_launch": conv_flops / max(launches, 1) / 1e9,
                         "share_of_step_time": conv_ms * 1e-3 / dt, "traffic": None},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.depth, args.size, args.threshold, sd)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
