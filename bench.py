#!/usr/bin/env python3
"""Headline benchmark: EM tiles/s (2048x2048, R101-FPN) on N MI355X.

One "step" = one pass of the hot path (resize -> R101-FPN -> RPN -> ROI heads -> mask paste
to bit-packed full-resolution masks) over one batch of synthetic 2048^2 tiles that is already
resident in HBM when the timed region starts.  One process per GPU; tiles are independent
units sharded over ranks (weak scaling).  See DESIGN.md "Measurement".

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N --steps K --warmup W        # starts the N ranks itself (a torch.distributed.run child)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Process roles (decided before anything touches the GPU):
* launcher   -- ``--gpus N > 1`` without WORLD_SIZE: starts the N ranks as a CHILD ``torch.distributed.run``, passes rank 0's JSON
                line and the exit code through, never initialises HIP itself; WORLD_SIZE != --gpus is refused (exit 2)
* rank       -- one per GPU: lane 0 of that GPU, all the side legs (roofline instrumentation, parity, CPU sample), the JSON line
* lane child -- ``--lanes L`` (default 2) in the default ``--lane-mode processes``: L - 1 further PROCESSES per GPU, each one software
                pipeline with its own interpreter, HIP context, engine and hipGraphs, started by the rank before it initialises
                the GPU and driven over a pipe (RUN <leg> ... / DONE ...): no interpreter lock shared between the lanes
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np
import torch

# dense MFMA peaks, MI355X_MICROARCH.md.  f16x2 (default) computes the f32 product with three fp16 MFMAs per tile (two
# scaled fp16 planes per operand): its roof in f32-equivalent FLOP/s is the fp16 peak / 3; f32x3 = six bf16 MFMAs per
# tile (three bf16 planes): bf16 peak / 6.
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f32x3": 2500.0 / 6, "bf16x2": 2500.0 / 3, "f16x2": 2500.0 / 3, "f16x2r": 2500.0 / 3,
               "f16": 2500.0}        # flagged single-plane mode: one fp16 MFMA per product, the native fp16 peak


CLASS_THRESHOLDS = {0: (0.3, 0.7), 1: (0.3, 0.5)}   # class -> (confidence, IoU) as in config.yaml class_0 / class_1
SMALL_CLASSES = {1}
NUMPY_TILES = 16


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(depth: int, size: int, thr: float, sd, tiles: int = 3):  # noqa: C901
    """Time the CPU oracle (a port of the reference's CPU path: Detectron2 predictor restatement + the dense numpy / scipy
    post-processing and measurements, ``oracle/tile_parity.py``) on a bounded sample of the workload: ``tiles`` tiles with
    all host threads of this GPU's share, plus the predictor alone with one thread (the post-processing is numpy / scipy
    and does not scale with torch threads).  Returns (the ``cpu_baseline`` object, the reference results of all sampled tiles)."""
    from deepemia_amd import synth
    from oracle import tile_parity as TP

    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    nthr = max(1, min(16, ncpu))            # the GPU box gives one GPU a 16-core share
    torch.set_num_threads(nthr)
    refs = []
    for i in range(tiles):
        refs.append(TP.reference_tile(synth.em_tile(i, size), sd, depth, thr, CLASS_THRESHOLDS, SMALL_CLASSES))
        # (a progress line per tile: a long sample -- BASELINE.md's 20 tiles take nine minutes -- must not look like a hung run)
        print(f"[cpu_baseline] tile {i + 1}/{tiles}: {refs[-1]['seconds']['total']:.1f} s", file=sys.stderr, flush=True)
    secs = {k: float(np.median([r["seconds"][k] for r in refs])) for k in refs[0]["seconds"]}
    tot = sorted(r["seconds"]["total"] for r in refs)
    torch.set_num_threads(1)
    from oracle import maskrcnn_ref
    t0 = time.perf_counter()
    maskrcnn_ref.predict(synth.em_tile(0, size), sd, depth, thr)
    pred1 = time.perf_counter() - t0
    torch.set_num_threads(nthr)
    total1 = pred1 + secs["total"] - secs["predictor"]
    base = {"value": 1.0 / secs["total"], "unit": "tiles/s", "cores": nthr, "kind": "port",
            "cpu_model": cpu_model(), "host_cores_visible": ncpu,
            "tiles_timed": tiles, "seconds_per_tile_median": secs["total"], "seconds_per_tile_min_max": [tot[0], tot[-1]],
            "seconds_per_tile_p10_p90": [float(np.percentile(tot, 10)), float(np.percentile(tot, 90))],
            "stage_seconds_median": {k: v for k, v in secs.items() if k != "total"},
            "single_thread": {"value": 1.0 / total1, "cores": 1, "predictor_seconds": pred1,
                              "note": "predictor timed with torch.set_num_threads(1) on tile 0; the numpy / scipy stages are "
                                      "single-threaded in both settings and taken from the median above"},
            "instances_per_tile": [len(r["masks"]) for r in refs],
            "csv_rows_per_tile": [sum(len(x) for x in r["rows"]) for r in refs],
            "sample": f"{tiles} synthetic {size}x{size} tiles (indices 0..{tiles - 1}) through the whole per-tile path: R{depth}-FPN fp32 "
                      f"torch-CPU restatement of DefaultPredictor (oracle/maskrcnn_ref.py) + class loop, mask morphology, dedup and "
                      f"contour measurements on dense masks (oracle/postproc_ref.py); a restatement, not Detectron2 itself -- the "
                      f"reference's own prose claim is 30-120 s per image on CPU (docs/gpu-check.md:250)"}
    return base, refs


def resolve_lanes(lanes: int, world: int, overlap: bool, graph: bool) -> int:
    """Pipelines in flight per GPU (1..4; one without overlap / hipGraph replay).  With WORLD_SIZE > 1 the lanes' host threads
    must NOT issue their steps' all-gathers themselves -- nothing orders them alike on every rank, and two ranks x two lanes hung
    in mismatched all-gathers until the lease ended (round 3, gpurun_out/two_rank_lanes.err): the steps' exchanges then go through
    :class:`OrderedExchange`, ONE thread per rank that issues them in step order."""
    if not overlap or not graph:
        return 1
    return max(1, min(int(lanes), 4))


class OrderedExchange:
    """WORLD_SIZE > 1 with several lanes: every rank has to issue the steps' all-gathers in the SAME order, but the lanes' host
    threads finish their steps in any order.  The lanes hand their step's table (and the event after which it is complete) to this
    object; its one thread issues the exchanges in step order 0, 1, 2, ... on a stream of its own -- from the collective library's
    point of view a single thread issuing one all-gather after the other, exactly as with one lane.  (The bench does not use the
    gathered table, so a lane does not wait for its exchange; ``finish`` -- inside the timed region -- does.)"""

    def __init__(self, dev_index: int, n_steps: int, exchange):
        import threading
        self.cv = threading.Condition()
        self.items, self.err, self.abort = {}, None, False
        self.thread = threading.Thread(target=self._run, args=(dev_index, n_steps, exchange), daemon=True)
        self.thread.start()

    def submit(self, i: int, hdr, pay, event) -> None:
        with self.cv:
            self.items[i] = (hdr, pay, event)
            self.cv.notify_all()

    def _run(self, dev_index, n_steps, exchange) -> None:
        try:
            torch.cuda.set_device(dev_index)
            stream = torch.cuda.Stream(device=f"cuda:{dev_index}")
            with torch.cuda.stream(stream):
                for i in range(n_steps):
                    with self.cv:
                        while i not in self.items and not self.abort:
                            self.cv.wait(timeout=1.0)
                        if self.abort:
                            return
                        hdr, pay, event = self.items.pop(i)
                    stream.wait_event(event)
                    exchange(hdr, pay)
        except BaseException as e:          # re-raised by finish()
            self.err = e

    def finish(self, failed: bool = False) -> None:
        if failed:
            with self.cv:
                self.abort = True
                self.cv.notify_all()
        self.thread.join()
        if self.err is not None:
            raise self.err


def load_or_make_device_tiles(first: int, n: int, size: int, dev, cache_dir):
    """``n`` device-generated tiles (synth.em_tiles_device, seeds first ..).  With ``cache_dir`` they are read from / written
    to ``<cache_dir>/tiles_<first>_<n>_<size>.npy``: a profiled run (rocprofv3 --pmc) then starts from a host copy and an
    H2D memcpy instead of the generator's torch kernels (a --pmc pass crashed inside one of those in round 3)."""
    from deepemia_amd import synth
    if cache_dir:
        path = Path(cache_dir) / f"tiles_{first}_{n}_{size}.npy"
        if path.exists():
            return torch.from_numpy(np.load(path)).to(dev)
        t = synth.em_tiles_device(range(first, first + n), size, dev)
        path.parent.mkdir(parents=True, exist_ok=True)
        np.save(path, t.cpu().numpy())
        return t
    return synth.em_tiles_device(range(first, first + n), size, dev)


def cli_leg(sd, depth: int, dev, n_images: int, value: float, tmp_root=None) -> dict:
    """Side leg (never `value`): the DROP-IN itself -- ``main.py --task inference`` run as a user runs it, a subprocess with its own
    config tree, Detectron2-layout checkpoint and image folder -- on (a) a folder of ``n_images`` synthetic 2048^2 micrographs
    (full-image pass + 9 overlapping tiles of 1024 each = 10 forwards of the 800-pixel network per image, containment + overlap rules,
    RLE and measurement CSVs) and (b) BASELINE configs[2]'s shape, ONE 8192^2 image (full-image pass + 16 tiles of 2048^2).  Reports
    the image loop's seconds per image and tile-forwards/s next to the headline `value` (same network work per forward).
    Reference loop: src/functions/inference.py:713-942, 2299-2485."""
    import csv
    import re
    import tempfile

    import yaml
    from PIL import Image

    from deepemia_amd import synth
    Image.MAX_IMAGE_PIXELS = None
    # the weights of the CLI parity cases and of scripts/gpu_cli_throughput.py (soft masks: mask_bias 0.5, mask_gain 6; the headline
    # weights' solid masks make every region program a large-region one and measure the morphology kernels, not the loop)
    sd = synth.random_d2_state_dict(depth, 2, seed=0, mask_bias=0.5, mask_gain=6.0)
    root = Path(tmp_root or tempfile.mkdtemp(prefix="deepemia_cli_leg_"))
    name = "benchfolder"

    def tree(sub, tile, images):
        base = root / sub
        (base / "cfg" / "datasets").mkdir(parents=True, exist_ok=True)
        split = base / "split_dir"
        cfg = {"bucket": None,
               "paths": {"split_dir": str(split), "category_json": str(base / "dataset_info.json"), "local_dataset_root": str(base)},
               "inference_settings": {"confidence_mode": "manual", "ensemble_settings": {"enabled": False},
                                      "spatial_constraints": {"default": {"enabled": False}}},
               "l4_performance_optimizations": {"enable_parallel_mask_processing": True}}
        (base / "cfg" / "config.yaml").write_text(yaml.safe_dump(cfg, sort_keys=False))
        ds = {"inference_overrides": {"confidence_mode": "manual",
                                      "class_specific_settings": {"class_0": {"confidence_threshold": 0.3, "iou_threshold": 0.6, "min_size": 25},
                                                                  "class_1": {"confidence_threshold": 0.35, "iou_threshold": 0.5, "min_size": 5}},
                                      "tile_settings": tile,
                                      "spatial_constraints": {"enabled": True, "containment_rules": {1: 0}, "containment_threshold": 0.5,
                                                              "overlap_rules": {0: {"allow_overlap": False, "max_iou_threshold": 0.3}}}}}
        (base / "cfg" / "datasets" / f"{name}.yaml").write_text(yaml.safe_dump(ds, sort_keys=False))
        (base / "dataset_info.json").write_text(json.dumps({name: ["imgs", "labels", ["pore", "throat"]]}))
        mdir = split / name / f"rcnn_r{depth}"
        mdir.mkdir(parents=True, exist_ok=True)
        synth.save_d2_checkpoint(str(mdir / f"model_final_r{depth}.pth"), sd)
        inf = base / "DATASET" / "INFERENCE"
        inf.mkdir(parents=True, exist_ok=True)
        for fn, arr in images:
            Image.fromarray(arr).save(inf / fn, compress_level=1)
        return base, split

    def run(base, split, workers):
        env = dict(os.environ, DEEPEMIA_CONFIG_DIR=str(base / "cfg"), DEEPEMIA_OFFLINE="1", DEEPEMIA_WORKERS=str(workers), DEEPEMIA_LOG_DIR=str(base))
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
            env.pop(k, None)
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, str(ROOT / "main.py"), "--task", "inference", "--dataset_name", name, "--threshold", "0.3", "--no-gpu-check"],
                           cwd=str(base), env=env, capture_output=True, text=True, timeout=300)
        wall = time.perf_counter() - t0
        log = r.stderr + r.stdout
        loops = [(int(a), float(b)) for a, b in re.findall(r"Inference complete: (\d+)/\d+ images, avg ([0-9.]+)s/image", log)]
        tm = re.search(r"Inference task finished in ([0-9.]+)s", log)
        rows = sum(1 for _ in csv.reader(open(split / "measurements_results.csv"))) - 1 if (split / "measurements_results.csv").exists() else 0
        if (r.returncode != 0 or os.environ.get("DEEPEMIA_KEEP_CLI_LOG")) and (ROOT / "gpurun_out").is_dir():
            (ROOT / "gpurun_out" / f"cli_leg_{'failed_' if r.returncode else ''}{base.name}_workers_{workers}.log").write_text(log)
        return {"rc": r.returncode, "processes_on_the_gpu": len(loops), "image_loop_s": max((a * b for a, b in loops), default=None),
                "task_s": float(tm.group(1)) if tm else None, "wall_s_incl_start_up": wall, "csv_rows": rows,
                **({} if r.returncode == 0 else {"stderr_tail": log[-800:]})}

    # (a) the folder: device-generated micrographs (gray, like the reference's inputs), written as PNG
    tiles = synth.em_tiles_device(range(300, 300 + n_images), 2048, dev)[..., 0].cpu().numpy()
    base, split = tree("folder", {"tile_size": 1024, "overlap_ratio": 0.125, "upscale_factor": 1.0, "edge_filter_enabled": True},
                       [(f"em_{i:03d}.png", tiles[i]) for i in range(n_images)])
    fwd_per_image = 10
    out = {"workload": f"main.py --task inference (subprocess) on {n_images} synthetic 2048^2 images: per image the full-image pass + 9 tiles of "
                       f"1024 (12.5 % overlap) = {fwd_per_image} forwards, class loops, 0.4 / 0.7 dedups, containment + overlap rules, RLE + measurement CSVs; "
                       f"R{depth} with the soft-mask weights of the CLI parity cases (mask_bias 0.5, mask_gain 6); one untimed run first",
           "folder": {}}
    run(base, split, "1")          # (untimed: the first CLI run on a fresh box pays the page-ins of the library, of torch and of the images)
    for label, workers in (("default", "auto"), ("one_process", "1")):
        rec = run(base, split, workers)
        if rec["image_loop_s"]:
            rec["ms_per_image"] = 1e3 * rec["image_loop_s"] / n_images
            rec["tile_forwards_per_s"] = n_images * fwd_per_image / rec["image_loop_s"]
            rec["frac_of_value"] = rec["tile_forwards_per_s"] / value if value else None
        out["folder"][label] = rec
    # (b) configs[2]: one 8192^2 image = 16 device-generated tiles side by side
    t16 = synth.em_tiles_device(range(100, 116), 2048, dev)[..., 0].cpu().numpy()
    big = np.concatenate([np.concatenate(list(t16[4 * r_:4 * r_ + 4]), axis=1) for r_ in range(4)], axis=0)
    base2, split2 = tree("c2", {"tile_size": 2048, "overlap_ratio": 0.0, "upscale_factor": 1.0, "edge_filter_enabled": True}, [("big.png", big)])
    rec = run(base2, split2, "1")
    if rec["image_loop_s"]:
        rec["tile_forwards_per_s"] = 17 / rec["image_loop_s"]
    rec["workload"] = "configs[2] shape: ONE 8192^2 image, full-image pass + 16 tiles of 2048^2 (17 forwards), one process"
    out["one_8192_image"] = rec
    import shutil
    shutil.rmtree(root, ignore_errors=True)
    return out


def free_port() -> int:
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return int(s_.getsockname()[1])


def launch_ranks(args, argv) -> int:
    """Launcher role: ``python bench.py --gpus N`` with N > 1 and no WORLD_SIZE in the environment.  Starts the N ranks as a CHILD
    process (``python -m torch.distributed.run``; rendezvous on 127.0.0.1, one rank per GPU) with this command's own arguments, lets
    the child write straight to this process's stdout / stderr (rank 0 prints the one JSON line) and returns its exit code.  This
    process never calls into HIP: it neither execs from a GPU-initialised process nor holds a context beside the ranks'."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(Path(__file__).resolve()), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL across processes)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    try:
        # this process's stdout carries rank 0's JSON line(s) only; anything else the ranks (or gloo's C++ side) print goes to stderr
        for line in proc.stdout:
            dst = sys.stdout if line.lstrip().startswith("{") else sys.stderr
            dst.write(line)
            dst.flush()
        return proc.wait()
    except BaseException:
        proc.terminate()
        try:
            proc.wait(timeout=30)
        except subprocess.TimeoutExpired:
            proc.kill()
        raise


class LaneChildren:
    """The rank's further lanes as PROCESSES (``--lane-mode processes``): each is this script again with ``--lane-child l``, one
    software pipeline with its own interpreter, HIP context, engine, hipGraphs and resident inputs.  Started before the rank makes
    its first GPU call; commands go down the child's stdin (INIT / RUN / EXIT), replies come back as JSON lines on a pipe of their
    own (the children's stdout goes to this process's stderr: the rank's stdout carries the one JSON line only).  A child that
    dies closes its pipe, which ``recv`` turns into an error; a rank that dies closes the children's stdin, which ends them."""

    def __init__(self, n: int, argv):
        self.procs, self.readers = [], []
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        for l_ in range(1, n + 1):
            r_, w_ = os.pipe()
            p_ = subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *argv, "--lane-child", str(l_), "--lane-fd", str(w_)],
                                  stdin=subprocess.PIPE, stdout=sys.stderr, pass_fds=(w_,), env=env, text=True)
            os.close(w_)
            self.procs.append(p_)
            self.readers.append(os.fdopen(r_, "r"))

    def send(self, line: str, only=None) -> None:
        for k, p_ in enumerate(self.procs):
            if only is None or k == only:
                p_.stdin.write(line + "\n")
                p_.stdin.flush()

    def recv(self, tag: str):
        out = []
        for k, rd in enumerate(self.readers):
            while True:
                line = rd.readline()
                if not line:
                    raise RuntimeError(f"lane child {k + 1} ended (exit code {self.procs[k].poll()}) while the rank waited for {tag}")
                if line.startswith(tag + " "):
                    out.append(json.loads(line[len(tag) + 1:]))
                    break
        return out

    def close(self, kill: bool = False) -> None:
        for p_ in self.procs:
            try:
                if p_.poll() is None and not kill:
                    p_.stdin.write("EXIT\n")
                    p_.stdin.flush()
                p_.stdin.close()
            except (BrokenPipeError, OSError, ValueError):
                pass
        for p_ in self.procs:
            try:
                p_.wait(timeout=5 if kill else 120)
            except subprocess.TimeoutExpired:
                p_.kill()          # (the exact child this object started)
                p_.wait()


def join_groups(args, world, rank, my_lane, children, backend, dev_index):
    """The process groups of a run: the ranks (lane 0 of every GPU) form the default group torch.distributed.run prepared;
    the lane-l children of all ranks form a group of their own, over a TCP store on a port rank 0 picks and every rank hands
    to its lane-l child (INIT <port>): each step's all-gather then has exactly one issuer per rank and group, whatever the
    lanes' relative timing.  Returns torch.distributed (None when N = 1; False when a lane child's rank went away)."""
    lane_port = 0
    if my_lane:
        # a lane child waits for the rank's INIT line: the rendezvous port of ITS process group (lane l of every rank), 0 if N = 1
        first = sys.stdin.readline().split()
        if not first or first[0] != "INIT":
            return False
        lane_port = int(first[1])
    if world == 1:
        if children is not None:
            children.send("INIT 0")
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    on_gpu = backend == "nccl"
    if on_gpu:
        torch.cuda.set_device(dev_index)
    kw = {}
    if my_lane:
        # an explicit store: under torch.distributed.run a tcp:// rendezvous would look for the elastic agent's store instead
        import datetime
        store = dist.TCPStore("127.0.0.1", lane_port, world, rank == 0, timeout=datetime.timedelta(seconds=180))
        kw = {"store": store, "rank": rank, "world_size": world, "timeout": datetime.timedelta(seconds=600)}
    if on_gpu:
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{dev_index}"), **kw)
    else:
        dist.init_process_group(backend, **kw)
    if children is not None:
        ports = [[free_port() for _ in range(args.lanes - 1)]] if rank == 0 else [None]
        dist.broadcast_object_list(ports, src=0, device=torch.device(f"cuda:{dev_index}" if on_gpu else "cpu"))
        for k, port in enumerate(ports[0]):
            children.send(f"INIT {port}", only=k)
    return dist


def parity_step_of(steps: int, n_batches: int, n_lanes: int, total_tiles: int) -> int:
    """The timed step whose tile 0 is synthetic tile 0 AND that lane 0 (the rank process itself) runs: the last multiple of
    lcm(distinct batches, lanes) below ``steps`` (a job of distinct tiles: step 0)."""
    if total_tiles:
        return 0
    import math
    q = math.lcm(max(1, n_batches), max(1, n_lanes))
    return ((steps - 1) // q) * q


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=48, help="tiles per GPU per step (48: the res3-res5 layers fill the 256 CUs for "
                    "several rounds per launch; 26 GiB of activations)")
    ap.add_argument("--precision", choices=["f32", "f32x3", "f16x2", "f16", "f16x2r", "bf16x2", "bf16"], default="f16x2",
                    help="f16x2 (default, parity); f16 = flagged NON-parity single-plane fp16 operands (the reference's autocast arithmetic)")
    ap.add_argument("--single-stages", default="", help="(flagged NON-parity opt-in) comma list of stages of the f16x2 forward on ONE MFMA per "
                    "product, e.g. mask_fcn,deconv: holds the eight-tile headline parity record, fails a soft-mask CLI parity case "
                    "(DESIGN.md section 7)")
    ap.add_argument("--depth", type=int, default=101)
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--threshold", type=float, default=0.3)
    ap.add_argument("--min-size-test", type=int, default=800, help="INPUT.MIN_SIZE_TEST; anything but 800 is the flagged NON-parity "
                    "native-resolution mode (not BASELINE.json's workload)")
    ap.add_argument("--max-size-test", type=int, default=1333)
    ap.add_argument("--total-tiles", type=int, default=0, help="BASELINE configs[4]: a job of this many DISTINCT tiles per GPU (e.g. 256), "
                    "walked in steps of --batch; overrides --steps")
    ap.add_argument("--distinct-batches", type=int, default=2, help="the timed loop alternates this many DISTINCT resident batches (step i "
                    "takes batch i %% n): batch 0 = the numpy tiles (tile 0 = the parity tile) + device-generated ones, the others all "
                    "device-generated with their own seeds -- the incremental paste and the plane pools see changing boxes every step. "
                    "1 = the same batch every step (also reported as the side field `same_batch_every_step`)")
    ap.add_argument("--tiles-cache", default="", help="directory of .npy copies of the device-generated tiles: written when missing, "
                    "read (numpy -> H2D copy, no generator kernels) when present -- make it with an unprofiled run, then profile")
    ap.add_argument("--cpu-tiles", type=int, default=3, help="tiles of the timed CPU sample (cpu_baseline); ~27 s each")
    ap.add_argument("--repeat-tiles", action="store_true", help="(profiling passes) fill a batch beyond the 16 numpy tiles with copies of "
                    "them instead of device-generated tiles: no torch compute kernels before the forward (rocprofv3 --pmc crashed in one)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the timed CPU sample AND the parity leg")
    ap.add_argument("--parity-only", action="store_true", help="run the CPU path on tile 0 only: the parity check of the timed result "
                    "without the 3-tile cpu_baseline timing (the -m gpu configs[4] job)")
    ap.add_argument("--no-h2d-leg", action="store_true", help="skip the second timed leg that uploads every step's tiles from pinned "
                    "host memory on a copy stream (reported as `h2d`, never as `value`)")
    ap.add_argument("--no-plane-pools", action="store_true", help="(A/B) gather mask sets into fresh zero-filled planes instead of the plane pools")
    ap.add_argument("--no-csv-text", action="store_true", help="leave the CSV text (a19) of every step out of the timed region")
    ap.add_argument("--lanes", type=int, default=2, help="independent software pipelines in flight on the GPU (default 2), each with its own "
                    "engine arena, hipGraphs, streams and host side; steps are dealt out round robin.  A second forward fills the CUs "
                    "the first leaves idle in the partly filled last round of every launch and in its HBM-bound layers; three lanes "
                    "are slower than one.  The line carries `one_lane` (the same K steps with lane 0 alone, same process) beside `value`")
    ap.add_argument("--lane-mode", choices=["processes", "threads"], default="processes", help="processes (default): lanes 1.. are "
                    "child PROCESSES of the rank (own interpreter and HIP context, started before the rank initialises the GPU; with "
                    "N > 1 the lane-l processes of all ranks form their own process group, so every step's all-gather has one issuer "
                    "per rank); threads: round 4's host threads in one interpreter (exchanges of N > 1 through OrderedExchange)")
    ap.add_argument("--lane-child", type=int, default=0, help=argparse.SUPPRESS)       # (internal) this process is lane l of its rank
    ap.add_argument("--lane-fd", type=int, default=-1, help=argparse.SUPPRESS)         # (internal) reply pipe of a lane child
    ap.add_argument("--no-cli-leg", action="store_true", help="skip the `cli` side leg (main.py --task inference as a subprocess on a folder "
                    "of 2048^2 images and on one 8192^2 image; ~40 s)")
    ap.add_argument("--cli-images", type=int, default=64, help="images of the `cli` side leg's folder")
    ap.add_argument("--no-one-lane-leg", action="store_true", help="skip the `one_lane` side leg (K more steps with lane 0 alone)")
    ap.add_argument("--rendezvous-only", action="store_true", help="(test hook, no GPU) ranks rendezvous over gloo, rank 0 prints "
                    "{n_gpus, rendezvous_only} and everything exits: checks the launcher role on a CPU box")
    ap.add_argument("--no-overlap", action="store_true", help="do not overlap batch i+1's network with batch i's post-processing")
    ap.add_argument("--forward-only", action="store_true", help="time predictor(tile) only, without the per-tile post-processing")
    ap.add_argument("--post-priority", choices=["default", "high", "low"], default="default", help="(experiment) priority of the post-processing stream")
    ap.add_argument("--no-conv-events", action="store_true", help="(experiment) do not bracket the conv launches with HIP events")
    ap.add_argument("--eager", action="store_true", help="launch the forward kernel by kernel instead of replaying one hipGraph per batch "
                    "(the default replays: +20 %% tiles/s; per-kernel HIP events cannot be taken inside a replayed graph, so the roofline "
                    "numbers of the default mode come from two instrumented eager steps of the same path, run right after the timed region)")
    args = ap.parse_args()

    # ---- role (before anything touches the GPU) ----
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: refusing to report a GPU count that is not the one running "
              f"(start it as `python bench.py --gpus N`, or under torch.distributed.run with --nproc-per-node equal to --gpus)", file=sys.stderr)
        sys.exit(2)
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.graph = not args.eager and args.precision in ("f16x2", "f16")
    args.lanes = resolve_lanes(args.lanes, world, overlap=not args.no_overlap, graph=bool(args.graph))
    my_lane = int(args.lane_child)                     # 0: the rank process itself
    import signal
    if my_lane:
        # a lane child must not outlive its rank (a rank that torch.distributed.run terminates cannot say goodbye, and a child that
        # sits in a collective of its lane group would hold the GPU until the collective's timeout): the kernel ends it with the rank
        try:
            import ctypes
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, int(signal.SIGKILL))        # PR_SET_PDEATHSIG
        except OSError:
            pass
    else:
        # SIGTERM (torch.distributed.run tearing the job down) becomes an exception, so that the `except` below ends the lane children
        signal.signal(signal.SIGTERM, lambda *_: sys.exit(143))
    proc_lanes = args.lane_mode == "processes" and args.lanes > 1
    children = None
    reply = os.fdopen(args.lane_fd, "w") if my_lane else None
    if proc_lanes and not my_lane:
        # the further lanes of this GPU, started BEFORE this process makes its first GPU call
        children = LaneChildren(args.lanes - 1, [a for a in sys.argv[1:]])
    if args.rendezvous_only:
        # (test hook, no GPU) the roles and the process groups only: every rank and every lane child joins its group over gloo
        # and takes part in one all-reduce of (lane + 1); rank 0 reports the sums it and its children saw
        try:
            dist = join_groups(args, world, rank, my_lane, children, "gloo", 0)
            if dist is False:
                return
            total = my_lane + 1
            if dist is not None:
                t_ = torch.tensor([my_lane + 1])
                dist.all_reduce(t_)
                total = int(t_.item())
            if my_lane:
                reply.write("READY " + json.dumps({"sum": total}) + "\n")
                reply.flush()
            else:
                sums = [total] + ([r_["sum"] for r_ in children.recv("READY")] if children is not None else [])
                if rank == 0:
                    print(json.dumps({"n_gpus": world, "rendezvous_only": True, "lanes": args.lanes, "lane_group_sums": sums}), flush=True)
            if dist is not None:
                dist.barrier()
                dist.destroy_process_group()
        except BaseException:
            if children is not None:
                children.close(kill=True)
            raise
        if children is not None:
            children.close()
        return
    try:
        run_rank(args, world, rank, local_rank, my_lane, proc_lanes, children, reply)
    except BaseException:
        if children is not None:
            children.close(kill=True)
        raise
    if children is not None:
        children.close()


def run_rank(args, world, rank, local_rank, my_lane, proc_lanes, children, reply) -> None:  # noqa: C901
    """A rank (lane 0 of its GPU, all side legs, the JSON line) or one of its lane children (``my_lane`` > 0: set-up, warm-up,
    then the RUN commands of the rank)."""
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, ncpu // max(world, 1))))      # N ranks share the host: no oversubscription
    # rehearsal of the N > 1 path on a one-GPU box: DEEPEMIA_BENCH_BACKEND=gloo DEEPEMIA_BENCH_ONE_DEVICE=1 (all ranks on cuda:0)
    backend = os.environ.get("DEEPEMIA_BENCH_BACKEND", "nccl")
    one_dev = os.environ.get("DEEPEMIA_BENCH_ONE_DEVICE", "0") == "1"
    dev_index = 0 if (world == 1 or one_dev) else local_rank
    if dev_index >= torch.cuda.device_count():          # (counting devices does not initialise HIP)
        print(f"bench.py: rank {rank} needs cuda:{dev_index} but {torch.cuda.device_count()} device(s) are visible (--gpus {args.gpus})", file=sys.stderr)
        sys.exit(2)
    dist = join_groups(args, world, rank, my_lane, children, backend, dev_index)
    if dist is False:
        return
    torch.cuda.set_device(dev_index)
    dev = f"cuda:{dev_index}"

    from deepemia_amd import parallel, synth
    from deepemia_amd.engine import MaskRCNNEngine
    from deepemia_amd.functions.inference import InferencePipeline, measurement_csv_text
    from deepemia_amd.predictor import Predictor

    sd = synth.random_d2_state_dict(args.depth, 2, seed=0)
    import types
    lanes = []                                   # the lanes THIS process runs: one in the processes mode, all in the threads mode
    for li in ([my_lane] if proc_lanes else range(args.lanes)):
        # a lane = one software pipeline (forward of batch i + L under the post-processing of batch i): its own engine, i.e. its
        # own intermediates arena and hipGraphs (the weights are duplicated: 0.5 GB), its own plane pools, streams and host thread
        e_ = MaskRCNNEngine(sd, args.depth, 2, args.threshold, dev, args.precision, args.min_size_test, args.max_size_test,
                            single_stages=tuple(t for t in args.single_stages.split(",") if t) if args.single_stages else None)
        p_ = InferencePipeline([Predictor(e_)], f"bench{li}", {}, {})
        p_.use_graphs = bool(args.graph)
        p_.graph_after = 1
        p_.forward_batch = args.batch             # ONE forward per step over the whole batch (the CLI default chunks at 16)
        # the loop below consumes batch i's detections before it launches the lane's forward after next: two graph slots, results read in place
        p_.graph_slots, p_.clone_graph_outputs = 2, False
        p_.pooled_planes = not args.no_plane_pools   # ... and its masks before the lane's next post-processing: plane pools
        lanes.append(types.SimpleNamespace(eng=e_, pipe=p_, last={}, net=None, post=None, lid=li))
    eng, pipe = lanes[0].eng, lanes[0].pipe       # (lane 0: the roofline instrumentation, the parity snapshot, the upload leg)
    # tiles 0..NUMPY_TILES-1 of the batch are the byte-reproducible numpy tiles (tile 0 is what the parity leg checks, the
    # CPU baseline times tiles 0..2); the rest of a large batch comes from the device generator (1.4 s of host numpy per tile)
    n_np = min(args.batch, NUMPY_TILES)
    tiles = np.stack([synth.em_tile(rank * args.batch + i, args.size) for i in range(n_np)])
    sha_tile0 = synth.hashlib.sha256(tiles[0].tobytes()).hexdigest()
    x = torch.empty((args.batch, args.size, args.size, 3), dtype=torch.uint8, device=dev)
    x[:n_np].copy_(torch.from_numpy(tiles))
    if args.batch > n_np and args.repeat_tiles:
        for k in range(n_np, args.batch, n_np):
            x[k:k + n_np].copy_(x[:min(n_np, args.batch - k)])
    elif args.batch > n_np:
        first = 50000 + rank * args.batch
        x[n_np:].copy_(load_or_make_device_tiles(first + n_np, args.batch - n_np, args.size, dev, args.tiles_cache))
    xs = [x]
    if args.total_tiles:
        # configs[4]: every step sees DIFFERENT tiles.  Step 0 holds the numpy tiles (tile 0 is the parity check's tile); the
        # other steps' tiles are generated on the device with the same recipe and their own seeds (synth.em_tiles_device:
        # 1.4 s of host numpy per tile otherwise) -- total_tiles distinct tiles per GPU, resident before the timed region
        assert args.total_tiles % args.batch == 0, "--total-tiles must be a multiple of --batch"
        args.steps = args.total_tiles // args.batch
        for k in range(1, args.steps):
            first = 100000 + (rank * args.steps + k) * args.batch
            xs.append(load_or_make_device_tiles(first, args.batch, args.size, dev, args.tiles_cache))
    elif not args.repeat_tiles:
        # the headline loop alternates DISTINCT resident batches: different boxes in every slot from one step to the next
        for k in range(1, max(1, args.distinct_batches)):
            first = 200000 + (rank * 16 + k) * args.batch
            xs.append(load_or_make_device_tiles(first, args.batch, args.size, dev, args.tiles_cache))
    # the last timed step that runs batch 0 (tile 0 = synthetic tile 0) on lane 0: the step the parity leg looks at
    parity_step = parity_step_of(args.steps, len(xs), args.lanes, args.total_tiles)
    n_par_tiles = max(1, min(1 if args.parity_only else args.cpu_tiles, n_np))      # tiles 0 .. n-1 of that step are checked
    cur = {"xs": xs}            # the resident batches of the running leg (the `same batch` side leg takes [x])

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    DIAG_POST = os.environ.get("DEEPEMIA_BENCH_POST", "")
    MIN_AREA = max(5, args.size * args.size * 0.000005 * 0.05)      # inference.py:1175-1190
    prio = {"default": 0, "high": -1, "low": 1}[args.post_priority]
    for ln in lanes:
        ln.net = torch.cuda.Stream(device=dev)                      # network of the lane's next batch ...
        ln.post = torch.cuda.Stream(device=dev, priority=prio)      # ... runs under the post-processing of its current one
    last = lanes[0].last
    import threading
    post_lock = threading.Lock()
    xchg = {}          # (lanes > 1) the start offset between lanes, set after the warm-up; N > 1: the ordered exchange of the running leg

    TRACE = [] if os.environ.get("DEEPEMIA_BENCH_TRACE") else None      # (diagnostic) host timestamps of every lane's steps

    def launch(i, ln=lanes[0]):
        if TRACE is not None:
            TRACE.append((ln.lid, i, "launch", time.perf_counter()))
        with torch.cuda.stream(ln.net):
            return ln.pipe.forward_async(0, cur["xs"][i % len(cur["xs"])] if i >= 0 else x)

    def step(i, handle=None, ln=lanes[0]):
        """One pass of the hot path over this rank's batch of tiles."""
        if args.forward_only:
            with torch.cuda.stream(ln.net):
                raw = (ln.eng.forward_graphed if ln.pipe.use_graphs else ln.eng.forward)(cur["xs"][i % len(cur["xs"])] if i >= 0 else x)
                return int(raw.count.sum().item()), 0
        with torch.cuda.stream(ln.post):
            return post(i, handle if handle is not None else launch(i, ln), ln)

    def post(i, handle, ln):
        tp0 = time.perf_counter()
        try:
            return _post(i, handle, ln)
        finally:
            ln.last["post_s"] = ln.last.get("post_s", 0.0) + time.perf_counter() - tp0

    def _post(i, handle, ln):
        dets = ln.pipe.finish_forward(handle)
        tq0 = time.perf_counter()
        if TRACE is not None:
            TRACE.append((ln.lid, i, "fwd_done", tq0))
        try:
            # one lane post-processes at a time: two host loops at once share the interpreter lock and their kernels the same CUs
            # (48-68 ms per pass instead of 22), and taking turns keeps the lanes half a period apart
            with post_lock:
                if TRACE is not None:
                    TRACE.append((ln.lid, i, "post_start", time.perf_counter()))
                return _post_after_forward(i, dets, ln)
        finally:
            if TRACE is not None:
                TRACE.append((ln.lid, i, "post_done", time.perf_counter()))
            ln.last["post_after_fwd_s"] = ln.last.get("post_after_fwd_s", 0.0) + time.perf_counter() - tq0

    def _post_after_forward(i, dets, ln):
        if DIAG_POST == "wait":          # (diagnostic: the forward's tables fetched, no post-processing kernels at all)
            time.sleep(0.02)
            return 0, 0
        res = ln.pipe.process_tile_batch(f"step{i}", cur["xs"][i % len(cur["xs"])] if i >= 0 else x, SMALL_CLASSES, CLASS_THRESHOLDS, dets=dets)
        if not my_lane and (i == parity_step or ln.last.get("keep_every")):
            # the step whose first tiles are synthetic tiles 0, 1, 2 (the parity check's references); with plane pools the masks
            # are views that the lane's next step overwrites, and later steps follow: keep the checked tiles' own copies
            ln.last["res"] = [((r[0].clone() if r[0] is not None else None),) + tuple(r[1:]) for r in res[:n_par_tiles]]
        n_inst = sum(0 if r[0] is None else int(r[0].shape[0]) for r in res)
        if args.no_csv_text:
            n_rows = sum(len(c) for r in res for c in r[3])
        else:
            # a19: the CSV text of the step's tiles (20 columns, what csv.writer writes, area gate of inference.py:1175-1190) into
            # memory: the float columns of all rows through one native call (measurement_csv_text; 4 ms per 2700 rows)
            text = measurement_csv_text([(f"tile{rank * args.batch + t_}.tif", r[2], r[3]) for t_, r in enumerate(res)],
                                        ("class_0", "class_1"), MIN_AREA)
            n_rows, ln.last["csv_bytes"] = text.count("\r\n"), len(text)
        if dist is not None:
            # the one exchange of the path: instance tables of every rank's tiles (unit id = global tile index);
            # areas / boxes come from the reductions the path has already done, the crop is one launch
            parts = [r[0] for r in res if r[0] is not None and r[0].shape[0]]
            if parts:
                packed = torch.cat(parts)
                scores = [s_ for r in res for s_ in r[1]]
                classes = [c_ for r in res for c_ in r[2]]
                units = [rank * args.batch + t for t, r in enumerate(res) for _ in r[1]]
                stats = [st for st, r in zip(ln.pipe.last_batch_stats, res) if r[0] is not None and r[0].shape[0]]
                area = np.concatenate([a for a, _ in stats])
                bbox = np.concatenate([b for _, b in stats])
                hdr, pay = parallel.encode_instance_table(packed, scores, classes, units, bbox, area)
            else:
                hdr = torch.zeros((0, parallel.HDR), dtype=torch.int32, device=dev)
                pay = torch.zeros((0,), dtype=torch.int32, device=dev)
            ox = xchg.get("ordered") if i >= 0 else None
            if ox is not None:           # several lanes: this step's exchange is issued in step order by the rank's exchange thread
                ev_ = torch.cuda.Event()
                ev_.record(torch.cuda.current_stream(dev))
                ox.submit(i, hdr, pay, ev_)
            else:
                parallel.all_gather_instance_tables(hdr, pay)
        return n_inst, n_rows

    def run_lane(ln, idxs, out, launcher=launch, stagger_s=0.0):
        """The lane's share of the timed steps, software-pipelined: the forward of its next batch is enqueued before the host
        starts the post-processing of its current one, so the MFMA-bound network hides the latency-bound mask work."""
        try:
            torch.cuda.set_device(dev_index)
            if ln.lid and stagger_s:
                # lanes start one after the other, a lane's share of the period apart: two forwards launched together run the same
                # layers side by side (and finish, post-process and drain together); half a period apart one is in its MFMA-bound
                # layers while the other is in its HBM-bound ones.  The GPU is not idle meanwhile: the earlier lanes are running.
                time.sleep(ln.lid * stagger_s)
            if args.forward_only or args.no_overlap:
                for i in idxs:
                    out[0] = step(i, None, ln)
                return
            handle = launcher(idxs[0], ln) if idxs else None
            for k, i in enumerate(idxs):
                nxt = launcher(idxs[k + 1], ln) if k + 1 < len(idxs) else None
                out[0] = step(i, handle, ln)
                handle = nxt
        except BaseException as e:      # re-raised by the caller
            out[1] = e

    h2d_state = {}

    def launcher_of(leg):
        """The legs a rank can ask of its lanes: main (the resident batches in turn), same (ONE resident batch every step),
        h2d (every step's tiles uploaded from pinned host memory on a copy stream)."""
        cur["xs"] = [x] if leg == "same" else xs
        if leg != "h2d":
            return launch
        if not h2d_state:
            # two device slots per lane; the forward waits for its upload, the upload of step i + L runs under the forward of step i
            h2d_state.update(pinned=[t_.cpu().pin_memory() for t_ in xs], stream=torch.cuda.Stream(device=dev), events=[])
            for ln in lanes:
                ln.up_slots, ln.up_free, ln.up_n = [torch.empty_like(x), torch.empty_like(x)], [None, None], 0

        def launch_up(i, ln=lanes[0]):
            sl = ln.up_n % 2
            ln.up_n += 1
            cs, pinned = h2d_state["stream"], h2d_state["pinned"]
            with torch.cuda.stream(cs):
                if ln.up_free[sl] is not None:
                    cs.wait_event(ln.up_free[sl])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cs)
                ln.up_slots[sl].copy_(pinned[i % len(pinned)], non_blocking=True)
                e1.record(cs)
            h2d_state["events"].append((e0, e1))
            with torch.cuda.stream(ln.net):
                ln.net.wait_event(e1)
                hd = ln.pipe.forward_async(0, ln.up_slots[sl])
            ln.up_free[sl] = hd[1]
            return hd
        return launch_up

    def reset_stats():
        for ln in lanes:
            ln.last["post_s"] = ln.last["post_after_fwd_s"] = 0.0
            ln.waits0 = ln.pipe.d2h_waits

    def local_stats():
        return {"d2h_waits": sum(ln.pipe.d2h_waits - ln.waits0 for ln in lanes),
                "post_s": sum(ln.last.get("post_s", 0.0) for ln in lanes),
                "post_after_fwd_s": sum(ln.last.get("post_after_fwd_s", 0.0) for ln in lanes)}

    def run_steps(n_steps, leg="main", n_lanes=None):
        """n_steps passes dealt out to the lanes round robin (step i -> lane i % L).  Processes mode: this process runs its own
        lane's share, the rank tells its lane children to run theirs (RUN ... / DONE ...) and adds up their counters; threads
        mode: one host thread per lane in this process.  ``n_lanes=1``: lane 0 alone (the `one_lane` side leg).
        Returns ((instances, rows) of the last step, counters over all lanes)."""
        L = args.lanes if n_lanes is None else n_lanes
        launcher = launcher_of(leg)
        stagger_s = xchg.get("stagger_s", 0.0) if L > 1 else 0.0
        reset_stats()
        outs = [[(0, 0), None] for _ in lanes]
        kids = children is not None and L > 1
        if kids:
            children.send(f"RUN {leg} {n_steps} {stagger_s:.6f}")
        if len(lanes) == 1 or L == 1:
            run_lane(lanes[0], list(range(lanes[0].lid if L > 1 else 0, n_steps, L)), outs[0], launcher, stagger_s)
        else:
            if dist is not None and not args.forward_only:
                xchg["ordered"] = OrderedExchange(dev_index, n_steps, parallel.all_gather_instance_tables)
            ths = [threading.Thread(target=run_lane, args=(ln, list(range(li, n_steps, len(lanes))), outs[li], launcher, stagger_s))
                   for li, ln in enumerate(lanes)]
            for t_ in ths:
                t_.start()
            for t_ in ths:
                t_.join()
            ox = xchg.pop("ordered", None)
            if ox is not None:
                ox.finish(failed=any(o[1] is not None for o in outs))      # the steps' exchanges belong to the timed region
        for o in outs:
            if o[1] is not None:
                raise o[1]
        stats = local_stats()
        last_out = outs[(n_steps - 1) % len(lanes)][0] if (len(lanes) > 1 and L > 1) else outs[0][0]
        if kids:
            torch.cuda.synchronize()
            for k, rep in enumerate(children.recv("DONE")):      # (a child replies after its own device work has finished)
                if rep.get("error"):
                    raise RuntimeError(f"lane child {k + 1}: {rep['error']}")
                for key in ("d2h_waits", "post_s", "post_after_fwd_s"):
                    stats[key] += rep[key]
                if (n_steps - 1) % L == k + 1:
                    last_out = tuple(rep["last"])
        return last_out, stats

    if my_lane:
        # ---- a lane child: warm up, report, then serve the rank's RUN commands until EXIT / end of input ----
        if os.environ.get("DEEPEMIA_BENCH_FAIL_LANE_CHILD"):       # (test hook: the rank must carry on with one lane)
            sys.exit(7)
        try:
            for i in range(max(1, args.warmup)):
                step(-1 - i, None, lanes[0])
                torch.cuda.synchronize()
            reply.write("READY {}\n")
            reply.flush()
            while True:
                cmd = sys.stdin.readline().split()
                if not cmd or cmd[0] == "EXIT":
                    break
                if cmd[0] != "RUN":
                    continue
                leg, n_steps, xchg["stagger_s"] = cmd[1], int(cmd[2]), float(cmd[3])
                try:
                    out_, st_ = run_steps(n_steps, leg)
                    torch.cuda.synchronize()
                    st_["last"] = list(out_)
                except BaseException as e:      # the rank raises it
                    st_ = {"error": f"{type(e).__name__}: {e}"}
                reply.write("DONE " + json.dumps(st_) + "\n")
                reply.flush()
        finally:
            if dist is not None:
                dist.destroy_process_group()
        return

    t_warm = None
    for ln in lanes:                          # W warm-up passes per lane (captures its graphs), one lane after the other
        for i in range(args.warmup):
            step(-1 - i, None, ln)
            torch.cuda.synchronize()
    lane_note = None
    if children is not None:
        try:
            children.recv("READY")            # the lane children have built their engines and warmed up meanwhile
        except RuntimeError as e:
            if world > 1:
                raise                         # (the lane groups of the other ranks would wait for this rank's child)
            # a lane child that cannot start (or dies while it warms up) must not take the measurement with it: the run goes on
            # with lane 0 alone and the line says so (`config.lanes` = 1, `config.lane_fallback`)
            print(f"bench.py: {e}; continuing with one lane", file=sys.stderr)
            children.close(kill=True)
            children, lane_note = None, str(e)
            args.lanes = 1
    if args.lanes > 1:
        tw0 = time.perf_counter()
        step(-1, None, lanes[0])
        torch.cuda.synchronize()
        t_warm = time.perf_counter() - tw0    # a pass ALONE on the GPU, forward and post-processing one after the other
    sync_all()
    # the start offset between lanes: with L forwards in flight a lane's forward takes about L single forwards, so the lanes sit
    # one single forward apart (a pass alone is the forward + ~1/4 of it for the post-processing)
    xchg["stagger_s"] = float(os.environ.get("DEEPEMIA_BENCH_STAGGER", "0.8")) * t_warm if (t_warm and args.lanes > 1) else 0.0      # (the factor: an A/B switch)
    eng.conv_events = None if (args.no_conv_events or args.graph) else []
    t0 = time.perf_counter()
    (det_total, rows_total), st_main = run_steps(args.steps)      # K complete passes
    sync_all()
    dt = time.perf_counter() - t0
    if TRACE is not None:
        for li_, i_, what, t_ in TRACE:
            if i_ >= 0:
                print(f"[trace] lane {li_} step {i_:3d} {what:10s} {1e3 * (t_ - t0):9.2f} ms", file=sys.stderr)
        TRACE.clear()
    d2h_waits_per_step = st_main["d2h_waits"] / max(args.steps, 1)
    post_wall_ms = st_main["post_s"] / max(args.steps, 1) * 1e3     # host wall time inside the post-processing of a step (incl. its waits)
    post_after_fwd_ms = st_main["post_after_fwd_s"] / max(args.steps, 1) * 1e3   # ... of which after the step's own forward had finished
    events, eng.conv_events = eng.conv_events or [], None

    def snapshot(res, t=0):
        """Tile t of a step's result as host data (dense masks, scores, classes, contour records)."""
        packed, scores, classes, recs = res[t]
        dense = pipe.ops.to_dense(packed, args.size) if packed is not None else np.zeros((0, args.size, args.size), dtype=bool)
        return np.array(dense, copy=True), [float(v) for v in scores], [int(c) for c in classes], recs

    # what the parity leg checks is the TIMED path's own result (graph replay in the default mode): taken here, before
    # anything else runs through the pipeline (total-tiles mode: step 0, the step whose tile 0 is synthetic tile 0)
    want_parity = rank == 0 and world == 1 and not args.no_cpu_baseline and not args.forward_only
    timed_snaps = [snapshot(last["res"], t) for t in range(len(last["res"]))] if (want_parity and "res" in last) else None
    # side leg: the same K passes with lane 0 ALONE (this process, the lane children idle): every line carries its own A/B of
    # the second pipeline, and the conv share below is taken against a one-lane step
    one_lane = None
    if args.lanes > 1 and not args.no_one_lane_leg:
        sync_all()
        to = time.perf_counter()
        run_steps(args.steps, "main", n_lanes=1)
        sync_all()
        dto = time.perf_counter() - to
        if dist is not None:
            tto = torch.tensor([dto], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tto, op=dist.ReduceOp.MAX)
            dto = float(tto.item())
        one_lane = {"value": world * args.batch * args.steps / dto, "ms_per_step": dto / args.steps * 1e3,
                    "note": "the same K passes, same process, lane 0 alone (run after the timed region)"}
    one_lane_step_ms = one_lane["ms_per_step"] if one_lane else (dt / args.steps * 1e3 if args.lanes == 1 else None)
    instrumented_s, eager_snap = None, None
    n_instr = 2
    if args.graph and not args.no_conv_events:
        # per-kernel HIP events cannot be taken inside a replayed graph: two eager, instrumented passes of the same path
        cur["xs"] = xs
        pipe.use_graphs = False
        last.pop("res", None)
        last["keep_every"] = True
        step(0)
        sync_all()
        last.pop("keep_every")
        if want_parity and "res" in last:
            eager_snap = snapshot(last["res"])      # same input as the timed snapshot (batch 0): eager launch vs graph replay
        eng.conv_events = []
        ti = time.perf_counter()
        for i in range(n_instr):
            step(i)
        sync_all()
        instrumented_s = time.perf_counter() - ti
        events, eng.conv_events = eng.conv_events, None
        pipe.use_graphs = True
    same_batch = None
    if len(xs) > 1 and not args.total_tiles and world == 1 and not args.no_h2d_leg:
        # side figure: the loop of rounds 1-3, the SAME resident batch every step (paste and pools see identical boxes)
        sync_all()
        ts = time.perf_counter()
        run_steps(args.steps, "same")
        sync_all()
        dts = time.perf_counter() - ts
        same_batch = {"value": args.batch * args.steps / dts, "ms_per_step": dts / args.steps * 1e3,
                      "note": "the same K passes over ONE resident batch repeated every step (the headline loop of rounds 1-3)"}
    h2d = None
    if world == 1 and not args.no_h2d_leg and not args.forward_only and not args.no_overlap:
        # second leg: the same K passes with every step's tiles UPLOADED from pinned host memory on a copy stream (two device
        # slots per lane; the forward waits for its upload, the upload of step i+1 runs under the forward of step i)
        launcher_of("h2d")           # (set-up outside the timed region; the lane children do theirs on their first h2d step)
        if children is not None:
            children.send("RUN h2d 0 0")
            children.recv("DONE")
        sync_all()
        th = time.perf_counter()
        run_steps(args.steps, "h2d")
        sync_all()
        dth = time.perf_counter() - th
        up_ms = [a.elapsed_time(b) for a, b in h2d_state["events"]]
        h2d = {"value_with_upload": args.batch * args.steps / dth, "ms_per_step_with_upload": dth / args.steps * 1e3,
               "h2d_ms_per_step": float(np.median(up_ms)), "bytes_per_step": int(x.numel()),
               "gbps": float(x.numel() / (np.median(up_ms) * 1e-3) / 1e9),
               "note": "same K passes, tiles uploaded from pinned host memory on a copy stream, overlapped with the previous "
                       "step's forward; reported beside `value` (inputs resident), never as `value`"}
    cur["xs"] = xs
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        all_conv_ms = sum(e[0].elapsed_time(e[1]) for e in events)
        # the dominant kernel = the conv kernel of the run's precision (f32x3: conv_igemm_split_kernel; the few layers
        # whose shape it does not take -- 15 / 11 / 2 output channels -- run on the exact-f32 kernel and are left out)
        # (f16x2 with --single-stages: the launches of the opted-in stages are the same kernel with one MFMA per product; they
        # carry the kind "f16" and are priced against the native fp16 peak)
        kinds = {args.precision} | ({"f16"} if args.precision == "f16x2" else set())
        dom = [e for e in events if e[3] in kinds]
        conv_ms = sum(e[0].elapsed_time(e[1]) for e in dom)
        conv_flops = sum(e[2] for e in dom)
        conv_bytes = sum(e[4] for e in dom)
        launches = len(dom)
        single_launches = sum(1 for e in dom if e[3] == "f16" and args.precision != "f16")
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        # the roof of the MIX of launches: every launch's algorithmic FLOPs at the dense peak of ITS arithmetic (FLOP-weighted
        # harmonic mean; = PEAK_TFLOPS[precision] when all launches compute alike)
        roof_s = sum(e[2] / (PEAK_TFLOPS[e[3]] * 1e12) for e in dom)
        peak = conv_flops / roof_s / 1e12 if roof_s > 0 else PEAK_TFLOPS[args.precision]
        mfmas_per_product = {"f16x2": 3, "f16x2r": 3, "bf16x2": 3, "f32x3": 6}
        executed_tflops = sum(e[2] * mfmas_per_product.get(e[3], 1) for e in dom) / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        # the same launches against the roof that bounds EACH of them: a launch cannot take less than its FLOPs at the MFMA
        # roof or its algorithmic bytes at the HBM roof (8 TB/s, MI355X_MICROARCH.md), whichever is longer
        attainable_ms = sum(max(e[2] / (PEAK_TFLOPS[e[3]] * 1e12), e[4] / 8.0e12) for e in dom) * 1e3
        hbm_bound = sum(1 for e in dom if e[4] / 8.0e12 > e[2] / (PEAK_TFLOPS[e[3]] * 1e12))
        # HBM bytes per conv launch from the PMC passes committed under profiles/ (collected with separate
        # `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this workload; FETCH_SIZE doubled per the gfx950 note)
        # a STORED value, not measured in this run: it is only reported when the stored pass was taken on this kernel
        traffic, traffic_source = None, None
        kernel_name = "conv_p32_kernel" if args.precision in ("f16x2", "f16") else ("conv_igemm_split_kernel" if args.precision in ("f16x2r", "f32x3", "bf16x2") else "conv_igemm_kernel")
        cands = sorted((ROOT / "profiles").glob(f"r*_conv_{args.precision}_b{args.batch}_pmc_traffic.json"))
        tf = cands[-1] if cands else ROOT / "profiles" / "none.json"          # the newest round's stored pass
        native = (args.min_size_test, args.max_size_test) != (800, 1333)
        if native:
            args.no_cpu_baseline = True        # the oracle sample below is the 800-pixel workload
        if tf.exists() and args.depth == 101 and args.size == 2048 and not native:
            rec = json.loads(tf.read_text())
            if rec.get("kernel") == kernel_name:
                traffic = rec.get("hbm_bytes_per_launch")
                traffic_source = f"stored PMC pass profiles/{tf.name} (kernel {rec.get('kernel')}, taken at {rec.get('head', '?')}); not re-measured by this run"
        line = {
            "metric": "EM tiles/s (2048x2048, R101-FPN)", "value": world * args.batch * args.steps / dt,
            "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": ("NON-PARITY single-plane fp16 arithmetic (flagged mode, parity bar not met -- see parity): " if args.precision == "f16" else "") +
                                   (f"FLAGGED opt-in, not the product default: stages {sorted(eng.single_stages)} on one MFMA per product: " if eng.single_stages else "") +
                                   ("NON-PARITY native-resolution mode, not configs[1]: " if native else
                                    (f"configs[4] (a job of {args.total_tiles} distinct tiles per GPU, {args.steps} steps): " if args.total_tiles else
                                     f"configs[1] ({len(xs)} distinct resident batches taken in turn): ")) +
                                   f"R{args.depth}-FPN, {args.size}x{args.size} synthetic EM tiles, "
                                   f"{args.batch} tiles per GPU per step; per tile: resize {args.min_size_test} -> backbone/FPN/RPN/ROI heads -> mask paste to "
                                   f"bit-packed {args.size}^2 masks" + ("" if args.forward_only else " -> class loop (fill holes, closing, overlap "
                                   "removal, component test, opening, greedy IoU dedup) -> cross-class dedup -> contour trace + 12 measurements" +
                                   ("" if args.no_csv_text else " -> measurement CSV text in memory (byte for byte csv.writer's, floats by the native demia_host_repr_rows)")) +
                                   f"; random-init Detectron2-layout weights, K=2, threshold {args.threshold}"
                                   + (f"; {args.lanes} pipelines in flight per GPU (steps dealt out round robin; lanes 1.. are "
                                      f"{'child processes of the rank' if proc_lanes else 'host threads'})" if args.lanes > 1 else "")
                                   + ("; all-gather of instance tables over ranks" if world > 1 and not args.forward_only else ""),
                       "tiles_per_step_per_gpu": args.batch, "instances_last_step_rank0": det_total,
                       "csv_rows_last_step_rank0": rows_total, "stage": "predictor only" if args.forward_only else "whole per-tile path",
                       "overlap": (not args.forward_only) and (not args.no_overlap), "hipgraph_forward": bool(args.graph),
                       "lanes": args.lanes, "lane_mode": (args.lane_mode if args.lanes > 1 else None),
                       **({"lane_fallback": lane_note} if lane_note else {}),
                       "post_d2h_waits_per_step": None if args.forward_only else d2h_waits_per_step,
                       "post_wall_ms_per_step": None if args.forward_only else post_wall_ms,
                       "post_after_forward_ms_per_step": None if args.forward_only else post_after_fwd_ms},
            "roofline": {"bound": "mfma", "kernel": ("conv_igemm_split_kernel (implicit-GEMM conv, f32 operands as 3 bf16 planes, 6 bf16 MFMAs "
                                                     "per product; peak = bf16 dense peak / 6)" if args.precision == "f32x3" else
                                                     "conv_p32_kernel (implicit-GEMM conv, both operands as 2 pre-scaled fp16 planes moved by LDS-DMA, "
                                                     "3 fp16 MFMAs per product: roof = fp16 dense peak / 3" +
                                                     (f"; {single_launches} of the {launches} timed launches (--single-stages) run ONE MFMA per product on the "
                                                      "high planes: roof = fp16 dense peak; `peak` = the FLOP-weighted roof of the mix)" if single_launches else ")")
                                                     if args.precision == "f16x2" else
                                                     "conv_p32_kernel in its flagged single-plane mode (fp16 operands, ONE fp16 MFMA per product, zero low "
                                                     "planes still moved: bytes as f16x2; peak = fp16 dense peak)" if args.precision == "f16" else
                                                     "conv_igemm_split_kernel (implicit-GEMM conv, f32 activations split into 2 scaled fp16 planes in "
                                                     "the K loop, 3 fp16 MFMAs per product; peak = fp16 dense peak / 3)" if args.precision == "f16x2r" else
                                                     "conv_igemm_kernel (implicit-GEMM conv)") + ", all tile configs",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "launches_per_step": launches // max(n_instr if instrumented_s else args.steps, 1),
                         "avg_launch_us": conv_ms * 1e3 / max(launches, 1),
                         "algorithmic_gflop_per_launch": conv_flops / max(launches, 1) / 1e9,
                         "algorithmic_bytes_per_launch": conv_bytes / max(launches, 1),
                         # conv time per step (from the instrumented single-lane passes) over a ONE-lane step of the same process (the
                         # `one_lane` leg; < 1 by construction), and -- a different thing -- over the period of the timed loop, in
                         # which L passes overlap (can pass 1 with L > 1); the eager figure beside them
                         "conv_share_of_one_lane_step": ((conv_ms / (n_instr if instrumented_s else args.steps)) / one_lane_step_ms
                                                         if one_lane_step_ms else None),
                         "all_conv_share_of_one_lane_step": ((all_conv_ms / (n_instr if instrumented_s else args.steps)) / one_lane_step_ms
                                                             if one_lane_step_ms else None),
                         "conv_alone_over_step_period": (conv_ms / (n_instr if instrumented_s else args.steps)) / (dt / args.steps * 1e3),
                         "share_of_eager_instrumented_step_time": conv_ms * 1e-3 / (instrumented_s or dt),
                         "measured_over": ("two instrumented eager steps after the timed region (the timed steps replay hipGraphs)" if instrumented_s
                                           else "the timed region"),
                         # executed MFMA work (every product counted with the MFMAs it costs) against the native 16-bit dense peak
                         "frac_of_native_peak": executed_tflops / 2500.0 if args.precision in ("f16x2", "f16x2r", "bf16x2", "f16", "f32x3") else None,
                         "single_plane_launches": single_launches,
                         # the same achieved (algorithmic, f32-equivalent) rate against the NATIVE 16-bit dense peak, i.e. without
                         # crediting the three MFMAs a product costs in this arithmetic
                         "algorithmic_frac_of_native_16bit_peak": achieved / 2500.0,
                         "frac_of_per_launch_roof": attainable_ms / conv_ms if conv_ms > 0 else None,
                         "per_launch_roof": f"sum over launches of max(FLOP / the launch's MFMA roof, algorithmic bytes / 8 TB/s) / measured time; "
                                            f"{hbm_bound} of {launches} launches are HBM-bound by that measure",
                         "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)"},
        }
        cands = sorted((ROOT / "profiles").glob(f"r*_conv_{args.precision}_sq.json"))
        sq = cands[-1] if cands else ROOT / "profiles" / "none.json"
        if h2d is not None:
            line["h2d"] = h2d
        if same_batch is not None:
            line["same_batch_every_step"] = same_batch
        if one_lane is not None:
            line["one_lane"] = one_lane
            line["value_over_one_lane"] = line["value"] / one_lane["value"]
        line["inputs"] = {"sha256_tile0": sha_tile0, "sha256_weights": synth.sha256_of_state(sd),
                          "distinct_resident_batches": len(xs),
                          "note": "tile 0 = synth.em_tile(0) (numpy PCG64, seed 1234); weights = synth.random_d2_state_dict(depth, 2, seed=0); "
                                  f"batch 0 = {n_np} numpy tiles + device-generated tiles, the other batches device-generated (seeded per tile)"}
        if sq.exists():
            rec = json.loads(sq.read_text())
            if rec.get("kernel") == kernel_name:
                line["roofline"]["mfma_busy_frac"] = rec.get("mfma_busy_frac")
                line["roofline"]["mfma_busy_source"] = f"stored SQ counter pass profiles/{sq.name}; not re-measured by this run"
        ok = True
        if want_parity:
            # parity is checked on every run before a throughput number is accepted -- tile 0 of a
            # TIMED step (snapshot taken right after the timed region) against the CPU path's result for the same tile
            from oracle import tile_parity as TP
            if args.parity_only:
                from deepemia_amd import synth as _synth
                refs = [TP.reference_tile(_synth.em_tile(t, args.size), sd, args.depth, args.threshold, CLASS_THRESHOLDS, SMALL_CLASSES)
                        for t in range(n_par_tiles)]
            else:
                line["cpu_baseline"], refs = cpu_baseline(args.depth, args.size, args.threshold, sd, args.cpu_tiles)
            # every tile the CPU sample went through is compared (tiles 0 .. n-1 of the parity step = synthetic tiles 0 .. n-1)
            pars = [TP.compare_tile(ref_, *snap_) for ref_, snap_ in zip(refs, timed_snaps)]
            dense, scores, classes, recs = timed_snaps[0]

            def worst(key, fn):
                vals = [p_[key] for p_ in pars if p_.get(key) is not None]
                return fn(vals) if vals else None
            line["parity"] = {"tiles_checked": len(pars), "ok": all(bool(p_["ok"]) for p_ in pars),
                              "mask_iou_min": worst("mask_iou_min", min), "tie_pixels_max": worst("tie_pixels_max", max),
                              "score_max_abs_err": worst("score_max_abs_err", max)}
            for k in ("csv_max_rel_err", "csv_max_rel_err_own_mask", "csv_max_rel_err_all"):
                line["parity"][k] = worst(k, max)
            for k in ("instances", "instances_ref", "masks_identical", "masks_with_tie_pixels", "csv_rows", "csv_rows_own_mask", "ellipse_rows_skipped"):
                line["parity"][k] = sum(int(p_[k]) for p_ in pars)
            line["parity"]["per_tile"] = [{k: p_.get(k) for k in ("ok", "instances", "instances_ref", "mask_iou_min", "masks_identical",
                                                                   "tie_pixels_max", "csv_max_rel_err", "csv_max_rel_err_own_mask",
                                                                   "csv_max_rel_err_all", "score_max_abs_err", "why") if k in p_} for p_ in pars]
            line["parity"]["checked"] = ((f"tiles 0..{len(pars) - 1} of timed step {parity_step} (the last one over batch 0 on lane 0)") +
                                         (" (hipGraph replay)" if args.graph else " (eager launches)") +
                                         ", snapshot taken before any other pass, vs oracle/tile_parity.py; bar: every mask IoU >= 0.999, CSV "
                                         "within 1e-4 relative on the instances whose mask equals the reference's bit for bit; the others "
                                         "differ by <= 8 threshold-tie pixels and their CSV rows are within 1e-4 of the oracle's measurement "
                                         "of the product's OWN mask (csv_max_rel_err_own_mask); csv_max_rel_err_all = vs the reference's masks")
            why = [f"tile {t}: {p_['why']}" for t, p_ in enumerate(pars) if "why" in p_]
            if why:
                line["parity"]["why"] = "; ".join(why)
            ok = bool(line["parity"]["ok"])
            if eager_snap is not None:
                # the eager launch sequence and its captured replay must give the same bits on the same input
                same = (eager_snap[0].shape == dense.shape and bool((eager_snap[0] == dense).all()) and eager_snap[1] == scores
                        and eager_snap[2] == classes)
                line["parity"]["eager_equals_replay"] = bool(same)
                ok = ok and same
                line["parity"]["ok"] = ok
            if not ok:
                line["value_rejected"] = line["value"]
                line["value"] = None          # a fast path whose results differ from the reference's is not measured
        elif world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], _ = cpu_baseline(args.depth, args.size, args.threshold, sd, args.cpu_tiles)
        if world == 1 and not args.no_cli_leg and not args.forward_only and not native and args.precision == "f16x2" and not args.single_stages \
                and (args.depth, args.size) == (101, 2048):
            # the drop-in CLI on its own workloads, as a subprocess -- with the GPU to itself: the lane children end and this process
            # hands its arenas, graphs and resident batches back first (the CLI's worker processes bring ~45 GiB each)
            try:
                if children is not None:
                    children.close()
                for ln in lanes:
                    ln.eng.release_cached_shapes()
                    ln.pipe.clear_cache()
                lanes.clear()
                xs.clear()
                h2d_state.clear()
                cur.clear()
                del x, pipe, eng
                import gc
                gc.collect()
                torch.cuda.empty_cache()
                line["cli"] = cli_leg(sd, args.depth, dev, args.cli_images, line["value"] or line.get("value_rejected") or 0.0)
            except Exception as e:          # a side leg never takes the headline down with it
                line["cli"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(line), flush=True)
        if not ok:
            sys.exit(3)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
