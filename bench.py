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
    and does not scale with torch threads).  Returns (the ``cpu_baseline`` object, the reference result of tile 0)."""
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
    return base, refs[0]


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
                    "engine arena, hipGraphs, streams and host thread; steps are dealt out round robin.  A second forward fills the CUs "
                    "the first leaves idle in the partly filled last round of every launch and in its HBM-bound layers: whole path +4.2 %% / "
                    "+6.1 %% on two boxes at the default K = 20, five of five runs (round 4); three lanes are slower than one.  With N > 1 "
                    "the steps' exchanges are issued in step order by one thread per rank (OrderedExchange)")
    ap.add_argument("--no-overlap", action="store_true", help="do not overlap batch i+1's network with batch i's post-processing")
    ap.add_argument("--forward-only", action="store_true", help="time predictor(tile) only, without the per-tile post-processing")
    ap.add_argument("--post-priority", choices=["default", "high", "low"], default="default", help="(experiment) priority of the post-processing stream")
    ap.add_argument("--no-conv-events", action="store_true", help="(experiment) do not bracket the conv launches with HIP events")
    ap.add_argument("--eager", action="store_true", help="launch the forward kernel by kernel instead of replaying one hipGraph per batch "
                    "(the default replays: +20 %% tiles/s; per-kernel HIP events cannot be taken inside a replayed graph, so the roofline "
                    "numbers of the default mode come from two instrumented eager steps of the same path, run right after the timed region)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, ncpu // max(world, 1))))      # N ranks share the host: no oversubscription
    # rehearsal of the N > 1 path on a one-GPU box: DEEPEMIA_BENCH_BACKEND=gloo DEEPEMIA_BENCH_ONE_DEVICE=1 (all ranks on cuda:0)
    backend = os.environ.get("DEEPEMIA_BENCH_BACKEND", "nccl")
    one_dev = os.environ.get("DEEPEMIA_BENCH_ONE_DEVICE", "0") == "1"
    dev_index = 0 if (world == 1 or one_dev) else local_rank
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{dev_index}"))
        else:
            dist.init_process_group(backend)
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = f"cuda:{dev_index}"

    from deepemia_amd import parallel, synth
    from deepemia_amd.engine import MaskRCNNEngine
    from deepemia_amd.functions.inference import InferencePipeline, measurement_csv_text
    from deepemia_amd.predictor import Predictor

    sd = synth.random_d2_state_dict(args.depth, 2, seed=0)
    args.graph = not args.eager and args.precision in ("f16x2", "f16")
    args.lanes = resolve_lanes(args.lanes, world, overlap=not args.no_overlap, graph=bool(args.graph))
    import types
    lanes = []
    for li in range(args.lanes):
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
        lanes.append(types.SimpleNamespace(eng=e_, pipe=p_, last={}, net=None, post=None))
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
    # the last timed step that runs batch 0 (tile 0 = synthetic tile 0): the step the parity leg looks at
    parity_step = 0 if args.total_tiles else ((args.steps - 1) // len(xs)) * len(xs)

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
    net_stream, post_stream = lanes[0].net, lanes[0].post
    last = lanes[0].last
    import threading
    post_lock = threading.Lock()
    xchg = {}          # (lanes > 1) the start offset between lanes, set after the warm-up; N > 1: the ordered exchange of the running leg

    TRACE = [] if os.environ.get("DEEPEMIA_BENCH_TRACE") else None      # (diagnostic) host timestamps of every lane's steps

    def launch(i, ln=lanes[0]):
        if TRACE is not None:
            TRACE.append((lanes.index(ln), i, "launch", time.perf_counter()))
        with torch.cuda.stream(ln.net):
            return ln.pipe.forward_async(0, xs[i % len(xs)] if i >= 0 else x)

    def step(i, handle=None, ln=lanes[0]):
        """One pass of the hot path over this rank's batch of tiles."""
        if args.forward_only:
            with torch.cuda.stream(ln.net):
                raw = (ln.eng.forward_graphed if ln.pipe.use_graphs else ln.eng.forward)(xs[i % len(xs)] if i >= 0 else x)
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
            TRACE.append((lanes.index(ln), i, "fwd_done", tq0))
        try:
            # one lane post-processes at a time: two host loops at once share the interpreter lock and their kernels the same CUs
            # (48-68 ms per pass instead of 22), and taking turns keeps the lanes half a period apart
            with post_lock:
                if TRACE is not None:
                    TRACE.append((lanes.index(ln), i, "post_start", time.perf_counter()))
                return _post_after_forward(i, dets, ln)
        finally:
            if TRACE is not None:
                TRACE.append((lanes.index(ln), i, "post_done", time.perf_counter()))
            ln.last["post_after_fwd_s"] = ln.last.get("post_after_fwd_s", 0.0) + time.perf_counter() - tq0

    def _post_after_forward(i, dets, ln):
        if DIAG_POST == "wait":          # (diagnostic: the forward's tables fetched, no post-processing kernels at all)
            time.sleep(0.02)
            return 0, 0
        res = ln.pipe.process_tile_batch(f"step{i}", xs[i % len(xs)] if i >= 0 else x, SMALL_CLASSES, CLASS_THRESHOLDS, dets=dets)
        if i == parity_step or ln.last.get("keep_every"):
            # the step whose tile 0 is synthetic tile 0 (the parity check's reference); with plane pools the masks are views
            # that the lane's next step overwrites, and later steps may follow: keep tile 0's own copy
            ln.last["res"] = [(res[0][0].clone() if (res[0][0] is not None and (i != args.steps - 1 or args.lanes > 1)) else res[0][0],) + tuple(res[0][1:])] + list(res[1:])
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

    def run_lane(ln, idxs, out, launcher=launch):
        """The lane's share of the timed steps, software-pipelined: the forward of its next batch is enqueued before the host
        starts the post-processing of its current one, so the MFMA-bound network hides the latency-bound mask work."""
        try:
            torch.cuda.set_device(dev_index)
            li = lanes.index(ln)
            if li and xchg.get("stagger_s"):
                # lanes start one after the other, a lane's share of the period apart: two forwards launched together run the same
                # layers side by side (and finish, post-process and drain together); half a period apart one is in its MFMA-bound
                # layers while the other is in its HBM-bound ones.  The GPU is not idle meanwhile: the earlier lanes are running.
                time.sleep(li * xchg["stagger_s"])
            if args.forward_only or args.no_overlap:
                for i in idxs:
                    out[0] = step(i, None, ln)
                return
            handle = launcher(idxs[0], ln) if idxs else None
            for k, i in enumerate(idxs):
                nxt = launcher(idxs[k + 1], ln) if k + 1 < len(idxs) else None
                out[0] = step(i, handle, ln)
                handle = nxt
        except BaseException as e:      # re-raised by the main thread
            out[1] = e

    def run_steps(n_steps, launcher=launch):
        """n_steps passes dealt out to the lanes round robin (step i -> lane i % L), one host thread per lane."""
        outs = [[(0, 0), None] for _ in lanes]
        if len(lanes) == 1:
            run_lane(lanes[0], list(range(n_steps)), outs[0], launcher)
        else:
            if dist is not None and not args.forward_only:
                xchg["ordered"] = OrderedExchange(dev_index, n_steps, parallel.all_gather_instance_tables)
            ths = [threading.Thread(target=run_lane, args=(ln, list(range(li, n_steps, len(lanes))), outs[li], launcher))
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
        return outs[(n_steps - 1) % len(lanes)][0]      # (instances, rows) of the last step

    det_total = 0
    waits0 = None
    t_warm = None
    for ln in lanes:                          # W warm-up passes per lane (captures its graphs), one lane after the other
        for i in range(args.warmup):
            tw0 = time.perf_counter()
            step(-1 - i, None, ln)
            torch.cuda.synchronize()
            t_warm = time.perf_counter() - tw0          # a pass alone, forward and post-processing one after the other
    sync_all()
    # the start offset between lanes: with L forwards in flight a lane's forward takes about L single forwards, so the lanes sit
    # one single forward apart (a warm-up pass is the forward + ~1/4 of it for the post-processing)
    xchg["stagger_s"] = 0.8 * t_warm if (t_warm and len(lanes) > 1) else 0.0
    eng.conv_events = None if (args.no_conv_events or args.graph) else []
    waits0 = sum(ln.pipe.d2h_waits for ln in lanes)
    for ln in lanes:
        ln.last["post_s"] = ln.last["post_after_fwd_s"] = 0.0
    t0 = time.perf_counter()
    det_total, rows_total = run_steps(args.steps)      # K complete passes
    sync_all()
    dt = time.perf_counter() - t0
    if TRACE is not None:
        for li_, i_, what, t_ in TRACE:
            if i_ >= 0:
                print(f"[trace] lane {li_} step {i_:3d} {what:10s} {1e3 * (t_ - t0):9.2f} ms", file=sys.stderr)
        TRACE.clear()
    d2h_waits_per_step = (sum(ln.pipe.d2h_waits for ln in lanes) - waits0) / max(args.steps, 1)
    post_wall_ms = sum(ln.last.get("post_s", 0.0) for ln in lanes) / max(args.steps, 1) * 1e3     # host wall time inside the post-processing of a step (incl. its waits)
    post_after_fwd_ms = sum(ln.last.get("post_after_fwd_s", 0.0) for ln in lanes) / max(args.steps, 1) * 1e3   # ... of which after the step's own forward had finished
    events, eng.conv_events = eng.conv_events or [], None

    def snapshot(res):
        """Tile 0 of a step's result as host data (dense masks, scores, classes, contour records)."""
        packed, scores, classes, recs = res[0]
        dense = pipe.ops.to_dense(packed, args.size) if packed is not None else np.zeros((0, args.size, args.size), dtype=bool)
        return np.array(dense, copy=True), [float(v) for v in scores], [int(c) for c in classes], recs

    # what the parity leg checks is the TIMED path's own result (graph replay in the default mode): taken here, before
    # anything else runs through the pipeline (total-tiles mode: step 0, the step whose tile 0 is synthetic tile 0)
    want_parity = rank == 0 and world == 1 and not args.no_cpu_baseline and not args.forward_only
    par_last = lanes[parity_step % len(lanes)].last          # the lane that ran the parity step
    timed_snap = snapshot(par_last["res"]) if (want_parity and "res" in par_last) else None
    instrumented_s, eager_snap = None, None
    n_instr = 2
    if args.graph and not args.no_conv_events:
        # per-kernel HIP events cannot be taken inside a replayed graph: two eager, instrumented passes of the same path
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
        xs_all, xs = xs, [x]
        sync_all()
        ts = time.perf_counter()
        run_steps(args.steps)
        sync_all()
        dts = time.perf_counter() - ts
        xs = xs_all
        same_batch = {"value": args.batch * args.steps / dts, "ms_per_step": dts / args.steps * 1e3,
                      "note": "the same K passes over ONE resident batch repeated every step (the headline loop of rounds 1-3)"}
    h2d = None
    if world == 1 and not args.no_h2d_leg and not args.forward_only and not args.no_overlap:
        # second leg: the same K passes with every step's tiles UPLOADED from pinned host memory on a copy stream (two device
        # slots; the forward waits for its upload, the upload of step i+1 runs under the forward of step i)
        pinned = [t_.cpu().pin_memory() for t_ in xs]
        copy_stream = torch.cuda.Stream(device=dev)
        h2d_events = []
        for ln in lanes:
            ln.up_slots = [torch.empty_like(x), torch.empty_like(x)]
            ln.up_free = [None, None]
            ln.up_n = 0

        def launch_up(i, ln=lanes[0]):
            sl = ln.up_n % 2
            ln.up_n += 1
            with torch.cuda.stream(copy_stream):
                if ln.up_free[sl] is not None:
                    copy_stream.wait_event(ln.up_free[sl])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(copy_stream)
                ln.up_slots[sl].copy_(pinned[i % len(pinned)], non_blocking=True)
                e1.record(copy_stream)
            h2d_events.append((e0, e1))
            with torch.cuda.stream(ln.net):
                ln.net.wait_event(e1)
                hd = ln.pipe.forward_async(0, ln.up_slots[sl])
            ln.up_free[sl] = hd[1]
            return hd

        sync_all()
        th = time.perf_counter()
        run_steps(args.steps, launcher=launch_up)
        sync_all()
        dth = time.perf_counter() - th
        up_ms = [a.elapsed_time(b) for a, b in h2d_events]
        h2d = {"value_with_upload": args.batch * args.steps / dth, "ms_per_step_with_upload": dth / args.steps * 1e3,
               "h2d_ms_per_step": float(np.median(up_ms)), "bytes_per_step": int(x.numel()),
               "gbps": float(x.numel() / (np.median(up_ms) * 1e-3) / 1e9),
               "note": "same K passes, tiles uploaded from pinned host memory on a copy stream, overlapped with the previous "
                       "step's forward; reported beside `value` (inputs resident), never as `value`"}
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
                                   + (f"; {args.lanes} pipelines in flight per GPU (steps dealt out round robin)" if args.lanes > 1 else "")
                                   + ("; all-gather of instance tables over ranks" if world > 1 and not args.forward_only else ""),
                       "tiles_per_step_per_gpu": args.batch, "instances_last_step_rank0": det_total,
                       "csv_rows_last_step_rank0": rows_total, "stage": "predictor only" if args.forward_only else "whole per-tile path",
                       "overlap": (not args.forward_only) and (not args.no_overlap), "hipgraph_forward": bool(args.graph),
                       "lanes": args.lanes,
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
                         # conv time per step (from the instrumented single-lane passes) over the TIMED step (with two lanes in flight two
                         # passes overlap, so this can pass 1); the eager figure beside it
                         "share_of_step_time": (conv_ms / (n_instr if instrumented_s else args.steps)) / (dt / args.steps * 1e3),
                         "all_conv_share_of_step_time": (all_conv_ms / (n_instr if instrumented_s else args.steps)) / (dt / args.steps * 1e3),
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
                ref0 = TP.reference_tile(_synth.em_tile(0, args.size), sd, args.depth, args.threshold, CLASS_THRESHOLDS, SMALL_CLASSES)
            else:
                line["cpu_baseline"], ref0 = cpu_baseline(args.depth, args.size, args.threshold, sd, args.cpu_tiles)
            dense, scores, classes, recs = timed_snap
            par = TP.compare_tile(ref0, dense, scores, classes, recs)
            line["parity"] = {k: par[k] for k in ("mask_iou_min", "csv_max_rel_err", "csv_max_rel_err_own_mask", "csv_max_rel_err_all",
                                                  "score_max_abs_err", "instances", "instances_ref", "masks_identical",
                                                  "masks_with_tie_pixels", "tie_pixels_max", "csv_rows", "csv_rows_own_mask",
                                                  "ellipse_rows_skipped", "ok")}
            line["parity"]["checked"] = ((f"tile 0 of timed step {parity_step} (the last one over batch 0)") +
                                         (" (hipGraph replay)" if args.graph else " (eager launches)") +
                                         ", snapshot taken before any other pass, vs oracle/tile_parity.py; bar: every mask IoU >= 0.999, CSV "
                                         "within 1e-4 relative on the instances whose mask equals the reference's bit for bit; the others "
                                         "differ by <= 8 threshold-tie pixels and their CSV rows are within 1e-4 of the oracle's measurement "
                                         "of the product's OWN mask (csv_max_rel_err_own_mask); csv_max_rel_err_all = vs the reference's masks")
            if "why" in par:
                line["parity"]["why"] = par["why"]
            ok = bool(par["ok"])
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
        print(json.dumps(line), flush=True)
        if not ok:
            sys.exit(3)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
