#!/usr/bin/env python3
"""``main.py --task inference`` drop-in for deepEMIA (reference ``main.py:241-342,456-528``).

Same flags and defaults as the reference CLI.  Only the ``inference`` task is implemented (the hot
path this build accelerates); ``prepare`` / ``train`` / ``evaluate`` / ``setup`` are reported as out
of scope.  Google Cloud Storage is optional: the reference shells out to ``gsutil`` unconditionally
(``main.py:383-398,473``) and ``--download`` / ``--upload`` cannot be switched off there; here the
GCS steps run only when ``gsutil`` exists and ``DEEPEMIA_OFFLINE`` is not set, so the CLI also runs
on a box without network.  Inference inputs are deleted after the run only if they were downloaded.
"""
from __future__ import annotations

import argparse
import glob
import logging
import os
import shutil
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="deepEMIA hot path on MI355X", formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--task", type=str, required=True, choices=["prepare", "train", "evaluate", "inference", "setup"])
    p.add_argument("--dataset_name", type=str, required=False)
    p.add_argument("--threshold", type=float, default=0.65)
    p.add_argument("--dataset_format", type=str, default="json", choices=["json", "coco"])
    p.add_argument("--visualize", action="store_true", default=False)
    p.add_argument("--download", action="store_true", default=True)
    p.add_argument("--upload", action="store_true", default=True)
    p.add_argument("--id", dest="draw_id", action="store_true")
    p.set_defaults(draw_id=False)
    p.add_argument("--rcnn", type=str, default="101", choices=["50", "101", "combo"])
    p.add_argument("--augment", action="store_true")
    p.add_argument("--optimize", action="store_true")
    p.add_argument("--n-trials", type=int, default=10)
    p.add_argument("--verbosity", type=str, default="info", choices=["debug", "info", "warning", "error"])
    p.add_argument("--no-gpu-check", action="store_true")
    p.add_argument("--draw-scalebar", action="store_true", default=False)
    return p


def gcs_available() -> bool:
    return not os.environ.get("DEEPEMIA_OFFLINE") and shutil.which("gsutil") is not None


def worker_processes(args) -> int:
    """How many processes share ONE GPU for this run (``DEEPEMIA_WORKERS``: a number, or ``auto`` = the default).  The image
    loop's post-processing is host-bound (class loops, dedups, writers: ~35 ms of interpreted host work per 2048^2 image against
    ~15 ms of network), so a folder of images goes faster with the images dealt out to two processes -- each with its own
    interpreter, engine and hipGraphs, exactly as with one process per GPU on a multi-GPU node (`run_inference` shards by image).
    ``auto``: when the run is local (no GCS download pending), 4 for a folder of at least 48 images, 3 from 24, 2 from 8, else 1
    (2048^2 images, round 5: 28.1 / 20.9 / 18.5 ms per image with 1 / 2 / 3 processes on 64 images, 19.0 / 16.7 / 24.3 ms with
    3 / 4 / 5 on 96 -- five is too many; byte-identical CSVs throughout)."""
    want = os.environ.get("DEEPEMIA_WORKERS", "auto").strip().lower()
    if want != "auto":
        try:
            return max(1, min(int(want), 6))
        except ValueError:
            return 1
    if args.task != "inference" or not args.dataset_name or (gcs_available() and args.download):
        return 1
    try:
        from deepemia_amd.functions.inference import get_image_folder_path, is_image_file
        n = sum(1 for f in os.listdir(get_image_folder_path()) if is_image_file(f))
    except Exception:
        return 1
    want = 4 if n >= 48 else (3 if n >= 24 else (2 if n >= 8 else 1))
    free = free_vram_gib()
    if free is not None:
        # every process brings its own arena (~26 GiB of activations for a 48-forward batch) and graphs: ~45 GiB each
        want = max(1, min(want, int(free // 45)))
    return want


def free_vram_gib():
    """Free device memory of the first GPU in GiB, read from sysfs (no HIP call: the launcher must not initialise the GPU);
    None when the files are not there."""
    try:
        cards = sorted(Path("/sys/class/drm").glob("card*/device/mem_info_vram_total"))
        if not cards:
            return None
        dev = cards[0].parent
        total = int((dev / "mem_info_vram_total").read_text())
        used = int((dev / "mem_info_vram_used").read_text())
        return (total - used) / 2 ** 30
    except (OSError, ValueError):
        return None


def launch_workers(workers: int, argv) -> int:
    """Start ``workers`` ranks of this CLI on ONE GPU as a child ``torch.distributed.run`` (gloo between them: RCCL refuses two
    ranks on one device; the only traffic is the rows rank 0 collects at the end) and return its exit code.  This process has
    not touched the GPU and does not: it only waits."""
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, DEEPEMIA_ONE_DEVICE="1", DEEPEMIA_DIST_BACKEND="gloo", DEEPEMIA_WORKERS="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={workers}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    proc = subprocess.Popen(cmd, env=env)
    try:
        return proc.wait()
    except BaseException:
        proc.terminate()
        try:
            proc.wait(timeout=30)
        except subprocess.TimeoutExpired:
            proc.kill()
        raise


def main(argv=None) -> int:
    args = build_parser().parse_args(argv)
    from deepemia_amd.utils.config import get_config
    from deepemia_amd.utils.gpu_check import check_gpu_availability, log_device_info
    from deepemia_amd.utils.logger_utils import set_console_log_level, system_logger

    set_console_log_level({"debug": logging.DEBUG, "info": logging.INFO, "warning": logging.WARNING,
                           "error": logging.ERROR}[args.verbosity])
    if args.task != "inference":
        system_logger.error(f"task '{args.task}' is outside the scope of this build (inference hot path only)")
        return 2
    if not args.dataset_name:
        system_logger.error("--dataset_name is required for --task inference")
        return 2
    if "WORLD_SIZE" not in os.environ:
        workers = worker_processes(args)
        if workers > 1:
            return launch_workers(workers, sys.argv[1:] if argv is None else list(argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dist = None
    if world > 1:
        # one process per GPU (torchrun) -- or DEEPEMIA_WORKERS processes on ONE GPU (`launch_workers`): a folder of images is
        # sharded by image, a single large image by tile (run_inference); rank 0 writes the outputs
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local_rank = 0 if os.environ.get("DEEPEMIA_ONE_DEVICE", "0") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local_rank)
        dist.init_process_group(os.environ.get("DEEPEMIA_DIST_BACKEND", "nccl"), device_id=torch.device(f"cuda:{local_rank}"))
    config = get_config()
    if not args.no_gpu_check:
        log_device_info()
    check_gpu_availability(require_gpu=True, interactive=False)
    split_dir = Path(config["paths"]["split_dir"]).expanduser().resolve()
    local_root = Path(config["paths"].get("local_dataset_root", "~")).expanduser().resolve()
    category_json = Path(config["paths"]["category_json"]).expanduser().resolve()
    bucket = config.get("bucket")
    online = gcs_available() and bool(bucket)
    downloaded = False

    def barrier():
        if dist is not None:
            dist.barrier()

    def all_ranks_ok(ok: bool) -> bool:
        """Rank 0's verdict on its host-side section, known to every rank: a failure there must end the whole job with a
        non-zero code instead of leaving the other ranks in a barrier until the RCCL timeout."""
        if dist is None:
            return ok
        import torch
        on_dev = dist.get_backend() == "nccl"
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=f"cuda:{local_rank}" if on_dev else "cpu")
        dist.broadcast(flag, src=0)
        return bool(int(flag.item()))

    # Host side effects (download, clean-up, upload, removing the inputs) belong to rank 0 alone; the other ranks wait at a
    # barrier until the inputs are in place and again until rank 0 has written every output that is uploaded or removed.
    prep_ok = True
    if rank != 0:
        pass
    elif online and args.download:
        try:
            subprocess.run(["gsutil", "cp", f"gs://{bucket}/dataset_info.json", str(category_json)], check=True)
            inf_dir = local_root / "DATASET" / "INFERENCE"
            if inf_dir.exists():
                shutil.rmtree(inf_dir)
            (local_root / "DATASET").mkdir(parents=True, exist_ok=True)
            subprocess.run(["gsutil", "-m", "cp", "-r", f"gs://{bucket}/DATASET/INFERENCE", str(local_root / "DATASET")], check=True)
            downloaded = True
        except (subprocess.CalledProcessError, OSError) as e:
            system_logger.error(f"GCS download failed: {e}")
            prep_ok = False
    else:
        system_logger.info("GCS disabled (no gsutil / DEEPEMIA_OFFLINE): using local dataset_info.json and DATASET/INFERENCE")
    if rank == 0 and prep_ok:
        for pattern in ("*.png", "*.csv", "*.jpg"):     # reference main.py:462-468
            for f in glob.glob(pattern):
                try:
                    os.remove(f)
                except OSError:
                    pass
    if not all_ranks_ok(prep_ok):               # (a status broadcast, not a bare barrier: every rank leaves together)
        if dist is not None:
            dist.destroy_process_group()
        return 1
    from deepemia_amd.functions.inference import OutputWriteError, run_inference

    t0 = time.perf_counter()
    run_ok = True
    try:
        run_inference(args.dataset_name, str(split_dir), visualize=args.visualize, threshold=args.threshold, draw_id=args.draw_id,
                      dataset_format=args.dataset_format, draw_scalebar=args.draw_scalebar)
        system_logger.info(f"Inference task finished in {time.perf_counter() - t0:.2f}s; results in {split_dir}")
    except OutputWriteError:
        # rank 0's output writers, AFTER the last collective of the image loop: tell the other ranks through the all-reduce
        # below instead of leaving them at a barrier.  (A failure of one rank's share of an image is agreed on inside the
        # image's own exchange -- every rank skips that image together, inference.py::PeerImageFailure.  Anything else -- set-up,
        # a collective itself -- is NOT caught: the process ends non-zero and torchrun tears the job down, because a rank that
        # carried on would meet its peers at the wrong collective.)
        system_logger.error("run_inference failed while writing the outputs", exc_info=True)
        run_ok = False
    if dist is not None:
        import torch
        on_dev = dist.get_backend() == "nccl"
        flag = torch.tensor([1 if run_ok else 0], dtype=torch.int32, device=f"cuda:{local_rank}" if on_dev else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)      # also the barrier: rank 0 has written the CSVs / overlays
        if not int(flag.item()):
            dist.destroy_process_group()
            return 1
    if rank == 0 and online and args.upload:
        try:
            stamp = time.strftime("%Y%m%d_%H%M%S")
            subprocess.run(["gsutil", "-m", "cp", str(split_dir / "*.csv"), f"gs://{bucket}/Archive/{stamp}_{args.dataset_name}/"],
                           check=False)
        except OSError as e:
            system_logger.warning(f"GCS upload skipped: {e}")
    if rank == 0 and downloaded:
        shutil.rmtree(local_root / "DATASET" / "INFERENCE", ignore_errors=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
