"""Dev script / bench side leg: `main.py --task inference` as a user runs it (a subprocess) on N synthetic 2048^2 images (R101,
full-image pass + 9 tiles of 1024 with 12.5 % overlap per image: 10 forwards of the 800-pixel network per image, containment +
overlap rules, both CSVs): seconds per image of the image loop and tile-forwards/s, for 1 .. W worker processes on the one GPU.

    python scripts/gpu_cli_throughput.py [n_images=32] [workers=1,2,3] [--profile] [--solid] [--rocprof]

--solid: the headline weights (mask predictor bias + 2: solid masks, large regions) instead of the CLI parity cases' soft masks;
--rocprof: one more run with one process under `rocprofv3 --kernel-trace --stats` (tables under gpurun_out/cli_rocprof/).
"""
import csv, hashlib, json, os, re, subprocess, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from PIL import Image
import test_gpu_pipeline_e2e as T
from deepemia_amd import synth

TILES_PER_IMAGE = 9


def make_tree(n):
    root = Path(tempfile.mkdtemp())
    ds_cfg = {"inference_overrides": {"confidence_mode": "manual",
                                      "class_specific_settings": {"class_0": {"confidence_threshold": 0.3, "iou_threshold": 0.6, "min_size": 25},
                                                                  "class_1": {"confidence_threshold": 0.35, "iou_threshold": 0.5, "min_size": 5}},
                                      "tile_settings": {"tile_size": 1024, "overlap_ratio": 0.125, "upscale_factor": 1.0, "edge_filter_enabled": True},
                                      "spatial_constraints": {"enabled": True, "containment_rules": {1: 0}, "containment_threshold": 0.5,
                                                              "overlap_rules": {0: {"allow_overlap": False, "max_iou_threshold": 0.3}}}}}
    cfgdir, split, sds, images = T._write_tree(root, [101], *((2.0, 1.0) if "--solid" in sys.argv else (0.5, 6.0)), 0, 512, ds_cfg)
    inf = root / "DATASET" / "INFERENCE"
    for i in range(n):
        Image.fromarray(synth.em_tile(300 + i, 2048)[:, :, ::-1]).save(inf / f"em_{i}.png")
    return root, cfgdir, split


def run_cli(root, cfgdir, split, workers, n, profile=False, rocprof=False):
    env = dict(os.environ, DEEPEMIA_CONFIG_DIR=str(cfgdir), DEEPEMIA_OFFLINE="1", DEEPEMIA_WORKERS=str(workers), DEEPEMIA_LOG_DIR=str(root))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable] + (["-m", "cProfile", "-o", str(root / "cli.prof")] if profile else []) + \
          [str(ROOT / "main.py"), "--task", "inference", "--dataset_name", T.DATASET, "--threshold", "0.3", "--no-gpu-check"]
    if rocprof:
        out = ROOT / "gpurun_out" / "cli_rocprof"
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", str(out), "-o", "cli", "--", "python3"] + cmd[1:]
        env["TMPDIR"] = "/tmp"
    t0 = time.time()
    r = subprocess.run(cmd, cwd=str(root), env=env, capture_output=True, text=True, timeout=1200)
    dt = time.time() - t0
    log = r.stderr + r.stdout
    loops = [(int(a), float(b)) for a, b in re.findall(r"Inference complete: (\d+)/\d+ images, avg ([0-9.]+)s/image", log)]
    loop_s = max((a * b for a, b in loops), default=None)          # the slowest rank's image loop
    tm = re.search(r"Inference task finished in ([0-9.]+)s", log)
    rows = list(csv.reader(open(split / "measurements_results.csv"))) if (split / "measurements_results.csv").exists() else []
    # (rows sorted: the folder's listing order differs from one temporary directory to the next, the rows themselves must not)
    sha = hashlib.sha256(b"".join(b"".join(sorted(open(split / f, "rb").readlines())) for f in ("measurements_results.csv", "R50_flip_results.csv"))).hexdigest() if rows else None
    return {"rc": r.returncode, "workers": workers, "images": n, "wall_s_incl_start_up": dt, "csv_rows": max(len(rows) - 1, 0),
            "image_loop_s": loop_s, "image_loop_ms_per_image": (1e3 * loop_s / n if loop_s else None),
            "tile_forwards_per_image": 1 + TILES_PER_IMAGE,
            "tile_forwards_per_s_image_loop": (n * (1 + TILES_PER_IMAGE) / loop_s if loop_s else None),
            "task_s": float(tm.group(1)) if tm else None, "outputs_sha256": sha,
            "stderr_tail": "" if r.returncode == 0 else log[-1500:]}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]) if args else 32
    ws = [int(w) for w in (args[1].split(",") if len(args) > 1 else ["1", "2", "3"])]
    root, cfgdir, split = make_tree(n)
    masks = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--masks=")]
    recs = []
    for mask in (masks[0] if masks else [""]):           # --masks=,mod:32:28,mod:8:7 : DEEPEMIA_NET_CU_MASK values ("" = none)
        if mask.startswith("env:"):                     # env:NAME=VALUE -- any A/B switch of the CLI
            k_, v_ = mask[4:].split("=", 1)
            os.environ[k_] = v_
        elif mask:
            os.environ["DEEPEMIA_NET_CU_MASK"] = mask
        else:
            os.environ.pop("DEEPEMIA_NET_CU_MASK", None)
        for w in ws:
            rec = run_cli(root, cfgdir, split, w, n)
            rec["net_cu_mask"] = mask
            recs.append(rec)
    os.environ.pop("DEEPEMIA_NET_CU_MASK", None)
    for rec in recs:
        print("CLI_THROUGHPUT " + json.dumps(rec), flush=True)
    print("CLI_SAME_OUTPUTS", len({r["outputs_sha256"] for r in recs}) == 1)
    if "--rocprof" in sys.argv:
        print("CLI_ROCPROF " + json.dumps(run_cli(root, cfgdir, split, 1, n, rocprof=True)), flush=True)
    if "--profile" in sys.argv:
        import pstats
        run_cli(root, cfgdir, split, 1, n, profile=True)
        pstats.Stats(str(root / "cli.prof")).sort_stats("cumulative").print_stats(45)
