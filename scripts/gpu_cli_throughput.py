"""Dev script: `main.py --task inference` on N synthetic 2048^2 images (R101, full-image pass + 9 tiles of 1024 with 12.5 % overlap per
image: 10 forwards of the 800-pixel network per image, containment + overlap rules, both CSVs): images/s and tile-forwards/s.

    python scripts/gpu_cli_throughput.py [n_images=16] [--profile]
"""
import csv, io, json, logging, os, re, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from PIL import Image
import test_gpu_pipeline_e2e as T
from deepemia_amd import synth
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 16
profile = "--profile" in sys.argv
root = Path(tempfile.mkdtemp())
ds_cfg = {"inference_overrides": {"confidence_mode": "manual",
                                  "class_specific_settings": {"class_0": {"confidence_threshold": 0.3, "iou_threshold": 0.6, "min_size": 25},
                                                              "class_1": {"confidence_threshold": 0.35, "iou_threshold": 0.5, "min_size": 5}},
                                  "tile_settings": {"tile_size": 1024, "overlap_ratio": 0.125, "upscale_factor": 1.0, "edge_filter_enabled": True},
                                  "spatial_constraints": {"enabled": True, "containment_rules": {1: 0}, "containment_threshold": 0.5,
                                                          "overlap_rules": {0: {"allow_overlap": False, "max_iou_threshold": 0.3}}}}}
cfgdir, split, sds, images = T._write_tree(root, [101], 0.5, 6.0, 0, 512, ds_cfg)
inf = root / "DATASET" / "INFERENCE"
for i in range(n):
    Image.fromarray(synth.em_tile(300 + i, 2048)[:, :, ::-1]).save(inf / f"em_{i}.png")
os.environ["DEEPEMIA_CONFIG_DIR"] = str(cfgdir); os.environ["DEEPEMIA_OFFLINE"] = "1"
os.chdir(root)
import main as cli
from deepemia_amd.utils.logger_utils import system_logger
buf = io.StringIO()
h = logging.StreamHandler(buf); h.setLevel(logging.INFO); system_logger.addHandler(h)
if profile:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
t0 = time.time()
rc = cli.main(["--task", "inference", "--dataset_name", T.DATASET, "--threshold", "0.3", "--no-gpu-check"])
dt = time.time() - t0
if profile:
    pr.disable()
rows = list(csv.reader(open(split / "measurements_results.csv")))
log = buf.getvalue()
m = re.search(r"Inference complete: (\d+)/(\d+) images, avg ([0-9.]+)s/image, (\d+) batched forwards", log)
tm = re.search(r"Inference task finished in ([0-9.]+)s", log)
loop_s = float(m.group(3)) * n if m else None
tiles_per_image = 9
rec = {"rc": rc, "images": n, "wall_s_incl_model_load": dt, "csv_rows": len(rows) - 1,
       "image_loop_s": loop_s, "image_loop_s_per_image": (loop_s / n if loop_s else None),
       "tile_forwards_per_image": 1 + tiles_per_image,
       "tile_forwards_per_s_image_loop": (n * (1 + tiles_per_image) / loop_s if loop_s else None),
       "task_s": float(tm.group(1)) if tm else None,
       "tile_forwards_per_s_whole_task": (n * (1 + tiles_per_image) / float(tm.group(1)) if tm else None),
       "batched_forwards": int(m.group(4)) if m else None}
print("CLI_THROUGHPUT " + json.dumps(rec))
if profile:
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(60); print(s.getvalue()[:12000])
