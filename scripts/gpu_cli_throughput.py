"""Dev script: `main.py --task inference` on N synthetic 2048^2 images (R101, tiles of 1024 with 10 % overlap): images/s."""
import csv, os, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from PIL import Image
import test_gpu_pipeline_e2e as T
from deepemia_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
t0 = time.time()
rc = cli.main(["--task", "inference", "--dataset_name", T.DATASET, "--threshold", "0.3", "--no-gpu-check"])
dt = time.time() - t0
pr.disable()
rows = list(csv.reader(open(split / "measurements_results.csv")))
print(f"rc={rc} {n} images of 2048^2: {dt:.2f} s wall ({n / dt:.2f} images/s incl. model load), {len(rows) - 1} CSV rows")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(40); print(s.getvalue()[:7000])
