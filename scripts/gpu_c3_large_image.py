"""Dev script: BASELINE configs[2] shape -- ONE 8192 x 8192 image -> 16 tiles of 2048 (overlap 0, upscale 1), R101,
through `main.py --task inference` (full-image pass + tile pass + dedup + constraints + CSV).  Prints wall time and rows."""
import csv, os, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from PIL import Image
Image.MAX_IMAGE_PIXELS = None
import test_gpu_pipeline_e2e as T
from deepemia_amd import synth

size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
root = Path(tempfile.mkdtemp())
spatial = {"enabled": True, "containment_rules": {1: 0}, "containment_threshold": 0.5,
           "overlap_rules": {0: {"allow_overlap": False, "max_iou_threshold": 0.3}, 1: {"allow_overlap": False, "max_iou_threshold": 0.5}}}
ds_cfg = {"inference_overrides": {"confidence_mode": "manual",
                                  "class_specific_settings": {"class_0": {"confidence_threshold": 0.3, "iou_threshold": 0.6, "min_size": 25},
                                                              "class_1": {"confidence_threshold": 0.35, "iou_threshold": 0.5, "min_size": 5}},
                                  "tile_settings": {"tile_size": 2048, "overlap_ratio": 0.0, "upscale_factor": 1.0, "edge_filter_enabled": True},
                                  "spatial_constraints": spatial}}
cfgdir, split, sds, images = T._write_tree(root, [101], 0.5, 6.0, 0, 512, ds_cfg)
inf = root / "DATASET" / "INFERENCE"
tiles = [[synth.em_tile(100 + 4 * r + c, 2048) for c in range(size // 2048)] for r in range(size // 2048)]
img = np.concatenate([np.concatenate(row, axis=1) for row in tiles], axis=0)
Image.fromarray(img[:, :, ::-1]).save(inf / "big.tif")
os.environ["DEEPEMIA_CONFIG_DIR"] = str(cfgdir); os.environ["DEEPEMIA_OFFLINE"] = "1"
os.chdir(root)
import main as cli
t0 = time.time()
rc = cli.main(["--task", "inference", "--dataset_name", T.DATASET, "--threshold", "0.3", "--no-gpu-check"])
dt = time.time() - t0
rows = list(csv.reader(open(split / "measurements_results.csv")))
print(f"rc={rc} {size}x{size} image: {dt:.1f} s wall, {len(rows) - 1} CSV rows, peak GPU mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
