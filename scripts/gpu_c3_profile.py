"""Dev script: configs[2] shape (ONE 8192^2 image, full-image pass + 16 tiles of 2048) through the CLI in-process under cProfile,
twice (the second run has its graphs and arenas)."""
import cProfile, io, os, pstats, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from PIL import Image
Image.MAX_IMAGE_PIXELS = None
import test_gpu_pipeline_e2e as T
from deepemia_amd import synth
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
t16 = synth.em_tiles_device(range(100, 116), 2048, "cuda:0")[..., 0].cpu().numpy()
big = np.concatenate([np.concatenate(list(t16[4 * r:4 * r + 4]), axis=1) for r in range(4)], axis=0)
for k in range(3):
    Image.fromarray(big).save(inf / f"big{k}.png", compress_level=1)
os.environ["DEEPEMIA_CONFIG_DIR"] = str(cfgdir); os.environ["DEEPEMIA_OFFLINE"] = "1"; os.environ["DEEPEMIA_WORKERS"] = "1"
os.chdir(root)
import main as cli
pr = cProfile.Profile(); pr.enable()
t0 = time.time()
rc = cli.main(["--task", "inference", "--dataset_name", T.DATASET, "--threshold", "0.3", "--no-gpu-check"])
dt = time.time() - t0
pr.disable()
print(f"rc={rc} three 8192^2 images: {dt:.2f} s wall, peak GPU mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(55); print(s.getvalue()[:11000])
