"""BASELINE.json configs that the oracle is too slow for at full size, checked through size-independent properties on the
GPU, plus the N > 1 launch path of ``bench.py`` rehearsed on the one-GPU box.

* configs[2]: ONE 8192 x 8192 image -> full-image pass + 16 tiles of 2048^2 (tile_size 2048, overlap 0, upscale 1) through
  ``main.py --task inference``: finishes, is deterministic, every instance lies inside the frame, RLE rows == instances,
  CSV rows >= instances; the same pipeline is compared with the oracle at 512^2 in ``test_gpu_pipeline_e2e.py``.
* ``bench.py --gpus 2`` under ``torch.distributed.run`` (both ranks on cuda:0, gloo for the exchange because RCCL refuses
  two ranks on one device): one JSON line, n_gpus 2, weak scaling, a positive whole-job rate.  No RCCL run has happened
  anywhere in this repository's records -- the 8-GPU scaling run is the driver's.
* configs[4] shape: ``bench.py --total-tiles 32`` walks distinct tiles and still passes its own parity check."""
import csv
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import yaml
from PIL import Image

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
DATASET = "bigimage"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_config2_one_8192_image_through_the_cli(tmp_path, monkeypatch, gpu_device):
    import main as cli
    from deepemia_amd import synth
    from deepemia_amd.utils import config as C

    Image.MAX_IMAGE_PIXELS = None
    cfgdir = tmp_path / "cfg"
    (cfgdir / "datasets").mkdir(parents=True)
    split = tmp_path / "split_dir"
    base = {"bucket": None,
            "paths": {"split_dir": str(split), "category_json": str(tmp_path / "dataset_info.json"), "local_dataset_root": str(tmp_path)},
            "inference_settings": {"confidence_mode": "manual", "ensemble_settings": {"enabled": False},
                                   "spatial_constraints": {"default": {"enabled": False}}},
            "l4_performance_optimizations": {"enable_parallel_mask_processing": True}}
    (cfgdir / "config.yaml").write_text(yaml.safe_dump(base, sort_keys=False))
    ds = {"inference_overrides": {"confidence_mode": "manual",
                                  "class_specific_settings": {"class_0": {"confidence_threshold": 0.3, "iou_threshold": 0.7},
                                                              "class_1": {"confidence_threshold": 0.3, "iou_threshold": 0.5}},
                                  "tile_settings": {"tile_size": 2048, "overlap_ratio": 0.0, "upscale_factor": 1.0, "edge_filter_enabled": True}}}
    (cfgdir / "datasets" / f"{DATASET}.yaml").write_text(yaml.safe_dump(ds, sort_keys=False))
    (tmp_path / "dataset_info.json").write_text(json.dumps({DATASET: ["imgs", "labels", ["pore", "throat"]]}))
    sd = synth.random_d2_state_dict(101, 2, seed=0)
    mdir = split / DATASET / "rcnn_r101"
    mdir.mkdir(parents=True)
    synth.save_d2_checkpoint(str(mdir / "model_final_r101.pth"), sd)
    inf = tmp_path / "DATASET" / "INFERENCE"
    inf.mkdir(parents=True)
    big = np.zeros((8192, 8192, 3), dtype=np.uint8)
    for ty in range(4):
        for tx in range(4):
            big[ty * 2048:(ty + 1) * 2048, tx * 2048:(tx + 1) * 2048] = synth.em_tile(ty * 4 + tx, 2048)
    Image.fromarray(big[:, :, 0]).save(inf / "big.png")          # 8-bit gray PNG: the CLI expands it to BGR
    monkeypatch.setenv("DEEPEMIA_CONFIG_DIR", str(cfgdir))
    monkeypatch.setenv("DEEPEMIA_OFFLINE", "1")
    monkeypatch.chdir(tmp_path)
    outs = []
    for _ in range(2):
        C.reset_cache()
        assert cli.main(["--task", "inference", "--dataset_name", DATASET, "--threshold", "0.3", "--no-gpu-check"]) == 0
        rows = list(csv.reader(open(split / "measurements_results.csv")))
        rle = list(csv.reader(open(split / "R50_flip_results.csv")))
        outs.append((rows, rle))
    C.reset_cache()
    rows, rle = outs[0]
    assert outs[1] == outs[0], "two runs over the same 8192^2 image differ"
    inst = {r[0] for r in rows[1:]}
    # every CSV instance has an RLE row; an instance whose contours all fail the area gate (inference.py:1175-1190) has none in the CSV
    assert len(rle) - 1 >= len(inst) > 300 and len(rle) - 1 - len(inst) <= 0.02 * len(inst), (len(rle) - 1, len(inst))
    assert all(np.isfinite(float(r[c])) for r in rows[1:] for c in range(3, 15))
    # every run of every mask lies inside the 8192 x 8192 frame (column-major 1-based starts)
    for _, runs in rle[1:50]:
        v = [int(t) for t in runs.split()]
        assert v and min(v[0::2]) >= 1 and max(a + b - 1 for a, b in zip(v[0::2], v[1::2])) <= 8192 * 8192


def test_bench_two_ranks_on_one_device_and_total_tiles_mode(gpu_device, tmp_path):
    env = dict(os.environ, DEEPEMIA_BENCH_ONE_DEVICE="1", DEEPEMIA_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 2 and line["value"] > 0
    assert "all-gather of instance tables" in line["config"]["workload"]
    assert line["roofline"]["bound"] == "mfma" and 0 < line["roofline"]["frac"] < 1
    # configs[4] shape on one rank: distinct tiles per step, parity checked against the CPU path on tile 0
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--total-tiles", "16", "--batch", "8", "--warmup", "1"], cwd=str(ROOT),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["steps"] == 2 and "configs[4]" in line["config"]["workload"] and line["parity"]["ok"] and line["value"] > 0
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["tiles_timed"] >= 3 and line["cpu_baseline"]["cpu_model"]
