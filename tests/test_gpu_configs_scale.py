"""BASELINE.json configs that the oracle is too slow for at full size, checked through size-independent properties on the
GPU, plus the N > 1 launch path of ``bench.py`` rehearsed on the one-GPU box.

* configs[2]: ONE 8192 x 8192 image -> full-image pass + 16 tiles of 2048^2 (tile_size 2048, overlap 0, upscale 1) through
  ``main.py --task inference``: finishes, is deterministic, every instance lies inside the frame, RLE rows == instances,
  CSV rows >= instances; the same pipeline is compared with the oracle at 512^2 in ``test_gpu_pipeline_e2e.py``.
* plain ``python bench.py --gpus 2`` -- the form of the driver's command -- starts its two ranks ITSELF (a child
  ``torch.distributed.run``; both ranks on cuda:0, gloo for the exchange because RCCL refuses two ranks on one device), with the
  default TWO lanes per rank, lanes 1.. being child processes of the rank whose exchanges run in a process group per lane: one JSON
  line, n_gpus 2, weak scaling, a positive whole-job rate; the same under round 4's thread lanes (``--lane-mode threads``: the
  configuration that hung in round 3 until the steps' exchanges were issued in step order by one thread per rank).  No RCCL run has
  happened anywhere in this repository's records -- the 8-GPU scaling run is the driver's.
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


@pytest.mark.parametrize("lane_mode", ["processes", "threads"])
def test_bench_two_ranks_on_one_device_and_total_tiles_mode(gpu_device, tmp_path, lane_mode):
    env = dict(os.environ, DEEPEMIA_BENCH_ONE_DEVICE="1", DEEPEMIA_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    # no torch.distributed.run here: `--gpus 2` makes bench.py start its ranks itself (default --lanes 2: two ranks x two lanes)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "1", "--batch", "4",
           "--no-cpu-baseline", "--lane-mode", lane_mode]
    r = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]          # stdout of the launcher = the one JSON line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 6 and line["value"] > 0
    assert line["config"]["lanes"] == 2 and line["config"]["lane_mode"] == lane_mode
    assert "all-gather of instance tables" in line["config"]["workload"]
    assert line["roofline"]["bound"] == "mfma" and 0 < line["roofline"]["frac"] < 1
    assert line["one_lane"]["value"] > 0 and abs(line["value_over_one_lane"] - line["value"] / line["one_lane"]["value"]) < 1e-9
    # a rank count that is not the one running is refused, not reported
    bad = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--no-cpu-baseline"], cwd=str(ROOT),
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode == 2 and "refusing" in bad.stderr and not bad.stdout.strip()


def test_bench_default_line_carries_its_contract(gpu_device):
    """The default bench command at a small batch: one JSON line with the driver's fields, two lanes, two distinct resident batches,
    the input hashes (tile 0 is byte-reproducible: numpy PCG64), the side figure of the repeated batch, the upload leg, and the
    parity leg on tile 0 of the last timed step over batch 0."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "6", "--warmup", "1", "--batch", "8", "--parity-only", "--cli-images", "8"],
                       cwd=str(ROOT), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "parity", "h2d", "same_batch_every_step", "inputs", "one_lane", "value_over_one_lane"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 6 and line["dtype"] == "f16x2" and line["vs_baseline"] is None and line["value"] > 0
    assert line["config"]["lanes"] == 2 and line["config"]["lane_mode"] == "processes" and "2 distinct resident batches" in line["config"]["workload"]
    assert line["one_lane"]["value"] > 0 and line["one_lane"]["ms_per_step"] > 0      # the same K steps with lane 0 alone, same process
    assert line["inputs"]["distinct_resident_batches"] == 2
    assert line["inputs"]["sha256_tile0"] == "f9c43b982cea3c5afa2d7573af7540bda988ebd59f9f5c099760f70176a33d87"
    assert len(line["inputs"]["sha256_weights"]) == 64
    assert line["parity"]["ok"] and line["parity"]["eager_equals_replay"] and "timed step 4" in line["parity"]["checked"]
    assert line["parity"]["tiles_checked"] == 1 and len(line["parity"]["per_tile"]) == 1      # --parity-only: tile 0 (the default run: 0..2)
    rf = line["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["single_plane_launches"] == 0
    assert line["config"]["post_d2h_waits_per_step"] <= 3.5        # three per step; the first batch of a lane copies its contour points separately
    # the drop-in CLI as a subprocess: a folder of 2048^2 images (two processes on the GPU by default) and one 8192^2 image
    cli = line["cli"]
    assert cli["folder"]["default"]["rc"] == 0 and cli["folder"]["one_process"]["rc"] == 0 and cli["one_8192_image"]["rc"] == 0, cli
    assert cli["folder"]["default"]["processes_on_the_gpu"] == 2 and cli["folder"]["one_process"]["processes_on_the_gpu"] == 1
    assert cli["folder"]["default"]["csv_rows"] == cli["folder"]["one_process"]["csv_rows"] > 100
    assert cli["folder"]["default"]["tile_forwards_per_s"] > 0 and cli["one_8192_image"]["csv_rows"] > 100
    # conv time of one pass over a ONE-lane step of the same process: a share
    assert 0.3 < rf["conv_share_of_one_lane_step"] < 1.0 and rf["conv_alone_over_step_period"] > 0


def test_bench_carries_on_with_one_lane_when_the_lane_child_dies(gpu_device):
    """A lane child that cannot start (here: made to exit before its warm-up) must not take the measurement with it: the rank
    continues with lane 0 alone and the line says so."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "4", "--warmup", "1", "--batch", "4", "--no-cpu-baseline", "--no-h2d-leg", "--no-cli-leg"],
                       cwd=str(ROOT), env=dict(os.environ, DEEPEMIA_BENCH_FAIL_LANE_CHILD="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["config"]["lanes"] == 1 and "lane child 1 ended" in line["config"]["lane_fallback"] and line["value"] > 0
    assert "one_lane" not in line and "continuing with one lane" in r.stderr


def test_config4_job_of_256_distinct_tiles_at_size(gpu_device):
    """BASELINE configs[4] at its stated size on one GPU: a job of 256 DISTINCT 2048^2 tiles (16 steps of 16; step 0 holds
    the numpy tiles, the rest are generated on the device), the whole per-tile path on every tile, the parity leg on tile 0
    of the timed job (graph replay) against the CPU path, and the eager launch sequence against its replay."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--total-tiles", "256", "--batch", "16", "--warmup", "1", "--parity-only", "--no-cli-leg"],
                       cwd=str(ROOT), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["steps"] == 16 and "configs[4]" in line["config"]["workload"] and "256 distinct tiles" in line["config"]["workload"]
    assert line["value"] > 0 and line["config"]["instances_last_step_rank0"] > 0
    par = line["parity"]
    assert par["ok"] and par["eager_equals_replay"] and par["mask_iou_min"] >= 0.999
    assert par["csv_max_rel_err"] <= 1e-4 and par["csv_max_rel_err_own_mask"] <= 1e-4
    assert "hipGraph replay" in par["checked"]
    assert line["h2d"]["value_with_upload"] > 0 and line["h2d"]["h2d_ms_per_step"] > 0
    assert 0.3 < line["roofline"]["conv_share_of_one_lane_step"] < 1.0


_RCCL_SCRIPT = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["DEEPEMIA_ROOT"])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from deepemia_amd import parallel as PL
from deepemia_amd.maskset import MaskOps
ops = MaskOps("cuda:0")
h = w = 512
ops.set_frame_width(w)
g = np.random.default_rng(5)
for step, n in enumerate((37, 30, 90)):
    masks = np.zeros((n, h, w), dtype=bool)
    for i in range(n):
        y0, x0 = g.integers(0, h - 40, 2)
        masks[i, y0:y0 + g.integers(4, 40), x0:x0 + g.integers(4, 40)] = True
    masks[3] = False                                        # an empty mask travels as a header row without payload
    packed = ops.from_dense(masks)
    area, bbox = ops.area_bbox(packed)
    scores = g.uniform(0.3, 1.0, n) * 0.6
    classes = g.integers(0, 2, n)
    units = np.sort(g.integers(0, 9, n))[::-1]              # descending: the merge has to permute the payload
    hdr, pay = PL.encode_instance_table(packed, scores, classes, units, bbox.cpu().numpy(), area.cpu().numpy())
    assert hdr.is_cuda and pay.is_cuda
    before = dict(PL.stats)
    gt = PL.all_gather_instance_tables(hdr, pay)        # world 1, backend nccl: goes through RCCL on the device
    assert gt.header.is_cuda and gt.payload.is_cuda and PL.stats["exchanges"] == before["exchanges"] + 1
    got, gs, gc, gu = PL.decode_instance_table(gt.header, gt.payload, h, w, "cuda:0", host_header=gt.host_header, offsets=gt.offsets)
    order = np.argsort(units, kind="stable")
    assert gu == [int(u) for u in units[order]] and gc == [int(c) for c in classes[order]]
    assert gs == [float(s) for s in scores[order]]
    assert bool((ops.to_dense(got, w) == masks[order]).all())
    if step == 1:
        assert PL.stats["host_syncs"] - before["host_syncs"] == 1 and PL.stats["size_exchanges"] == before["size_exchanges"]
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_instance_tables_go_through_rccl_on_the_device(gpu_device):
    """The exchange of the multi-GPU path on the backend it will use: `nccl` (= RCCL) with world size 1 on the one GPU of
    this box -- device tensors through ``all_gather_into_tensor``, the agreed-capacity protocol, the device crop / unpack
    kernels and the payload permutation.  (More than one GPU is not reachable from the build pool: no scaling number.)"""
    env = dict(os.environ, DEEPEMIA_ROOT=str(ROOT), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT], cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stderr[-3000:] + r.stdout[-1000:]
