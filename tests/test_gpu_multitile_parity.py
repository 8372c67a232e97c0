"""The headline configuration (BASELINE.json configs[1]: R101-FPN, 2048 x 2048 tiles, threshold 0.3, K = 2) under the
oracle on EIGHT tiles, in the default ``f16x2`` arithmetic and -- as the control that separates "the fp16 split" from
"fp32 sums taken in another order" -- in the exact-f32 MFMA kernel (``--precision f32``).

What is asserted is what is true of an fp32-sized arithmetic difference against the fp32 CPU path (``oracle/maskrcnn_ref.py``
<-> reference ``src/functions/inference.py:1395-1403``), tile by tile:
 * the product's instances are a permutation of the oracle's (same class, box within 0.5 px);
 * two detections change places only where the ORACLE's own scores differ by <= 2e-6;
 * scores within 1e-4 of the matched instance;
 * every pixel on which a mask differs is a tie of the paste threshold in the oracle's own sampled probability
   (|p - 0.5| <= 3e-4, ``paste_masks(soft=True)``);
 * a mask may fall below IoU 0.999 (north_star's bar) only as DESIGN.md section 7 describes it: at most 12 differing pixels,
   every one of them a threshold tie, and at most 2 such masks among the 800 (a mask of a few thousand pixels with a
   handful of tie pixels on its border falls a hair below the bar in ANY fp32-sized arithmetic, the exact-f32 kernel included).
The per-tile records go to ``gpurun_out/multitile_parity_<precision>.json`` (copied to ``profiles/`` per round).

Plus, on the same eight tiles: the WHOLE per-tile path (class loop, dedup, contours, 12 measurements, CSV rows) against the
dense CPU pipeline (``oracle/tile_parity.py::compare_tile``, reference ``src/functions/inference.py:1395-1461, 2552-2677``,
``src/utils/measurements.py:114-233``) -- ~450 final instances instead of the 56 of tile 0 -- and the ``f32x3`` column of
DESIGN.md's table under the DEV build of the library, which this module loads in a child process of its own."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

THR, TILES = 0.3, 8


@pytest.fixture(scope="module")
def oracle_tiles(gpu_device):
    from deepemia_amd import synth
    from oracle import maskrcnn_ref

    torch.set_num_threads(min(16, torch.get_num_threads()))
    sd = synth.random_d2_state_dict(101, 2, seed=0)
    tiles = [synth.em_tile(i, 2048) for i in range(TILES)]
    refs = [maskrcnn_ref.predict(t, sd, 101, THR) for t in tiles]
    return sd, tiles, refs


def run_precision(sd, tiles, refs, precision, device, single_stages=None, tag=None):
    from deepemia_amd.engine import MaskRCNNEngine
    from deepemia_amd.predictor import Predictor
    from oracle import tile_parity as TP

    pred = Predictor(MaskRCNNEngine(sd, 101, 2, THR, device, precision, single_stages=single_stages))
    rows = []
    for t, r in zip(tiles, refs):
        inst = pred(t)["instances"].to("cpu")
        boxes = inst.pred_boxes if torch.is_tensor(inst.pred_boxes) else inst.pred_boxes.tensor
        rows.append(TP.compare_predictor(r, boxes, inst.scores, inst.pred_classes, inst.pred_masks))
    masks = sum(r["instances"] for r in rows)
    summary = dict(precision=precision, single_stages=sorted(pred.engine.single_stages), tiles=len(rows), masks=masks, masks_ge_0999=sum(r["masks_ge_0999"] for r in rows),
                   masks_identical=sum(r["masks_identical"] for r in rows),
                   tiles_in_oracle_order=sum(1 for r in rows if r["bijection"] and not r["moved_positions"]),
                   order_gap_max=max(r["order_gap_max"] for r in rows),
                   score_max_abs_err=max(r["score_max_abs_err"] or 0.0 for r in rows),
                   box_max_abs_err=max(r["box_max_abs_err"] or 0.0 for r in rows),
                   tie_dist_max=max(r["tie_dist_max"] for r in rows),
                   iou_min=min(r["iou_min"] if r["iou_min"] is not None else 0.0 for r in rows), per_tile=rows,
                   config=f"R101-FPN, synthetic 2048^2 tiles 0..{len(rows) - 1}, threshold {THR}, K=2, seeded random Detectron2-layout weights; "
                          "oracle = fp32 torch-CPU restatement (oracle/maskrcnn_ref.py)")
    del pred
    torch.cuda.empty_cache()
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/multitile_parity_{tag or precision}.json", "w") as f:
        json.dump(summary, f, indent=1)
    return summary


@pytest.mark.parametrize("precision", ["f16x2", "f16x2_mask_head_single_plane", "f32"])
def test_eight_headline_tiles_against_the_oracle(oracle_tiles, gpu_device, precision):
    """f16x2 = the product default (three MFMAs per product in every stage); the opt-in with the mask head on ONE MFMA per
    product (engine.MASK_HEAD_STAGES: holds this bar, not that of the soft-mask CLI cases -- DESIGN.md section 7); exact f32."""
    from deepemia_amd.engine import MASK_HEAD_STAGES
    sd, tiles, refs = oracle_tiles
    single = precision == "f16x2_mask_head_single_plane"
    s = run_precision(sd, tiles, refs, "f16x2" if single else precision, gpu_device, single_stages=MASK_HEAD_STAGES if single else None, tag=precision)
    print({k: v for k, v in s.items() if k != "per_tile"})
    for i, r in enumerate(s["per_tile"]):
        assert r["instances"] == r["instances_ref"] == 100, (i, r["instances"], r["instances_ref"])
        assert r["bijection"], (i, r.get("why"))
        assert r["order_gap_max"] <= 2e-6, (i, r["moved_positions"])
        assert r["score_max_abs_err"] <= 1e-4, (i, r["score_max_abs_err"])
        assert r["tie_dist_max"] <= 3e-4, (i, r["differing"])
    check_masks_below_the_bar(s)


def check_masks_below_the_bar(s, most=2, pixels=12):
    """What DESIGN.md section 7 claims of the masks under IoU 0.999: few (<= 2 of 800), small differences (<= 12 pixels), ties only
    (``tie_dist`` <= 3e-4 was asserted per tile) -- not a count that happens to hold today."""
    below = [(t, d) for t, r in enumerate(s["per_tile"]) for d in r["differing"] if d["iou"] < 0.999]
    assert len(below) <= most * max(1, s["masks"] // 800), below
    for t, d in below:
        assert d["pixels"] <= pixels and d["tie_dist"] <= 3e-4, (t, d)
    assert s["masks_ge_0999"] == s["masks"] - len(below)


def test_whole_tile_path_on_eight_headline_tiles(gpu_device):
    """``compare_tile`` on tiles 0..7: the same final instances (an order swap only between instances whose REFERENCE scores are
    within 2e-6), every mask at IoU >= 0.999 or a tie-pixel mask, every CSV number within 1e-4 on the bit-identical masks and
    within 1e-4 of the oracle's measurement of the product's own mask on the others.  The eight CPU references run in eight
    spawned processes (~25 s of single-threaded numpy / scipy each)."""
    from deepemia_amd import synth
    from deepemia_amd.engine import MaskRCNNEngine
    from deepemia_amd.functions.inference import InferencePipeline
    from deepemia_amd.predictor import Predictor
    from oracle import tile_parity as TP

    class_thr, small = {0: (0.3, 0.7), 1: (0.3, 0.5)}, {1}
    refs = TP.reference_tiles_parallel(range(TILES), 2048, 101, THR, class_thr, small, workers=8, threads=2)
    sd = synth.random_d2_state_dict(101, 2, seed=0)
    pipe = InferencePipeline([Predictor(MaskRCNNEngine(sd, 101, 2, THR, gpu_device, "f16x2"))], "eight", {}, {})
    pipe.forward_batch = TILES
    x = torch.from_numpy(np.stack([synth.em_tile(t, 2048) for t in range(TILES)])).to(gpu_device)
    out = pipe.process_tile_batch("eight", x, small, class_thr)
    rows, total, identical, csv_rows = [], 0, 0, 0
    for t in range(TILES):
        packed, scores, classes, recs = out[t]
        dense = pipe.ops.to_dense(packed, 2048)
        r = TP.compare_tile(refs[t], dense, scores, classes, recs, order_gap=2e-6)
        rows.append(r)
        assert r["ok"], (t, r)
        total += r["instances"]
        identical += r["masks_identical"]
        csv_rows += r["csv_rows"] + r["csv_rows_own_mask"]
    summary = dict(tiles=TILES, instances=total, masks_identical=identical, csv_rows_checked=csv_rows,
                   mask_iou_min=min(r["mask_iou_min"] for r in rows), tie_pixels_max=max(r["tie_pixels_max"] for r in rows),
                   csv_max_rel_err=max(r["csv_max_rel_err"] for r in rows), csv_max_rel_err_own_mask=max(r["csv_max_rel_err_own_mask"] for r in rows),
                   csv_max_rel_err_all=max(r["csv_max_rel_err_all"] for r in rows), score_max_abs_err=max(r["score_max_abs_err"] for r in rows),
                   moved=[(t, r["moved_positions"]) for t, r in enumerate(rows) if r.get("moved_positions")], per_tile=rows)
    print({k: v for k, v in summary.items() if k != "per_tile"})
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/multitile_whole_path_f16x2.json", "w") as f:
        json.dump(summary, f, indent=1)
    assert total > 300 and summary["mask_iou_min"] >= 0.999 and summary["csv_max_rel_err"] <= 1e-4 and summary["csv_max_rel_err_own_mask"] <= 1e-4


_F32X3_CHILD = r"""
import json, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import torch
import test_gpu_multitile_parity as T
from deepemia_amd import _lib, synth
from oracle import maskrcnn_ref
assert _lib.is_dev_build(), "the child must have loaded libdeepemia_hip_dev.so"
torch.set_num_threads(min(16, torch.get_num_threads()))
sd = synth.random_d2_state_dict(101, 2, seed=0)
tiles = [synth.em_tile(i, 2048) for i in range(T.TILES)]
refs = [maskrcnn_ref.predict(t, sd, 101, T.THR) for t in tiles]
s = T.run_precision(sd, tiles, refs, "f32x3", "cuda:0")
print("SUMMARY " + json.dumps({k: v for k, v in s.items() if k != "per_tile"}))
"""


def test_f32x3_column_under_the_dev_library(gpu_device):
    """The product library computes f16x2 / f16 / exact f32 only; the ``f32x3`` row of DESIGN.md section 7 needs the dev build
    (``make DEV=1`` -> ``libdeepemia_hip_dev.so``, built by ``__graft_entry__.build()`` beside the product library).  This test
    sets ``DEEPEMIA_DEV_LIB=1`` ITSELF for one child process, so the row is checked by whoever runs the suite."""
    root = Path(__file__).resolve().parent.parent
    if not (root / "deepemia_amd" / "csrc" / "libdeepemia_hip_dev.so").exists():
        pytest.fail("libdeepemia_hip_dev.so is missing: run __graft_entry__.build()")
    r = subprocess.run([sys.executable, "-c", _F32X3_CHILD, str(root)], cwd=str(root), env=dict(os.environ, DEEPEMIA_DEV_LIB="1"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("SUMMARY ")][-1]
    head = json.loads(line[len("SUMMARY "):])
    s = json.load(open(root / "gpurun_out" / "multitile_parity_f32x3.json"))
    print(head)
    for i, t in enumerate(s["per_tile"]):
        assert t["instances"] == t["instances_ref"] == 100 and t["bijection"], (i, t.get("why"))
        assert t["order_gap_max"] <= 2e-6 and t["score_max_abs_err"] <= 1e-4 and t["tie_dist_max"] <= 3e-4, (i, t["moved_positions"], t["differing"])
    check_masks_below_the_bar(s)
