"""ORACLE-SIDE STUDY (test infrastructure, never shipped): would fp32 Winograd F(2x2, 3x3) for the 3x3 stride-1 layers hold
the parity bar an exact-f32 GPU kernel holds?  (decide on the CPU before building a kernel.)

For each headline tile the plain oracle (``maskrcnn_ref.predict``, direct sums) is the reference and the SAME oracle with
``WINOGRAD_STAGES`` set is the candidate; ``tile_parity.compare_predictor`` reports instance sets, order swaps, score error,
masks at IoU >= 0.999 and the tie distance of every differing pixel -- the columns of DESIGN.md section 7's table.

    python -m oracle.winograd_probe [n_tiles=8] [stages=backbone,fpn,rpn,mask | mask | fpn,rpn,mask ...]
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from deepemia_amd import synth
from oracle import maskrcnn_ref as R
from oracle import tile_parity as TP


def main() -> None:
    n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    stage_sets = [s.split(",") for s in sys.argv[2:]] or [["mask"], ["fpn", "rpn", "mask"], ["backbone", "fpn", "rpn", "mask"],
                                                         ["f16x2", "mask"], ["f16x2", "fpn", "rpn", "mask"], ["f16x2", "backbone", "fpn", "rpn", "mask"]]
    depth, thr, size = 101, 0.3, 2048
    sd = synth.random_d2_state_dict(depth, 2, seed=0)
    # unit check of the transform itself: one random layer against the direct sum in float64
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 64, 37, 50, generator=g)
    w = torch.randn(48, 64, 3, 3, generator=g) * 0.05
    ref64 = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    e_w = float((R.conv3x3_winograd(x, w).double() - ref64).abs().max() / ref64.abs().max())
    e_d = float((torch.nn.functional.conv2d(x, w, padding=1).double() - ref64).abs().max() / ref64.abs().max())
    out = {"unit_check_rel_err_vs_f64": {"winograd_f32": e_w, "direct_f32": e_d}, "tiles": n_tiles, "runs": []}
    refs = []
    for t in range(n_tiles):
        R.WINOGRAD_STAGES = set()
        refs.append(R.predict(synth.em_tile(t, size), sd, depth, thr))
    for stages in stage_sets:
        t0 = time.perf_counter()
        rec = {"stages": stages, "per_tile": []}
        for t in range(n_tiles):
            R.WINOGRAD_STAGES = set(stages) - {"f16x2"}
            R.WINOGRAD_F16X2 = "f16x2" in stages        # transformed operands as two fp16 planes, three products (the product path's arithmetic)
            cand = R.predict(synth.em_tile(t, size), sd, depth, thr)
            R.WINOGRAD_STAGES, R.WINOGRAD_F16X2 = set(), False
            c = TP.compare_predictor(refs[t], cand["pred_boxes"], cand["scores"], cand["pred_classes"], cand["pred_masks"])
            c["in_order"] = bool(c["bijection"] and not c["moved_positions"])
            rec["per_tile"].append(c)
        pt = rec["per_tile"]
        rec["summary"] = {"tiles_with_the_plain_oracles_instance_set": sum(bool(c["bijection"]) for c in pt),
                          "tiles_in_order": sum(c["in_order"] for c in pt),
                          "order_gap_max": max(c["order_gap_max"] for c in pt),
                          "score_max_abs_err": max((c["score_max_abs_err"] or 0.0) for c in pt),
                          "masks_ge_0999": sum(c["masks_ge_0999"] for c in pt), "masks": sum(c["instances_ref"] for c in pt),
                          "masks_identical": sum(c["masks_identical"] for c in pt),
                          "tie_dist_max": max(c["tie_dist_max"] for c in pt),
                          "worst_mask": min(((d["iou"], t, d["position"], d["pixels"], d["area"]) for t, c in enumerate(pt) for d in c["differing"]),
                                            default=None),
                          "ok_by_the_8_tile_tests_rule": all(c["ok"] and c["bijection"] for c in pt)
                                                          and sum(c["masks_ge_0999"] for c in pt) >= int(np.ceil(0.998 * sum(c["instances_ref"] for c in pt))),
                          "seconds": time.perf_counter() - t0}
        print(json.dumps({"stages": stages, **rec["summary"]}), flush=True)
        out["runs"].append(rec)
    R.WINOGRAD_STAGES = set()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/winograd_probe.json", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
