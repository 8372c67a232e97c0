"""ORACLE (test infrastructure, never shipped): the whole per-tile path of the headline workload on the CPU, and the
comparison of a product result with it.

One "tile" of BASELINE.json configs[1] goes through: ``predictor(tile)`` (``oracle/maskrcnn_ref.py``, the Detectron2
restatement) -> per class ``single_model_class_pass`` (inference.py:1384-1461) -> ``deduplicate_masks_smart`` at 0.7
(inference.py:859) -> contours + ``calculate_measurements`` per instance (inference.py:1148-1230).  Used by
``tests/test_gpu_headline_parity.py`` and by ``bench.py``'s ``cpu_baseline`` leg, which times it and then checks the
timed GPU result against it (parity is checked on every run before a throughput is accepted).
"""
from __future__ import annotations

import time
from typing import Dict, Sequence, Tuple

import numpy as np

from . import maskrcnn_ref
from . import postproc_ref as P

MEASURES = ("major_axis_length", "minor_axis_length", "eccentricity", "Length", "Width", "CircularED", "Aspect_Ratio",
            "Circularity", "Chords", "Feret_diam", "Roundness", "Sphericity")      # order of demia_contour_measure's 12 values


def reference_tile(img: np.ndarray, sd, depth: int, thr: float, class_thresholds: Dict[int, Tuple[float, float]],
                   small_classes, um_pix: float = 1.0) -> dict:
    """The CPU path for one tile + its stage timings (seconds)."""
    t0 = time.perf_counter()
    out = maskrcnn_ref.predict(img, sd, depth, thr)
    t1 = time.perf_counter()
    pm, ps, pc = out["pred_masks"].numpy(), out["scores"].numpy(), out["pred_classes"].numpy()
    masks, scores, classes = [], [], []
    for cls, (conf, iou_thr) in class_thresholds.items():
        m, s, c = P.single_model_class_pass(pm, ps, pc, img.shape[:2], cls, small_classes, conf, iou_thr, None, True)
        masks += list(m)
        scores += list(s)
        classes += list(c)
    t2 = time.perf_counter()
    masks, scores, classes = P.deduplicate_masks_smart(masks, scores, classes, 0.7)
    t3 = time.perf_counter()
    rows = [P.measure_mask(np.asarray(m) > 0, um_pix) for m in masks]
    t4 = time.perf_counter()
    return {"masks": [np.asarray(m) > 0 for m in masks], "scores": [float(s) for s in scores], "classes": [int(c) for c in classes],
            "rows": rows, "raw": out,
            "seconds": {"predictor": t1 - t0, "class_passes": t2 - t1, "dedup": t3 - t2, "measurements": t4 - t3, "total": t4 - t0}}


def compare_tile(ref: dict, masks: np.ndarray, scores: Sequence[float], classes: Sequence[int], records) -> dict:
    """Product result of the same tile (dense bool masks [n, H, W], scores, classes, per instance the list of contour
    records with ``values`` = the 12 measurements) against :func:`reference_tile`.  Instances are compared in order: the
    path is deterministic, so the same instances come out in the same order or parity is lost.  Returns the numbers the
    bench line carries; ``ok`` = north_star's bar (mask IoU >= 0.999, every CSV number within 1e-4 relative)."""
    n_ref, n = len(ref["masks"]), int(len(scores))
    res = {"instances": n, "instances_ref": n_ref, "mask_iou_min": None, "csv_max_rel_err": None, "score_max_abs_err": None,
           "csv_rows": 0, "ellipse_rows_skipped": 0, "ok": False}
    if n != n_ref or list(int(c) for c in classes) != ref["classes"]:
        res["why"] = "instance count / classes differ"
        return res
    if n == 0:
        res.update(mask_iou_min=1.0, csv_max_rel_err=0.0, score_max_abs_err=0.0, ok=True)
        return res
    iou_min, err_max, rows, skipped = 1.0, 0.0, 0, 0
    for i in range(n):
        a, b = np.asarray(masks[i]) > 0, ref["masks"][i]
        union = int((a | b).sum())
        iou = 1.0 if union == 0 else int((a & b).sum()) / union
        iou_min = min(iou_min, iou)
        got, want = records[i], ref["rows"][i]
        # the product hands every contour back; the CSV (and the reference rows) keep those that pass the area gate
        h, w = b.shape
        min_area = max(5, h * w * 0.000005 * 0.05)
        got = [r for r in got if r["area"] >= min_area]
        if len(got) != len(want):
            res["why"] = f"instance {i}: {len(got)} CSV rows, reference {len(want)}"
            return res
        for g, r in zip(got, want):
            rows += 1
            for k, name in enumerate(MEASURES):
                if r["_ellipse_unstable"] and k < 3:
                    skipped += (k == 0)
                    continue          # fitEllipse on a degenerate contour: rounding of OpenCV's own SVD decides
                x, y = float(g["values"][k]), float(r[name])
                err_max = max(err_max, abs(x - y) / max(abs(y), 1e-12))
    smax = float(np.max(np.abs(np.asarray(scores, dtype=np.float64) - np.asarray(ref["scores"], dtype=np.float64))))
    res.update(mask_iou_min=iou_min, csv_max_rel_err=err_max, score_max_abs_err=smax, csv_rows=rows, ellipse_rows_skipped=skipped,
               ok=bool(iou_min >= 0.999 and err_max <= 1e-4 and smax <= 1e-4))
    return res
