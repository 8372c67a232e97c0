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


def _reference_tile_worker(job):
    """(spawned process) one tile through :func:`reference_tile`; the dense masks come back bit-packed, ``raw`` stays behind."""
    t, size, depth, k, seed, thr, class_thresholds, small_classes, threads = job
    import torch

    from deepemia_amd import synth
    torch.set_num_threads(threads)
    sd = synth.random_d2_state_dict(depth, k, seed=seed)
    ref = reference_tile(synth.em_tile(t, size), sd, depth, thr, class_thresholds, small_classes)
    ref.pop("raw")
    ref["masks"] = [np.packbits(m, axis=1) for m in ref["masks"]]
    ref["width"] = size
    return t, ref


def reference_tiles_parallel(tiles: Sequence[int], size: int, depth: int, thr: float, class_thresholds, small_classes,
                             workers: int = 8, threads: int = 2, k: int = 2, seed: int = 0) -> Dict[int, dict]:
    """:func:`reference_tile` for several synthetic tiles at once, one SPAWNED process per tile (fresh interpreters: the caller
    may hold a GPU context): the dense numpy / scipy post-processing is single-threaded and takes ~25 s per 2048^2 tile, so
    eight tiles cost about what one does.  Every worker rebuilds the seeded weights itself (``synth.random_d2_state_dict``)."""
    import multiprocessing as mp
    jobs = [(int(t), size, depth, k, seed, thr, dict(class_thresholds), set(small_classes), threads) for t in tiles]
    with mp.get_context("spawn").Pool(min(workers, len(jobs))) as pool:
        out = dict(pool.map(_reference_tile_worker, jobs, chunksize=1))
    for ref in out.values():
        w = ref.pop("width")
        ref["masks"] = [np.unpackbits(m, axis=1)[:, :w].astype(bool) for m in ref["masks"]]
    return out


def match_near_tie_swaps(ref: dict, masks, classes: Sequence[int], window: int = 3):
    """The permutation that maps product position i to the reference instance it IS when near-tied detections came out in
    another order: same class, the highest mask IoU among the reference positions within ``window`` of i.  Returns (perm,
    [(position, reference position, |reference score gap|)]) or (None, why)."""
    n = len(classes)
    perm, used = [], set()
    for i in range(n):
        a = np.asarray(masks[i]) > 0
        best, bj = -1.0, None
        for j in range(max(0, i - window), min(n, i + window + 1)):
            if j in used or ref["classes"][j] != int(classes[i]):
                continue
            b = ref["masks"][j]
            union = int((a | b).sum())
            iou = 1.0 if union == 0 else int((a & b).sum()) / union
            if iou > best:
                best, bj = iou, j
        if bj is None or best < 0.9:
            return None, f"instance {i}: no reference instance of its class within {window} positions (best IoU {best:.3f})"
        used.add(bj)
        perm.append(bj)
    moved = [(i, j, abs(ref["scores"][i] - ref["scores"][j])) for i, j in enumerate(perm) if i != j]
    return perm, moved


def compare_tile(ref: dict, masks: np.ndarray, scores: Sequence[float], classes: Sequence[int], records, order_gap: float = 0.0) -> dict:
    """Product result of the same tile (dense bool masks [n, H, W], scores, classes, per instance the list of contour
    records with ``values`` = the 12 measurements) against :func:`reference_tile`.  Instances are compared in order: the
    path is deterministic, so the same instances come out in the same order or parity is lost.

    north_star's bar is mask IoU >= 0.999 and every CSV number within 1e-4 relative.  The two interact: the paste
    thresholds a bilinear sample at 0.5, so an fp32-rounding-sized difference in a mask probability can flip ONE pixel,
    and the reference then truncates ``minAreaRect``'s corners to integers (measurements.py:140) -- a one-pixel change of
    a small mask moves Length / Width / Feret by a whole pixel.  So the CSV bound is asserted on the instances whose masks
    are IDENTICAL to the reference's, and the others must be tie pixels: IoU >= 0.999, at most 8 differing pixels, and at most
    a tenth of the instances.  (With ~30 000 border pixels per tile and probabilities that agree to ~1e-5, a handful of raw
    pixels per tile sit within rounding of the threshold -- tests/test_gpu_headline_parity.py checks on the raw masks that
    every differing pixel has |p - 0.5| <= 1e-4 in the CPU path's own sampled probability -- and fill-holes / closing /
    opening of the class pass can turn one raw pixel into a few.)
    No CSV row is exempt from a check, though: for every instance whose mask is NOT bit-identical the oracle's
    ``measure_mask`` runs on the PRODUCT's own mask, and the product's contour rows must agree with that within 1e-4
    (``csv_max_rel_err_own_mask``; row count included) -- the measurement kernels are verified on exactly the masks
    they saw.  ``csv_max_rel_err_all`` reports the error against the reference's masks over all instances for the record.

    ``order_gap`` > 0 (the eight-tile test; ``bench.py`` keeps 0): instances may come out in another order where the
    REFERENCE's own scores of the instances that changed places differ by at most ``order_gap`` (near-tied fp32 scores sort
    either way in an arithmetic that adds in another order) -- the comparison then runs against the matched instance and
    ``moved_positions`` says which ones moved."""
    n_ref, n = len(ref["masks"]), int(len(scores))
    res = {"instances": n, "instances_ref": n_ref, "mask_iou_min": None, "csv_max_rel_err": None, "csv_max_rel_err_all": None,
           "score_max_abs_err": None, "masks_identical": 0, "masks_with_tie_pixels": 0, "tie_pixels_max": 0,
           "csv_rows": 0, "ellipse_rows_skipped": 0, "csv_max_rel_err_own_mask": None, "csv_rows_own_mask": 0, "ok": False}
    if n != n_ref or (order_gap <= 0 and list(int(c) for c in classes) != ref["classes"]):
        res["why"] = "instance count / classes differ"
        return res
    if order_gap > 0 and n:
        perm, moved = match_near_tie_swaps(ref, masks, classes)
        if perm is None:
            res["why"] = moved
            return res
        res["moved_positions"] = [{"position": i, "reference_position": j, "reference_score_gap": g} for i, j, g in moved]
        if any(g > order_gap for _, _, g in moved):
            res["why"] = f"instances changed places whose reference scores differ by more than {order_gap}: {res['moved_positions']}"
            return res
        if moved:
            ref = dict(ref, masks=[ref["masks"][j] for j in perm], scores=[ref["scores"][j] for j in perm],
                       classes=[ref["classes"][j] for j in perm], rows=[ref["rows"][j] for j in perm])
    if n == 0:
        res.update(mask_iou_min=1.0, csv_max_rel_err=0.0, csv_max_rel_err_all=0.0, csv_max_rel_err_own_mask=0.0, score_max_abs_err=0.0, ok=True)
        return res
    iou_min, err_same, err_all, rows, skipped, same, tie_max = 1.0, 0.0, 0.0, 0, 0, 0, 0
    err_own, rows_own = 0.0, 0
    for i in range(n):
        a, b = np.asarray(masks[i]) > 0, ref["masks"][i]
        diff = int((a ^ b).sum())
        union = int((a | b).sum())
        iou = 1.0 if union == 0 else 1.0 - diff / union
        iou_min = min(iou_min, iou)
        same += diff == 0
        tie_max = max(tie_max, diff)
        got, want = records[i], ref["rows"][i]
        # the product hands every contour back; the CSV (and the reference rows) keep those that pass the area gate
        h, w = b.shape
        min_area = max(5, h * w * 0.000005 * 0.05)
        got = [r for r in got if r["area"] >= min_area]
        if diff != 0:
            # a tie-pixel mask: the product's rows against the oracle's measurement of the product's OWN mask
            own = P.measure_mask(a, 1.0)
            if len(got) != len(own):
                res["why"] = f"instance {i} (tie-pixel mask): {len(got)} CSV rows, oracle on the same mask {len(own)}"
                return res
            for g, r in zip(got, own):
                rows_own += 1
                for k, name in enumerate(MEASURES):
                    if r["_ellipse_unstable"] and k < 3:
                        continue
                    x, y = float(g["values"][k]), float(r[name])
                    err_own = max(err_own, abs(x - y) / max(abs(y), 1e-12))
        if len(got) != len(want):
            if diff == 0:
                res["why"] = f"instance {i}: {len(got)} CSV rows, reference {len(want)}"
                return res
            continue
        for g, r in zip(got, want):
            rows += 1
            for k, name in enumerate(MEASURES):
                if r["_ellipse_unstable"] and k < 3:
                    skipped += (k == 0)
                    continue          # fitEllipse on a degenerate contour: rounding of OpenCV's own SVD decides
                x, y = float(g["values"][k]), float(r[name])
                e = abs(x - y) / max(abs(y), 1e-12)
                err_all = max(err_all, e)
                if diff == 0:
                    err_same = max(err_same, e)
    smax = float(np.max(np.abs(np.asarray(scores, dtype=np.float64) - np.asarray(ref["scores"], dtype=np.float64))))
    res.update(mask_iou_min=iou_min, csv_max_rel_err=err_same, csv_max_rel_err_all=err_all, score_max_abs_err=smax, csv_rows=rows,
               csv_max_rel_err_own_mask=err_own, csv_rows_own_mask=rows_own,
               ellipse_rows_skipped=skipped, masks_identical=int(same), masks_with_tie_pixels=int(n - same), tie_pixels_max=int(tie_max),
               ok=bool(iou_min >= 0.999 and err_same <= 1e-4 and err_own <= 1e-4 and smax <= 1e-4 and tie_max <= 8 and (n - same) <= max(3, n // 10)))
    return res


def compare_predictor(raw: dict, boxes, scores, classes, masks, soft_tol: float = 3e-4, order_gap: float = 2e-6) -> dict:
    """``predictor(tile)`` of the product (boxes [n, 4], scores [n], classes [n], dense bool masks [n, H, W]; torch CPU
    tensors in detector order) against :func:`maskrcnn_ref.predict` of the same tile, WITHOUT assuming that the two lists
    come in the same order: near-tied scores (the oracle's own fp32 scores a few 1e-7 apart) may legitimately sort the
    other way round in an arithmetic that adds in a different order (inference.py:1395-1403 hands the list on in score order).

    Every product instance is matched to the oracle instance of the same class with the nearest box; the result says
    whether that is a bijection, which positions moved and how far apart the ORACLE's scores of the swapped detections are,
    the score error and mask IoU against the matched instance, and for every mask that is not bit-identical the largest
    distance from the 0.5 paste threshold of the oracle's own sampled probability (``paste_masks(soft=True)``) over the
    differing pixels: a difference that is a threshold tie has |p - 0.5| of the size of the arithmetic error."""
    import torch

    rb, rs, rc, rm = raw["pred_boxes"], raw["scores"], raw["pred_classes"], raw["pred_masks"]
    n, nr = int(scores.shape[0]), int(rs.shape[0])
    res = {"instances": n, "instances_ref": nr, "bijection": False, "moved_positions": [], "order_gap_max": 0.0,
           "score_max_abs_err": None, "box_max_abs_err": None, "masks_identical": 0, "masks_ge_0999": 0, "iou_min": None,
           "differing": [], "tie_dist_max": 0.0, "ok": False}
    if n != nr:
        res["why"] = "instance count differs"
        return res
    if n == 0:
        res.update(bijection=True, score_max_abs_err=0.0, box_max_abs_err=0.0, iou_min=1.0, ok=True)
        return res
    d = (boxes[:, None, :].float() - rb[None, :, :].float()).abs().amax(2)
    d = d + (classes[:, None].long() != rc[None, :].long()).float() * 1e6
    perm = d.argmin(1)
    res["bijection"] = bool(perm.unique().numel() == n and float(d[torch.arange(n), perm].max()) < 0.5)
    if not res["bijection"]:
        res["why"] = "the product's instances are not a permutation of the oracle's (class + box within 0.5 px)"
        return res
    moved = [int(i) for i in (perm != torch.arange(n)).nonzero().flatten()]
    gaps = [abs(float(rs[i]) - float(rs[int(perm[i])])) for i in moved]
    res["moved_positions"] = [{"position": i, "oracle_position": int(perm[i]), "oracle_score_gap": g} for i, g in zip(moved, gaps)]
    res["order_gap_max"] = max(gaps, default=0.0)
    res["score_max_abs_err"] = float((scores.float() - rs[perm].float()).abs().max())
    res["box_max_abs_err"] = float((boxes.float() - rb[perm].float()).abs().max())
    h, w = rm.shape[1:]
    iou_min, same, good, tie_max = 1.0, 0, 0, 0.0
    for i in range(n):
        j = int(perm[i])
        a, b = masks[i], rm[j]
        if torch.equal(a, b):
            same += 1
            good += 1
            continue
        diff = a != b
        union = int((a | b).sum())
        iou = 1.0 - int(diff.sum()) / max(union, 1)
        soft = maskrcnn_ref.paste_masks(raw["mask_probs28"][j:j + 1], rb[j:j + 1], h, w, soft=True)[0]
        tie = float((soft[diff] - 0.5).abs().max())
        res["differing"].append({"position": i, "oracle_position": j, "iou": iou, "area": int(b.sum()), "pixels": int(diff.sum()),
                                 "tie_dist": tie, "box_abs_err": float((boxes[i].float() - rb[j].float()).abs().max())})
        iou_min, tie_max = min(iou_min, iou), max(tie_max, tie)
        good += iou >= 0.999
    res.update(masks_identical=same, masks_ge_0999=int(good), iou_min=iou_min, tie_dist_max=tie_max)
    res["ok"] = bool(res["order_gap_max"] <= order_gap and res["score_max_abs_err"] <= 1e-4 and tie_max <= soft_tol)
    return res
