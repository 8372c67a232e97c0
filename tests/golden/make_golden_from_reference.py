#!/usr/bin/env python3
"""Generates the golden fixtures in this directory by IMPORTING the importable modules of the
reference (build container only; /root/reference never travels to the GPU box).

Run (SURVEY.md section 8(c) recipe):

    mkdir -p /tmp/refhome/deepEMIA && cp -r /root/reference/config /tmp/refhome/deepEMIA/
    HOME=/tmp/refhome PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference \
        python3 tests/golden/make_golden_from_reference.py spatial
    HOME=/tmp/refhome PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference \
        /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_from_reference.py maskutils

`spatial`   -> src.utils.spatial_constraints + src.utils.config   (python 3.10)
`maskutils` -> src.utils.mask_utils + the scipy / scikit-image primitives it calls (python 3.9,
               skimage 0.18.3 / scipy 1.7.1; binary morphology is version-stable)

The fixtures hold only inputs and expected outputs (data), no reference source.
"""
import json
import sys
from pathlib import Path

import numpy as np

OUT = Path(__file__).resolve().parent


def blobs(rng, n, h, w, rmin=4, rmax=22):
    yy, xx = np.mgrid[0:h, 0:w]
    masks = []
    for _ in range(n):
        cy, cx = rng.uniform(0, h), rng.uniform(0, w)
        a, b = rng.uniform(rmin, rmax, size=2)
        th = rng.uniform(0, np.pi)
        u = ((xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)) / a
        v = (-(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)) / b
        masks.append((u * u + v * v) <= 1.0)
    return np.stack(masks)


def pack(masks):
    return np.packbits(masks.astype(np.uint8), axis=-1, bitorder="little")


def gen_spatial():
    from src.utils import spatial_constraints as sc
    from src.utils.config import get_config

    cfg = get_config("polyhipes_tommy")
    sp = sc.load_spatial_constraints("polyhipes_tommy")
    base = get_config()
    (OUT / "config_polyhipes_tommy.json").write_text(json.dumps(
        {"merged": cfg, "spatial": sp, "base_inference_settings": base.get("inference_settings", {}),
         "base_l4": base.get("l4_performance_optimizations", {})}, indent=1, sort_keys=True, default=str))
    cases = {}
    h, w = 96, 128
    for seed in range(8):
        rng = np.random.default_rng(100 + seed)
        n = int(rng.integers(6, 16))
        m = blobs(rng, n, h, w)
        if seed == 3:  # parents that really contain children + an empty mask
            m[0] = False
            m[0, 10:80, 10:110] = True
            m[1] = False
            m[1, 20:40, 20:50] = True
            m[2] = False
        if seed == 5:
            m[:] = m[0]  # all identical -> everything after the first is removed
        scores = rng.uniform(0.3, 1.0, size=n).astype(np.float32)
        if seed == 6:
            scores[1] = scores[0]  # a tie
        classes = rng.integers(0, 2, size=n).astype(np.int64)
        if seed == 3:
            classes[0], classes[1], classes[2] = 0, 1, 1
        if seed == 7:
            classes[:] = 1  # children only, no parent instance -> all removed
        masks = [x for x in m]
        sl, cl = [float(s) for s in scores], [int(c) for c in classes]
        bb = [sc.get_mask_bbox(x) for x in masks]
        bbox = np.array([[-1] * 4 if b is None else list(map(int, b)) for b in bb], dtype=np.int64)
        pair_ov = np.zeros((n, n), dtype=bool)
        pair_iou = np.zeros((n, n), dtype=np.float64)
        pair_con = np.zeros((n, n), dtype=np.float64)
        for i in range(n):
            for j in range(n):
                pair_ov[i, j] = sc.bboxes_overlap(bb[i], bb[j])
                pair_iou[i, j] = sc.calculate_iou(masks[i], masks[j])
                pair_con[i, j] = sc.calculate_containment(masks[i], masks[j])
        rules_o = {0: {"allow_overlap": False, "max_iou_threshold": 0.3}, 1: {"allow_overlap": True, "max_iou_threshold": 0.5}}
        _, _, _, rem_o = sc.filter_by_overlap_rules(masks, sl, cl, rules_o)
        rules_skip = {0: {"allow_overlap": True, "max_iou_threshold": 0.95}}
        _, _, _, rem_skip = sc.filter_by_overlap_rules(masks, sl, cl, rules_skip)
        _, _, _, rem_c = sc.filter_by_containment_rules(masks, sl, cl, {1: 0}, 0.95)
        _, _, _, rem_c50 = sc.filter_by_containment_rules(masks, sl, cl, {1: 0}, 0.5)
        fm, fs, fc = sc.apply_spatial_constraints(masks, sl, cl, "polyhipes_tommy")
        kept_apply = [sl.index(s) for s in fs] if len(set(sl)) == len(sl) else [-1]
        key = f"s{seed}_"
        cases.update({key + "masks": pack(m), key + "scores": scores, key + "classes": classes, key + "bbox": bbox,
                      key + "pair_overlap": pair_ov, key + "pair_iou": pair_iou, key + "pair_containment": pair_con,
                      key + "removed_overlap": np.array(sorted(rem_o), dtype=np.int64),
                      key + "removed_overlap_skip": np.array(sorted(rem_skip), dtype=np.int64),
                      key + "removed_containment": np.array(sorted(rem_c), dtype=np.int64),
                      key + "removed_containment50": np.array(sorted(rem_c50), dtype=np.int64),
                      key + "kept_apply": np.array(kept_apply, dtype=np.int64),
                      key + "n_apply": np.array([len(fm)], dtype=np.int64)})
    cases["shape"] = np.array([h, w])
    np.savez_compressed(OUT / "spatial_constraints.npz", **cases)
    print("wrote spatial_constraints.npz,", len(cases), "arrays")


def gen_maskutils():
    from scipy.ndimage import binary_fill_holes
    from skimage.measure import label
    from skimage.morphology import dilation, disk, erosion

    from src.utils.mask_utils import postprocess_masks, rle_encoding

    out = {}
    # ---- rle_encoding -------------------------------------------------------------------------
    x = np.zeros((6, 6), dtype=np.uint8)
    x[1:4, 2:5] = 1
    out["rle0_in"], out["rle0_out"] = x, np.array(rle_encoding(x), dtype=np.int64)
    rng = np.random.default_rng(7)
    for i in range(1, 5):
        x = (rng.random((17, 23)) > 0.6).astype(np.uint8)
        if i == 4:
            x[:] = 0
        out[f"rle{i}_in"], out[f"rle{i}_out"] = x, np.array(rle_encoding(x), dtype=np.int64)
    # ---- morphology primitives ------------------------------------------------------------------
    h, w = 64, 96
    for i in range(6):
        rng = np.random.default_rng(200 + i)
        m = blobs(rng, 5, h, w, 3, 18).any(0)
        holes = blobs(rng, 6, h, w, 1, 4).any(0)
        m = m & ~holes
        if i == 4:
            m[0, :] = True
            m[:, 0] = True
            m[5:20, 5:20] = False
        if i == 5:
            m = rng.random((h, w)) > 0.55
        k = f"morph{i}_"
        out[k + "in"] = pack(m)
        out[k + "fill"] = pack(binary_fill_holes(m))
        out[k + "erode_disk1"] = pack(erosion(m.astype(np.uint8), disk(1)) > 0)
        out[k + "dilate_disk1"] = pack(dilation(m.astype(np.uint8), disk(1)) > 0)
        out[k + "erode_default"] = pack(erosion(m.astype(np.uint8)) > 0)
        out[k + "dilate_default"] = pack(dilation(m.astype(np.uint8)) > 0)
        out[k + "closing_default"] = pack(erosion(dilation(m.astype(np.uint8))) > 0)
        out[k + "nlabels8"] = np.array([int(label(m).max())], dtype=np.int64)
    out["morph_shape"] = np.array([h, w])
    out["disk1"] = disk(1).astype(np.uint8)
    # ---- postprocess_masks (legacy path, mask_utils.py:38-84) -------------------------------------
    img = np.zeros((h, w, 3), dtype=np.uint8)
    for i in range(6):
        rng = np.random.default_rng(300 + i)
        n = int(rng.integers(3, 9))
        m = blobs(rng, n, h, w, 4, 16)
        if i == 1:  # column-count truncation quirk: 5 masks, only 2 populated columns
            m = np.zeros((5, h, w), dtype=bool)
            for q in range(5):
                m[q, 5 + 8 * q: 11 + 8 * q, 10:12] = True
            n = 5
        if i == 2:  # heavy overlap
            m[1] = m[0]
            m[2, :, : w // 2] |= m[0, :, : w // 2]
        if i == 3:  # a mask that splits into two components after overlap removal, and holes
            m = np.zeros((3, h, w), dtype=bool)
            m[0, 20:40, 40:50] = True
            m[1, 25:35, 20:80] = True
            m[2, 5:15, 5:30] = True
            m[2, 8:11, 10:14] = False
            n = 3
        if i == 4:
            m[:] = False  # nothing populated -> []
        scores = np.sort(rng.uniform(0.3, 1.0, size=n))[::-1].astype(np.float32)
        min_size = [2, 5, 2, 2, 2, 25][i]
        res = postprocess_masks(m, scores, img, min_crys_size=min_size)
        k = f"pp{i}_"
        out[k + "in"] = pack(m)
        out[k + "scores"] = scores
        out[k + "min_size"] = np.array([min_size])
        out[k + "n_out"] = np.array([len(res)])
        out[k + "out"] = pack(np.stack(res) > 0) if len(res) else np.zeros((0, h, w // 8), dtype=np.uint8)
        out[k + "out_dtype_is_uint8"] = np.array([all(r.dtype == np.uint8 for r in res)])
    np.savez_compressed(OUT / "mask_utils.npz", **out)
    print("wrote mask_utils.npz,", len(out), "arrays")


if __name__ == "__main__":
    {"spatial": gen_spatial, "maskutils": gen_maskutils}[sys.argv[1]]()
