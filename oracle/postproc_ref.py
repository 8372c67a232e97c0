"""ORACLE (test infrastructure, never shipped, never imported by the product path).

CPU restatement (numpy / scipy, dense masks) of the host-side stages the reference runs
after ``predictor(image)``.  Each function cites the reference lines it follows.

Pinned by fixtures generated from the reference itself (tests/golden/, produced by
``tests/golden/make_golden_from_reference.py`` importing ``src.utils.spatial_constraints``,
``src.utils.config`` and ``src.utils.mask_utils``): rows a9, a15, a16 and the scipy /
scikit-image primitives under a10 / a11.  The OpenCV / imutils geometry (contours, arcLength,
minAreaRect, boxPoints, order_points, fitEllipse -- rows a14, a17, a18) restates the published
algorithms of opencv-python-headless 4.11.0.86 / imutils and is **parity unpinned** (no
OpenCV here, no reference fixtures); it is anchored by closed-form
known-answer shapes in the tests.
"""
from __future__ import annotations

import ctypes
import math
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
from scipy import ndimage as ndi

_CROSS = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=bool)  # skimage disk(1) == default footprint
_EIGHT = np.ones((3, 3), dtype=bool)


# ------------------------------------------------------------------ morphology primitives
def fill_holes(m: np.ndarray) -> np.ndarray:
    """scipy.ndimage.binary_fill_holes (default cross structure): background not
    4-connected to the outside of the image becomes foreground."""
    return ndi.binary_fill_holes(m.astype(bool))


def erode_cross(m: np.ndarray) -> np.ndarray:
    """skimage.morphology.erosion(m, disk(1)) == erosion(m): grey erosion, 3x3 cross,
    scipy mode='reflect' (the frame does not erode)."""
    return ndi.grey_erosion(m.astype(np.uint8), footprint=_CROSS, mode="reflect").astype(bool)


def dilate_cross(m: np.ndarray) -> np.ndarray:
    return ndi.grey_dilation(m.astype(np.uint8), footprint=_CROSS, mode="reflect").astype(bool)


def n_components8(m: np.ndarray) -> int:
    """skimage.measure.label(m).max(): 8-connected components."""
    return int(ndi.label(m.astype(bool), structure=_EIGHT)[1])


# ------------------------------------------------------------------ a16 rle
def rle_encoding(x: np.ndarray) -> List[int]:
    """mask_utils.py:17-35 (column-major, 1-based start/length pairs)."""
    dots = np.where(x.T.flatten() == 1)[0]
    out: List[int] = []
    prev = -2
    for b in dots:
        if b > prev + 1:
            out.extend((int(b) + 1, 0))
        out[-1] += 1
        prev = b
    return out


# ------------------------------------------------------------------ a9
def postprocess_masks(ori_mask: np.ndarray, ori_score: np.ndarray, image_hw: Tuple[int, int], min_crys_size: int = 2):
    """mask_utils.py:38-84, quirks included (per-COLUMN counts, truncation to the number of
    populated columns, accumulator-based overlap removal, multi-component masks zeroed but kept)."""
    height, width = image_hw
    if len(ori_mask) == 0 or ori_score.all() < 0.5:
        return []
    keep_ind = np.where(np.sum(ori_mask, axis=(0, 1)) > min_crys_size)[0]
    if len(keep_ind) < len(ori_mask):
        if keep_ind.shape[0] != 0:
            ori_mask = ori_mask[: keep_ind.shape[0]]
            ori_score = ori_score[: keep_ind.shape[0]]
        else:
            return []
    overlap = np.zeros([height, width])
    masks = []
    for i in range(len(ori_mask)):
        mask = fill_holes(ori_mask[i]).astype(np.uint8)
        mask = erode_cross(dilate_cross(mask)).astype(np.uint8)
        overlap += mask
        mask[overlap > 1] = 0
        if n_components8(mask) > 1:
            mask[:] = 0
        masks.append(mask)
    return masks


# ------------------------------------------------------------------ a10 / a11
def universal_min_size(image_hw, is_small: bool) -> int:
    area = image_hw[0] * image_hw[1]
    return max(3, int(area * 0.000005)) if is_small else max(25, int(area * 0.0001))


def postprocess_masks_universal(ori_mask: np.ndarray, image_hw, is_small_class: bool, min_crys_size=None):
    """inference.py:1739-1813."""
    if len(ori_mask) == 0:
        return []
    if min_crys_size is None:
        min_crys_size = universal_min_size(image_hw, is_small_class)
    out = []
    for mask in ori_mask:
        filled = fill_holes(mask)
        final = erode_cross(filled) if is_small_class else dilate_cross(erode_cross(filled))
        if int(final.sum()) >= min_crys_size:
            out.append(final.astype(bool))
    return out


def process_single_mask(mask: np.ndarray) -> np.ndarray:
    """inference.py:190-203 (fill -> erosion disk(1) -> dilation disk(1)), uint8 result."""
    return dilate_cross(erode_cross(fill_holes(mask))).astype(np.uint8)


# ------------------------------------------------------------------ a12
def iou(m1: np.ndarray, m2: np.ndarray) -> float:
    """inference.py:422-435."""
    inter = np.logical_and(m1, m2).sum()
    union = np.logical_or(m1, m2).sum()
    return inter / union if union > 0 else 0


def greedy_dedup(processed: Sequence[np.ndarray], scores: Sequence[float], target_class: int, thr: float):
    """inference.py:1446-1459."""
    um, us, uc = [], [], []
    for i, mask in enumerate(processed):
        if not any(iou(mask, k) > thr for k in um):
            um.append(mask)
            us.append(scores[i])
            uc.append(target_class)
    return um, us, uc


def single_model_class_pass(pred_masks, pred_scores, pred_classes, image_hw, target_class, small_classes,
                            confidence_threshold, iou_threshold, class_specific_settings=None,
                            parallel_mask_processing=True):
    """inference.py:1400-1461 applied to one predictor output."""
    cm = pred_classes == target_class
    masks, scores = pred_masks[cm], pred_scores[cm]
    conf = scores >= confidence_threshold
    masks, scores = masks[conf], scores[conf]
    if len(masks) == 0:
        return [], [], []
    is_small = target_class in small_classes
    cfg = (class_specific_settings or {}).get(f"class_{target_class}", {})
    min_size = cfg.get("min_size", 5 if is_small else 25)
    processed = postprocess_masks(masks, scores, image_hw, min_crys_size=min_size)
    if len(processed) > 2 and parallel_mask_processing:
        processed = [process_single_mask(m) for m in processed]
    thr = 0.5 if is_small else iou_threshold
    return greedy_dedup(processed, scores, target_class, thr) if processed else ([], [], [])


# ------------------------------------------------------------------ a1 / a13
def generate_tiles_with_overlap(image: np.ndarray, tile_size: int, overlap_ratio: float):
    """inference.py:2488-2519."""
    h, w = image.shape[:2]
    stride = int(tile_size * (1 - overlap_ratio))
    tiles = []
    for y in range(0, h, stride):
        for x in range(0, w, stride):
            tile = image[y:min(y + tile_size, h), x:min(x + tile_size, w)]
            if tile.shape[0] < tile_size or tile.shape[1] < tile_size:
                padded = np.zeros((tile_size, tile_size, 3), dtype=image.dtype)
                padded[: tile.shape[0], : tile.shape[1]] = tile
                tile = padded
            tiles.append((tile, x, y))
    return tiles


def is_edge_mask(mask: np.ndarray, tile_size: int, overlap_ratio: float) -> bool:
    """inference.py:2522-2549."""
    edge = int(tile_size * overlap_ratio / 2)
    coords = np.argwhere(mask)
    if len(coords) == 0:
        return True
    y_min, x_min = coords.min(axis=0)
    y_max, x_max = coords.max(axis=0)
    return bool(y_min < edge or y_max > tile_size - edge or x_min < edge or x_max > tile_size - edge)


def resize_nearest(mask: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(INTER_NEAREST): src index = min(floor(dst * src / dst_size), src - 1)."""
    h, w = mask.shape
    # OpenCV: inv_scale = dsize / ssize (double); ifx = 1 / inv_scale; sx = min(cvFloor(x * ifx), ssize - 1)
    ys = np.minimum(np.floor(np.arange(out_h) * (1.0 / (out_h / h))).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(out_w) * (1.0 / (out_w / w))).astype(np.int64), w - 1)
    return mask[ys][:, xs]


# ------------------------------------------------------------------ contours (a17) via the C tracer
_LIB = None


def _contour_lib():
    global _LIB
    if _LIB is None:
        p = Path(__file__).resolve().parent / "_build" / "libcontours_ref.so"
        if not p.exists():
            import subprocess
            subprocess.run(["make", "-C", str(p.parent.parent)], check=True, capture_output=True)
        _LIB = ctypes.CDLL(str(p))
        _LIB.demia_ref_find_external_contours.restype = ctypes.c_int
        _LIB.demia_ref_find_external_contours.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                          ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    return _LIB


def find_external_contours(mask: np.ndarray) -> List[np.ndarray]:
    """cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) -> list of (P, 2) int32 (x, y),
    in the order OpenCV's Python binding returns them (reverse discovery order)."""
    m = np.ascontiguousarray((mask != 0).astype(np.uint8))
    h, w = m.shape
    max_pts = 4 * h * w + 16
    max_c = h * w // 2 + 2
    pts = np.zeros((max_pts, 2), dtype=np.int32)
    offs = np.zeros((max_c + 1,), dtype=np.int32)
    n = _contour_lib().demia_ref_find_external_contours(m.ctypes.data, h, w, pts.ctypes.data, max_pts, offs.ctypes.data, max_c)
    if n < 0:
        raise RuntimeError("contour capacity exceeded")
    cs = [pts[offs[i]: offs[i + 1]].copy() for i in range(n)]
    return cs[::-1]


def contour_area(c: np.ndarray) -> float:
    """cv2.contourArea: 0.5 * |sum(prev.x * cur.y - prev.y * cur.x)| over float32 points, in double."""
    p = c.astype(np.float32).astype(np.float64)
    if len(p) == 0:
        return 0.0
    prev = np.roll(p, 1, axis=0)
    return float(abs(np.sum(prev[:, 0] * p[:, 1] - prev[:, 1] * p[:, 0])) * 0.5)


def arc_length(c: np.ndarray, closed: bool = True) -> float:
    """cv2.arcLength: per-segment float32 sqrt(dx*dx + dy*dy), accumulated in double."""
    p = c.astype(np.float32)
    if len(p) <= 1:
        return 0.0
    prev = np.roll(p, 1, axis=0)
    d = p - prev
    seg = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32)).astype(np.float32)
    if not closed:
        seg = seg[1:]
    return float(np.sum(seg.astype(np.float64)))


# ------------------------------------------------------------------ convex hull + rotating calipers (a18)
def _sklansky(pts, order, start, end, nsign, sign2):
    incr = 1 if end > start else -1
    pprev, pcur = start, start + incr
    pnext = pcur + incr
    P = lambda i: pts[order[i]]
    if start == end or (P(start)[0] == P(end)[0] and P(start)[1] == P(end)[1]):
        return [start]
    stack = [pprev, pcur, pnext]
    end += incr
    sgn = lambda v: (v > 0) - (v < 0)
    while pnext != end:
        cury, nexty = int(P(pcur)[1]), int(P(pnext)[1])
        by = nexty - cury
        if sgn(by) != nsign:
            ax = int(P(pcur)[0]) - int(P(pprev)[0])
            bx = int(P(pnext)[0]) - int(P(pcur)[0])
            ay = cury - int(P(pprev)[1])
            convexity = ay * bx - ax * by
            if sgn(convexity) == sign2 and (ax != 0 or ay != 0):
                pprev, pcur = pcur, pnext
                pnext += incr
                stack.append(pnext)
            else:
                if pprev == start:
                    pcur = pnext
                    stack[1] = pcur
                    pnext += incr
                    stack[2] = pnext
                else:
                    stack[-2] = pnext
                    pcur = pprev
                    pprev = stack[-4]
                    stack.pop()
        else:
            pnext += incr
            stack[-1] = pnext
    return stack[:-1]


def convex_hull(points: np.ndarray) -> np.ndarray:
    """cv::convexHull(points, clockwise=false, returnPoints=true) for integer points
    (Sklansky on the x-then-y sorted set, OpenCV's output order incl. the final cyclic shift)."""
    pts = [(int(x), int(y)) for x, y in points]
    total = len(pts)
    if total == 0:
        return np.zeros((0, 2), dtype=np.int32)
    order = sorted(range(total), key=lambda i: (pts[i][0], pts[i][1], i))
    # std::sort is not stable, but equal points are interchangeable for the hull coordinates
    miny = maxy = 0
    for i in range(1, total):
        y = pts[order[i]][1]
        if pts[order[miny]][1] > y:
            miny = i
        if pts[order[maxy]][1] < y:
            maxy = i
    hull: List[int] = []
    if pts[order[0]] == pts[order[total - 1]]:
        hull.append(0)
    else:
        tl = _sklansky(pts, order, 0, maxy, -1, 1)
        tr = _sklansky(pts, order, total - 1, maxy, -1, -1)
        tl, tr = tr, tl  # !clockwise
        hull += tl[:-1]
        hull += tr[:0:-1]
        stop_idx = tr[1] if len(tr) > 2 else (tl[len(tl) - 2] if len(tl) > 2 else -1)
        bl = _sklansky(pts, order, 0, miny, 1, -1)
        br = _sklansky(pts, order, total - 1, miny, 1, 1)
        if stop_idx >= 0:
            if len(bl) > 2:
                check_idx = bl[1]
            elif len(bl) + len(br) > 2:
                check_idx = br[2 - len(bl)]
            else:
                check_idx = -1
            if check_idx == stop_idx or (check_idx >= 0 and pts[order[check_idx]] == pts[order[stop_idx]]):
                bl = bl[: min(len(bl), 2)]
                br = br[: min(len(br), 2)]
        hull += bl[:-1]
        hull += br[:0:-1]
    idx = [order[i] for i in hull]
    # OpenCV then tries a cyclic shift that makes the ORIGINAL indices ascending / descending
    nout = len(idx)
    if nout >= 3:
        min_i = max_i = lt = 0
        for i in range(1, nout):
            v = idx[i]
            lt += idx[i - 1] < v
            if lt > 1 and lt <= i - 2:
                break
            if v < idx[min_i]:
                min_i = i
            if v > idx[max_i]:
                max_i = i
        mmdist = abs(max_i - min_i)
        if (mmdist == 1 or mmdist == nout - 1) and (lt <= 1 or lt >= nout - 2):
            ascending = (max_i + 1) % nout == min_i
            i0 = min_i if ascending else max_i
            if i0 > 0:
                j, tmp, ok = i0, [], True
                for i in range(nout):
                    cur = idx[j]
                    tmp.append(cur)
                    nj = j + 1 if j + 1 < nout else 0
                    if i < nout - 1 and (ascending != (cur < idx[nj])):
                        ok = False
                        break
                    j = nj
                if ok:
                    idx = tmp
    return np.array([pts[i] for i in idx], dtype=np.int32).reshape(-1, 2)


def min_area_rect(points: np.ndarray):
    """cv2.minAreaRect: convex hull + float32 rotating calipers -> ((cx, cy), (w, h), angle_deg)."""
    f32 = np.float32
    hull = convex_hull(points).astype(np.float32)
    n = len(hull)
    if n > 2:
        out = _rotating_calipers(hull)
        cx = f32(out[0] + (out[2] + out[4]) * f32(0.5))
        cy = f32(out[1] + (out[3] + out[5]) * f32(0.5))
        w = f32(math.sqrt(float(out[2]) * float(out[2]) + float(out[3]) * float(out[3])))
        h = f32(math.sqrt(float(out[4]) * float(out[4]) + float(out[5]) * float(out[5])))
        ang = f32(math.atan2(float(out[3]), float(out[2])))
    elif n == 2:
        cx = f32((hull[0, 0] + hull[1, 0]) * f32(0.5))
        cy = f32((hull[0, 1] + hull[1, 1]) * f32(0.5))
        dx, dy = float(hull[1, 0] - hull[0, 0]), float(hull[1, 1] - hull[0, 1])
        w, h = f32(math.sqrt(dx * dx + dy * dy)), f32(0)
        ang = f32(math.atan2(dy, dx))
    else:
        cx, cy = (hull[0, 0], hull[0, 1]) if n == 1 else (f32(0), f32(0))
        w = h = ang = f32(0)
    return (float(cx), float(cy)), (float(w), float(h)), float(f32(float(ang) * 180 / math.pi))


def _rotating_calipers(points: np.ndarray) -> np.ndarray:
    f32 = np.float32
    n = len(points)
    vect = np.zeros((n, 2), dtype=np.float32)
    inv_len = np.zeros(n, dtype=np.float32)
    left = bottom = right = top = 0
    pt0 = points[0]
    left_x = right_x = pt0[0]
    top_y = bottom_y = pt0[1]
    for i in range(n):
        if pt0[0] < left_x:
            left_x, left = pt0[0], i
        if pt0[0] > right_x:
            right_x, right = pt0[0], i
        if pt0[1] > top_y:
            top_y, top = pt0[1], i
        if pt0[1] < bottom_y:
            bottom_y, bottom = pt0[1], i
        pt = points[(i + 1) if i + 1 < n else 0]
        dx, dy = float(pt[0]) - float(pt0[0]), float(pt[1]) - float(pt0[1])
        vect[i] = (f32(dx), f32(dy))
        inv_len[i] = f32(1.0 / math.sqrt(dx * dx + dy * dy))
        pt0 = pt
    orientation = f32(0)
    ax, ay = float(vect[n - 1, 0]), float(vect[n - 1, 1])
    for i in range(n):
        bx, by = float(vect[i, 0]), float(vect[i, 1])
        conv = ax * by - ay * bx
        if conv != 0:
            orientation = f32(1) if conv > 0 else f32(-1)
            break
        ax, ay = bx, by
    base_a, base_b = orientation, f32(0)
    seq = [bottom, right, top, left]
    minarea = f32(np.finfo(np.float32).max)
    buf = None
    for _ in range(n):
        dp = [f32(+base_a * vect[seq[0], 0] + base_b * vect[seq[0], 1]),
              f32(-base_b * vect[seq[1], 0] + base_a * vect[seq[1], 1]),
              f32(-base_a * vect[seq[2], 0] - base_b * vect[seq[2], 1]),
              f32(+base_b * vect[seq[3], 0] - base_a * vect[seq[3], 1])]
        maxcos = f32(dp[0] * inv_len[seq[0]])
        main = 0
        for i in range(1, 4):
            c = f32(dp[i] * inv_len[seq[i]])
            if c > maxcos:
                main, maxcos = i, c
        pi = seq[main]
        lead_x = f32(vect[pi, 0] * inv_len[pi])
        lead_y = f32(vect[pi, 1] * inv_len[pi])
        if main == 0:
            base_a, base_b = lead_x, lead_y
        elif main == 1:
            base_a, base_b = lead_y, f32(-lead_x)
        elif main == 2:
            base_a, base_b = f32(-lead_x), f32(-lead_y)
        else:
            base_a, base_b = f32(-lead_y), lead_x
        seq[main] = 0 if seq[main] + 1 == n else seq[main] + 1
        dx = f32(points[seq[1], 0] - points[seq[3], 0])
        dy = f32(points[seq[1], 1] - points[seq[3], 1])
        width = f32(f32(dx * base_a) + f32(dy * base_b))
        dx = f32(points[seq[2], 0] - points[seq[0], 0])
        dy = f32(points[seq[2], 1] - points[seq[0], 1])
        height = f32(f32(-dx * base_b) + f32(dy * base_a))
        area = f32(width * height)
        if area <= minarea:
            minarea = area
            buf = (seq[3], base_a, width, base_b, height, seq[0])
    li, A1, wdt, B1, hgt, bi = buf
    A2, B2 = f32(-B1), A1
    C1 = f32(f32(A1 * points[li, 0]) + f32(points[li, 1] * B1))
    C2 = f32(f32(A2 * points[bi, 0]) + f32(points[bi, 1] * B2))
    idet = f32(f32(1) / f32(f32(A1 * B2) - f32(A2 * B1)))
    px = f32(f32(f32(C1 * B2) - f32(C2 * B1)) * idet)
    py = f32(f32(f32(A1 * C2) - f32(A2 * C1)) * idet)
    return np.array([px, py, f32(A1 * wdt), f32(B1 * wdt), f32(A2 * hgt), f32(B2 * hgt)], dtype=np.float32)


def box_points(rect) -> np.ndarray:
    """cv2.boxPoints (RotatedRect::points), float32."""
    f32 = np.float32
    (cx, cy), (w, h), ang = rect
    cx, cy, w, h = f32(cx), f32(cy), f32(w), f32(h)
    a_ = float(f32(ang)) * math.pi / 180.0
    b = f32(f32(math.cos(a_)) * f32(0.5))
    a = f32(f32(math.sin(a_)) * f32(0.5))
    p0 = (f32(cx - f32(a * h) - f32(b * w)), f32(cy + f32(b * h) - f32(a * w)))
    p1 = (f32(cx + f32(a * h) - f32(b * w)), f32(cy - f32(b * h) - f32(a * w)))
    p2 = (f32(f32(2) * cx - p0[0]), f32(f32(2) * cy - p0[1]))
    p3 = (f32(f32(2) * cx - p1[0]), f32(f32(2) * cy - p1[1]))
    return np.array([p0, p1, p2, p3], dtype=np.float32)


def order_points(pts: np.ndarray) -> np.ndarray:
    """imutils.perspective.order_points: sort by x; left pair by y -> (tl, bl); of the right pair
    the one farther from tl is br.  Returns float32 [tl, tr, br, bl]."""
    pts = np.asarray(pts)
    xs = pts[np.argsort(pts[:, 0]), :]
    left, right = xs[:2, :], xs[2:, :]
    left = left[np.argsort(left[:, 1]), :]
    tl, bl = left
    d = np.sqrt(((right - tl[None, :]) ** 2).sum(axis=1).astype(np.float64))
    br, tr = right[np.argsort(d)[::-1], :]
    return np.array([tl, tr, br, bl], dtype="float32")


def _jacobi_lstsq(A: np.ndarray, b: np.ndarray):
    """Least squares through a one-sided Jacobi (Hestenes) SVD -- the algorithm behind cv::SVD::compute --
    and cv::SVD::backSubst's rule that singular values <= 2 * DBL_EPSILON * sum(w) count as zero.
    Returns (x, singular values)."""
    A = A.astype(np.float64).copy()
    n, K = A.shape
    V = np.eye(K)
    eps = np.finfo(np.float64).eps
    for _ in range(40):
        changed = False
        for p in range(K - 1):
            for q in range(p + 1, K):
                a = float(np.sum(A[:, p] * A[:, p]))
                bb = float(np.sum(A[:, q] * A[:, q]))
                g = float(np.sum(A[:, p] * A[:, q]))
                if abs(g) <= eps * math.sqrt(a * bb):
                    continue
                changed = True
                z = (bb - a) / (2.0 * g)
                with np.errstate(over="ignore"):
                    zz = np.float64(z) * np.float64(z)
                t = (1.0 if z >= 0 else -1.0) / (abs(z) + math.sqrt(1.0 + zz))
                c = 1.0 / math.sqrt(1.0 + t * t)
                sn = c * t
                u, v = A[:, p].copy(), A[:, q].copy()
                A[:, p], A[:, q] = c * u - sn * v, sn * u + c * v
                u, v = V[:, p].copy(), V[:, q].copy()
                V[:, p], V[:, q] = c * u - sn * v, sn * u + c * v
        if not changed:
            break
    w = np.sqrt(np.sum(A * A, axis=0))
    utb = A.T @ b
    thr = w.sum() * 2 * eps
    x = np.zeros(K)
    for j in range(K):
        if w[j] > thr:
            x += utb[j] / (w[j] * w[j]) * V[:, j]
    return x, w


def fit_ellipse_ex(c: np.ndarray):
    """cv2.fitEllipse (fitEllipseNoDirect, OpenCV 4.x): ((cx, cy), (w, h), angle) with w <= h, plus a
    flag telling that the fit is numerically unstable (OpenCV's own degenerate branch fired, or one of
    the two least-squares systems has a singular-value ratio below 1e-6): there the result depends on
    the rounding of the SVD, in OpenCV as much as here, and parity checks skip it."""
    pts = c.reshape(-1, 2).astype(np.float32)
    n = len(pts)
    assert n >= 5
    f32 = np.float32
    csum = np.zeros(2, dtype=np.float32)
    for p in pts:
        csum = (csum + p).astype(np.float32)
    cx, cy = f32(csum[0] / f32(n)), f32(csum[1] / f32(n))
    q = (pts - np.array([cx, cy], dtype=np.float32)).astype(np.float32)
    s = float(np.sum(np.abs(q[:, 0].astype(np.float64)) + np.abs(q[:, 1].astype(np.float64))))
    eps32 = float(np.finfo(np.float32).eps)
    scale = 100.0 / (s if s > eps32 else eps32)

    def design(qq):
        px = qq[:, 0].astype(np.float64) * scale
        py = qq[:, 1].astype(np.float64) * scale
        return np.stack([-px * px, -py * py, -px * py, px, py], axis=1), px, py

    A, px, py = design(q)
    gfp, w5 = _jacobi_lstsq(A, np.full(n, 10000.0))
    unstable = False
    if w5.max() * eps32 > w5.min():
        unstable = True
        eps = f32(s / (n * 2) * 1e-3)
        i = np.arange(n)
        ofs = np.stack([((i & 1) * 2 - 1) * eps, ((i & 2) - 1) * eps], axis=1).astype(np.float32)
        pts = (pts + ofs).astype(np.float32)
        q = (pts - np.array([cx, cy], dtype=np.float32)).astype(np.float32)
        A, px, py = design(q)
        gfp, w5 = _jacobi_lstsq(A, np.full(n, 10000.0))
    M = np.array([[2 * gfp[0], gfp[2]], [gfp[2], 2 * gfp[1]]])
    rp = np.zeros(5)
    rp[:2], w2 = _jacobi_lstsq(M, np.array([gfp[3], gfp[4]]))
    A3 = np.stack([(px - rp[0]) ** 2, (py - rp[1]) ** 2, (px - rp[0]) * (py - rp[1])], axis=1)
    g, w3 = _jacobi_lstsq(A3, np.ones(n))
    for wv in (w5, w2, w3):
        if wv.max() == 0 or wv.min() / wv.max() < 1e-6:
            unstable = True
    rp[4] = -0.5 * math.atan2(g[2], g[1] - g[0])
    t = g[2] / math.sin(-2.0 * rp[4]) if abs(g[2]) > 1e-8 else g[1] - g[0]
    rp[2] = abs(g[0] + g[1] - t)
    if rp[2] > 1e-8:
        rp[2] = math.sqrt(2.0 / rp[2])
    rp[3] = abs(g[0] + g[1] + t)
    if rp[3] > 1e-8:
        rp[3] = math.sqrt(2.0 / rp[3])
    with np.errstate(over="ignore", invalid="ignore"):
        bcx = f32(f32(rp[0] / scale) + cx)
        bcy = f32(f32(rp[1] / scale) + cy)
        bw = f32(rp[2] * 2 / scale)
        bh = f32(rp[3] * 2 / scale)
    ang = f32(0)
    if bw > bh:
        bw, bh = bh, bw
        ang = f32(90 + rp[4] * 180 / math.pi)
    if ang < -180:
        ang = f32(ang + 360)
    if ang > 360:
        ang = f32(ang - 360)
    return ((float(bcx), float(bcy)), (float(bw), float(bh)), float(ang)), unstable


def fit_ellipse(c: np.ndarray):
    return fit_ellipse_ex(c)[0]


# ------------------------------------------------------------------ a18
def midpoint(a, b):
    return ((a[0] + b[0]) * 0.5, (a[1] + b[1]) * 0.5)


def calculate_measurements(c: np.ndarray, um_pix: float = 1.0, pixels_per_metric: float = 1.0) -> Dict[str, float]:
    """measurements.py:114-233 (contrast distribution off, config.yaml:37)."""
    c = c.reshape(-1, 2)
    area = contour_area(c)
    perimeter = arc_length(c, True)
    box = box_points(min_area_rect(c))
    box = np.array(box, dtype="int")
    tl, tr, br, bl = order_points(box)
    tltr, blbr = midpoint(tl, tr), midpoint(bl, br)
    tlbl, trbr = midpoint(tl, bl), midpoint(tr, br)
    dA = math.sqrt((tltr[0] - blbr[0]) ** 2 + (tltr[1] - blbr[1]) ** 2)
    dB = math.sqrt((tlbl[0] - trbr[0]) ** 2 + (tlbl[1] - trbr[1]) ** 2)
    dimA, dimB = dA / pixels_per_metric, dB / pixels_per_metric
    dimArea, dimPerimeter = area / pixels_per_metric, perimeter / pixels_per_metric
    diaFeret = max(dimA, dimB)
    aspect = max(dimB, dimA) / min(dimA, dimB) if (dimA and dimB) != 0 else 0
    length, width = min(dimA, dimB) * um_pix, max(dimA, dimB) * um_pix
    circ_ed = math.sqrt(4 * area / math.pi) * um_pix
    chords = perimeter * um_pix
    roundness = 1 / aspect if aspect != 0 else 0
    sphericity = (2 * math.sqrt(math.pi * dimArea)) / dimPerimeter * um_pix if dimPerimeter != 0 else 0
    circularity = 4 * math.pi * (dimArea / dimPerimeter ** 2) * um_pix if dimPerimeter != 0 else 0
    feret = diaFeret * um_pix
    unstable = False
    if len(c) >= 5:
        (_, (maj, mnr), _), unstable = fit_ellipse_ex(c)
        a, b = (maj / 2.0, mnr / 2.0) if maj > mnr else (mnr / 2.0, maj / 2.0)
        ecc = math.sqrt(1 - (b ** 2 / a ** 2)) if a != 0 else 0
        maj_l, min_l = maj / pixels_per_metric * um_pix, mnr / pixels_per_metric * um_pix
    else:
        ecc = maj_l = min_l = 0
    return {"major_axis_length": maj_l, "minor_axis_length": min_l, "eccentricity": ecc, "Length": length,
            "Width": width, "CircularED": circ_ed, "Aspect_Ratio": aspect, "Circularity": circularity,
            "Chords": chords, "Feret_diam": feret, "Roundness": roundness, "Sphericity": sphericity,
            "contrast_d10": None, "contrast_d50": None, "contrast_d90": None, "_ellipse_unstable": unstable}


def bgr_to_gray(image: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(image, COLOR_BGR2GRAY) on uint8 [3P, OpenCV 4.11 color_yuv / RGB2Gray<uchar>]: fixed point with
    B 1868, G 9617, R 4899 (sum 2^14) and round-half-up descale by 14 bits; 2-D input is returned as is
    (measurements.py:198-203, inference.py:267-271)."""
    if image.ndim == 2:
        return image
    b, g, r = (image[:, :, i].astype(np.int64) for i in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def contrast_distribution(gray: np.ndarray, mask: np.ndarray):
    """measurements.py:195-215: d10 / d50 / d90 of the gray levels under the WHOLE instance mask (``single_im_mask``, not
    the contour): density histogram over 256 bins on [0, 255], cumulative sum normalised by its last value, np.interp
    over the left bin edges.  Returns (d10, d50, d90) or (None, None, None) for an empty mask."""
    particle_pixels = gray[mask > 0]
    if len(particle_pixels) == 0:
        return None, None, None
    hist, bin_edges = np.histogram(particle_pixels, bins=256, range=(0, 255), density=True)
    cdf = np.cumsum(hist)
    cdf /= cdf[-1]
    return tuple(np.interp(q, cdf, bin_edges[:-1]) for q in (0.10, 0.50, 0.90))


def contrast_from_counts(counts: np.ndarray):
    """The same three values from the INTEGER histogram (bin i = pixels of gray level i, which is what np.histogram
    counts for integer data on these edges) through numpy's own density / cumsum / interp arithmetic -- the host half of
    the product path, restated here so that the tests can pin it against :func:`contrast_distribution`."""
    counts = np.asarray(counts, dtype=np.int64)
    if counts.sum() == 0:
        return None, None, None
    bin_edges = np.linspace(0, 255, 257)
    db = np.array(np.diff(bin_edges), float)
    hist = counts / db / counts.sum()
    cdf = np.cumsum(hist)
    cdf /= cdf[-1]
    return tuple(np.interp(q, cdf, bin_edges[:-1]) for q in (0.10, 0.50, 0.90))


def calculate_image_quality_score(image: np.ndarray) -> float:
    """inference.py:256-285."""
    gray = bgr_to_gray(image)
    brightness = np.mean(gray) / 255.0
    contrast = np.std(gray) / 128.0
    return float(np.clip(0.4 * brightness + 0.6 * contrast, 0.0, 1.0))


def get_confidence_threshold(image: np.ndarray, target_class: int, small_classes, global_config: dict) -> float:
    """inference.py:288-362 (get_confidence_threshold -> adaptive_confidence_threshold): base threshold AND the mode are
    read from the GLOBAL config (302, 352-359); quality < 0.3 -> x 0.7, < 0.5 -> x 0.85."""
    inf = global_config.get("inference_settings", {})
    class_config = inf.get("class_specific_settings", {}).get(f"class_{target_class}", {})
    base = class_config.get("confidence_threshold", 0.3 if target_class in small_classes else 0.5)
    if inf.get("confidence_mode", "auto") == "manual":
        return base
    q = calculate_image_quality_score(image)
    if q < 0.3:
        return base * 0.7
    if q < 0.5:
        return base * 0.85
    return base


def measure_mask(mask: np.ndarray, um_pix: float = 1.0, image: Optional[np.ndarray] = None,
                 measure_contrast_distribution: bool = False) -> List[Dict[str, float]]:
    """inference.py:1148-1230 for one instance: contours, area gate, measurements per contour (every contour of an
    instance carries the contrast values of the whole instance mask, measurements.py:204)."""
    h, w = mask.shape
    min_area = max(5, h * w * 0.000005 * 0.05)
    rows = []
    contrast = (None, None, None)
    if measure_contrast_distribution and image is not None:
        contrast = contrast_distribution(bgr_to_gray(image), mask)
    for c in find_external_contours(mask):
        if contour_area(c) < min_area:
            continue
        r = calculate_measurements(c, um_pix=um_pix)
        r["contrast_d10"], r["contrast_d50"], r["contrast_d90"] = contrast
        rows.append(r)
    return rows


# ------------------------------------------------------------------ a14
def _smart_bbox(mask):
    rows, cols = np.any(mask, axis=1), np.any(mask, axis=0)
    if not rows.any() or not cols.any():
        return None
    y_min, y_max = np.where(rows)[0][[0, -1]]
    x_min, x_max = np.where(cols)[0][[0, -1]]
    return (y_min, y_max, x_min, x_max)  # NOTE the order: inference.py:2635 stores it like this ...


def _bboxes_overlap_literal(b1, b2):
    if b1 is None or b2 is None:
        return False
    y1_min, x1_min, y1_max, x1_max = b1  # ... and inference.py:2685 unpacks it like this (N6)
    y2_min, x2_min, y2_max, x2_max = b2
    if x1_max < x2_min or x2_max < x1_min:
        return False
    if y1_max < y2_min or y2_max < y1_min:
        return False
    return True


def _calc_iou_literal(m1, m2, b1, b2):
    if not _bboxes_overlap_literal(b1, b2):
        return 0.0
    inter = np.count_nonzero(m1 & m2)
    if inter == 0:
        return 0.0
    union = np.count_nonzero(m1 | m2)
    return inter / union if union else 0.0


def deduplicate_masks_smart(masks, scores, classes, iou_threshold=0.4):
    """inference.py:2552-2677, bug-for-bug (N6): mixed-axis bbox pre-filter, candidate slice by
    MASK INDEX (``sorted_indices[idx+1:]``), compactness from ``contours[0]`` only.  Score ties:
    ``np.argsort(kind='stable')[::-1]`` (numpy's default is unspecified on ties)."""
    if len(masks) == 0:
        return [], [], []
    keep0 = []
    for idx, mask in enumerate(masks):
        rows, cols = np.any(mask, axis=1), np.any(mask, axis=0)
        if not rows.any() or not cols.any():
            continue
        area = np.sum(mask)
        cs = find_external_contours(mask.astype(np.uint8))
        if len(cs) > 0:
            per = arc_length(cs[0], True)
            if per > 0 and (4 * np.pi * area) / (per ** 2) < 0.15:
                continue
        keep0.append(idx)
    masks = [masks[i] for i in keep0]
    scores = [scores[i] for i in keep0]
    classes = [classes[i] for i in keep0]
    if len(masks) == 0:
        return [], [], []
    bboxes = [_smart_bbox(m) for m in masks]
    sorted_indices = np.argsort(np.asarray(scores), kind="stable")[::-1]
    keep, removed = [], set()
    for idx in sorted_indices:
        if idx in removed:
            continue
        keep.append(idx)
        for other in sorted_indices[idx + 1:]:
            if other in removed or classes[other] != classes[idx]:
                continue
            if not _bboxes_overlap_literal(bboxes[idx], bboxes[other]):
                continue
            if _calc_iou_literal(masks[idx], masks[other], bboxes[idx], bboxes[other]) > iou_threshold:
                removed.add(other)
    return [masks[i] for i in keep], [scores[i] for i in keep], [classes[i] for i in keep]


# ------------------------------------------------------------------ a15
def get_mask_bbox(mask):
    rows, cols = np.any(mask, axis=1), np.any(mask, axis=0)
    if not rows.any() or not cols.any():
        return None
    y_min, y_max = np.where(rows)[0][[0, -1]]
    x_min, x_max = np.where(cols)[0][[0, -1]]
    return (y_min, x_min, y_max, x_max)


def bboxes_overlap(b1, b2):
    if b1 is None or b2 is None:
        return False
    y1_min, x1_min, y1_max, x1_max = b1
    y2_min, x2_min, y2_max, x2_max = b2
    if x1_max < x2_min or x2_max < x1_min:
        return False
    if y1_max < y2_min or y2_max < y1_min:
        return False
    return True


def calculate_iou(m1, m2, b1=None, b2=None):
    b1 = get_mask_bbox(m1) if b1 is None else b1
    b2 = get_mask_bbox(m2) if b2 is None else b2
    if not bboxes_overlap(b1, b2):
        return 0.0
    inter = np.count_nonzero(m1 & m2)
    if inter == 0:
        return 0.0
    union = np.count_nonzero(m1 | m2)
    return inter / union if union else 0.0


def calculate_containment(child, parent, cb=None, pb=None):
    cb = get_mask_bbox(child) if cb is None else cb
    pb = get_mask_bbox(parent) if pb is None else pb
    if not bboxes_overlap(cb, pb):
        return 0.0
    ca = np.count_nonzero(child)
    if ca == 0:
        return 0.0
    return np.count_nonzero(child & parent) / ca


def filter_by_overlap_rules(masks, scores, classes, overlap_rules):
    """spatial_constraints.py:192-277 -> set of removed indices."""
    removed = set()
    if not overlap_rules:
        return removed
    bboxes = [get_mask_bbox(m) for m in masks]
    groups: Dict[int, List[int]] = {}
    for idx, cls in enumerate(classes):
        groups.setdefault(cls, []).append(idx)
    for cls, indices in groups.items():
        if cls not in overlap_rules:
            continue
        rule = overlap_rules[cls]
        if rule.get("allow_overlap", True) and rule.get("max_iou_threshold", 0.5) >= 0.9:
            continue
        max_iou = rule.get("max_iou_threshold", 0.5)
        order = sorted(indices, key=lambda i: scores[i], reverse=True)
        for i, i1 in enumerate(order):
            if i1 in removed:
                continue
            for i2 in order[i + 1:]:
                if i2 in removed or not bboxes_overlap(bboxes[i1], bboxes[i2]):
                    continue
                if calculate_iou(masks[i1], masks[i2], bboxes[i1], bboxes[i2]) > max_iou:
                    removed.add(i2)
    return removed


def filter_by_containment_rules(masks, scores, classes, containment_rules, thr=0.95):
    """spatial_constraints.py:280-398 -> set of removed indices."""
    removed = set()
    if not containment_rules:
        return removed
    bboxes = [get_mask_bbox(m) for m in masks]
    by_class: Dict[int, List[int]] = {}
    for idx, cls in enumerate(classes):
        by_class.setdefault(cls, []).append(idx)
    for child_class, parent_class in containment_rules.items():
        if child_class not in by_class:
            continue
        if parent_class not in by_class:
            removed.update(by_class[child_class])
            continue
        parents = [p for p in by_class[parent_class] if p not in removed and bboxes[p] is not None]
        for ch in by_class[child_class]:
            if ch in removed:
                continue
            if bboxes[ch] is None:
                removed.add(ch)
                continue
            best = 0.0
            for p in parents:
                if p in removed or not bboxes_overlap(bboxes[ch], bboxes[p]):
                    continue
                best = max(best, calculate_containment(masks[ch], masks[p], bboxes[ch], bboxes[p]))
            if best < thr:
                removed.add(ch)
    return removed


def apply_spatial_constraints(masks, scores, classes, cfg: dict):
    """spatial_constraints.py:401-460 with the already-loaded constraint dict."""
    if not masks or not cfg.get("enabled", False):
        return masks, scores, classes
    rules = cfg.get("overlap_rules", {})
    if rules:
        rem = filter_by_overlap_rules(masks, scores, classes, rules)
        keep = [i for i in range(len(masks)) if i not in rem]
        masks, scores, classes = [masks[i] for i in keep], [scores[i] for i in keep], [classes[i] for i in keep]
    crules = cfg.get("containment_rules", {})
    if crules:
        rem = filter_by_containment_rules(masks, scores, classes, crules, cfg.get("containment_threshold", 0.95))
        keep = [i for i in range(len(masks)) if i not in rem]
        masks, scores, classes = [masks[i] for i in keep], [scores[i] for i in keep], [classes[i] for i in keep]
    return masks, scores, classes


# ---------------------------------------------------------------------------------------------------------------------
# f4: dense twins of the two flagged NON-parity modes of the product (they exist in north_star, not in the reference's
# live path -- SURVEY N1 / N2); checkers of the product's device versions, nothing else.
def soft_nms_masks(masks: Sequence[np.ndarray], scores: Sequence[float], classes: Sequence[int], sigma: float = 0.5,
                   score_threshold: float = 0.001):
    """Gaussian soft-NMS on mask IoU, per class (Bodla et al. 2017); ties: lower index first.  Returns the kept indices in
    the order of selection and their decayed scores."""
    import math
    sc = [float(s) for s in scores]
    alive = [s >= score_threshold for s in sc]
    order = []
    while any(alive):
        i = max((k for k in range(len(sc)) if alive[k]), key=lambda k: (sc[k], -k))
        order.append(i)
        alive[i] = False
        for j in range(len(sc)):
            if alive[j] and classes[j] == classes[i]:
                v = iou(np.asarray(masks[i]) > 0, np.asarray(masks[j]) > 0)
                if v > 0:
                    sc[j] *= math.exp(-(v * v) / sigma)
                    if sc[j] < score_threshold:
                        alive[j] = False
    return order, [sc[i] for i in order]


def multiscale_merge(per_scale: Dict[float, Tuple[List[np.ndarray], List[float]]], order_of_scales: Sequence[float]):
    """The cross-scale step of inference.py:1951-1977: all masks (already in the original frame) in descending score order
    (stable), a mask is kept unless its IoU with an already kept one exceeds 0.4."""
    masks, scores = [], []
    for s in order_of_scales:
        m, sc = per_scale[s]
        masks += list(m)
        scores += list(sc)
    order = np.argsort(-np.asarray(scores, dtype=np.float64), kind="stable")
    kept = []
    for idx in order:
        if not any(iou(masks[idx], masks[k]) > 0.4 for k in kept):
            kept.append(int(idx))
    return [masks[k] for k in kept], [scores[k] for k in kept]
