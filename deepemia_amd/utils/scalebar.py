"""Scale-bar LINE detection without OCR.

The reference (``src/utils/scalebar_ocr.py:72-364``) reads the bar's label with EasyOCR and only then looks for the bar
itself: ``cv2.Canny(gray_roi, 50, 150, apertureSize=3)`` (:197), ``cv2.HoughLinesP(edges, 1, pi/180, threshold=50,
minLineLength=20, maxLineGap=10)`` (:204-211), a filter on angle / ROI margins / brightness / distance to the text box
(:219-255), ``merge_collinear_segments`` (:376-462) and the choice of the longest surviving segment (:259-300);
``um_pix = float(label) / length`` (:358).  EasyOCR is out of scope here, so the label and,
optionally, the position of its text box come from the configuration::

    scale_bar:
      label: "500"              # what the OCR would have read (digits are kept, as scalebar_ocr.py:173 does)
      text_center: [120, 30]    # optional, ROI coordinates of the label's box centre; default = centre of the ROI
      # um_per_pixel: 0.8       # alternatively the calibration itself (no image processing at all)

Everything else follows the reference line by line.  The two OpenCV primitives are host-side numpy restatements of
OpenCV 4.11.0's published algorithms (``opencv-python-headless==4.11.0.86``, reference ``requirements.txt:31``):
``modules/imgproc/src/canny.cpp`` (3x3 Sobel with replicated border, L1 magnitude, the fixed-point tan(22.5) sector test,
hysteresis over 8 neighbours) and ``modules/imgproc/src/hough.cpp::HoughLinesProbabilistic`` (the point order drawn
from ``cv::RNG((uint64)-1)``, float trig table, 16-bit fixed-point line walk, vote removal of accepted lines).  OpenCV is
not installable in the build container, so these are checked against the literal loop restatement in
``oracle/scalebar_ref.py`` and closed-form cases, not against OpenCV itself (PARITY UNPINNED).
This is per-image host logic on a region of a few hundred pixels a side -- like the reference's, it runs on the CPU.
"""
from __future__ import annotations

import math
from math import sqrt
from typing import List, Optional, Sequence, Tuple

import numpy as np

TG22 = 13573            # (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5)


def sobel3_s16(gray: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """cv::Sobel(src, CV_16S, 1, 0, 3) and (0, 1, 3) with BORDER_REPLICATE (what Canny calls)."""
    g = np.pad(gray.astype(np.int32), 1, mode="edge")
    dx = (g[:-2, 2:] + 2 * g[1:-1, 2:] + g[2:, 2:]) - (g[:-2, :-2] + 2 * g[1:-1, :-2] + g[2:, :-2])
    dy = (g[2:, :-2] + 2 * g[2:, 1:-1] + g[2:, 2:]) - (g[:-2, :-2] + 2 * g[:-2, 1:-1] + g[:-2, 2:])
    return dx, dy


def canny(gray: np.ndarray, low: float = 50, high: float = 150) -> np.ndarray:
    """cv2.Canny(gray, low, high, apertureSize=3, L2gradient=False) -> u8 {0, 255}."""
    from scipy import ndimage

    if low > high:
        low, high = high, low
    lo, hi = int(math.floor(low)), int(math.floor(high))
    dx, dy = sobel3_s16(gray)
    mag = np.abs(dx) + np.abs(dy)
    m = np.pad(mag, 1)                                        # zero magnitude outside the image
    c = m[1:-1, 1:-1]
    x = np.abs(dx).astype(np.int64)
    y = np.abs(dy).astype(np.int64) << 15
    tg22x = x * TG22
    tg67x = tg22x + (x << 16)
    horiz = y < tg22x                                         # gradient along x: compare with the left / right neighbours
    vert = ~horiz & (y > tg67x)
    diag = ~horiz & ~vert
    s = np.where((dx ^ dy) < 0, 1, -1)                        # canny.cpp: `int s = (xs ^ ys) < 0 ? 1 : -1`
    h, w = gray.shape
    ii, jj = np.mgrid[0:h, 0:w]
    prev_d = m[ii, jj + 1 - s]                                # _mag_p[j - s] (row i - 1; +1 = padding)
    next_d = m[ii + 2, jj + 1 + s]                            # _mag_n[j + s] (row i + 1)
    is_max = (horiz & (c > m[1:-1, :-2]) & (c >= m[1:-1, 2:])) | (vert & (c > m[:-2, 1:-1]) & (c >= m[2:, 1:-1])) | \
             (diag & (c > prev_d) & (c > next_d))
    cand = is_max & (c > lo)
    strong = cand & (c > hi)
    lab, n = ndimage.label(cand, structure=np.ones((3, 3), dtype=bool))
    keep = np.zeros(n + 1, dtype=bool)
    keep[np.unique(lab[strong])] = True
    keep[0] = False
    return np.where(keep[lab], 255, 0).astype(np.uint8)


class CvRNG:
    """cv::RNG: multiply-with-carry, `state = (uint32)state * 4164903690 + (state >> 32)`."""

    def __init__(self, state: int = 0xFFFFFFFFFFFFFFFF):
        self.state = state & 0xFFFFFFFFFFFFFFFF

    def next(self) -> int:
        self.state = ((self.state & 0xFFFFFFFF) * 4164903690 + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return self.state & 0xFFFFFFFF

    def uniform(self, a: int, b: int) -> int:
        return a if a == b else a + self.next() % (b - a)


def hough_lines_p(edges: np.ndarray, rho: float = 1.0, theta: float = math.pi / 180, threshold: int = 50, min_line_length: int = 20,
                  max_line_gap: int = 10, lines_max: int = 2 ** 31 - 1) -> List[Tuple[int, int, int, int]]:
    """cv2.HoughLinesP -> [(x1, y1, x2, y2)] in OpenCV's order of discovery."""
    height, width = edges.shape
    irho = 1.0 / rho
    numangle = int(np.rint(math.pi / theta))
    numrho = int(np.rint(((width + height) * 2 + 1) / rho))
    ang = np.arange(numangle, dtype=np.float64) * theta
    tcos = (np.cos(ang) * irho).astype(np.float32)
    tsin = (np.sin(ang) * irho).astype(np.float32)
    accum = np.zeros((numangle, numrho), dtype=np.int32)
    mask = (edges != 0)
    ys, xs = np.nonzero(mask)                                  # row-major, like the collection loop
    nz = list(zip(xs.tolist(), ys.tolist()))
    mask = mask.copy()
    rng = CvRNG()
    lines: List[Tuple[int, int, int, int]] = []
    shift = 16
    half = (numrho - 1) // 2
    nidx = np.arange(numangle)

    def rows_of(j: int, i: int) -> np.ndarray:
        # cvRound(j * ttab[2n] + i * ttab[2n + 1]) in float arithmetic, round half to even
        return np.rint(np.float32(j) * tcos + np.float32(i) * tsin).astype(np.int64) + half

    count = len(nz)
    while count > 0:
        idx = rng.uniform(0, count)
        j, i = nz[idx]
        nz[idx] = nz[count - 1]
        count -= 1
        if not mask[i, j]:
            continue
        r = rows_of(j, i)
        accum[nidx, r] += 1
        vals = accum[nidx, r]
        max_val = int(vals.max())
        if max_val < threshold:
            continue
        max_n = int(np.argmax(vals))                           # the first angle that reaches the maximum
        a, b = -float(tsin[max_n]), float(tcos[max_n])
        x0, y0 = j, i
        if abs(a) > abs(b):
            xflag = True
            dx0 = 1 if a > 0 else -1
            dy0 = int(np.rint(b * (1 << shift) / abs(a)))
            y0 = (y0 << shift) + (1 << (shift - 1))
        else:
            xflag = False
            dy0 = 1 if b > 0 else -1
            dx0 = int(np.rint(a * (1 << shift) / abs(b)))
            x0 = (x0 << shift) + (1 << (shift - 1))
        line_end = [(0, 0), (0, 0)]
        for k in range(2):
            gap, x, y = 0, x0, y0
            dx, dy = (dx0, dy0) if k == 0 else (-dx0, -dy0)
            while True:
                if xflag:
                    j1, i1 = x, y >> shift
                else:
                    j1, i1 = x >> shift, y
                if j1 < 0 or j1 >= width or i1 < 0 or i1 >= height:
                    break
                if mask[i1, j1]:
                    gap = 0
                    line_end[k] = (j1, i1)
                else:
                    gap += 1
                    if gap > max_line_gap:
                        break
                x += dx
                y += dy
        good = abs(line_end[1][0] - line_end[0][0]) >= min_line_length or abs(line_end[1][1] - line_end[0][1]) >= min_line_length
        for k in range(2):
            x, y = x0, y0
            dx, dy = (dx0, dy0) if k == 0 else (-dx0, -dy0)
            while True:
                if xflag:
                    j1, i1 = x, y >> shift
                else:
                    j1, i1 = x >> shift, y
                if mask[i1, j1]:
                    if good:
                        accum[nidx, rows_of(j1, i1)] -= 1
                    mask[i1, j1] = False
                if i1 == line_end[k][1] and j1 == line_end[k][0]:
                    break
                x += dx
                y += dy
        if good:
            lines.append((line_end[0][0], line_end[0][1], line_end[1][0], line_end[1][1]))
            if len(lines) >= lines_max:
                return lines
    return lines


def line_mean_intensity(gray: np.ndarray, x1: int, y1: int, x2: int, y2: int) -> float:
    """``cv2.mean(gray_roi, mask=<cv2.line(zeros, p1, p2, 255, 2)>)[0]`` (scalebar_ocr.py:246-248).  The mask of a
    thickness-2 line is approximated by the pixels whose centre lies within one pixel of the segment (OpenCV fills a
    sub-pixel quadrilateral plus a radius-1 disc at each end); the value is compared with a brightness threshold on a bar
    that is several pixels thick, where the two rasterisations agree to a fraction of a grey level."""
    h, w = gray.shape
    xa, xb = max(min(x1, x2) - 2, 0), min(max(x1, x2) + 3, w)
    ya, yb = max(min(y1, y2) - 2, 0), min(max(y1, y2) + 3, h)
    if xa >= xb or ya >= yb:
        return 0.0
    yy, xx = np.mgrid[ya:yb, xa:xb].astype(np.float64)
    vx, vy = float(x2 - x1), float(y2 - y1)
    L2 = vx * vx + vy * vy
    t = np.zeros_like(xx) if L2 == 0 else np.clip(((xx - x1) * vx + (yy - y1) * vy) / L2, 0.0, 1.0)
    d2 = (xx - (x1 + t * vx)) ** 2 + (yy - (y1 + t * vy)) ** 2
    sel = d2 <= 1.0 + 1e-9
    if not sel.any():
        return 0.0
    return float(gray[ya:yb, xa:xb][sel].astype(np.float64).mean())


def merge_segment_group(group: Sequence[dict]) -> dict:
    """scalebar_ocr.py:430-462."""
    if len(group) == 1:
        return group[0]
    all_x = [s["x1"] for s in group] + [s["x2"] for s in group]
    all_y = [s["y1"] for s in group] + [s["y2"] for s in group]
    x1, x2 = min(all_x), max(all_x)
    y1 = y2 = int(sum(all_y) / len(all_y))
    total = sum(s["length"] for s in group)
    return {"x1": x1, "y1": y1, "x2": x2, "y2": y2, "length": sqrt((x2 - x1) ** 2 + (y2 - y1) ** 2),
            "intensity": sum(s["intensity"] * s["length"] for s in group) / total,
            "dist_to_text": sum(s["dist_to_text"] * s["length"] for s in group) / total, "line_idx": -1}


def merge_collinear_segments(segments: Sequence[dict], max_gap: int = 15, angle_tolerance: int = 5, y_tolerance: int = 5) -> List[dict]:
    """scalebar_ocr.py:376-427 (``angle_tolerance`` is unused there too)."""
    if not segments:
        return []
    ordered = sorted(segments, key=lambda s: min(s["x1"], s["x2"]))
    merged, group = [], [ordered[0]]
    for seg in ordered[1:]:
        last = group[-1]
        gap = min(seg["x1"], seg["x2"]) - max(last["x1"], last["x2"])
        y_offset = abs((seg["y1"] + seg["y2"]) / 2 - (last["y1"] + last["y2"]) / 2)
        if gap <= max_gap and y_offset <= y_tolerance:
            group.append(seg)
        else:
            merged.append(merge_segment_group(group))
            group = [seg]
    merged.append(merge_segment_group(group))
    return merged


def bgr_to_gray_u8(img: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(BGR2GRAY) on u8: (B 1868 + G 9617 + R 4899 + 8192) >> 14."""
    if img.ndim == 2:
        return img
    b, g, r = (img[..., k].astype(np.int32) for k in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)


def find_scale_bar_line(image: np.ndarray, roi_config: dict, label: str, text_center: Optional[Sequence[int]] = None,
                        intensity_threshold: float = 200, proximity_threshold: float = 50, merge_gap: int = 15,
                        min_line_length: int = 30, edge_margin_factor: float = 0.1) -> dict:
    """``detect_scale_bar`` (scalebar_ocr.py:72-364) from the ROI crop on, with the OCR result supplied by the caller.
    Returns {"psum", "um_pix", "line" (full-image coordinates or None), "roi", "segments" (merged candidates, ROI coords)}."""
    for key in ("x_start_factor", "y_start_factor", "width_factor", "height_factor"):
        if key not in roi_config:
            raise ValueError(f"ROI config missing key: {key}")
    h, w = image.shape[:2]
    x_start = int(w * roi_config["x_start_factor"])
    y_start = int(h * roi_config["y_start_factor"])
    x_end = int(x_start + w * roi_config["width_factor"])
    y_end = int(y_start + h * roi_config["height_factor"])
    gray_roi = bgr_to_gray_u8(image[y_start:y_end, x_start:x_end])
    roi_h, roi_w = gray_roi.shape[:2]
    out = {"psum": "0", "um_pix": 1.0, "line": None, "roi": (x_start, y_start, min(x_end, w), min(y_end, h)), "segments": []}
    digits = "".join(ch for ch in str(label) if ch.isdigit())            # re.sub("[^0-9]", "", text) (:173)
    if not digits or roi_h == 0 or roi_w == 0:
        return out
    x_margin, y_margin = int(roi_w * edge_margin_factor), int(roi_h * edge_margin_factor)
    tc = (roi_w // 2, roi_h // 2) if text_center is None else (int(text_center[0]), int(text_center[1]))
    edges = canny(gray_roi, 50, 150)
    raw = []
    for idx, (x1, y1, x2, y2) in enumerate(hough_lines_p(edges, 1, math.pi / 180, 50, 20, 10)):
        angle = abs(np.arctan2(y2 - y1, x2 - x1) * 180 / np.pi)
        if 10 < angle < 170:
            continue
        if min(x1, x2) < x_margin or max(x1, x2) > roi_w - x_margin or min(y1, y2) < y_margin or max(y1, y2) > roi_h - y_margin:
            continue
        center = ((x1 + x2) // 2, (y1 + y2) // 2)
        raw.append({"x1": x1, "y1": y1, "x2": x2, "y2": y2, "length": sqrt((x2 - x1) ** 2 + (y2 - y1) ** 2),
                    "intensity": line_mean_intensity(gray_roi, x1, y1, x2, y2),
                    "dist_to_text": sqrt((center[0] - tc[0]) ** 2 + (center[1] - tc[1]) ** 2), "line_idx": idx})
    longest, max_length = None, 0.0
    for seg in merge_collinear_segments(raw, merge_gap):
        x1, y1, x2, y2 = seg["x1"], seg["y1"], seg["x2"], seg["y2"]
        near_edge = (min(x1, x2) < x_margin or max(x1, x2) > roi_w - x_margin or min(y1, y2) < y_margin or max(y1, y2) > roi_h - y_margin)
        out["segments"].append(dict(seg, near_edge=near_edge))
        if seg["dist_to_text"] < proximity_threshold and seg["intensity"] > intensity_threshold and seg["length"] > min_line_length \
                and not near_edge and seg["length"] > max_length:
            max_length, longest = seg["length"], (x1, y1, x2, y2)
    if longest is not None:
        out["psum"] = digits
        out["um_pix"] = float(digits) / max_length if max_length > 0 else 1.0
        out["line"] = (longest[0] + x_start, longest[1] + y_start, longest[2] + x_start, longest[3] + y_start)
        out["length"] = max_length
    return out
