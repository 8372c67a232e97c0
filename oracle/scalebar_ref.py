"""ORACLE (test infrastructure, never shipped): literal, loop-by-loop restatements of the two OpenCV primitives the
reference's scale-bar detector calls (``src/utils/scalebar_ocr.py:197`` ``cv2.Canny(gray_roi, 50, 150, apertureSize=3)`` and
``:204-211`` ``cv2.HoughLinesP(edges, 1, np.pi / 180, threshold=50, minLineLength=20, maxLineGap=10)``), written from the
published source of the pinned dependency ``opencv-python-headless==4.11.0.86`` (reference ``requirements.txt:31``):
``modules/imgproc/src/canny.cpp`` (serial path) and ``modules/imgproc/src/hough.cpp::HoughLinesProbabilistic``, plus the
reference's own ``merge_collinear_segments`` / ``merge_segment_group`` (``scalebar_ocr.py:376-462``) and the selection loop
(``:219-300``).

PARITY UNPINNED: OpenCV cannot be installed in the build container and the reference holds no fixture for this path, so
this file pins the product's vectorised numpy version (``deepemia_amd/utils/scalebar.py``) against an independent
scalar transcription and closed-form cases only.  Pure-Python loops: small regions only.
"""
from __future__ import annotations

import math
from math import sqrt
from typing import List, Tuple

import numpy as np

CANNY_SHIFT = 15
TG22 = int(0.4142135623730950488016887242097 * (1 << CANNY_SHIFT) + 0.5)


def canny(gray: np.ndarray, low_thresh: float, high_thresh: float) -> np.ndarray:
    h, w = gray.shape
    if low_thresh > high_thresh:
        low_thresh, high_thresh = high_thresh, low_thresh
    low, high = int(math.floor(low_thresh)), int(math.floor(high_thresh))
    src = gray.astype(int)

    def px(i, j):                                   # BORDER_REPLICATE
        return int(src[min(max(i, 0), h - 1), min(max(j, 0), w - 1)])

    dx = [[0] * w for _ in range(h)]
    dy = [[0] * w for _ in range(h)]
    for i in range(h):
        for j in range(w):
            dx[i][j] = (px(i - 1, j + 1) + 2 * px(i, j + 1) + px(i + 1, j + 1)) - (px(i - 1, j - 1) + 2 * px(i, j - 1) + px(i + 1, j - 1))
            dy[i][j] = (px(i + 1, j - 1) + 2 * px(i + 1, j) + px(i + 1, j + 1)) - (px(i - 1, j - 1) + 2 * px(i - 1, j) + px(i - 1, j + 1))

    def mag(i, j):                                  # zero outside the image (the mag buffer's border rows / columns)
        if i < 0 or i >= h or j < 0 or j >= w:
            return 0
        return abs(dx[i][j]) + abs(dy[i][j])

    # map: 1 = not an edge, 0 = might be an edge, 2 = edge; one-pixel border of 1s
    pmap = [[1] * (w + 2) for _ in range(h + 2)]
    stack: List[Tuple[int, int]] = []
    for i in range(h):
        for j in range(w):
            m = mag(i, j)
            if m > low:
                xs, ys = dx[i][j], dy[i][j]
                x = abs(xs)
                y = abs(ys) << CANNY_SHIFT
                tg22x = x * TG22
                is_max = False
                if y < tg22x:
                    is_max = m > mag(i, j - 1) and m >= mag(i, j + 1)
                else:
                    tg67x = tg22x + (x << (CANNY_SHIFT + 1))
                    if y > tg67x:
                        is_max = m > mag(i - 1, j) and m >= mag(i + 1, j)
                    else:
                        s = 1 if (xs ^ ys) < 0 else -1
                        is_max = m > mag(i - 1, j - s) and m > mag(i + 1, j + s)
                if is_max:
                    if m > high:
                        pmap[i + 1][j + 1] = 2
                        stack.append((i + 1, j + 1))
                    else:
                        pmap[i + 1][j + 1] = 0
    while stack:                                    # hysteresis: 8 neighbours
        i, j = stack.pop()
        for di in (-1, 0, 1):
            for dj in (-1, 0, 1):
                if (di or dj) and pmap[i + di][j + dj] == 0:
                    pmap[i + di][j + dj] = 2
                    stack.append((i + di, j + dj))
    out = np.zeros((h, w), dtype=np.uint8)
    for i in range(h):
        for j in range(w):
            if pmap[i + 1][j + 1] == 2:
                out[i, j] = 255
    return out


def cv_round(v: float) -> int:
    """cvRound: nearest integer, ties to even (lrint in the default rounding mode)."""
    return int(np.rint(v))


class RNG:
    def __init__(self, state=0xFFFFFFFFFFFFFFFF):
        self.state = state

    def next(self):
        self.state = ((self.state & 0xFFFFFFFF) * 4164903690 + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return self.state & 0xFFFFFFFF

    def uniform(self, a, b):
        return a if a == b else a + self.next() % (b - a)


def hough_lines_p(image: np.ndarray, rho: float, theta: float, threshold: int, line_length: int, line_gap: int,
                  lines_max: int = 2 ** 31 - 1) -> List[Tuple[int, int, int, int]]:
    height, width = image.shape
    irho = 1.0 / rho
    numangle = cv_round(math.pi / theta)
    numrho = cv_round(((width + height) * 2 + 1) / rho)
    f32 = np.float32
    trigtab = []
    for n in range(numangle):
        trigtab.append(f32(math.cos(float(n) * theta) * irho))
        trigtab.append(f32(math.sin(float(n) * theta) * irho))
    accum = [[0] * numrho for _ in range(numangle)]
    mask = [[0] * width for _ in range(height)]
    nzloc = []
    for y in range(height):                          # stage 1: collect the non-zero points
        for x in range(width):
            if image[y, x]:
                mask[y][x] = 1
                nzloc.append((x, y))
    rng = RNG()
    lines = []
    shift = 16
    count = len(nzloc)
    while count > 0:                                 # stage 2: points in random order
        idx = rng.uniform(0, count)
        max_val, max_n = threshold - 1, 0
        j, i = nzloc[idx]
        nzloc[idx] = nzloc[count - 1]
        count -= 1
        if not mask[i][j]:
            continue
        for n in range(numangle):
            r = cv_round(f32(f32(j) * trigtab[2 * n]) + f32(f32(i) * trigtab[2 * n + 1]))
            r += (numrho - 1) // 2
            accum[n][r] += 1
            val = accum[n][r]
            if max_val < val:
                max_val, max_n = val, n
        if max_val < threshold:
            continue
        a, b = -float(trigtab[2 * max_n + 1]), float(trigtab[2 * max_n])
        x0, y0 = j, i
        if abs(a) > abs(b):
            xflag = 1
            dx0 = 1 if a > 0 else -1
            dy0 = cv_round(b * (1 << shift) / abs(a))
            y0 = (y0 << shift) + (1 << (shift - 1))
        else:
            xflag = 0
            dy0 = 1 if b > 0 else -1
            dx0 = cv_round(a * (1 << shift) / abs(b))
            x0 = (x0 << shift) + (1 << (shift - 1))
        line_end = [[0, 0], [0, 0]]
        for k in range(2):
            gap, x, y, dx, dy = 0, x0, y0, dx0, dy0
            if k > 0:
                dx, dy = -dx, -dy
            while True:
                if xflag:
                    j1, i1 = x, y >> shift
                else:
                    j1, i1 = x >> shift, y
                if j1 < 0 or j1 >= width or i1 < 0 or i1 >= height:
                    break
                if mask[i1][j1]:
                    gap = 0
                    line_end[k] = [j1, i1]
                else:
                    gap += 1
                    if gap > line_gap:
                        break
                x += dx
                y += dy
        good_line = abs(line_end[1][0] - line_end[0][0]) >= line_length or abs(line_end[1][1] - line_end[0][1]) >= line_length
        for k in range(2):
            x, y, dx, dy = x0, y0, dx0, dy0
            if k > 0:
                dx, dy = -dx, -dy
            while True:
                if xflag:
                    j1, i1 = x, y >> shift
                else:
                    j1, i1 = x >> shift, y
                if mask[i1][j1]:
                    if good_line:
                        for n in range(numangle):
                            r = cv_round(f32(f32(j1) * trigtab[2 * n]) + f32(f32(i1) * trigtab[2 * n + 1]))
                            r += (numrho - 1) // 2
                            accum[n][r] -= 1
                    mask[i1][j1] = 0
                if i1 == line_end[k][1] and j1 == line_end[k][0]:
                    break
                x += dx
                y += dy
        if good_line:
            lines.append((line_end[0][0], line_end[0][1], line_end[1][0], line_end[1][1]))
            if len(lines) >= lines_max:
                return lines
    return lines


def merge_segment_group(group):
    if len(group) == 1:
        return group[0]
    all_x = [seg["x1"] for seg in group] + [seg["x2"] for seg in group]
    all_y = [seg["y1"] for seg in group] + [seg["y2"] for seg in group]
    x1, x2 = min(all_x), max(all_x)
    y_avg = sum(all_y) / len(all_y)
    y1 = y2 = int(y_avg)
    length = sqrt((x2 - x1) ** 2 + (y2 - y1) ** 2)
    total_length = sum(seg["length"] for seg in group)
    return {"x1": x1, "y1": y1, "x2": x2, "y2": y2, "length": length,
            "intensity": sum(seg["intensity"] * seg["length"] for seg in group) / total_length,
            "dist_to_text": sum(seg["dist_to_text"] * seg["length"] for seg in group) / total_length, "line_idx": -1}


def merge_collinear_segments(segments, max_gap=15, angle_tolerance=5, y_tolerance=5):
    if not segments:
        return []
    sorted_segments = sorted(segments, key=lambda s: min(s["x1"], s["x2"]))
    merged = []
    current_group = [sorted_segments[0]]
    for seg in sorted_segments[1:]:
        last = current_group[-1]
        last_right_x = max(last["x1"], last["x2"])
        last_y = (last["y1"] + last["y2"]) / 2
        curr_left_x = min(seg["x1"], seg["x2"])
        curr_y = (seg["y1"] + seg["y2"]) / 2
        if curr_left_x - last_right_x <= max_gap and abs(curr_y - last_y) <= y_tolerance:
            current_group.append(seg)
        else:
            merged.append(merge_segment_group(current_group))
            current_group = [seg]
    if current_group:
        merged.append(merge_segment_group(current_group))
    return merged
