"""Host half of the contrast-distribution columns (``src/utils/measurements.py:195-215``).

The reference masks the gray image with the instance mask, takes ``np.histogram(..., bins=256, range=(0, 255),
density=True)``, the cumulative sum normalised by its last value and three ``np.interp`` calls.  Here the 256 counts come
from ``demia_mask_gray_histogram`` (one workgroup per instance over its bounding box, OpenCV's fixed-point BGR -> gray);
what is left is arithmetic on 256 numbers, done with the same numpy calls in the same order as the reference so that the
doubles agree to the last bit.  The 12 geometric measurements live in ``csrc/contours.hip``.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

_BIN_EDGES = np.linspace(0, 255, 257)          # np.histogram(bins=256, range=(0, 255)) edges
_BIN_WIDTH = np.array(np.diff(_BIN_EDGES), float)


def contrast_percentiles(counts: np.ndarray) -> Tuple[Optional[float], Optional[float], Optional[float]]:
    """(d10, d50, d90) from the integer gray-level histogram of one instance; ``(None, None, None)`` for an empty mask
    (``len(particle_pixels) == 0``, measurements.py:205).  For 8-bit data bin ``i`` of the reference's histogram holds
    exactly the pixels of gray level ``i`` (255 falls into the closed last bin), so the counts are its ``hist`` before
    ``density`` normalisation."""
    counts = np.asarray(counts, dtype=np.int64)
    total = counts.sum()
    if total == 0:
        return None, None, None
    hist = counts / _BIN_WIDTH / total              # density=True: n / db / n.sum()
    cdf = np.cumsum(hist)
    cdf /= cdf[-1]
    left = _BIN_EDGES[:-1]
    return (float(np.interp(0.10, cdf, left)), float(np.interp(0.50, cdf, left)), float(np.interp(0.90, cdf, left)))
