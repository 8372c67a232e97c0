"""Mask utilities (reference ``src/utils/mask_utils.py``) on device-resident packed masks."""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from .constants import DefaultThresholds


def rle_encoding(x: np.ndarray) -> List[int]:
    """Column-major run-length encoding, 1-based ``start length`` pairs (``mask_utils.py:17-35``);
    vectorised, same output as the reference's per-pixel loop."""
    flat = (np.asarray(x).T.reshape(-1) == 1)
    if not flat.any():
        return []
    d = np.diff(np.concatenate(([0], flat.astype(np.int8), [0])))
    starts = np.nonzero(d == 1)[0]
    ends = np.nonzero(d == -1)[0]
    out = np.empty(2 * len(starts), dtype=np.int64)
    out[0::2] = starts + 1
    out[1::2] = ends - starts
    return out.tolist()


def postprocess_masks_device(ops, packed: torch.Tensor, scores: np.ndarray, min_crys_size: Optional[int] = None,
                             bbox: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """``postprocess_masks`` (``mask_utils.py:38-84``) on packed device masks, quirks included.

    * ``np.sum(ori_mask, axis=(0, 1))`` is a per-COLUMN count over all masks; if fewer columns
      exceed ``min_crys_size`` than there are masks, only the first that-many masks are kept
      (none -> ``[]``);
    * per mask, in score order: fill holes -> closing (dilation then erosion, 3x3 cross) -> remove
      every pixel an earlier (closed) mask covers -> zero the mask if it has more than one
      8-connected component (the zeroed mask is still returned).
    ``packed`` is CONSUMED (the stages run in place on it).  ``bbox``: boxes containing the masks, if known.
    Returns ``[N', H, W/32]`` or ``None`` for the reference's ``[]``."""
    if min_crys_size is None:
        min_crys_size = DefaultThresholds.MIN_CRYSTAL_SIZE
    n = int(packed.shape[0])
    scores = np.asarray(scores)
    if n == 0 or bool(scores.all()) < 0.5:
        return None
    if bbox is None:
        _, bbox = ops.area_bbox(packed)
    n_cols = int((ops.column_counts(packed, bbox=bbox) > min_crys_size).sum().item())
    if n_cols < n:
        if n_cols == 0:
            return None
        packed, bbox = packed[:n_cols].contiguous(), bbox[:n_cols].contiguous()
    _, bbox, _ = ops.program_(packed, ["fill", "dilate", "erode"], bbox)
    ops.overlap_prefix_(packed, None, bbox)
    ops.program_(packed, ["drop_multi"], bbox)
    return packed


def process_masks_device(ops, packed: torch.Tensor) -> torch.Tensor:
    """``process_masks_parallel`` (``inference.py:170-213``): fill holes -> erosion -> dilation (in place)."""
    ops.program_(packed, ["fill", "erode", "dilate"])
    return packed


def postprocess_masks_universal_device(ops, packed: torch.Tensor, image_hw, is_small_class: bool, min_crys_size=None,
                                       bbox: Optional[torch.Tensor] = None):
    """``postprocess_masks_universal`` (``inference.py:1739-1813``): returns (packed_kept, kept_indices);
    ``packed`` is consumed."""
    n = int(packed.shape[0])
    if n == 0:
        return packed, []
    area_img = image_hw[0] * image_hw[1]
    if min_crys_size is None:
        min_crys_size = max(3, int(area_img * 0.000005)) if is_small_class else max(25, int(area_img * 0.0001))
    area, _, _ = ops.program_(packed, ["fill", "erode"] if is_small_class else ["fill", "erode", "dilate"], bbox)
    keep = (area >= min_crys_size).nonzero().flatten()
    return packed[keep].contiguous(), keep.cpu().tolist()
