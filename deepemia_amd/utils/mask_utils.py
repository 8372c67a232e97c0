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


def rle_from_crop(sub: np.ndarray, y0: int, x0: int, H: int) -> List[int]:
    """The same encoding from the mask's bounding-box crop ``sub = mask[y0:y0+h, x0:x0+w]`` of a frame with ``H`` rows:
    runs cannot leave the box, so a 16-Mpixel frame costs a few thousand operations instead of a transposed copy."""
    sub = np.asarray(sub) != 0
    h, w = sub.shape
    if h == 0 or w == 0 or not sub.any():
        return []
    pad = np.zeros((w, h + 2), dtype=np.int8)
    pad[:, 1:-1] = sub.T
    d = np.diff(pad, axis=1)
    cs, ys = np.nonzero(d == 1)            # row-major: by column, then by row -- the column-major scan order
    ce, ye = np.nonzero(d == -1)
    starts = (x0 + cs).astype(np.int64) * H + (y0 + ys) + 1
    lens = (ye - ys).astype(np.int64)
    if h == H and len(starts) > 1:         # a run that reaches the last row continues at the top of the next column
        glue = starts[1:] == starts[:-1] + lens[:-1]
        if glue.any():
            first = np.concatenate(([True], ~glue))
            grp = np.cumsum(first) - 1
            lens = np.bincount(grp, weights=lens).astype(np.int64)
            starts = starts[first]
    out = np.empty(2 * len(starts), dtype=np.int64)
    out[0::2] = starts
    out[1::2] = lens
    return out.tolist()


def mask_crops(ops, packed: torch.Tensor):
    """Per packed device mask its bounding-box crop on the host: ``(y0, x0, bool[h, w])`` or ``None`` for an empty
    mask.  Only the bounding-box words cross PCIe (one crop launch for the whole set)."""
    from .. import parallel

    n = int(packed.shape[0])
    if n == 0:
        return []
    area, bbox = ops.area_bbox(packed)
    bb = bbox.cpu().numpy()
    _, pay = parallel.encode_instance_table(packed, [0.0] * n, [0] * n, [0] * n, bb, area.cpu().numpy())
    pay = pay.cpu().numpy().view(np.uint32)
    off, out = 0, []
    for i in range(n):
        y0, x0, y1, x1 = (int(v) for v in bb[i])
        if y0 < 0:
            out.append(None)
            continue
        rows, c0, cols = y1 - y0 + 1, x0 >> 5, (x1 >> 5) - (x0 >> 5) + 1
        words = pay[off: off + rows * cols].reshape(rows, cols)
        off += rows * cols
        bits = np.unpackbits(words.view(np.uint8), axis=1, bitorder="little")       # [rows, cols * 32]
        out.append((y0, x0, bits[:, x0 - 32 * c0: x1 - 32 * c0 + 1].astype(bool)))
    return out


def rle_encoding_packed(ops, packed: torch.Tensor, W: int) -> List[List[int]]:
    """``rle_encoding`` of every packed device mask, from its bounding-box crop."""
    H = int(packed.shape[1])
    return [[] if c is None else rle_from_crop(c[2], c[0], c[1], H) for c in mask_crops(ops, packed)]


def rle_crop_launch(ops, packed: torch.Tensor, area, bbox):
    """First half of :func:`rle_text_packed`: the crop launch.  Returns (payload on the device, boxes, offsets) -- the caller brings
    the payload to the host together with whatever else it waits for and hands it to :func:`rle_text_from_payload`."""
    from .. import parallel

    n = int(packed.shape[0])
    bb = np.ascontiguousarray(bbox, dtype=np.int32).reshape(n, 4)
    _, pay = parallel.encode_instance_table(packed, [0.0] * n, [0] * n, [0] * n, bb, np.asarray(area))
    hd = np.zeros((n, parallel.HDR), dtype=np.int32)
    hd[:, 4:8] = bb
    offs = np.ascontiguousarray(parallel._offsets(parallel._payload_lengths(hd)), dtype=np.int64)
    return pay, bb, offs


def rle_text_from_payload(pay: np.ndarray, bb: np.ndarray, offs: np.ndarray, H: int) -> List[str]:
    """Second half: ONE native call for all masks (``demia_host_rle_text``) over the cropped words on the host."""
    import ctypes as C

    from .. import _lib

    n = int(bb.shape[0])
    pay = np.ascontiguousarray(pay)
    toff = np.zeros(n + 1, dtype=np.int64)
    cap = max(1 << 16, 16 * int(pay.size) + 64)
    lib = _lib.load()
    while True:
        out = C.create_string_buffer(cap)
        got = int(lib.demia_host_rle_text(pay.ctypes.data, bb.ctypes.data, offs.ctypes.data, n, H, out, cap, toff.ctypes.data))
        if got >= 0:
            break
        cap = -got + 64
    raw = out.raw[:got].decode("ascii")
    return [raw[toff[i]:toff[i + 1]] for i in range(n)]


def rle_text_packed(ops, packed: torch.Tensor, area=None, bbox=None) -> List[str]:
    """The ``EncodedPixels`` text of every packed device mask (``" ".join(map(str, rle_encoding(mask)))``, reference
    ``inference.py:917-925``): one crop launch, one device-to-host copy of the bounding-box words, ONE native call for all masks
    (``demia_host_rle_text``) -- the per-mask numpy encoding + str() of ~40 000 integers per image was 14 ms of a 70-ms image.
    ``area`` / ``bbox`` (host arrays of the tight boxes) save the reduction when the caller has them."""
    n = int(packed.shape[0])
    if n == 0:
        return []
    if bbox is None or area is None:
        a, b = ops.area_bbox(packed)
        area, bbox = a.cpu().numpy(), b.cpu().numpy()
    pay, bb, offs = rle_crop_launch(ops, packed, area, bbox)
    return rle_text_from_payload(pay.cpu().numpy(), bb, offs, int(packed.shape[1]))


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
