"""Index-based view of a set of device-resident packed masks for the host-side decision logic.

The reference's dedup / constraint code asks three things of a mask: its bounding box, its pixel
count and its intersection count with another mask.  ``DeviceMaskAlgebra`` answers them from HIP
reductions (``demia_mask_area_bbox`` / ``demia_mask_pair_intersections``); the greedy decision
loops (which are inherently sequential and tiny) then run on the host over these integers exactly as
the reference runs them over dense arrays.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch


def _overlap(b1, b2) -> bool:
    return not (b1[3] < b2[1] or b2[3] < b1[1] or b1[2] < b2[0] or b2[2] < b1[0])


class DeviceMaskAlgebra:
    def __init__(self, ops, packed: torch.Tensor, area=None, bbox=None, blocks=None):
        """``area`` / ``bbox`` (device tensors or host arrays): the exact pixel counts and tight boxes when an earlier
        kernel has already reduced them (``MaskOps.program_`` does) -- otherwise one ``demia_mask_area_bbox`` launch.
        ``blocks``: index arrays (e.g. the masks of one tile each); pairs are only ever asked for INSIDE a block, so the
        box-disjointness table is built block by block (16 x 100^2 instead of 1600^2 entries for a 16-tile batch)."""
        self.ops = ops
        self.packed = packed
        self.n = int(packed.shape[0])
        if self.n:
            if area is None or bbox is None:
                area, bbox = ops.area_bbox(packed)
            if isinstance(bbox, torch.Tensor):
                self._bbox_dev = bbox
                self.bbox = bbox.cpu().numpy().astype(np.int64)
            else:
                self.bbox = np.asarray(bbox, dtype=np.int64).reshape(-1, 4)
                self._bbox_dev = torch.from_numpy(self.bbox.astype(np.int32)).to(packed.device)
            self.area = (area.cpu().numpy() if isinstance(area, torch.Tensor) else np.asarray(area)).astype(np.int64)
        else:
            self._bbox_dev = torch.zeros((0, 4), dtype=torch.int32, device=packed.device)
            self.area = np.zeros((0,), dtype=np.int64)
            self.bbox = np.zeros((0, 4), dtype=np.int64)
        self._cache: Dict[Tuple[int, int], int] = {}
        # dense view for the vectorised greedy loops: I[i, j] = |mask_i & mask_j| where known (pairs whose
        # boxes cannot overlap are known to be 0 without asking the GPU)
        self.I = np.zeros((self.n, self.n), dtype=np.int64)
        self.known = np.zeros((self.n, self.n), dtype=bool)
        if self.n:
            for blk in ([np.arange(self.n)] if blocks is None else blocks):
                blk = np.asarray(blk, dtype=np.int64)
                if len(blk) == 0:
                    continue
                b = self.bbox[blk]
                emp = b[:, 0] < 0
                far = ((b[:, None, 3] < b[None, :, 1]) | (b[None, :, 3] < b[:, None, 1]) |
                       (b[:, None, 2] < b[None, :, 0]) | (b[None, :, 2] < b[:, None, 0]) | emp[:, None] | emp[None, :])
                self.known[np.ix_(blk, blk)] = far
            d = np.arange(self.n)
            self.I[d, d] = self.area
            self.known[d, d] = True

    def preload(self, inter: np.ndarray) -> None:
        """Every pair's intersection count at once (a symmetric [n, n] matrix with the pixel counts on its diagonal, e.g. from the
        all-pairs pair matrix of the 0.7 pass): nothing is asked of the GPU afterwards."""
        assert inter.shape == (self.n, self.n)
        self.I = np.ascontiguousarray(inter, dtype=np.int64)
        self.known[:] = True

    def view(self, indices: Sequence[int]) -> "AlgebraView":
        """The same answers for a subset of the masks, renumbered 0..len(indices)-1 (no copy, no launch)."""
        return AlgebraView(self, indices)

    def bbox_of(self, i: int) -> Optional[Tuple[int, int, int, int]]:
        """(y_min, x_min, y_max, x_max) or None for an empty mask (spatial_constraints.py:70-89)."""
        b = self.bbox[i]
        return None if b[0] < 0 else (int(b[0]), int(b[1]), int(b[2]), int(b[3]))

    def intersections(self, pi: Sequence[int], pj: Sequence[int]) -> np.ndarray:
        pi = np.asarray(pi, dtype=np.int64)
        pj = np.asarray(pj, dtype=np.int64)
        out = self.ops.pair_intersections(self.packed, self.packed, self._bbox_dev, self._bbox_dev, pi, pj)
        for a, b, v in zip(pi.tolist(), pj.tolist(), out.tolist()):
            self._cache[(a, b)] = v
            self._cache[(b, a)] = v
        self.I[pi, pj] = out
        self.I[pj, pi] = out
        self.known[pi, pj] = True
        self.known[pj, pi] = True
        return out

    def inter_row(self, i: int, js: np.ndarray) -> np.ndarray:
        """|mask_i & mask_j| for an index array (one GPU launch for whatever is not known yet)."""
        js = np.asarray(js, dtype=np.int64)
        miss = js[~self.known[i, js]]
        if len(miss):
            self.intersections(np.full(len(miss), i, dtype=np.int64), miss)
        return self.I[i, js]

    def prefetch_overlapping_pairs(self, groups: Optional[Iterable[Sequence[int]]] = None) -> None:
        """One launch for every bbox-overlapping pair (within each index group, or all-vs-all)."""
        if self.n < 2 or bool(self.known.all()):
            return
        groups = [list(range(self.n))] if groups is None else [list(g) for g in groups]
        pi: List[int] = []
        pj: List[int] = []
        for g in groups:
            g = [i for i in g if self.bbox[i, 0] >= 0]
            if len(g) < 2:
                continue
            b = self.bbox[g]
            ov = ~((b[:, None, 3] < b[None, :, 1]) | (b[None, :, 3] < b[:, None, 1]) |
                   (b[:, None, 2] < b[None, :, 0]) | (b[None, :, 2] < b[:, None, 0]))
            ii, jj = np.nonzero(np.triu(ov, 1))
            for a, c in zip(ii.tolist(), jj.tolist()):
                if (g[a], g[c]) not in self._cache:
                    pi.append(g[a])
                    pj.append(g[c])
        if pi:
            self.intersections(pi, pj)

    def inter(self, i: int, j: int) -> int:
        """|mask_i & mask_j| (0 without touching the GPU when the boxes cannot overlap)."""
        if i == j:
            return int(self.area[i])
        bi, bj = self.bbox[i], self.bbox[j]
        if bi[0] < 0 or bj[0] < 0 or not _overlap(bi, bj):
            return 0
        if self.known[i, j]:
            return int(self.I[i, j])
        v = self._cache.get((i, j))
        if v is None:
            v = int(self.intersections([i], [j])[0])
        return v

    def union(self, i: int, j: int) -> int:
        return int(self.area[i]) + int(self.area[j]) - self.inter(i, j)

    def iou(self, i: int, j: int) -> float:
        """``iou`` of inference.py:422-435 from the integer counts."""
        u = self.union(i, j)
        return self.inter(i, j) / u if u > 0 else 0


class AlgebraView:
    """Index-translated window on a :class:`DeviceMaskAlgebra` (what the spatial-constraint pass needs of it)."""

    def __init__(self, base: DeviceMaskAlgebra, indices: Sequence[int]):
        self.base = base
        self.idx = np.asarray(list(indices), dtype=np.int64)
        self.n = len(self.idx)
        self.area = base.area[self.idx]
        self.bbox = base.bbox[self.idx]

    def bbox_of(self, i: int):
        return self.base.bbox_of(int(self.idx[i]))

    def inter(self, i: int, j: int) -> int:
        return self.base.inter(int(self.idx[i]), int(self.idx[j]))

    def union(self, i: int, j: int) -> int:
        return self.base.union(int(self.idx[i]), int(self.idx[j]))

    def prefetch_overlapping_pairs(self, groups: Optional[Iterable[Sequence[int]]] = None) -> None:
        groups = [list(range(self.n))] if groups is None else groups
        self.base.prefetch_overlapping_pairs([[int(self.idx[i]) for i in g] for g in groups])

    def intersections(self, pi: Sequence[int], pj: Sequence[int]) -> np.ndarray:
        return self.base.intersections(self.idx[np.asarray(pi, dtype=np.int64)], self.idx[np.asarray(pj, dtype=np.int64)])

    @property
    def _cache(self):
        return _TranslatedKeys(self.base._cache, self.idx)


class _TranslatedKeys:
    """``(i, j) in view._cache`` in the view's numbering."""

    def __init__(self, cache, idx):
        self.cache, self.idx = cache, idx

    def __contains__(self, key) -> bool:
        return (int(self.idx[key[0]]), int(self.idx[key[1]])) in self.cache
