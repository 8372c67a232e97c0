"""Device-resident sets of bit-packed instance masks and the HIP operations on them.

Everything the reference does with dense ``(H, W)`` numpy masks after ``predictor(image)``
(hole filling, cross erosion / dilation, overlap removal, connected-component test, pair
intersection counts, tile placement, contour tracing and the measurement reductions) runs here on
``[M, H, W/32]`` int32 tensors through the C ABI.  Only small tables (areas, boxes, pair counts,
contour points, measurement rows) ever reach the host.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import os
import numpy as np
import torch

from . import _lib


# (A/B switch) 0: the large-region variants of the per-mask kernels run over all masks again instead of over a worklist
_WORKLISTS = os.environ.get("DEEPEMIA_MASK_WORKLISTS", "1") != "0"


class PlanePool:
    """Mask planes that stay ZERO outside per-slot boxes (``demia_mask_gather_regions_pooled``): a gather into the pool
    writes the union of a slot's previous box and its new one instead of whole 512-KiB planes.  The planes handed out are
    views of the pool -- valid until the pool is used again (callers that keep masks longer take fresh planes)."""

    def __init__(self, device, H: int, wpr: int, cap: int):
        self.H, self.wpr, self.cap = int(H), int(wpr), int(cap)
        self.planes = torch.zeros((self.cap, self.H, self.wpr), dtype=torch.int32, device=device)
        self.prev = torch.full((self.cap, 4), -1, dtype=torch.int32, device=device)

    def fits(self, n: int, H: int, wpr: int) -> bool:
        return n <= self.cap and (H, wpr) == (self.H, self.wpr)


class MaskOps:
    """Thin, stateless wrapper of the packed-mask entry points for one device."""

    def __init__(self, device: str = "cuda:0"):
        if not torch.cuda.is_available():
            raise _lib.HipExtensionMissing("no HIP device visible: packed-mask ops have no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self._frame_w = 0
        self._points_hint = 0             # contour points the previous trace used (+ slack): sizes ContourSet.fetch(with_points=True)
        self._protected = []              # (first byte, last byte) of planes the in-place ops must never touch (see protect)

    def protect(self, planes: torch.Tensor) -> None:
        """Register planes that consumers may only READ: a captured forward's own output planes stay zero outside the boxes its
        next replay knows about (incremental paste), so an in-place stage on them would leave bits no replay ever clears."""
        import weakref
        lo = int(planes.data_ptr())
        rng = (lo, lo + planes.numel() * planes.element_size())
        if not any(r[:2] == rng and r[2]() is not None for r in self._protected):
            # the registration lives as long as the planes' own tensor (the graph's static buffer, `_base` of the view handed out):
            # when the engine evicts the shape and the memory is reused, the range is no longer protected
            owner = planes._base if planes._base is not None else planes
            self._protected.append(rng + (weakref.ref(owner),))

    def unprotect(self, planes: torch.Tensor) -> None:
        lo = int(planes.data_ptr())
        self._protected = [r for r in self._protected if r[0] != lo]

    def _writable(self, packed: torch.Tensor, what: str) -> None:
        """Every in-place op (``program_``, ``overlap_prefix_``) calls this first."""
        if not self._protected:
            return
        self._protected = [r for r in self._protected if r[2]() is not None]
        p = int(packed.data_ptr())
        for lo, hi, _ in self._protected:
            if lo <= p < hi:
                raise RuntimeError(f"{what}: in-place stage on the read-only output planes of a captured forward -- gather a copy first")

    def set_frame_width(self, W: int) -> None:
        """True pixel width of the full-frame masks in flight.  Packed rows hold ceil(W / 32) words; kernels need
        the true W for the right border (replicate rule of the morphology, image-frame seeds of the hole fill).
        Tensors whose row length does not match ceil(W / 32) (tile-level masks) are taken as 32 * words wide."""
        self._frame_w = int(W)

    def _w(self, packed: torch.Tensor) -> int:
        wpr = int(packed.shape[-1])
        return self._frame_w if (self._frame_w and (self._frame_w + 31) // 32 == wpr) else wpr * 32

    def _stream(self) -> int:
        return int(torch.cuda.current_stream(self.device).cuda_stream)

    def upload(self, arr: np.ndarray) -> torch.Tensor:
        """Host table -> device WITHOUT queueing the host behind the current stream's backlog: a copy from pageable memory
        returns only when it has run, and on the post-processing stream that means after every kernel already enqueued
        there (each of which waits for a CU beside the resident convolution workgroups).  The copy goes through a side
        stream that is always empty; the current stream waits for its event."""
        t = torch.from_numpy(np.ascontiguousarray(arr))
        cur = torch.cuda.current_stream(self.device)
        if getattr(self, "_up_stream", None) is None:
            self._up_stream = torch.cuda.Stream(device=self.device)
        # (set_stream, not `with torch.cuda.stream(...)`: that context asks torch.cuda.is_available() -- hipGetDeviceCount, ~0.1 ms --
        # three times per use, and this runs two dozen times per image: 5 ms of a 36-ms image in the round-5 profile)
        torch.cuda.set_stream(self._up_stream)
        try:
            d = t.to(self.device)
            ev = torch.cuda.Event()
            ev.record(self._up_stream)
        finally:
            torch.cuda.set_stream(cur)
        cur.wait_event(ev)
        d.record_stream(cur)
        return d

    # -- construction -------------------------------------------------------------------------
    def from_dense(self, masks: np.ndarray) -> torch.Tensor:
        """(M, H, W) bool/uint8 host array -> packed device tensor (test / interop helper)."""
        m = np.ascontiguousarray(masks != 0)
        M, H, W = m.shape
        wpr = (W + 31) // 32
        if W % 32:
            m = np.concatenate([m, np.zeros((M, H, wpr * 32 - W), dtype=bool)], axis=2)
        packed = np.packbits(m.reshape(M, H, wpr, 32), axis=-1, bitorder="little").view(np.uint32)
        return torch.from_numpy(packed.reshape(M, H, wpr).view(np.int32)).to(self.device)

    def to_dense(self, packed: torch.Tensor, W: int) -> np.ndarray:
        M, H, wpr = packed.shape
        out = torch.empty((M, H, W), dtype=torch.bool, device=self.device)
        _lib.check(self.lib.demia_unpack_masks(_lib.ptr(packed), _lib.ptr(out), M, H, W, self._stream()), "demia_unpack_masks")
        return out.cpu().numpy()

    # -- reductions -----------------------------------------------------------------------------
    def area_bbox(self, packed: torch.Tensor, hint: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Pixel count and tight bbox per mask.  ``hint`` ([M, 4] int32 boxes known to contain the masks, e.g. the
        paste boxes) restricts the scan to those regions instead of the whole frame."""
        M, H, wpr = packed.shape
        area = torch.empty((M,), dtype=torch.int32, device=self.device)
        bbox = torch.empty((M, 4), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.demia_mask_area_bbox(_lib.ptr(packed), _lib.ptr(hint), _lib.ptr(area), _lib.ptr(bbox), M, H,
                                                 self._w(packed), self._stream()), "demia_mask_area_bbox")
        return area, bbox

    def column_counts(self, packed: torch.Tensor, seg: Optional[torch.Tensor] = None, n_seg: int = 1,
                      bbox: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Per-column pixel counts over all masks ([W]), or per segment ([n_seg, W]) when ``seg`` (int32,
        non-decreasing segment id per mask) is given."""
        M, H, wpr = packed.shape
        W = self._w(packed)
        shape = (W,) if seg is None else (n_seg, W)
        counts = torch.zeros(shape, dtype=torch.int32, device=self.device)
        if bbox is None:
            _, bbox = self.area_bbox(packed)
        _lib.check(self.lib.demia_mask_column_counts(_lib.ptr(packed), _lib.ptr(seg), _lib.ptr(bbox), M, H, W,
                                                     _lib.ptr(counts), self._stream()), "demia_mask_column_counts")
        return counts

    def pair_intersections(self, a: torch.Tensor, b: torch.Tensor, bbox_a: torch.Tensor, bbox_b: torch.Tensor,
                           pi: np.ndarray, pj: np.ndarray) -> np.ndarray:
        P = len(pi)
        if P == 0:
            return np.zeros((0,), dtype=np.int64)
        _, H, wpr = a.shape
        ti = torch.from_numpy(np.ascontiguousarray(pi, dtype=np.int32)).to(self.device)
        tj = torch.from_numpy(np.ascontiguousarray(pj, dtype=np.int32)).to(self.device)
        out = torch.empty((P,), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.demia_mask_pair_intersections(_lib.ptr(a), _lib.ptr(b), _lib.ptr(ti), _lib.ptr(tj), _lib.ptr(bbox_a),
                                                          _lib.ptr(bbox_b), _lib.ptr(out), P, H, self._w(a), self._stream()),
                   "demia_mask_pair_intersections")
        return out.cpu().numpy().astype(np.int64)

    def pair_matrix(self, packed: torch.Tensor, bbox: torch.Tensor, first: np.ndarray, count: np.ndarray,
                    label: Optional[np.ndarray] = None, ld: Optional[int] = None) -> torch.Tensor:
        """[M, ld] int32 ON THE DEVICE: row i, column j - first[i] = |mask_i & mask_j| for every j > i of mask i's segment
        (``first[i]`` / ``count[i]``: start and length of that segment; ``label``: equal-label pairs only).  The pair
        list is made on the device from the boxes -- no host round trip between the stage that reduces the boxes and the
        counts (``demia_mask_pair_matrix``).  The caller fetches it together with whatever else it waits for."""
        M, H, wpr = packed.shape
        if ld is None:
            ld = max(1, int(np.max(count)) if M else 1)
        out = torch.zeros((M, ld), dtype=torch.int32, device=self.device)
        if M == 0:
            return out
        tab = np.stack([np.asarray(first, dtype=np.int32), np.asarray(count, dtype=np.int32),
                        np.asarray(label if label is not None else np.zeros(M), dtype=np.int32)])
        tt = self.upload(tab)
        _lib.check(self.lib.demia_mask_pair_matrix(_lib.ptr(packed), _lib.ptr(bbox), _lib.ptr(tt[0]), _lib.ptr(tt[1]),
                                                   _lib.ptr(tt[2]) if label is not None else 0, _lib.ptr(out), M, ld, H, self._w(packed),
                                                   self._stream()), "demia_mask_pair_matrix")
        return out

    def gray_histogram(self, packed: torch.Tensor, image: torch.Tensor, bbox: Optional[torch.Tensor] = None) -> np.ndarray:
        """[M, 256] gray-level counts of ``image`` ([H, W, 3] BGR or [H, W] gray, uint8, on the device) under each mask:
        what ``np.histogram(cv2.cvtColor(image, BGR2GRAY)[mask > 0], bins=256, range=(0, 255))`` counts
        (measurements.py:197-205)."""
        M, H, wpr = packed.shape
        if M == 0:
            return np.zeros((0, 256), dtype=np.int64)
        W = self._w(packed)
        assert image.dtype == torch.uint8 and image.is_contiguous() and tuple(image.shape[:2]) == (H, W), (image.shape, H, W)
        ch = 1 if image.dim() == 2 else int(image.shape[2])
        if bbox is None:
            _, bbox = self.area_bbox(packed)
        hist = torch.empty((M, 256), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.demia_mask_gray_histogram(_lib.ptr(packed), _lib.ptr(bbox), _lib.ptr(image), ch, M, H, W,
                                                      _lib.ptr(hist), self._stream()), "demia_mask_gray_histogram")
        return hist.cpu().numpy().astype(np.int64)

    # -- morphology -----------------------------------------------------------------------------
    def program_(self, packed: torch.Tensor, stages: Sequence[str], bbox: Optional[torch.Tensor] = None,
                 active: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Run per-mask stages IN PLACE on the bbox region of every mask (``demia_mask_program``): ``stages`` from
        ``fill``, ``dilate``, ``erode``, ``drop_multi``, ``flag_multi``, ``gate`` (masks with ``active[m] == 0`` stop
        there).  ``bbox`` may be any superset of the tight boxes.  Returns (area, tight bbox, multi-component flag) of
        the result."""
        M, H, wpr = packed.shape
        assert len(stages) <= 8 and packed.is_contiguous()
        prog = 0
        for i, st in enumerate(stages):
            prog |= _lib.MOP[st] << (4 * i)
        if bbox is None:
            _, bbox = self.area_bbox(packed)
        area = torch.empty((M,), dtype=torch.int32, device=self.device)
        bbox_out = torch.empty((M, 4), dtype=torch.int32, device=self.device)
        flag = torch.empty((M,), dtype=torch.int32, device=self.device)
        scratch = torch.empty_like(packed)       # only touched by regions that do not fit in LDS
        worklist = torch.empty((M + 2,), dtype=torch.int32, device=self.device) if _WORKLISTS else None    # masks for the large-region variant
        _lib.check(self.lib.demia_mask_program_wl(_lib.ptr(packed), _lib.ptr(scratch), _lib.ptr(bbox), _lib.ptr(active), prog, M, H,
                                                  self._w(packed), _lib.ptr(area), _lib.ptr(bbox_out), _lib.ptr(flag),
                                                  _lib.ptr(worklist), self._stream()), "demia_mask_program_wl")
        return area, bbox_out, flag

    def fill_holes(self, packed: torch.Tensor, bbox: Optional[torch.Tensor] = None) -> torch.Tensor:
        out = packed.clone()
        self.program_(out, ["fill"], bbox)
        return out

    def erode(self, packed: torch.Tensor, bbox: Optional[torch.Tensor] = None) -> torch.Tensor:
        out = packed.clone()
        self.program_(out, ["erode"], bbox)
        return out

    def dilate(self, packed: torch.Tensor, bbox: Optional[torch.Tensor] = None) -> torch.Tensor:
        out = packed.clone()
        self.program_(out, ["dilate"], bbox)
        return out

    def overlap_prefix_(self, packed: torch.Tensor, seg: Optional[torch.Tensor] = None,
                        bbox: Optional[torch.Tensor] = None) -> torch.Tensor:
        M, H, wpr = packed.shape
        self._writable(packed, "overlap_prefix_")
        _lib.check(self.lib.demia_mask_overlap_prefix(_lib.ptr(packed), _lib.ptr(seg), _lib.ptr(bbox), M, H, self._w(packed),
                                                      self._stream()), "demia_mask_overlap_prefix")
        return packed

    def components_gt1(self, packed: torch.Tensor, bbox: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``label(mask).max() > 1`` per mask (the masks are not changed)."""
        return self.program_(packed, ["flag_multi"], bbox)[2]

    def pool(self, name: str, n: int, H: int, wpr: int) -> PlanePool:
        """The named plane pool, (re)allocated when it is too small or of another frame size."""
        pools = self.__dict__.setdefault("_pools", {})
        p = pools.get(name)
        if p is None or not p.fits(n, H, wpr):
            pools.pop(name, None)
            p = pools[name] = PlanePool(self.device, H, wpr, max(int(n * 1.25) + 64, 256))
        return p

    def gather_regions_pooled(self, src: torch.Tensor, index, bbox, pool: PlanePool, first: int = 0, grow: int = 2) -> torch.Tensor:
        """``src[index]`` into slots ``first ..`` of ``pool`` (see :class:`PlanePool`); returns the view of those slots."""
        idx = index if torch.is_tensor(index) else self.upload(np.asarray(index, dtype=np.int64))
        idx = idx.to(device=self.device, dtype=torch.int64).contiguous()
        bb = bbox if torch.is_tensor(bbox) else self.upload(np.ascontiguousarray(bbox, dtype=np.int32))
        bb = bb.to(device=self.device, dtype=torch.int32).contiguous()
        n = int(idx.shape[0])
        _, H, wpr = src.shape
        assert bb.shape == (n, 4) and src.is_contiguous() and src.dtype == torch.int32 and pool.fits(first + n, H, wpr)
        out = pool.planes[first:first + n]
        _lib.check(self.lib.demia_mask_gather_regions_pooled(_lib.ptr(src), _lib.ptr(idx), _lib.ptr(bb), n, H, self._w(src), _lib.ptr(out),
                                                             _lib.ptr(pool.prev[first:first + n]), int(grow), self._stream()),
                   "demia_mask_gather_regions_pooled")
        return out

    def gather_regions(self, src: torch.Tensor, index, bbox, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``src[index]`` for masks that are zero outside ``bbox`` (a superset of their tight boxes; -1 = empty): reads the
        boxes only, writes each destination plane once.  ``index``: int sequence / array (or an i64 device tensor), ``bbox``
        [n, 4] numpy or i32 device tensor, row i belonging to ``index[i]``."""
        idx = index if torch.is_tensor(index) else self.upload(np.asarray(index, dtype=np.int64))
        idx = idx.to(device=self.device, dtype=torch.int64).contiguous()
        bb = bbox if torch.is_tensor(bbox) else self.upload(np.ascontiguousarray(bbox, dtype=np.int32))
        bb = bb.to(device=self.device, dtype=torch.int32).contiguous()
        n = int(idx.shape[0])
        assert bb.shape == (n, 4) and src.is_contiguous() and src.dtype == torch.int32
        _, H, wpr = src.shape
        if out is None:
            out = torch.empty((n, H, wpr), dtype=torch.int32, device=self.device)
        assert out.is_contiguous() and tuple(out.shape) == (n, H, wpr)
        _lib.check(self.lib.demia_mask_gather_regions(_lib.ptr(src), _lib.ptr(idx), _lib.ptr(bb), n, H, self._w(src), _lib.ptr(out),
                                                      self._stream()), "demia_mask_gather_regions")
        return out

    def place_tiles(self, src: torch.Tensor, x_off: Sequence[int], y_off: Sequence[int], tile_h: int, tile_w: int,
                    H: int, W: int, src_w: Optional[int] = None) -> torch.Tensor:
        """``cv2.resize(mask, (tile_w, tile_h), INTER_NEAREST)`` + paste at (x_off, y_off) into a zero (H, W) frame.
        ``src_w``: true pixel width of the source masks (default: 32 x their words per row)."""
        T, sh, swpr = src.shape
        dst = torch.empty((T, H, (W + 31) // 32), dtype=torch.int32, device=self.device)
        xo = torch.tensor(list(x_off), dtype=torch.int32, device=self.device)
        yo = torch.tensor(list(y_off), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.demia_mask_place_tiles(_lib.ptr(src), _lib.ptr(dst), _lib.ptr(xo), _lib.ptr(yo), T, sh,
                                                   swpr * 32 if src_w is None else int(src_w), tile_h, tile_w, H, W, self._stream()), "demia_mask_place_tiles")
        return dst

    # -- contours + measurements ----------------------------------------------------------------
    def trace(self, packed: torch.Tensor, max_contours: int = 64, max_points: Optional[int] = None,
              bbox: Optional[torch.Tensor] = None, total_area: Optional[int] = None) -> "ContourSet":
        """cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) + contourArea + arcLength for every mask, once;
        the returned set can be measured later for any subset of its masks without tracing again.  ``bbox`` (any
        superset of the tight boxes) and ``total_area`` (sizes the point pool) save a reduction when known."""
        M, H, wpr = packed.shape
        W = self._w(packed)
        if bbox is None or (max_points is None and total_area is None):
            area, bbox = self.area_bbox(packed, bbox)
            total_area = int(area.sum().item())
        if max_points is None:
            max_points = int(min(max(4 * int(total_area) // 8 + 4096 * M, 1 << 16), 1 << 26))
        C = max_contours
        cs = ContourSet(self, M, C, max_points)
        scratch = torch.empty_like(packed)       # only touched by regions that do not fit in LDS
        worklist = torch.empty((M + 2,), dtype=torch.int32, device=self.device) if _WORKLISTS else None    # masks for the large-region variant
        _lib.check(self.lib.demia_mask_contours_wl(_lib.ptr(packed), _lib.ptr(scratch), _lib.ptr(bbox), M, H, W, C, max_points,
                                                   _lib.ptr(cs.count), _lib.ptr(cs.info), _lib.ptr(cs.red), _lib.ptr(cs.points),
                                                   _lib.ptr(cs.counters), _lib.ptr(worklist), self._stream()), "demia_mask_contours_wl")
        return cs

    def contours(self, packed: torch.Tensor, max_contours: int = 64, max_points: Optional[int] = None,
                 um_pix: float = 1.0, measure: bool = True, bbox: Optional[torch.Tensor] = None, total_area: Optional[int] = None,
                 extra: Optional[Sequence[torch.Tensor]] = None):
        """Per mask: external contours in OpenCV's order, with area, perimeter and the 12 measurement
        values.  Returns a list (per mask) of lists of dicts with keys ``points`` (P, 2) int32,
        ``area``, ``perimeter`` and (if ``measure``) ``values`` (12,) float64.  ``extra``: int32 device tensors that come to the
        host in the SAME copy (e.g. the cropped words the RLE texts are made of); the call then returns (records, host arrays)."""
        if int(packed.shape[0]) == 0:
            return [] if extra is None else ([], [e.cpu().numpy() for e in extra])
        cs = self.trace(packed, max_contours, max_points, bbox=bbox, total_area=total_area)
        # ONE device-to-host wait: the measurements of every mask's first contours are enqueued right behind the trace and come
        # over with the counts, the contour tables and the points (a mask with more than four contours falls back to the
        # sliced copies of ContourSet.host(): five waits, rare)
        if measure:
            cs.launch_measure(um_pix, slots=4)
        got = cs.fetch(extra=extra, with_points=True)
        recs = cs.records(um_pix=um_pix, measure=measure)
        return recs if extra is None else (recs, list(got))


class ContourSet:
    """Device-resident result of one contour trace over M masks (see :meth:`MaskOps.trace`)."""

    def __init__(self, ops: MaskOps, M: int, C: int, max_points: int):
        dev = ops.device
        self.ops, self.M, self.C, self.max_points = ops, M, C, max_points
        self.count = torch.zeros((M,), dtype=torch.int32, device=dev)
        self.info = torch.zeros((M, C, 4), dtype=torch.int32, device=dev)
        self.red = torch.zeros((M, C, 2), dtype=torch.float64, device=dev)
        self.points = torch.empty((max_points, 2), dtype=torch.int32, device=dev)
        self.counters = torch.zeros((4,), dtype=torch.int32, device=dev)
        self._host = None
        self._pts_host = None             # contour points brought over by fetch(with_points=True)
        self.blocking_point_copies = 0    # times records() had to fetch the points with a device-to-host copy of its own

    def host(self):
        if self._host is None:
            cnt = self.counters.cpu().numpy()
            if cnt[1] != 0:
                raise _lib.HipKernelError(f"contour extraction overflow (flags {int(cnt[1])}): raise max_contours / max_points")
            self.walker_stats = (int(cnt[3]), int(cnt[2]))      # masks traced by parallel walkers, of which fell back
            count = self.count.cpu().numpy()
            # only the contour slots in use cross PCIe ([M, C, ...] with C = 256 is mostly empty: usually one per mask)
            self.max_count = mc = max(1, int(count.max()) if self.M else 1)
            self._host = (count, self.info[:, :mc].contiguous().cpu().numpy(), self.red[:, :mc].contiguous().cpu().numpy(),
                          int(cnt[0]))
        return self._host

    def launch_measure(self, um_pix: float = 1.0, slots: int = 4) -> None:
        """Enqueue the measurements of the first ``slots`` contours of EVERY mask right behind the trace, without waiting
        for the contour counts: :meth:`fetch` then brings counts, contour tables and values to the host with ONE wait.
        (Masks with more contours than ``slots`` -- rare -- are measured again by :meth:`measure` on demand.)"""
        ops = self.ops
        M, C, mp = self.M, self.C, self.max_points
        self._mslots, self._m_um = int(slots), float(um_pix)
        self._wi = torch.empty((int(ops.lib.demia_contour_work_ints(M, C, mp)),), dtype=torch.int32, device=ops.device)
        self._wf = torch.empty((int(ops.lib.demia_contour_work_floats(M, C, mp)),), dtype=torch.float32, device=ops.device)
        self._wd = torch.empty((int(ops.lib.demia_contour_work_doubles(M, C, mp)),), dtype=torch.float64, device=ops.device)
        self._vals_dev = torch.zeros((M, self._mslots, 12), dtype=torch.float64, device=ops.device)
        _lib.check(ops.lib.demia_contour_measure(0, _lib.ptr(self.count), _lib.ptr(self.info), _lib.ptr(self.red),
                                                 _lib.ptr(self.points), M, C, mp, _lib.ptr(self._wi), _lib.ptr(self._wf), _lib.ptr(self._wd),
                                                 float(um_pix), _lib.ptr(self._vals_dev), self._mslots, ops._stream()), "demia_contour_measure")

    def fetch(self, extra: Optional[Sequence[torch.Tensor]] = None, with_points: bool = False):
        """ONE device-to-host wait for everything the host decides on: counters, contour counts, the first ``slots``
        contour slots of every mask (info, area / perimeter, measurement values when :meth:`launch_measure` ran) and the
        int32 device tensors in ``extra`` (returned as numpy arrays).  Falls back to the sliced copy of :meth:`host` when a
        mask has more contours than the slots fetched.  ``with_points``: the contour POINTS come over in the same copy --
        how many are in use is only known after the wait, so the copy takes as many as the previous trace of this
        ``MaskOps`` used (+ 25 %); when that was too few (the first call of a job) :meth:`records` copies them itself and
        counts it in ``blocking_point_copies``."""
        M = self.M
        k = getattr(self, "_mslots", 4)
        k = min(k, self.C)
        parts = [self.counters, self.count, self.info[:, :k].reshape(-1)]
        f64 = [self.red[:, :k].reshape(-1)]
        if getattr(self, "_vals_dev", None) is not None:
            f64.append(self._vals_dev.reshape(-1))
        extra = list(extra or [])
        parts += [e.reshape(-1) for e in extra]
        n_pts = min(int(getattr(self.ops, "_points_hint", 0)), self.max_points) if with_points else 0
        if n_pts:
            parts.append(self.points[:n_pts].reshape(-1))
        if sum(int(t.numel()) for t in parts) % 2:            # keep the float64 block 8-byte aligned in the host copy
            parts.append(torch.zeros((1,), dtype=torch.int32, device=self.ops.device))
        ints = torch.cat(parts)
        dbl = torch.cat(f64).view(torch.int32)
        host = torch.cat([ints, dbl]).cpu().numpy()            # the wait
        pos = 0
        cnt = host[pos:pos + 4]; pos += 4
        count = host[pos:pos + M]; pos += M
        info = host[pos:pos + M * k * 4].reshape(M, k, 4); pos += M * k * 4
        outs = []
        for e in extra:
            outs.append(host[pos:pos + e.numel()].reshape(tuple(e.shape)))
            pos += e.numel()
        if with_points:
            used = int(cnt[0])
            if n_pts >= used:
                self._pts_host = host[pos:pos + 2 * used].reshape(used, 2)
            self.ops._points_hint = used + used // 4 + 4096
        d = host[int(ints.numel()):].view(np.float64)
        red = d[:M * k * 2].reshape(M, k, 2)
        if cnt[1] != 0:
            raise _lib.HipKernelError(f"contour extraction overflow (flags {int(cnt[1])}): raise max_contours / max_points")
        self.walker_stats = (int(cnt[3]), int(cnt[2]))
        mc = max(1, int(count.max()) if M else 1)
        if mc <= k:
            self.max_count = mc
            self._host = (count, info[:, :mc], red[:, :mc], int(cnt[0]))
            if getattr(self, "_vals_dev", None) is not None:
                self._vals_host = d[M * k * 2:].reshape(M, k, 12)[:, :mc]
        else:                                                     # a mask with many contours: the general (two-wait) path
            self._host = None
            self._vals_host = None
            self.host()
        return outs

    def first_contour_perimeter(self) -> np.ndarray:
        """Perimeter of ``contours[0]`` (OpenCV order: the contour with the greatest (start y, start x)) per mask,
        -1 where a mask has no contour -- what ``deduplicate_masks_smart``'s compactness test reads."""
        count, info, red, _ = self.host()
        out = np.full(self.M, -1.0)
        one = count == 1                                          # the usual case: no choice to make
        out[one] = red[one, 0, 1]
        for m in np.nonzero(count > 1)[0]:
            c = int(count[m])
            k = np.lexsort((info[m, :c, 0], info[m, :c, 1]))[-1]
            out[m] = red[m, k, 1]
        return out

    def measure(self, um_pix: float = 1.0, select: Optional[Sequence[int]] = None) -> np.ndarray:
        """[M, max contours per mask, 12] measurement values (only for the selected masks when ``select`` is given)."""
        ops = self.ops
        self.host()
        pre = getattr(self, "_vals_host", None)
        if pre is not None and abs(getattr(self, "_m_um", um_pix) - um_pix) == 0.0:
            return pre                                    # measured for every mask right behind the trace (launch_measure)
        M, C, mp, mc = self.M, self.C, self.max_points, self.max_count
        sel_t = None
        if select is not None:
            flags = np.zeros(M, dtype=np.int32)
            flags[np.asarray(list(select), dtype=np.int64)] = 1
            sel_t = torch.from_numpy(flags).to(ops.device)
        wi = torch.empty((int(ops.lib.demia_contour_work_ints(M, C, mp)),), dtype=torch.int32, device=ops.device)
        wf = torch.empty((int(ops.lib.demia_contour_work_floats(M, C, mp)),), dtype=torch.float32, device=ops.device)
        wd = torch.empty((int(ops.lib.demia_contour_work_doubles(M, C, mp)),), dtype=torch.float64, device=ops.device)
        vals = torch.zeros((M, mc, 12), dtype=torch.float64, device=ops.device)
        _lib.check(ops.lib.demia_contour_measure(_lib.ptr(sel_t), _lib.ptr(self.count), _lib.ptr(self.info), _lib.ptr(self.red),
                                                 _lib.ptr(self.points), M, C, mp, _lib.ptr(wi), _lib.ptr(wf), _lib.ptr(wd),
                                                 float(um_pix), _lib.ptr(vals), mc, ops._stream()), "demia_contour_measure")
        return vals.cpu().numpy()

    def records(self, um_pix: float = 1.0, measure: bool = True, select: Optional[Sequence[int]] = None, with_points: bool = True):
        """Per (selected) mask the list of contour dicts in OpenCV's order (reverse raster order of the start)."""
        vals = self.measure(um_pix, select) if measure else None
        count, info, red, used = self.host()
        pts = None
        if with_points:
            if self._pts_host is not None and self._pts_host.shape[0] == used:
                pts = self._pts_host                      # came over with fetch(with_points=True): no wait of its own
            else:
                pts = self.points[:used].cpu().numpy()
                self.blocking_point_copies += 1
        out = []
        for m in (range(self.M) if select is None else select):
            recs = []
            for c in range(int(count[m])):
                sx, sy, n, off = (int(v) for v in info[m, c])
                rec = dict(start=(sx, sy), area=float(red[m, c, 0]), perimeter=float(red[m, c, 1]))
                if pts is not None:
                    rec["points"] = pts[off: off + n]             # views of this call's own host copies
                if vals is not None:
                    rec["values"] = vals[m, c]
                recs.append(rec)
            if len(recs) > 1:
                recs.sort(key=lambda r: (r["start"][1], r["start"][0]), reverse=True)
            out.append(recs)
        return out
