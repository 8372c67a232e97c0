"""Multi-GPU layer: one process per GPU, units (tiles / images) sharded round-robin, and ONE exchange
step -- an all-gather of per-tile *instance tables* -- before the global duplicate / containment
filters.  The reference is single-process (``models.py:140``); nothing here
translates an existing call pattern.

Instance table (what crosses xGMI instead of N x H x W masks):
  header  [n, 10] int32: unit id (0 = full-image pass, 1 + t = tile t), class, score (f64 bits, lo / hi word),
                         bbox y0, x0, y1, x1 (global frame, inclusive; -1 for an empty mask), area, flags
  payload [sum] int32  : each mask's bit-packed words cropped to its bbox rows and word columns
A 2048^2 tile with 100 instances is a few hundred KB at most; the collective is latency-bound, so it is
a single ``all_gather`` of sizes followed by one padded ``all_gather`` (direct, one hop on the fully
connected xGMI mesh) -- no ring of per-instance sends.  Every rank ends with the same global table in
the same order (unit id, then detector order), so the greedy filters that follow are deterministic and
identical on all ranks.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


HDR = 10


def shard_indices(n_units: int, rank: int, world: int) -> List[int]:
    """Unit i is owned by rank ``i % world`` (16 tiles on 8 GPUs -> 2 each)."""
    return list(range(rank, n_units, world))


def _crop_words(bbox_row) -> Tuple[int, int, int, int]:
    y0, x0, y1, x1 = (int(v) for v in bbox_row)
    return y0, y1 + 1, x0 >> 5, (x1 >> 5) + 1


def encode_instance_table(packed: Optional[torch.Tensor], scores: Sequence[float], classes: Sequence[int], unit_ids: Sequence[int],
                          bbox: np.ndarray, area: np.ndarray) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (header [n, 10] int32, payload [L] int32) on the masks' device."""
    n = 0 if packed is None else int(packed.shape[0])
    dev = packed.device if packed is not None else torch.device("cpu")
    hdr = np.zeros((n, HDR), dtype=np.int32)
    chunks = []
    for i in range(n):
        hdr[i, 0] = int(unit_ids[i])
        hdr[i, 1] = int(classes[i])
        hdr[i, 2:4] = np.array([scores[i]], dtype=np.float64).view(np.int32)   # exact: ensemble scores are f64 products
        hdr[i, 4:8] = bbox[i]
        hdr[i, 8] = int(area[i])
        if bbox[i, 0] >= 0:
            r0, r1, c0, c1 = _crop_words(bbox[i])
            chunks.append(packed[i, r0:r1, c0:c1].reshape(-1))
    payload = torch.cat(chunks) if chunks else torch.zeros((0,), dtype=torch.int32, device=dev)
    return torch.from_numpy(hdr).to(dev), payload.to(torch.int32)


def decode_instance_table(header: torch.Tensor, payload: torch.Tensor, H: int, W: int, device=None):
    """-> (packed [n, H, W/32] int32, scores list, classes list, unit ids list)."""
    device = header.device if device is None else device
    hdr = header.cpu().numpy()
    n = hdr.shape[0]
    packed = torch.zeros((n, H, (W + 31) // 32), dtype=torch.int32, device=device)
    payload = payload.to(device)
    off = 0
    for i in range(n):
        if hdr[i, 4] < 0:
            continue
        r0, r1, c0, c1 = _crop_words(hdr[i, 4:8])
        cnt = (r1 - r0) * (c1 - c0)
        packed[i, r0:r1, c0:c1] = payload[off:off + cnt].view(r1 - r0, c1 - c0)
        off += cnt
    scores = [float(v) for v in np.ascontiguousarray(hdr[:, 2:4]).view(np.float64).reshape(-1)] if n else []
    return packed, scores, [int(v) for v in hdr[:, 1]], [int(v) for v in hdr[:, 0]]


def all_gather_instance_tables(header: torch.Tensor, payload: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """All ranks contribute their table; every rank returns the GLOBAL table ordered by (unit id, local
    order).  Works with RCCL (device tensors) and gloo (host tensors)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return _sort_by_unit(header, payload)
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    comm_dev = header.device if backend == "nccl" else torch.device("cpu")
    h = header.to(comm_dev).contiguous()
    p = payload.to(comm_dev).contiguous()
    sizes = torch.tensor([h.shape[0], p.shape[0]], dtype=torch.int64, device=comm_dev)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    max_n = max(int(s[0]) for s in all_sizes)
    max_p = max(int(s[1]) for s in all_sizes)
    buf = torch.zeros((max_n * HDR + max_p,), dtype=torch.int32, device=comm_dev)
    buf[: h.numel()] = h.reshape(-1)
    buf[max_n * HDR: max_n * HDR + p.numel()] = p
    gathered = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf, group=group)
    hs, ps = [], []
    for r in range(world):
        n_r, p_r = int(all_sizes[r][0]), int(all_sizes[r][1])
        hs.append(gathered[r][: n_r * HDR].view(n_r, HDR))
        ps.append(gathered[r][max_n * HDR: max_n * HDR + p_r])
    out_h, out_p = _merge_tables(hs, ps)
    return out_h.to(header.device), out_p.to(payload.device)


def _payload_lengths(hdr: np.ndarray) -> np.ndarray:
    lens = np.zeros(hdr.shape[0], dtype=np.int64)
    ok = hdr[:, 4] >= 0
    lens[ok] = (hdr[ok, 6] - hdr[ok, 4] + 1) * ((hdr[ok, 7] >> 5) - (hdr[ok, 5] >> 5) + 1)
    return lens


def _merge_tables(headers: List[torch.Tensor], payloads: List[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """Concatenate rank tables and order instances by unit id (stable: local order within a unit)."""
    recs = []
    for r, (h, p) in enumerate(zip(headers, payloads)):
        hn = h.cpu().numpy()
        offs = np.concatenate(([0], np.cumsum(_payload_lengths(hn))))
        for i in range(hn.shape[0]):
            recs.append((int(hn[i, 0]), r, i, h[i], p[int(offs[i]): int(offs[i + 1])]))
    recs.sort(key=lambda t: (t[0], t[1], t[2]))
    if not recs:
        return headers[0][:0], payloads[0][:0]
    return torch.stack([t[3] for t in recs]), torch.cat([t[4] for t in recs])


def _sort_by_unit(header: torch.Tensor, payload: torch.Tensor):
    return _merge_tables([header], [payload])
