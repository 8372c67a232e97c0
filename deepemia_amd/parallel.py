"""Multi-GPU layer: one process per GPU, units (tiles / images) sharded round-robin, and ONE exchange
step -- an all-gather of per-tile *instance tables* -- before the global duplicate / containment
filters.  The reference is single-process (``models.py:140``); nothing here
translates an existing call pattern.

Instance table (what crosses xGMI instead of N x H x W masks):
  header  [n, 10] int32: unit id (0 = full-image pass, 1 + t = tile t), class, score (f64 bits, lo / hi word),
                         bbox y0, x0, y1, x1 (global frame, inclusive; -1 for an empty mask), area, flags
  payload [sum] int32  : each mask's bit-packed words cropped to its bbox rows and word columns
A 2048^2 tile with 100 instances is a few hundred KB at most; the collective is latency-bound, so it is
ONE padded ``all_gather`` per exchange (direct, one hop on the fully connected xGMI mesh; the per-rank capacity
is agreed without talking, see ``all_gather_instance_tables``) -- no ring of per-instance sends.  Every rank ends with the same global table in
the same order (unit id, then detector order), so the greedy filters that follow are deterministic and
identical on all ranks.
"""
from __future__ import annotations

import time
from typing import Dict, List, NamedTuple, Optional, Sequence, Tuple

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


def _payload_lengths(hdr: np.ndarray) -> np.ndarray:
    """Words of payload per instance: bbox rows x word columns (0 for an empty mask)."""
    lens = np.zeros(hdr.shape[0], dtype=np.int64)
    ok = hdr[:, 4] >= 0
    lens[ok] = (hdr[ok, 6].astype(np.int64) - hdr[ok, 4] + 1) * ((hdr[ok, 7] >> 5) - (hdr[ok, 5] >> 5) + 1)
    return lens


def _offsets(lens: np.ndarray) -> np.ndarray:
    return np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.int64) if len(lens) else np.zeros((0,), dtype=np.int64)


def encode_instance_table(packed: Optional[torch.Tensor], scores: Sequence[float], classes: Sequence[int], unit_ids: Sequence[int],
                          bbox: np.ndarray, area: np.ndarray) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (header [n, 10] int32, payload [L] int32) on the masks' device.  Device masks are cropped by ONE launch
    (``demia_mask_crop_pack``); host masks (the gloo tests) by numpy slicing."""
    n = 0 if packed is None else int(packed.shape[0])
    dev = packed.device if packed is not None else torch.device("cpu")
    hdr = np.zeros((n, HDR), dtype=np.int32)
    if n == 0:
        return torch.from_numpy(hdr).to(dev), torch.zeros((0,), dtype=torch.int32, device=dev)
    hdr[:, 0] = np.asarray(unit_ids, dtype=np.int32)
    hdr[:, 1] = np.asarray(classes, dtype=np.int32)
    hdr[:, 2:4] = np.asarray(scores, dtype=np.float64).reshape(n, 1).view(np.int32)   # exact: ensemble scores are f64 products
    hdr[:, 4:8] = np.asarray(bbox, dtype=np.int32).reshape(n, 4)
    hdr[:, 8] = np.asarray(area, dtype=np.int32)
    lens = _payload_lengths(hdr)
    offs = _offsets(lens)
    total = int(lens.sum())
    if packed.is_cuda:
        from . import _lib

        payload = torch.empty((total,), dtype=torch.int32, device=dev)
        hdr_t = torch.from_numpy(hdr).to(dev)
        if total:
            pk = packed.contiguous()
            bb = hdr_t[:, 4:8].contiguous()
            of = torch.from_numpy(offs).to(dev)
            H, W = int(pk.shape[1]), int(pk.shape[2]) * 32
            _lib.check(_lib.load().demia_mask_crop_pack(_lib.ptr(pk), _lib.ptr(bb), _lib.ptr(of), n, H, W, _lib.ptr(payload),
                                                        int(torch.cuda.current_stream(dev).cuda_stream)), "demia_mask_crop_pack")
        return hdr_t, payload
    pk = packed.numpy()
    out = np.empty((total,), dtype=np.int32)
    for i in range(n):
        if lens[i]:
            r0, r1, c0, c1 = _crop_words(hdr[i, 4:8])
            out[offs[i]: offs[i] + lens[i]] = pk[i, r0:r1, c0:c1].reshape(-1)
    return torch.from_numpy(hdr), torch.from_numpy(out)


def decode_header(hdr: np.ndarray):
    """Host side of the decode: (scores, classes, unit ids) as lists + payload lengths / offsets per instance -- numpy
    conversions, no Python loop over instances (eight ranks x ~2 800 rows per step)."""
    n = hdr.shape[0]
    lens = _payload_lengths(hdr)
    scores = np.ascontiguousarray(hdr[:, 2:4]).view(np.float64).reshape(-1).tolist() if n else []
    return scores, hdr[:, 1].tolist(), hdr[:, 0].tolist(), lens, _offsets(lens)


def decode_instance_table(header: torch.Tensor, payload: torch.Tensor, H: int, W: int, device=None, host_header: Optional[np.ndarray] = None,
                          offsets: Optional[np.ndarray] = None):
    """-> (packed [n, H, W/32] int32, scores list, classes list, unit ids list).  ``host_header``: the header's host copy
    when the exchange has already brought it over (``GlobalTable.host_header``) -- no second device-to-host wait.
    ``offsets``: where each instance's words start in ``payload`` (``GlobalTable.offsets``; default: back to back in
    header order)."""
    device = torch.device(header.device if device is None else device)
    hdr = host_header
    if hdr is None or hdr.shape[0] != header.shape[0]:
        hdr = header.cpu().numpy()
    n = hdr.shape[0]
    wpr = (W + 31) // 32
    scores, classes, units, lens, offs = decode_header(hdr)
    if offsets is not None:
        offs = np.ascontiguousarray(offsets, dtype=np.int64)
    if device.type == "cuda":
        from . import _lib

        packed = torch.zeros((n, H, wpr), dtype=torch.int32, device=device)
        if n and int(lens.sum()):
            pay = payload.to(device).contiguous()
            bb = torch.from_numpy(np.ascontiguousarray(hdr[:, 4:8])).to(device)
            of = torch.from_numpy(offs).to(device)
            _lib.check(_lib.load().demia_mask_crop_unpack(_lib.ptr(pay), _lib.ptr(bb), _lib.ptr(of), n, H, W, _lib.ptr(packed),
                                                          int(torch.cuda.current_stream(device).cuda_stream)), "demia_mask_crop_unpack")
    else:
        pk = np.zeros((n, H, wpr), dtype=np.int32)
        pay = payload.cpu().numpy()
        for i in range(n):
            if lens[i]:
                r0, r1, c0, c1 = _crop_words(hdr[i, 4:8])
                pk[i, r0:r1, c0:c1] = pay[offs[i]: offs[i] + lens[i]].reshape(r1 - r0, c1 - c0)
        packed = torch.from_numpy(pk)
    return packed, scores, classes, units


class GlobalTable(NamedTuple):
    """What an exchange returns on every rank: the global table ordered by (unit id, rank, local order)."""
    header: torch.Tensor                   # [n, 10] int32 on the caller's device
    payload: torch.Tensor                  # [L] int32
    host_header: Optional[np.ndarray]      # the header's host copy (the exchange's one host wait brought it over)
    offsets: Optional[np.ndarray]          # [n] int64: first payload word of each instance (None: back to back in header order)
    status: np.ndarray                     # [world] int: every rank's status word of THIS exchange (0 = fine)


class ExchangeState:
    """Reserved capacities and counters of the exchanges of ONE job.  The single collective of an exchange is only correct
    while every rank reserves the same capacity, i.e. while every rank has taken part in exactly the same sequence of
    exchanges of a process group: the capacity is therefore kept per GROUP OBJECT (a re-initialised default group, or a
    subgroup, starts from its own size exchange), and a pipeline / job may own its state instead of the module default."""

    def __init__(self):
        self.caps: Dict[object, Tuple[int, int]] = {}
        self.stats = {"exchanges": 0, "size_exchanges": 0, "host_syncs": 0}
        self.last_merge_ms = 0.0          # host time of the last exchange AFTER its collective: slot views, merge by unit id

    def reset(self) -> None:
        self.caps.clear()


_default_state = ExchangeState()
stats = _default_state.stats              # (module-level view of the default state's counters)


def _grow(v: int, q: int) -> int:
    """Capacity for the next exchange: half as much again as this one's largest table, in steps of ``q``."""
    return (int(v * 1.5) // q + 1) * q


SLOT_WORDS = 3        # every rank's slot starts with {header rows, payload words, status}


def all_gather_instance_tables(header: torch.Tensor, payload: torch.Tensor, group=None, status: int = 0,
                               state: Optional[ExchangeState] = None) -> GlobalTable:
    """All ranks contribute their table; every rank returns the GLOBAL table ordered by (unit id, local
    order).  Works with RCCL (device tensors) and gloo (host tensors).

    ONE collective and ONE device-to-host wait per call in the steady state: every rank reserves the same per-rank capacity
    (1.5 x the largest table of the previous exchange -- all ranks saw the same sizes, so they agree without talking), each
    rank's true sizes travel in the first words of its slot, and the sizes + headers of all ranks come to the host in
    one copy (the greedy filters that follow run there).  Only the first exchange of a job, and one whose table outgrew the
    reserved capacity (every rank sees that in the same words and repeats it together), pay a separate size exchange.

    ``status`` is this rank's verdict on its own share of the unit of work the exchange closes (one image of the CLI loop):
    a rank whose local passes FAILED still takes part -- with an empty table and a non-zero status -- so the collective
    sequence stays aligned, and every rank reads every rank's status in the same words and skips or merges the image
    TOGETHER (``GlobalTable.status``).  Without it a rank that skipped an image would meet its peers' next all-gather with
    the tables of another image."""
    st = _default_state if state is None else state
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1 and dist.get_backend(group) != "nccl":
        h, p, host = _merge_tables([header], [payload])
        return GlobalTable(h, p, host, None, np.asarray([int(status)], dtype=np.int64))
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    comm_dev = header.device if backend == "nccl" else torch.device("cpu")
    h = header.to(comm_dev).contiguous()
    p = payload.to(comm_dev).contiguous()
    n, pl = int(h.shape[0]), int(p.shape[0])
    key = (backend, world, dist.group.WORLD if group is None else group)
    cap = st.caps.get(key)
    st.stats["exchanges"] += 1
    S = SLOT_WORDS
    while True:
        if cap is None:
            sizes = torch.tensor([n, pl], dtype=torch.int64).to(comm_dev)
            all_sizes = torch.zeros((world, 2), dtype=torch.int64, device=comm_dev)
            dist.all_gather_into_tensor(all_sizes, sizes, group=group) if backend == "nccl" else \
                dist.all_gather(list(all_sizes.unbind(0)), sizes, group=group)
            all_sizes = all_sizes.cpu().numpy()
            st.stats["size_exchanges"] += 1
            st.stats["host_syncs"] += 1
            cap = (int(all_sizes[:, 0].max()), int(all_sizes[:, 1].max()))
        cn, cp = cap
        nn, pp = min(n, cn), min(pl, cp)
        buf = torch.zeros((S + cn * HDR + cp,), dtype=torch.int32, device=comm_dev)
        buf[:S] = torch.tensor([n, pl, int(status)], dtype=torch.int32).to(comm_dev, non_blocking=True)
        buf[S: S + nn * HDR] = h[:nn].reshape(-1)
        buf[S + cn * HDR: S + cn * HDR + pp] = p[:pp]
        gathered = torch.empty((world, buf.numel()), dtype=torch.int32, device=comm_dev)
        if backend == "nccl":
            dist.all_gather_into_tensor(gathered, buf, group=group)      # one direct all-gather over the xGMI mesh
        else:
            dist.all_gather(list(gathered.unbind(0)), buf, group=group)
        head = gathered[:, : S + cn * HDR].cpu().numpy()                 # THE host wait: every rank's sizes + status + header rows
        st.stats["host_syncs"] += 1
        sz = head[:, :2].astype(np.int64)
        if bool((sz[:, 0] <= cn).all() and (sz[:, 1] <= cp).all()):
            break
        cap = None            # some rank's table outgrew the reservation: all ranks read the same words and redo it exactly
    t_host = time.perf_counter()
    st.caps[key] = (_grow(int(sz[:, 0].max()), 16), _grow(int(sz[:, 1].max()), 4096))
    # The merged table WITHOUT moving a payload word: the gathered buffer itself is the payload, every instance's words stay
    # where its rank put them and the merged header carries their position (``offsets``).  Only the header rows (40 bytes
    # each, already on the host) are ordered by unit id -- a stable sort of (rank, local order).  Eight ranks x ~2 800
    # instances: ~4 ms of numpy; permuting the ~18 MB of payload took 60+ ms per exchange (gloo world-8 test).
    slot = int(buf.numel())
    hn = np.concatenate([head[r, S: S + int(sz[r, 0]) * HDR].reshape(-1, HDR) for r in range(world)], axis=0)
    offs = np.concatenate([_offsets(_payload_lengths(head[r, S: S + int(sz[r, 0]) * HDR].reshape(-1, HDR))) + (r * slot + S + cn * HDR)
                           for r in range(world)]) if hn.shape[0] else np.zeros((0,), dtype=np.int64)
    order = np.argsort(hn[:, 0], kind="stable")
    if not np.array_equal(order, np.arange(hn.shape[0])):
        hn, offs = np.ascontiguousarray(hn[order]), offs[order]
    st.last_merge_ms = (time.perf_counter() - t_host) * 1e3
    return GlobalTable(torch.from_numpy(hn).to(header.device), gathered.view(-1).to(payload.device), hn, offs, head[:, 2].astype(np.int64))


def reset_capacity() -> None:
    """Forget the default state's reserved capacities (tests; a new job whose tables have nothing to do with the last one's)."""
    _default_state.reset()


def _merge_tables(headers: List[torch.Tensor], payloads: List[torch.Tensor], host: Optional[List[np.ndarray]] = None):
    """Concatenate rank tables and order instances by unit id (stable: rank, then local order within a unit).
    Round-robin / blocked unit assignment usually leaves the concatenation already ordered; otherwise the payload
    segments are permuted with one index gather (the index built without a Python loop over instances: eight ranks x
    ~2 800 instances per step).  ``host``: the headers' host copies when the caller already has them (no second
    device-to-host wait).  -> (header, payload, the merged header's host copy)."""
    hcat = torch.cat(headers, dim=0) if len(headers) > 1 else headers[0]
    pcat = torch.cat(payloads, dim=0) if len(payloads) > 1 else payloads[0]
    hn = np.concatenate(host, axis=0) if host is not None else hcat.cpu().numpy()
    n = hn.shape[0]
    if n == 0:
        return hcat, pcat, hn
    order = np.argsort(hn[:, 0], kind="stable")                   # concatenation order = (rank, local index): stable sort keeps it
    if np.array_equal(order, np.arange(n)):
        return hcat, pcat, hn
    lens = _payload_lengths(hn)
    offs = _offsets(lens)
    lo = lens[order]
    total = int(lo.sum())
    idx = (np.repeat(offs[order] - _offsets(lo), lo) + np.arange(total, dtype=np.int64)) if total else np.zeros((0,), dtype=np.int64)
    ot = torch.from_numpy(order).to(hcat.device)
    return hcat[ot].contiguous(), pcat[torch.from_numpy(idx).to(pcat.device)].contiguous(), np.ascontiguousarray(hn[order])
