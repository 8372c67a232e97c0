"""P32 activations: the HBM format of the f16x2 product path (see ``csrc/conv_p32.hip``).

A tensor of pixels x C channels (C % 32 == 0) is kept as two fp16 planes of ``x * s`` -- ``s`` an exact power of two,
``x * s = h + l`` with ``h = half(x s)``, ``l = half(x s - h)`` (22 significand bits) -- laid out so that the 32
channels of one group of one pixel are ONE 128-byte line: ``[pixels][C / 32][2][32]`` fp16, preceded by a 128-byte zero
header that padding taps gather.  ``meta`` holds device floats ``[groups][2]``: per scale group the measured ``max |x|``
(accumulated by the producing kernel) and ``s`` (written by the producing kernel).  The engine uses one group per IMAGE
of a batch (``pixels / groups`` consecutive pixels each): the planes of an image -- and with them every result -- are
then the same whatever else is in the batch.

The conversions in this module are torch ops: they serve the tests, the debug dumps and the host-side packing of
constants -- the product path never converts, its kernels read and write P32 directly.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

import torch

HEADER_HALFS = 64          # 128 zero bytes


@dataclass
class P32:
    buf: torch.Tensor      # fp16, 1-D: HEADER_HALFS + pixels * C * 2
    meta: torch.Tensor     # f32 [groups, 2]: max |x|, s of each scale group
    shape: Tuple[int, ...]  # logical (..., C), product of the leading dims = pixels

    @property
    def pixels(self) -> int:
        n = 1
        for d in self.shape[:-1]:
            n *= int(d)
        return n

    @property
    def channels(self) -> int:
        return int(self.shape[-1])

    @property
    def groups(self) -> int:
        return int(self.meta.shape[0])

    def view(self, *shape) -> "P32":
        """Same bytes under another logical shape (channel counts must stay multiples of 32 and keep the group order,
        e.g. [R, 7, 7, 256] -> [R, 1, 1, 12544] or [M, 1024] -> [4 M, 256])."""
        n = 1
        for d in shape:
            n *= int(d)
        assert n == self.pixels * self.channels and shape[-1] % 32 == 0, (shape, self.shape)
        return P32(self.buf, self.meta, tuple(int(d) for d in shape))


def alloc(shape, device, meta: torch.Tensor | None = None, groups: int = 1) -> P32:
    n = 1
    for d in shape:
        n *= int(d)
    assert shape[-1] % 32 == 0, shape
    buf = torch.empty(HEADER_HALFS + 2 * n, dtype=torch.float16, device=device)
    buf[:HEADER_HALFS].zero_()
    if meta is None:
        meta = torch.zeros((groups, 2), dtype=torch.float32, device=device)
    return P32(buf, meta, tuple(int(d) for d in shape))


def plane_scale(bound: float) -> float:
    """2^(14 - ilogb(bound)): brings |x| <= bound below 2^15 (the rule of the kernels' epilogues)."""
    import math
    if not (bound > 0.0) or math.isinf(bound):
        return 1.0
    return math.ldexp(1.0, 14 - (math.frexp(bound)[1] - 1))


def from_f32(x: torch.Tensor, bound: float | None = None, groups: int = 1) -> P32:
    """[..., C] f32 -> P32 with s from ``bound`` (default: each group's own max |x|); ``groups`` equal runs of pixels."""
    x = x.to(torch.float32)
    c = int(x.shape[-1])
    xg = x.reshape(groups, -1)
    amax = xg.abs().amax(dim=1) if x.numel() else torch.zeros(groups, device=x.device)
    s = torch.tensor([plane_scale(float(a) if bound is None else bound) for a in amax.tolist()], dtype=torch.float32, device=x.device)
    y = (xg * s[:, None]).reshape(-1, c // 32, 32)
    h = y.to(torch.float16)
    l = (y - h.to(torch.float32)).to(torch.float16)
    out = alloc(tuple(x.shape), x.device, groups=groups)
    out.buf[HEADER_HALFS:] = torch.stack([h, l], dim=2).reshape(-1)
    out.meta[:, 0] = amax
    out.meta[:, 1] = s
    return out


def to_f32(t: P32) -> torch.Tensor:
    c = t.channels
    v = t.buf[HEADER_HALFS:HEADER_HALFS + 2 * t.pixels * c].view(t.pixels, c // 32, 2, 32).to(torch.float32)
    x = (v[:, :, 0, :] + v[:, :, 1, :]).reshape(t.groups, -1) / t.meta[:, 1:2]
    return x.reshape(t.shape)
