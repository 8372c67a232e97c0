"""Host driver of the MI355X Mask R-CNN forward: packs Detectron2 weights into the NHWC /
K-contiguous layouts the HIP kernels want and sequences the C-ABI calls on one HIP stream.

This is what replaces ``DefaultPredictor.__call__`` -> ``GeneralizedRCNN.inference``
(Detectron2 0.6) behind the reference's ``predictor(image)`` call
(``src/functions/inference.py:1395,1398,1507,1669``; built at ``src/data/models.py:107``).
Semantics follow SURVEY.md Appendix A; every compute step is a HIP kernel from
``libdeepemia_hip.so`` -- torch is used for device buffers and the stream only.

Layout in HBM (per batch of B equally sized images):
  activations  NHWC, bf16 (default) or f32 (parity mode), B outermost so B tiles form ONE
               implicit-GEMM M dimension (B*Ho*Wo pixels) -- this is what fills 256 CUs on
               the 25^2..200^2 feature maps;
  weights      [CoutPad, KH, KW, Cin] (K contiguous per output channel), FrozenBN kept as
               per-channel f32 scale/bias applied in the conv epilogue;
  masks        bit-packed, 1 bit per pixel ([B, D, H, W/32] u32): 512 KiB per 2048^2 mask.
"""
from __future__ import annotations

import os

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, p32
from ._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, BF16, BF16X2, F16X2, F32, F32X3, RES_NONE, RES_SAME, RES_UP2

RES_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}
PIXEL_MEAN = (103.530, 116.280, 123.675)
ANCHOR_SIZES = (32, 64, 128, 256, 512)
ANCHOR_RATIOS = (0.5, 1.0, 2.0)
STRIDES = (4, 8, 16, 32, 64)
# convolution stages that can be run single-plane one at a time (MaskRCNNEngine(single_stages=...))
# ... and the ones that run single-plane by DEFAULT in the f16x2 path: none.  The per-stage map (DESIGN.md section 7,
# profiles/r04_precision_map.json) shows that only the mask head (4 x conv3x3 + the deconv GEMM, 22 % of a tile's FLOPs) can run
# on one MFMA per product without changing an instance list -- it feeds nothing but the 0.5 threshold of the paste -- and on the
# eight headline tiles it keeps the parity record (799 / 800 masks at IoU >= 0.999, +8 % tiles/s).  But the headline weights carry a
# mask-predictor bias that keeps pixels away from the threshold; on the softer masks of the CLI parity cases
# (tests/test_gpu_pipeline_e2e.py, ensemble R50 + R101) an 11-bit mask head left one instance 10 pixels of 2016 away from the
# oracle's (IoU 0.995 < 0.999).  Parity is the first gate, so it is an OPT-IN: MaskRCNNEngine(single_stages=("mask_fcn", "deconv")),
# DEEPEMIA_SINGLE_STAGES=mask_fcn,deconv, bench.py --single-stages mask_fcn,deconv (flagged in the bench line).
DEFAULT_SINGLE_STAGES = ()
MASK_HEAD_STAGES = ("mask_fcn", "deconv")
STAGES = ("res2", "res3", "res4", "res5", "fpn_lateral", "fpn_output", "rpn_conv", "rpn_pred", "fc1", "fc2", "box_pred", "mask_fcn", "deconv")
BN_EPS = 1e-5
PRE_NMS_TOPK = 1000
POST_NMS_TOPK = 1000
RPN_NMS_THRESH = 0.7
DET_NMS_THRESH = 0.5
DETS_PER_IMAGE = 100

EXPECTED_IGNORED_PREFIXES = ("proposal_generator.anchor_generator.", "pixel_mean", "pixel_std")


def resize_shape(h: int, w: int, short: int = 800, max_size: int = 1333) -> Tuple[int, int]:
    """``ResizeShortestEdge.get_output_shape`` (Detectron2 0.6)."""
    scale = short * 1.0 / min(h, w)
    if h < w:
        newh, neww = short, scale * w
    else:
        newh, neww = scale * h, short
    if max(newh, neww) > max_size:
        s = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * s, neww * s
    return int(newh + 0.5), int(neww + 0.5)


def pil_bilinear_tables(in_size: int, out_size: int):
    """Pillow ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the triangle filter
    (PRECISION_BITS = 22).  Host-side table; the separable integer passes run on the GPU."""
    if in_size == out_size:
        return (np.arange(out_size, dtype=np.int32), np.ones(out_size, dtype=np.int32),
                np.full((out_size, 1), 1 << 22, dtype=np.int32))
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size)
    xsize = xmax - xmin
    j = np.arange(ksize, dtype=np.float64)[None, :]
    arg = (j + xmin[:, None] - center[:, None] + 0.5) * (1.0 / filterscale)
    wgt = np.where(np.abs(arg) < 1.0, 1.0 - np.abs(arg), 0.0)
    wgt = np.where(j < xsize[:, None], wgt, 0.0)
    # Pillow accumulates ww sequentially in double; cumulative sum reproduces that order
    ww = np.zeros(out_size, dtype=np.float64)
    for c in range(ksize):
        ww = ww + wgt[:, c]
    kk = np.where(ww[:, None] != 0.0, wgt / np.where(ww == 0.0, 1.0, ww)[:, None], wgt)
    ik = np.where(kk < 0, np.trunc(-0.5 + kk * (1 << 22)), np.trunc(0.5 + kk * (1 << 22))).astype(np.int32)
    return xmin.astype(np.int32), xsize.astype(np.int32), np.ascontiguousarray(ik)


def cv_linear_tables(in_size: int, out_size: int, vertical: bool = False):
    """OpenCV ``resize(INTER_LINEAR)`` 8-bit tables: source index pair and 11-bit coefficient pair per
    output coordinate.  ``f = (float)((d + 0.5) * scale - 0.5)``; horizontally an index outside the
    row zeroes the fraction and clamps, vertically the fraction is kept and only the ROW indices are
    clipped (``resizeGeneric_`` / ``resizeGeneric_Invoker``); coefficients = ``saturate_cast<short>(c * 2048)``."""
    inv_scale = float(out_size) / float(in_size)
    scale = 1.0 / inv_scale
    d = np.arange(out_size, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if vertical:
        s0 = np.clip(s, 0, in_size - 1)
        s1 = np.clip(s + 1, 0, in_size - 1)
    else:
        neg = s < 0
        f[neg] = 0.0
        s[neg] = 0
        hi = s >= in_size - 1
        f[hi] = 0.0
        s[hi] = in_size - 1
        s0, s1 = s, np.minimum(s + 1, in_size - 1)
    ofs = np.stack([s0, s1], axis=1).astype(np.int32)
    c = np.stack([np.float32(1.0) - f, f], axis=1).astype(np.float32) * np.float32(2048.0)
    coef = np.clip(np.rint(c), -32768, 32767).astype(np.int16)
    return np.ascontiguousarray(ofs), np.ascontiguousarray(coef)


def split3_bf16(w: torch.Tensor) -> torch.Tensor:
    """x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (round to nearest even, as the
    device cast does): the three planes carry the full 24-bit significand of an f32 value."""
    w = w.to(torch.float32)
    h = w.to(torch.bfloat16)
    r1 = w - h.to(torch.float32)
    m = r1.to(torch.bfloat16)
    l = (r1 - m.to(torch.float32)).to(torch.bfloat16)
    return torch.stack([h, m, l], dim=0)


def tile_weight_planes(w3: torch.Tensor, bk: int = 32) -> torch.Tensor:
    """[NP, CoutPad, KH, KW, Cin] 2-byte planes -> the layout ``demia_conv2d_nhwc`` reads for DEMIA_F32X3 / DEMIA_BF16X2
    (bk = 32) and DEMIA_F16X2 (bk = 64): [CoutPad / 64, ksteps, NP, 64, bk] with K = (kh, kw, cin) walked in steps of
    bk, so that the 64 x bk piece of one plane that a K-step needs is contiguous (``include/deepemia_hip.h``)."""
    npl, cout_pad = int(w3.shape[0]), int(w3.shape[1])
    k = w3[0, 0].numel()
    assert cout_pad % 64 == 0 and k % bk == 0, (cout_pad, k, bk)
    return w3.reshape(npl, cout_pad // 64, 64, k // bk, bk).permute(1, 3, 0, 2, 4).contiguous()


def tile_weight_planes_p32(planes: torch.Tensor) -> torch.Tensor:
    """[2, CoutPad, KH, KW, Cin] fp16 planes -> the layout ``demia_conv2d_p32`` streams by LDS-DMA:
    [CoutPad / 64, ksteps, 64, 2, 32] with K walked channel-group OUTER, tap inner (``include/deepemia_hip.h``): the 64
    rows x (32 high + 32 low halves) of one K-step are 8 KiB contiguous, one 128-byte line per output channel."""
    _, cout_pad, kh, kw, cin = (int(d) for d in planes.shape)
    assert cout_pad % 64 == 0 and cin % 32 == 0, (cout_pad, cin)
    t = planes.reshape(2, cout_pad // 64, 64, kh * kw, cin // 32, 32)
    return t.permute(1, 4, 3, 2, 0, 5).contiguous()        # [n64, cg, tap, 64, plane, 32]


def split2_f16_scaled(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """f16x2 weights: per output channel (dim 0) an exact power-of-two scale ``sw`` that brings max |w| into
    [2^14, 2^15), then ``w * sw = h + l`` with h = half(w * sw), l = half(w * sw - h) (22 significand bits).
    Returns (planes [2, ...] fp16, sw [Cout] f32); the caller divides the layer's output scale by ``sw``."""
    w = w.to(torch.float32)
    amax = w.abs().flatten(1).amax(dim=1)
    _, ex = torch.frexp(amax)                       # amax = m * 2^ex, m in [0.5, 1)
    sw = torch.ldexp(torch.ones_like(amax), 15 - ex)
    ws = w * sw.view(-1, *([1] * (w.dim() - 1)))
    h = ws.to(torch.float16)
    l = (ws - h.to(torch.float32)).to(torch.float16)
    return torch.stack([h, l], dim=0), sw


def stem_weight_planes(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Stem weights [64, 3, 7, 7] for ``demia_stem_conv_mfma``: K ordered (kh, kw, c) with kw padded to 8 and c to 4 (zero
    weights), one K-step of 32 per kernel row; per output channel an exact power of two brings max |w| into [2^14, 2^15), then
    ``w * 2^e = h + l`` in fp16.  Returns (planes [2, 7, 64, 32] fp16, the per-channel factors [64] f32)."""
    w = w.to(torch.float32)
    cout = w.shape[0]
    k = torch.zeros((cout, 7, 8, 4), dtype=torch.float32)
    k[:, :, :7, :3] = w.permute(0, 2, 3, 1)                       # [co, kh, kw, c]
    planes, sw = split2_f16_scaled(k.reshape(cout, 7 * 32))
    planes = planes.reshape(2, cout, 7, 32).permute(0, 2, 1, 3).contiguous()      # [plane, kh, co, 32]
    return planes, sw


def cell_anchor_table() -> np.ndarray:
    out = np.zeros((5, 3, 4), dtype=np.float32)
    for l, size in enumerate(ANCHOR_SIZES):
        area = size ** 2.0
        for a, r in enumerate(ANCHOR_RATIOS):
            w = math.sqrt(area / r)
            h = r * w
            out[l, a] = (-w / 2.0, -h / 2.0, w / 2.0, h / 2.0)
    return out


@dataclass
class ConvLayer:
    w: torch.Tensor
    scale: Optional[torch.Tensor]
    bias: Optional[torch.Tensor]
    cin: int
    cout: int
    cout_pad: int
    kh: int
    kw: int
    stride: int
    pad: int
    w3: Optional[torch.Tensor] = None   # f32x3 / bf16x2 mode: the three / two bf16 planes of w, tiled [CoutPad/64, ksteps, NP, 64, 32]
    scale3: Optional[torch.Tensor] = None   # f16x2: the output scale divided by the per-channel weight scale
    wbound: float = 0.0                 # f16x2: max_co(|scale_co| * sum_k |w_co,k|) and max |bias| -- the a-priori bound of
    bbound: float = 0.0                 #        |out| from which the epilogue derives the scale of its P32 output
    single: int = 0                     # demia_conv_p32_desc.single of this layer's launches (0 = three MFMAs per product)


@dataclass
class RawDetections:
    """Device-resident result of one batched forward (nothing has crossed PCIe yet)."""
    boxes: torch.Tensor      # [B, D, 4] f32 output-image coords
    scores: torch.Tensor     # [B, D] f32
    classes: torch.Tensor    # [B, D] i32
    valid: torch.Tensor      # [B, D] u8   (i < count and box non-empty)
    count: torch.Tensor      # [B] i32
    packed: torch.Tensor     # [B, D, H, W/32] i32 (bit-packed masks)
    height: int
    width: int
    bbox: Optional[torch.Tensor] = None   # [B, D, 4] i32: the box each paste could write (superset of the tight mask bbox)


class MaskRCNNEngine:
    def __init__(self, state_dict: Dict[str, torch.Tensor], depth: int, num_classes: int, score_thresh: float,
                 device: str = "cuda:0", precision: str = "f16x2", min_size_test: int = 800, max_size_test: int = 1333,
                 single_stages: Optional[Sequence[str]] = None):
        if depth not in RES_BLOCKS:
            raise ValueError(f"unsupported ResNet depth {depth}")
        if not torch.cuda.is_available():
            raise _lib.HipExtensionMissing("no HIP device visible: the deepEMIA hot path has no CPU fallback")
        self.lib = _lib.load()
        self.depth = depth
        self.K = int(num_classes)
        self.score_thresh = float(score_thresh)
        # demia_box_detections keeps at most 4096 above-threshold candidates per image in LDS; a softmax row has at most
        # floor(1 / thresh) of them, so every class count works for thresholds above 0.2 (the reference's default is 0.65)
        per_row = min(self.K, int(1.0 / self.score_thresh)) if self.score_thresh > 0 else self.K
        if self.score_thresh <= 0 or POST_NMS_TOPK * per_row > 4096:
            raise ValueError(f"{self.K} classes at score threshold {self.score_thresh}: up to {POST_NMS_TOPK} x {per_row} candidates per "
                             f"image exceed the 4096 the detection kernel sorts -- raise --threshold above {1.0 / (4096 // POST_NMS_TOPK + 1):.2f} "
                             f"or use at most {4096 // POST_NMS_TOPK} classes")
        self.device = torch.device(device)
        self.precision = precision
        # INPUT.MIN_SIZE_TEST / MAX_SIZE_TEST of the model-zoo config the reference loads (800 / 1333, never overridden
        # there: models.py:134-144).  Other values are the flagged "native resolution" mode of SURVEY 8(f)4.
        self.min_size_test, self.max_size_test = int(min_size_test), int(max_size_test)
        if self.min_size_test < 32 or self.max_size_test < self.min_size_test:
            raise ValueError(f"min_size_test / max_size_test = {min_size_test} / {max_size_test}")
        if precision not in ("f32", "f32x3", "f16x2", "f16", "f16x2r", "bf16x2", "bf16"):
            raise ValueError("precision must be 'f16x2' (default: f32-sized error on the fp16 pipe, activations kept as two "
                             "pre-scaled fp16 planes in HBM, LDS-DMA fed kernel), 'f16x2r' (the same arithmetic with f32 "
                             "activations split in the K loop: round 1's kernel), 'f32' (exact-f32 MFMA), 'f32x3' (f32 on the "
                             "bf16 pipe, 3-way split), 'bf16x2' (16-bit operands on the bf16 pipe), 'bf16', or 'f16' (flagged NON-parity: "
                             "the f16x2 path with single-plane fp16 operands, one MFMA per product -- the reference's autocast "
                             "arithmetic, inference.py:1390-1395)")
        if precision not in ("f16x2", "f16", "f32") and not _lib.is_dev_build():
            raise ValueError(f"precision '{precision}' needs the dev build of the library (make -C deepemia_amd/csrc DEV=1, then "
                             f"DEEPEMIA_DEV_LIB=1): the product library computes f16x2 (default), f16 and exact f32 only")
        self.p32 = precision in ("f16x2", "f16")          # activations travel as P32 planes (deepemia_amd/p32.py)
        self.single_plane = precision == "f16"            # ... with a zero low plane, every layer (conv desc `single` = 2)
        # per-STAGE single-plane arithmetic inside the f16x2 path (the precision map): the convolutions of
        # the named stages issue ONE MFMA per product on the high planes (conv desc `single` = 1); their outputs keep both planes
        if single_stages is None:
            env = os.environ.get("DEEPEMIA_SINGLE_STAGES")
            single_stages = () if precision != "f16x2" else (DEFAULT_SINGLE_STAGES if env is None else tuple(t for t in env.split(",") if t))
        unknown = set(single_stages) - set(STAGES)
        if unknown or (single_stages and precision != "f16x2"):
            raise ValueError(f"single_stages {sorted(unknown)}: choose from {STAGES}, with precision 'f16x2'")
        self.single_stages = frozenset(single_stages)
        self._amax_buf: Optional[torch.Tensor] = None     # f16x2r: per-forward pool of |activation| bounds
        self._amax_i = 0
        self._meta_pool: Optional[torch.Tensor] = None    # f16x2: {max |x|, s} per activation tensor and image, zeroed once per forward
        self._meta_pools: Dict[tuple, torch.Tensor] = {}  # one pool per input shape (captured graphs keep pointers into theirs)
        self._groups = 1
        self._meta_i = 0
        self._graphs: Dict[tuple, dict] = {}              # captured forwards per input shape (forward_graphed)
        self._arena: Dict[tuple, list] = {}               # intermediate buffers per input shape, reused by later forwards
        self._arena_key: Optional[tuple] = None
        self._arena_i = 0
        self._paste_static = None                         # (planes, previous boxes) while a forward is being captured
        # f16x2 path: the stem on the matrix pipe, fused with the max pool (DEEPEMIA_STEM = fused | mfma | valu for A/B)
        self.stem_mode = os.environ.get("DEEPEMIA_STEM", "fused") if self.p32 else "valu"
        self.stem_on_mfma = self.stem_mode in ("fused", "mfma")
        # arenas, meta pools and captured graphs are per input shape and GBs each (16 x 2048^2 R101: 8.6 GiB): at most this
        # many shapes stay resident, least recently used first out -- a folder of differently sized micrographs must not
        # accumulate one arena per size (the reference handles arbitrary sizes, inference.py:2299-2485)
        self.max_cached_shapes = max(1, int(os.environ.get("DEEPEMIA_MAX_CACHED_SHAPES", "3")))
        self._shape_lru: List[tuple] = []
        self.evictions = 0
        self.dt = BF16 if precision == "bf16" else F32
        self.tdt = torch.bfloat16 if precision == "bf16" else torch.float32
        self._tables: Dict[Tuple[int, int], dict] = {}
        self._cell = cell_anchor_table()
        # (A/B switch, default OFF) ROIAlign workgroups in spatial order per XCD (demia_roi_order).  Measured in round 5 on a 48-tile
        # forward: L2 hit rate of roi_align_kernel 33.5 % -> 50.9 % (TCC_HIT / TCC_MISS), kernel time 1 079 -> 1 108 us per launch --
        # the misses were Infinity-Cache hits at the same cost; the kernel is bound by bytes through L1, not by where they come from
        self.roi_order = os.environ.get("DEEPEMIA_ROI_ORDER", "0") == "1"
        self.conv_events = None   # bench hook: list of (start_event, end_event, algorithmic_flops, kernel kind, algorithmic_bytes)
        self.unmatched_keys: List[str] = []
        self._used = set()
        self._pack(state_dict)

    # ------------------------------------------------------------------ weight packing
    def _get(self, sd, key):
        if key not in sd:
            raise KeyError(f"checkpoint is missing '{key}' (expected Detectron2 0.6 Mask R-CNN R{self.depth}-FPN layout)")
        self._used.add(key)
        return sd[key].detach().to(torch.float32).cpu()

    def _conv(self, sd, prefix, stride=1, pad=0, norm=False, bias=False, weight=None, bias_t=None, stage: str = "") -> ConvLayer:
        w = self._get(sd, prefix + ".weight") if weight is None else weight
        cout, cin, kh, kw = w.shape
        cout_pad = (cout + 63) // 64 * 64 if self.p32 else (cout + 31) // 32 * 32
        wp = torch.zeros((cout_pad, kh, kw, cin), dtype=torch.float32)
        wp[:cout] = w.permute(0, 2, 3, 1)
        scale = b = None
        if norm:
            g = self._get(sd, prefix + ".norm.weight")
            beta = self._get(sd, prefix + ".norm.bias")
            rm = self._get(sd, prefix + ".norm.running_mean")
            rv = self._get(sd, prefix + ".norm.running_var")
            scale = g * (rv + BN_EPS).rsqrt()
            b = beta - rm * scale
        elif bias:
            b = self._get(sd, prefix + ".bias") if bias_t is None else bias_t
        dev = self.device
        if self.p32:
            # f16x2: every layer runs on demia_conv2d_p32 -- two fp16 planes of w * 2^e(co), tiled for the LDS-DMA stream;
            # the per-channel power of two is divided out of the epilogue scale; wbound / bbound = the a-priori bound of |out|
            if cin % 32:
                raise ValueError(f"{prefix or 'layer'}: Cin = {cin} is not a multiple of 32")
            planes, sw = split2_f16_scaled(wp.to(dev))
            if self.single_plane:
                planes[1].zero_()                             # weights rounded to ONE fp16 plane
            base = torch.ones(cout, dtype=torch.float32) if scale is None else scale
            l1 = w.abs().flatten(1).sum(1)
            return ConvLayer(None, None if scale is None else scale.to(dev).contiguous(), None if b is None else b.to(dev).contiguous(),
                             cin, cout, cout_pad, kh, kw, stride, pad, tile_weight_planes_p32(planes),
                             (base.to(dev) / sw[:cout]).contiguous(), float((base.abs() * l1).max()),
                             0.0 if b is None else float(b.abs().max()), 2 if self.single_plane else int(stage in self.single_stages))
        w3 = None
        if self.precision in ("f32x3", "bf16x2") and cout_pad % 64 == 0 and cin % 32 == 0:
            w3 = split3_bf16(wp.to(dev))                       # split on the device: same round-to-nearest casts
            w3 = tile_weight_planes(w3 if self.precision == "f32x3" else w3[:2])
        scale3 = None
        f16_bk = int(self.lib.demia_conv_f16x2_kstep())           # the tiling this build of the library reads
        if self.precision == "f16x2r" and cout_pad % 64 == 0 and cin % f16_bk == 0:
            planes, sw = split2_f16_scaled(wp.to(dev))
            w3 = tile_weight_planes(planes, f16_bk)
            base = torch.ones(cout, dtype=torch.float32, device=dev) if scale is None else scale.to(dev)
            scale3 = (base / sw[:cout]).contiguous()
        return ConvLayer(wp.to(dev, self.tdt).contiguous(),
                         None if scale is None else scale.to(dev).contiguous(),
                         None if b is None else b.to(dev).contiguous(), cin, cout, cout_pad, kh, kw, stride, pad, w3, scale3)

    def _pack(self, sd):
        bu = "backbone.bottom_up."
        w = self._get(sd, bu + "stem.conv1.weight")  # [64, 3, 7, 7]
        ws = torch.zeros((7, 8, 4, 64), dtype=torch.float32)
        ws[:, :7, :3, :] = w.permute(2, 3, 1, 0)
        g = self._get(sd, bu + "stem.conv1.norm.weight")
        beta = self._get(sd, bu + "stem.conv1.norm.bias")
        rm = self._get(sd, bu + "stem.conv1.norm.running_mean")
        rv = self._get(sd, bu + "stem.conv1.norm.running_var")
        sc = g * (rv + BN_EPS).rsqrt()
        self.stem_w = ws.to(self.device).contiguous()
        self.stem_scale = sc.to(self.device).contiguous()
        self.stem_bias = (beta - rm * sc).to(self.device).contiguous()
        # |pixel - mean| <= 255 - min(mean): the a-priori bound of the stem's (pooled) output, from which the P32 scale of the
        # first activation tensor is derived on the host
        self.stem_bound = float(((255.0 - min(PIXEL_MEAN)) * sc.abs() * w.abs().flatten(1).sum(1) + (beta - rm * sc).abs()).max())
        if self.p32:
            # the stem on the matrix pipe (demia_stem_conv_mfma): weight planes, the input's plane scale (|pixel - mean| <=
            # 255 - min(mean): a constant, so results do not depend on the image or the batch), FrozenBN scale with both powers
            # of two divided out
            planes, sw = stem_weight_planes(w)
            if self.single_plane:
                planes[1].zero_()
            self.stem_planes = planes.to(self.device).contiguous()
            self.stem_s_in = p32.plane_scale(255.0 - min(PIXEL_MEAN))
            self.stem_scale_mfma = (sc / (sw * self.stem_s_in)).to(self.device).contiguous()
        self.blocks = []
        for stage, nblk in zip((2, 3, 4, 5), RES_BLOCKS[self.depth]):
            stage_blocks = []
            for i in range(nblk):
                p = f"{bu}res{stage}.{i}."
                stride = 2 if (i == 0 and stage > 2) else 1
                blk = {}
                if (p + "shortcut.weight") in sd:
                    blk["shortcut"] = self._conv(sd, p + "shortcut", stride=stride, norm=True, stage=f"res{stage}")
                blk["conv1"] = self._conv(sd, p + "conv1", stride=stride, norm=True, stage=f"res{stage}")
                blk["conv2"] = self._conv(sd, p + "conv2", pad=1, norm=True, stage=f"res{stage}")
                blk["conv3"] = self._conv(sd, p + "conv3", norm=True, stage=f"res{stage}")
                stage_blocks.append(blk)
            self.blocks.append(stage_blocks)
        self.fpn_lateral = {l: self._conv(sd, f"backbone.fpn_lateral{l}", bias=True, stage="fpn_lateral") for l in (2, 3, 4, 5)}
        self.fpn_output = {l: self._conv(sd, f"backbone.fpn_output{l}", pad=1, bias=True, stage="fpn_output") for l in (2, 3, 4, 5)}
        rp = "proposal_generator.rpn_head."
        self.rpn_conv = self._conv(sd, rp + "conv", pad=1, bias=True, stage="rpn_conv")
        wobj, wdel = self._get(sd, rp + "objectness_logits.weight"), self._get(sd, rp + "anchor_deltas.weight")
        bobj, bdel = self._get(sd, rp + "objectness_logits.bias"), self._get(sd, rp + "anchor_deltas.bias")
        self.rpn_pred = self._conv(sd, "", bias=True, weight=torch.cat([wobj, wdel], 0), bias_t=torch.cat([bobj, bdel], 0), stage="rpn_pred")
        bh = "roi_heads.box_head."
        w1 = self._get(sd, bh + "fc1.weight")
        w1 = w1.view(w1.shape[0], 256, 7, 7).permute(0, 2, 3, 1).reshape(w1.shape[0], 12544, 1, 1)
        self.fc1 = self._conv(sd, "", bias=True, weight=w1, bias_t=self._get(sd, bh + "fc1.bias"), stage="fc1")
        w2 = self._get(sd, bh + "fc2.weight")
        self.fc2 = self._conv(sd, "", bias=True, weight=w2[:, :, None, None], bias_t=self._get(sd, bh + "fc2.bias"), stage="fc2")
        bp = "roi_heads.box_predictor."
        wc, wb = self._get(sd, bp + "cls_score.weight"), self._get(sd, bp + "bbox_pred.weight")
        bc, bb = self._get(sd, bp + "cls_score.bias"), self._get(sd, bp + "bbox_pred.bias")
        if wc.shape[0] != self.K + 1 or wb.shape[0] != 4 * self.K:
            raise ValueError(f"checkpoint has {wc.shape[0] - 1} classes, dataset metadata says {self.K}")
        self.box_pred = self._conv(sd, "", bias=True, weight=torch.cat([wc, wb], 0)[:, :, None, None],
                                   bias_t=torch.cat([bc, bb], 0), stage="box_pred")
        mh = "roi_heads.mask_head."
        self.mask_fcn = [self._conv(sd, f"{mh}mask_fcn{i}", pad=1, bias=True, stage="mask_fcn") for i in range(1, 5)]
        wd = self._get(sd, mh + "deconv.weight")  # [Cin, Cout, 2, 2]
        wd = wd.permute(2, 3, 1, 0).reshape(4 * wd.shape[1], wd.shape[0], 1, 1)
        self.deconv = self._conv(sd, "", bias=True, weight=wd, bias_t=self._get(sd, mh + "deconv.bias").repeat(4), stage="deconv")
        self.mask_pred = self._conv(sd, mh + "predictor", bias=True)
        # f32 copy of the class predictor for the fused deconv + predictor launch of the f16x2 path (K <= 4 classes)
        self.mask_pred_w32 = self._get(sd, mh + "predictor.weight").reshape(-1, 256).to(self.device).contiguous()
        self.mask_pred_b32 = self._get(sd, mh + "predictor.bias").to(self.device).contiguous()
        self.unmatched_keys = sorted(k for k in sd if k not in self._used
                                     and not k.startswith(EXPECTED_IGNORED_PREFIXES))

    # ------------------------------------------------------------------ kernel helpers
    def _stream(self) -> int:
        return int(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- f16x2 bookkeeping: intermediate buffers and {max |x|, s} slots -------------------------------------------
    def _begin_forward(self, key: tuple, ph: int, pw: int) -> None:
        """Intermediates of one forward come from an arena keyed by the input shape: the first forward of a shape
        allocates them (P32 headers zeroed once), later ones reuse them in call order -- no allocation, no memset but the
        one that clears the meta pool."""
        self._touch_shape(key)
        self._arena_key, self._arena_i = key, 0
        # one {max |x|, s} pair per activation tensor and SCALE GROUP = image of the batch: an image's planes, and so its
        # results, do not depend on its batch neighbours.  The conv epilogue needs >= 128 rows per group; the smallest
        # tensor with one group per image is p6 (the RPN runs on it), so tiny inputs fall back to one group per tensor.
        b = key[0]
        self._groups = b if (b > 1 and ((ph // 32 - 1) // 2 + 1) * ((pw // 32 - 1) // 2 + 1) >= 128) else 1
        pool = self._meta_pools.get(key)
        if pool is None:
            pool = self._meta_pools[key] = torch.zeros((256, self._groups, 2), dtype=torch.float32, device=self.device)
        else:
            pool.zero_()
        self._meta_pool = pool
        self._meta_i = 0

    def _touch_shape(self, key: tuple) -> None:
        """LRU bookkeeping of the per-shape caches.  An eviction drops the shape's arena, meta pool and graphs TOGETHER
        (the graphs hold raw pointers into the other two) after a device synchronize -- it only happens when a new shape
        arrives, never inside a capture (the eager warm-up forward has touched the key before)."""
        lru = self._shape_lru
        if lru and lru[-1] == key:
            return
        if key in lru:
            lru.remove(key)
        lru.append(key)
        while len(lru) > self.max_cached_shapes:
            self._evict_shape(lru.pop(0))

    def _evict_shape(self, key: tuple) -> None:
        torch.cuda.synchronize(self.device)
        self._graphs.pop(key, None)
        self._arena.pop(key, None)
        self._meta_pools.pop(key, None)
        self.evictions += 1

    def release_cached_shapes(self) -> None:
        """Drop every per-shape cache (arenas, meta pools, graphs)."""
        while self._shape_lru:
            self._evict_shape(self._shape_lru.pop(0))
        self._arena_key = None

    def _scratch(self, numel: int, dtype, zero: bool = False, zero_head: int = 0) -> torch.Tensor:
        """Flat intermediate buffer number ``_arena_i`` of the current forward.  ``zero`` / ``zero_head``: zeroed (whole /
        first elements) when it is first allocated -- for buffers whose zero regions no kernel ever writes."""
        if self._arena_key is None:                    # a stage called on its own (tests): plain allocation
            t = torch.zeros(numel, dtype=dtype, device=self.device) if zero else torch.empty(numel, dtype=dtype, device=self.device)
            if zero_head and not zero:
                t[:zero_head].zero_()
            return t
        slots = self._arena.setdefault(self._arena_key, [])
        i = self._arena_i
        self._arena_i += 1
        if i < len(slots) and slots[i].numel() == numel and slots[i].dtype == dtype:
            return slots[i]
        t = torch.zeros(numel, dtype=dtype, device=self.device) if zero else torch.empty(numel, dtype=dtype, device=self.device)
        if zero_head and not zero:
            t[:zero_head].zero_()
        if i < len(slots):
            slots[i] = t
        else:
            slots.append(t)
        return t

    def _meta_slot(self, groups: int) -> torch.Tensor:
        if self._arena_key is None:                    # a stage called on its own (tests): plain allocation
            return torch.zeros((groups, 2), dtype=torch.float32, device=self.device)
        if groups != self._groups:                     # not a tensor of the forward the pool was laid out for (tests)
            return torch.zeros((groups, 2), dtype=torch.float32, device=self.device)
        if self._meta_i >= self._meta_pool.shape[0]:
            raise RuntimeError("activation meta pool exhausted inside one forward")
        m = self._meta_pool[self._meta_i]
        self._meta_i += 1
        return m

    def new_p32(self, shape, groups: Optional[int] = None) -> p32.P32:
        n = 1
        for d in shape:
            n *= int(d)
        buf = self._scratch(p32.HEADER_HALFS + 2 * n, torch.float16, zero_head=p32.HEADER_HALFS)
        return p32.P32(buf, self._meta_slot(self._groups if groups is None else groups), tuple(int(d) for d in shape))

    def dense(self, x) -> torch.Tensor:
        """An activation as a plain f32 tensor (tests / debug dumps; the product path never converts)."""
        return p32.to_f32(x) if isinstance(x, p32.P32) else x.float()

    def conv_p32(self, x: p32.P32, L: ConvLayer, act=ACT_NONE, residual: Optional[p32.P32] = None, res_mode=RES_NONE,
                 out_f32: bool = False, out_ld: int = 0, tile_hint: int = 0, head=None):
        """``demia_conv2d_p32``: P32 in, P32 out (or plain f32 [.., out_ld] for the prediction heads).  ``head`` =
        (weights [hn, 256] f32, bias [hn] f32, ld, activation): the fused 1x1 head -- returns f32 [pixels * Cout / 256, ld]
        instead of the layer's own output, which is never written."""
        n, h, w, cin = x.shape
        assert cin == L.cin, (cin, L.cin)
        ho = (h + 2 * L.pad - L.kh) // L.stride + 1
        wo = (w + 2 * L.pad - L.kw) // L.stride + 1
        hw_ptr = hb_ptr = ho_ptr = hn = hld = hact = 0
        if head is not None:
            hw_t, hb_t, hld, hact = head
            hn = int(hw_t.shape[0])
            out = self._scratch(n * ho * wo * (L.cout // 256) * hld, torch.float32).view(n * ho * wo * (L.cout // 256), hld)
            hw_ptr, hb_ptr, ho_ptr = _lib.ptr(hw_t), _lib.ptr(hb_t), _lib.ptr(out)
            optr, ometa = 0, 0
        elif out_f32:
            ld = out_ld if out_ld > 0 else L.cout
            out = self._scratch(n * ho * wo * ld, torch.float32).view(n, ho, wo, ld)
            optr, ometa = _lib.ptr(out), 0
        else:
            out = self.new_p32((n, ho, wo, L.cout), x.groups)
            optr, ometa = _lib.ptr(out.buf), _lib.ptr(out.meta)
        # the kernel addresses its input with 32-bit byte offsets: a pointwise layer over more than 4 GiB of planes (the mask
        # head's last two layers at 64 tiles per forward) goes in pixel chunks -- rows are independent, and without
        # padding taps the only rows that gather the 128 bytes in front of a chunk are the tail tile's rows >= M, whose
        # results are never stored
        pix_bytes = cin * 4
        limit = (1 << 32) - (1 << 20)
        chunks = [(0, n)]
        if 128 + n * h * w * pix_bytes >= limit:
            if not (h == 1 and w == 1 and L.kh == 1 and L.kw == 1 and L.pad == 0 and L.stride == 1 and res_mode != RES_UP2):
                raise ValueError(f"{n}x{h}x{w}x{cin} planes exceed 4 GiB: lower the batch size")
            step = (limit - 128) // pix_bytes // 256 * 256
            chunks = [(i, min(step, n - i)) for i in range(0, n, step)]
        ev = self.conv_events
        if ev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
        groups = x.groups
        group_rows = (n * ho * wo) // groups
        assert groups == 1 or (group_rows * groups == n * ho * wo and (residual is None or residual.groups == groups)), (x.shape, groups)
        for c0, cn in chunks:
            d = _lib.ConvP32Desc(_lib.ptr(x.buf) + c0 * pix_bytes, _lib.ptr(x.meta), _lib.ptr(L.w3), _lib.ptr(L.scale3), _lib.ptr(L.bias),
                                 0 if residual is None else _lib.ptr(residual.buf) + c0 * L.cout * 4,
                                 0 if residual is None else _lib.ptr(residual.meta),
                                 (optr + c0 * (ld if out_f32 else L.cout) * 4) if optr else 0, ometa, L.wbound, L.bbound, cn, h, w, cin,
                                 ho, wo, L.cout, L.cout_pad, L.kh, L.kw, L.stride, L.pad, act, res_mode, 1 if out_f32 else 0, out_ld,
                                 tile_hint, hw_ptr, hb_ptr, (ho_ptr + c0 * (L.cout // 256) * hld * 4) if ho_ptr else 0, hn, hld, hact,
                                 groups, group_rows, c0, L.single)
            _lib.check(self.lib.demia_conv2d_p32(C.byref(d), self._stream()), "demia_conv2d_p32")
        if ev is not None:
            e1.record(torch.cuda.current_stream(self.device))
            # every operand once (planes are 4 bytes per element, like f32): the algorithmic traffic
            nbytes = (n * h * w * cin * 4 + L.cout_pad * L.kh * L.kw * cin * 4 + n * ho * wo * L.cout * 4 +
                      (0 if residual is None else residual.pixels * residual.channels * 4))
            ev.append((e0, e1, 2.0 * n * ho * wo * L.cout * L.kh * L.kw * cin, "f16" if L.single else self.precision, nbytes))
        return out

    def conv(self, x, L: ConvLayer, act=ACT_NONE, residual=None, res_mode=RES_NONE,
             out_dtype=None, out: Optional[torch.Tensor] = None, out_ld: int = 0, tile_hint: int = 0):
        if self.p32:
            return self.conv_p32(x, L, act, residual, res_mode, out_f32=out_dtype is not None, out_ld=out_ld, tile_hint=tile_hint)
        n, h, w, cin = x.shape
        assert cin == L.cin, (cin, L.cin)
        ho = (h + 2 * L.pad - L.kh) // L.stride + 1
        wo = (w + 2 * L.pad - L.kw) // L.stride + 1
        odt = self.tdt if out_dtype is None else out_dtype
        ld = out_ld if out_ld > 0 else L.cout
        if out is None:
            out = torch.empty((n, ho, wo, ld), dtype=odt, device=self.device)
        use3 = L.w3 is not None and odt == torch.float32
        f16 = use3 and L.w3.dtype == torch.float16
        kind = ("f16x2r" if f16 else ("f32x3" if L.w3.shape[2] == 3 else "bf16x2")) if use3 else ("bf16" if self.dt == BF16 else "f32")
        amax_in = self.amax_of(x) if f16 else None
        amax_out = self._amax_slot() if self.precision == "f16x2r" else None
        d = _lib.ConvDesc(_lib.ptr(x), _lib.ptr(L.w3 if use3 else L.w), _lib.ptr(L.scale3 if f16 else L.scale), _lib.ptr(L.bias),
                          _lib.ptr(residual), _lib.ptr(out), n, h, w, cin, ho, wo, L.cout, L.cout_pad, L.kh, L.kw, L.stride, L.pad,
                          {"f16x2r": F16X2, "f32x3": F32X3, "bf16x2": BF16X2}.get(kind, self.dt), BF16 if odt == torch.bfloat16 else F32,
                          act, res_mode, ld, tile_hint, _lib.ptr(amax_in), _lib.ptr(amax_out))
        if amax_out is not None:
            out._amax = amax_out
        ev = self.conv_events
        if ev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
        _lib.check(self.lib.demia_conv2d_nhwc(C.byref(d), self._stream()), "demia_conv2d_nhwc")
        if ev is not None:
            e1.record(torch.cuda.current_stream(self.device))
            esz = 2 if self.dt == BF16 else 4
            osz = 2 if odt == torch.bfloat16 else 4
            nbytes = (n * h * w * cin * esz + L.cout_pad * L.kh * L.kw * cin * (2 * int(L.w3.shape[2]) if use3 else esz) + n * ho * wo * L.cout * osz +
                      (0 if residual is None else residual.numel() * osz))     # every operand once: the algorithmic traffic
            ev.append((e0, e1, 2.0 * n * ho * wo * L.cout * L.kh * L.kw * cin, kind, nbytes))
        return out

    # f16x2: every activation tensor carries a device scalar bounding |x| (``_amax``), accumulated by the producing conv's
    # epilogue or derived from its inputs' bounds; the consuming conv turns it into a power-of-two operand scale.
    def _amax_slot(self) -> torch.Tensor:
        if self._amax_buf is None or self._amax_i >= self._amax_buf.numel():
            self._amax_buf = torch.zeros(512, dtype=torch.float32, device=self.device)
            self._amax_i = 0
        slot = self._amax_buf[self._amax_i:self._amax_i + 1]
        self._amax_i += 1
        return slot

    def amax_of(self, x: torch.Tensor) -> torch.Tensor:
        am = getattr(x, "_amax", None)
        if am is None:                                  # not produced by a tracked kernel: one reduction pass
            mn, mx = torch.aminmax(x)
            am = torch.maximum(mx, -mn).to(torch.float32).reshape(1)
            x._amax = am
        return am

    @staticmethod
    def view_as(x, *shape):
        """``x.view(shape)`` that keeps the |x| bound attached."""
        if isinstance(x, p32.P32):
            return x.view(*shape)
        y = x.view(*shape)
        am = getattr(x, "_amax", None)
        if am is not None:
            y._amax = am
        return y

    def _resize_tables(self, h: int, w: int):
        key = (h, w)
        if key not in self._tables:
            newh, neww = resize_shape(h, w, self.min_size_test, self.max_size_test)
            xm, xs, xk = pil_bilinear_tables(w, neww)
            ym, ys, yk = pil_bilinear_tables(h, newh)
            t = lambda a: torch.from_numpy(a).to(self.device)
            self._tables[key] = dict(newh=newh, neww=neww, xm=t(xm), xs=t(xs), xk=t(xk), ksx=xk.shape[1],
                                     ym=t(ym), ys=t(ys), yk=t(yk), ksy=yk.shape[1], need_h=(w != neww))
        return self._tables[key]

    def resize_linear_u8(self, images: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
        """``cv2.resize(img, (out_w, out_h), INTER_LINEAR)`` for a [N, H, W, 3] u8 device batch."""
        n, h, w, _ = images.shape
        xo, xa = cv_linear_tables(w, out_w)
        yo, ya = cv_linear_tables(h, out_h, vertical=True)
        t = lambda a: torch.from_numpy(a).to(self.device)
        xo, xa, yo, ya = t(xo), t(xa), t(yo), t(ya)
        out = torch.empty((n, out_h, out_w, 3), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.demia_resize_linear_u8(_lib.ptr(images.contiguous()), _lib.ptr(out), n, h, w, out_h, out_w, _lib.ptr(xo),
                                                   _lib.ptr(xa), _lib.ptr(yo), _lib.ptr(ya), self._stream()),
                   "demia_resize_linear_u8")
        return out

    # ------------------------------------------------------------------ stages
    def preprocess(self, images: torch.Tensor):
        """[B, H, W, 3] u8 BGR (device) -> zero-bordered f32 stem input, (newh, neww, PH, PW)."""
        b, h, w, _ = images.shape
        t = self._resize_tables(h, w)
        newh, neww = t["newh"], t["neww"]
        ph, pw = (newh + 31) // 32 * 32, (neww + 31) // 32 * 32
        if self.p32:
            self._begin_forward((b, h, w), ph, pw)          # the first stage of every forward
        st = self._stream()
        if t["need_h"]:
            tmp = (self._scratch(b * h * neww * 3, torch.uint8).view(b, h, neww, 3) if self.p32 else
                   torch.empty((b, h, neww, 3), dtype=torch.uint8, device=self.device))
            _lib.check(self.lib.demia_resize_h_u8(_lib.ptr(images), _lib.ptr(tmp), b, h, w, neww, _lib.ptr(t["xm"]),
                                                  _lib.ptr(t["xs"]), _lib.ptr(t["xk"]), t["ksx"], st), "demia_resize_h_u8")
        else:
            tmp = images
        # border, padding and the fourth channel are zero and never written: zeroed once when the buffer is first allocated
        dst = (self._scratch(b * (ph + 6) * (pw + 8) * 4, torch.float32, zero=True).view(b, ph + 6, pw + 8, 4) if self.p32 else
               torch.zeros((b, ph + 6, pw + 8, 4), dtype=torch.float32, device=self.device))
        mean = (C.c_float * 3)(*PIXEL_MEAN)
        _lib.check(self.lib.demia_resize_v_norm(_lib.ptr(tmp), _lib.ptr(dst), b, h, neww, newh, ph, pw, _lib.ptr(t["ym"]),
                                                _lib.ptr(t["ys"]), _lib.ptr(t["yk"]), t["ksy"], mean, F32, st),
                   "demia_resize_v_norm")
        return dst, newh, neww, ph, pw

    def backbone(self, xin: torch.Tensor, ph: int, pw: int) -> Dict[str, torch.Tensor]:
        b = xin.shape[0]
        st = self._stream()
        if self.p32:
            if self.stem_mode == "fused":
                # stem + max pool in one launch, planes out: the 2 GB f32 stem output of a 48-tile batch is never written
                x = self.new_p32((b, ph // 4, pw // 4, 64))
                _lib.check(self.lib.demia_stem_pool_mfma(_lib.ptr(xin), _lib.ptr(self.stem_planes), _lib.ptr(self.stem_scale_mfma),
                                                         _lib.ptr(self.stem_bias), _lib.ptr(x.buf), _lib.ptr(x.meta), b, ph, pw, self.stem_s_in,
                                                         p32.plane_scale(self.stem_bound), x.groups, int(self.single_plane), st), "demia_stem_pool_mfma")
                mid = None
            else:
                mid = self._scratch(b * (ph // 2) * (pw // 2) * 64, torch.float32)
            if mid is None:
                pass
            elif self.stem_on_mfma:
                _lib.check(self.lib.demia_stem_conv_mfma(_lib.ptr(xin), _lib.ptr(self.stem_planes), _lib.ptr(self.stem_scale_mfma),
                                                         _lib.ptr(self.stem_bias), _lib.ptr(mid), b, ph, pw, self.stem_s_in, st), "demia_stem_conv_mfma")
            else:
                _lib.check(self.lib.demia_stem_conv(_lib.ptr(xin), _lib.ptr(self.stem_w), _lib.ptr(self.stem_scale),
                                                    _lib.ptr(self.stem_bias), _lib.ptr(mid), b, ph, pw, F32, st), "demia_stem_conv")
            if mid is not None:
                x = self.new_p32((b, ph // 4, pw // 4, 64))
                _lib.check(self.lib.demia_maxpool3x3s2_p32(_lib.ptr(mid), _lib.ptr(x.buf), _lib.ptr(x.meta), p32.plane_scale(self.stem_bound),
                                                           b, ph // 2, pw // 2, 64, x.groups, int(self.single_plane), st), "demia_maxpool3x3s2_p32")
        else:
            mid = torch.empty((b, ph // 2, pw // 2, 64), dtype=self.tdt, device=self.device)
            _lib.check(self.lib.demia_stem_conv(_lib.ptr(xin), _lib.ptr(self.stem_w), _lib.ptr(self.stem_scale),
                                                _lib.ptr(self.stem_bias), _lib.ptr(mid), b, ph, pw, self.dt, st), "demia_stem_conv")
            x = torch.empty((b, ph // 4, pw // 4, 64), dtype=self.tdt, device=self.device)
            _lib.check(self.lib.demia_maxpool3x3s2(_lib.ptr(mid), _lib.ptr(x), b, ph // 2, pw // 2, 64, self.dt, st),
                       "demia_maxpool3x3s2")
        feats = {"stem": x}
        for si, stage_blocks in enumerate(self.blocks):
            for blk in stage_blocks:
                sc = self.conv(x, blk["shortcut"]) if "shortcut" in blk else x
                o = self.conv(x, blk["conv1"], act=ACT_RELU)
                o = self.conv(o, blk["conv2"], act=ACT_RELU)
                x = self.conv(o, blk["conv3"], act=ACT_RELU, residual=sc, res_mode=RES_SAME)
            feats[f"res{si + 2}"] = x
        prev = None
        for lvl in (5, 4, 3, 2):
            if prev is None:
                lat = self.conv(feats[f"res{lvl}"], self.fpn_lateral[lvl])
            else:
                lat = self.conv(feats[f"res{lvl}"], self.fpn_lateral[lvl], residual=prev, res_mode=RES_UP2)
            prev = lat
            feats[f"p{lvl}"] = self.conv(lat, self.fpn_output[lvl])
        p5 = feats["p5"]
        h5, w5 = p5.shape[1], p5.shape[2]
        if self.p32:
            # a P32 pixel of C channels is C 4-byte words: the plain strided copy moves whole pixels; same scale, |x| no larger
            p6 = self.new_p32((b, (h5 - 1) // 2 + 1, (w5 - 1) // 2 + 1, 256))
            p6 = p32.P32(p6.buf, p5.meta, p6.shape)
            _lib.check(self.lib.demia_subsample2(_lib.ptr(p5.buf) + 128, _lib.ptr(p6.buf) + 128, b, h5, w5, 256, F32, st), "demia_subsample2")
            feats["p6"] = p6
            return feats
        p6 = torch.empty((b, (h5 - 1) // 2 + 1, (w5 - 1) // 2 + 1, 256), dtype=self.tdt, device=self.device)
        _lib.check(self.lib.demia_subsample2(_lib.ptr(p5), _lib.ptr(p6), b, h5, w5, 256, self.dt, st), "demia_subsample2")
        feats["p6"] = p6
        return feats

    def rpn(self, feats, newh: int, neww: int):
        b = feats["p2"].shape[0]
        heads = []
        for name in ("p2", "p3", "p4", "p5", "p6"):
            t = self.conv(feats[name], self.rpn_conv, act=ACT_RELU)
            heads.append(self.conv(t, self.rpn_pred, out_dtype=torch.float32, out_ld=16))
        boxes = torch.empty((b, POST_NMS_TOPK, 4), dtype=torch.float32, device=self.device)
        scores = torch.empty((b, POST_NMS_TOPK), dtype=torch.float32, device=self.device)
        count = torch.empty((b,), dtype=torch.int32, device=self.device)
        ws = torch.empty((int(self.lib.demia_rpn_workspace_bytes(b)),), dtype=torch.uint8, device=self.device)
        d = _lib.RpnDesc()
        for i, hd in enumerate(heads):
            d.head[i] = _lib.ptr(hd)
            d.H[i], d.W[i], d.stride[i] = hd.shape[1], hd.shape[2], STRIDES[i]
        d.cell_anchors = self._cell.ctypes.data
        d.head_ld, d.N, d.img_h, d.img_w = 16, b, newh, neww
        d.pre_topk, d.post_topk, d.nms_thresh = PRE_NMS_TOPK, POST_NMS_TOPK, RPN_NMS_THRESH
        d.out_boxes, d.out_scores, d.out_count, d.workspace = _lib.ptr(boxes), _lib.ptr(scores), _lib.ptr(count), _lib.ptr(ws)
        _lib.check(self.lib.demia_rpn_proposals(C.byref(d), self._stream()), "demia_rpn_proposals")
        self._dbg_heads = heads
        return boxes, scores, count

    def roi_align(self, feats, boxes: torch.Tensor, count: torch.Tensor, P: int) -> torch.Tensor:
        b, r, _ = boxes.shape
        d = _lib.RoiAlignDesc()
        if self.roi_order and r <= 1024:
            # launch order: an image's ROIs sorted by (level, row band, column) and dealt to the XCDs in runs (demia_roi_order);
            # output rows stay where they are -- bit-identical results
            order = torch.empty((b * r,), dtype=torch.int32, device=self.device)
            _lib.check(self.lib.demia_roi_order(_lib.ptr(boxes), _lib.ptr(count), b, r, _lib.ptr(order), self._stream()), "demia_roi_order")
            d.order = _lib.ptr(order)
        if isinstance(feats["p2"], p32.P32):
            out = self.new_p32((b, r, P, P, 256), feats["p2"].groups)
            for i, name in enumerate(("p2", "p3", "p4", "p5")):
                f = feats[name]
                d.feat[i], d.meta[i] = _lib.ptr(f.buf), _lib.ptr(f.meta)
                d.H[i], d.W[i] = f.shape[1], f.shape[2]
            d.N, d.R, d.C, d.P, d.dtype, d.groups, d.single = b, r, 256, P, _lib.P32, out.groups, int(self.single_plane)
            d.boxes, d.count, d.out, d.out_meta = _lib.ptr(boxes), _lib.ptr(count), _lib.ptr(out.buf), _lib.ptr(out.meta)
            _lib.check(self.lib.demia_roi_align(C.byref(d), self._stream()), "demia_roi_align")
            return out
        out = torch.empty((b, r, P, P, 256), dtype=self.tdt, device=self.device)
        for i, name in enumerate(("p2", "p3", "p4", "p5")):
            f = feats[name]
            d.feat[i] = _lib.ptr(f)
            d.H[i], d.W[i] = f.shape[1], f.shape[2]
        d.N, d.R, d.C, d.P, d.dtype = b, r, 256, P, self.dt
        d.boxes, d.count, d.out = _lib.ptr(boxes), _lib.ptr(count), _lib.ptr(out)
        _lib.check(self.lib.demia_roi_align(C.byref(d), self._stream()), "demia_roi_align")
        if self.precision == "f16x2r":                 # bilinear taps and bin averages are convex combinations
            out._amax = torch.cat([self.amax_of(feats[n]) for n in ("p2", "p3", "p4", "p5")]).amax().reshape(1)
        return out

    def box_head(self, pooled: torch.Tensor) -> torch.Tensor:
        b, r = pooled.shape[:2]
        x = self.view_as(pooled, b * r, 1, 1, 12544)
        x = self.conv(x, self.fc1, act=ACT_RELU)
        x = self.conv(x, self.fc2, act=ACT_RELU)
        ld = (5 * self.K + 1 + 3) // 4 * 4
        return self.conv(x, self.box_pred, out_dtype=torch.float32, out_ld=ld).view(b, r, ld)

    def detections(self, logits, props, prop_count, newh, neww):
        b, r, ld = logits.shape
        D = DETS_PER_IMAGE
        det_boxes = torch.empty((b, D, 4), dtype=torch.float32, device=self.device)
        det_scores = torch.empty((b, D), dtype=torch.float32, device=self.device)
        det_classes = torch.empty((b, D), dtype=torch.int32, device=self.device)
        det_count = torch.empty((b,), dtype=torch.int32, device=self.device)
        d = _lib.DetsDesc(_lib.ptr(logits), ld, _lib.ptr(props), _lib.ptr(prop_count), b, r, self.K, newh, neww,
                          self.score_thresh, DET_NMS_THRESH, D, _lib.ptr(det_boxes), _lib.ptr(det_scores),
                          _lib.ptr(det_classes), _lib.ptr(det_count))
        _lib.check(self.lib.demia_box_detections(C.byref(d), self._stream()), "demia_box_detections")
        return det_boxes, det_scores, det_classes, det_count

    def mask_head(self, mpooled: torch.Tensor) -> torch.Tensor:
        b, dd = mpooled.shape[:2]
        x = self.view_as(mpooled, b * dd, 14, 14, 256)
        for L in self.mask_fcn:
            x = self.conv(x, L, act=ACT_RELU)
        ld = (self.K + 3) // 4 * 4
        if self.p32 and self.K <= 4:
            # ConvTranspose2d(k2, s2) as a GEMM onto (dy, dx, co) + ReLU with the class predictor + sigmoid fused into its
            # epilogue: the [.., 1024] deconv output (1.3 GB per 16-tile batch) is never written or re-read
            return self.conv_p32(self.view_as(x, b * dd * 196, 1, 1, 256), self.deconv, act=ACT_RELU,
                                 head=(self.mask_pred_w32, self.mask_pred_b32, ld, ACT_SIGMOID)).view(b * dd * 196 * 4, 1, 1, ld)
        x = self.conv(self.view_as(x, b * dd * 196, 1, 1, 256), self.deconv, act=ACT_RELU)        # [.., 1024] = (dy,dx,co)
        x = self.conv(self.view_as(x, b * dd * 196 * 4, 1, 1, 256), self.mask_pred, act=ACT_SIGMOID,
                      out_dtype=torch.float32, out_ld=(self.K + 3) // 4 * 4)
        return x  # [(i*196 + cell)*4 + sub, 1, 1, ld] f32 probabilities

    def paste(self, mask_prob, det_boxes, det_classes, det_count, newh, neww, out_h, out_w):
        b, D = det_boxes.shape[:2]
        out_boxes = torch.empty((b, D, 4), dtype=torch.float32, device=self.device)
        valid = torch.empty((b, D), dtype=torch.uint8, device=self.device)
        bbox = torch.empty((b, D, 4), dtype=torch.int32, device=self.device)
        static = self._paste_static
        if static is not None:
            # a captured forward pastes into ITS OWN planes every replay: they were zeroed once, `prev` holds the boxes the
            # last replay could set, so only the union of old and new boxes is written (incremental paste) -- the whole-plane
            # paste of a 48-tile batch is a 2.5 GB memset with ~1 % payload
            packed, prev = static
            assert tuple(packed.shape) == (b, D, out_h, (out_w + 31) // 32)
        else:
            packed, prev = torch.empty((b, D, out_h, (out_w + 31) // 32), dtype=torch.int32, device=self.device), None
        d = _lib.PasteDesc(_lib.ptr(mask_prob), mask_prob.shape[-1], _lib.ptr(det_boxes), _lib.ptr(det_classes),
                           _lib.ptr(det_count), b, D, newh, neww, out_h, out_w, _lib.ptr(out_boxes), _lib.ptr(valid),
                           _lib.ptr(packed), _lib.ptr(bbox), _lib.ptr(prev))
        _lib.check(self.lib.demia_paste_masks(C.byref(d), self._stream()), "demia_paste_masks")
        if prev is not None:
            prev.copy_(bbox)
        return out_boxes, valid, packed, bbox

    # ------------------------------------------------------------------ whole forward
    @torch.no_grad()
    def forward(self, images: torch.Tensor, keep_intermediates: bool = False) -> RawDetections:
        """images: [B, H, W, 3] uint8 BGR on the device."""
        assert images.dtype == torch.uint8 and images.dim() == 4 and images.shape[3] == 3
        images = images.contiguous()
        b, h, w, _ = images.shape
        self._amax_buf = None            # f16x2r: a fresh (zeroed) pool of |activation| bounds per forward
        xin, newh, neww, ph, pw = self.preprocess(images)
        feats = self.backbone(xin, ph, pw)
        props, pscores, pcount = self.rpn(feats, newh, neww)
        pooled = self.roi_align(feats, props, pcount, 7)
        logits = self.box_head(pooled)
        det_boxes, det_scores, det_classes, det_count = self.detections(logits, props, pcount, newh, neww)
        mpooled = self.roi_align(feats, det_boxes, det_count, 14)
        mask_prob = self.mask_head(mpooled)
        out_boxes, valid, packed, bbox = self.paste(mask_prob, det_boxes, det_classes, det_count, newh, neww, h, w)
        res = RawDetections(out_boxes, det_scores, det_classes, valid, det_count, packed, h, w, bbox)
        if keep_intermediates:
            res.dbg = dict(xin=xin, feats=feats, props=props, pscores=pscores, pcount=pcount, pooled=pooled,
                           logits=logits, det_boxes=det_boxes, mpooled=mpooled, mask_prob=mask_prob,
                           heads=self._dbg_heads, newh=newh, neww=neww)
        return res

    # ------------------------------------------------------------------ hipGraph replay of the forward
    @torch.no_grad()
    def forward_graphed(self, images: torch.Tensor, slots: int = 2) -> RawDetections:
        """The same forward as ONE hipGraph launch per batch shape: ~150 kernel launches (every entry point of the C ABI is
        allocation-free and capturable) are captured once and replayed, so the host thread that also drives the
        post-processing spends microseconds, not milliseconds, per batch.  ``slots`` graphs are captured per shape and
        used in turn: a replay overwrites its own outputs, and the caller may still be reading the previous batch's on
        another stream -- results stay valid until the same slot comes round again (``slots - 1`` later forwards)."""
        if not self.p32:
            return self.forward(images)
        images = images.contiguous()
        key = tuple(int(d) for d in images.shape[:3])
        self._touch_shape(key)                        # may evict ANOTHER shape's arena + graphs (LRU)
        st = self._graphs.get(key)
        cur = torch.cuda.current_stream(self.device)
        if st is None:
            self.forward(images)                      # eager once: arena, tables, kernel attributes
            st = {"graphs": [], "next": 0}
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(slots):
                    static_in = torch.empty_like(images)
                    b, h, w = key
                    planes = torch.zeros((b, DETS_PER_IMAGE, h, (w + 31) // 32), dtype=torch.int32, device=self.device)
                    prev = torch.full((b, DETS_PER_IMAGE, 4), -1, dtype=torch.int32, device=self.device)
                    g = torch.cuda.CUDAGraph()
                    self._paste_static = (planes, prev)
                    try:
                        with torch.cuda.graph(g, stream=side):
                            out = self.forward(static_in)
                    finally:
                        self._paste_static = None
                    out._slot = (key, len(st["graphs"]))      # lets release_outputs find the slot these outputs live in
                    st["graphs"].append((g, static_in, out, planes, prev))
            st["released"] = [None] * len(st["graphs"])
            cur.wait_stream(side)
            self._graphs[key] = st
        slot = st["next"]
        g, static_in, out = st["graphs"][slot][:3]
        st["next"] = (slot + 1) % len(st["graphs"])
        ev = st["released"][slot]
        if ev is not None:
            # the consumer of this slot's previous outputs said (release_outputs) after which point of ITS stream nothing reads
            # them any more: the replay that overwrites them waits for that point, whatever the host did in between
            cur.wait_event(ev)
            st["released"][slot] = None
        static_in.copy_(images, non_blocking=True)
        g.replay()
        return out

    def release_outputs(self, raw: RawDetections) -> None:
        """Called by the consumer of a replayed forward's outputs, on the stream that reads them, once its LAST read of them is
        enqueued: the next replay into the same slot waits for this point.  (No-op for eager results.)"""
        slot = getattr(raw, "_slot", None)
        if slot is None:
            return
        st = self._graphs.get(slot[0])
        if st is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            st["released"][slot[1]] = ev

    def unpack(self, packed: torch.Tensor, h: int, w: int) -> torch.Tensor:
        """bit-packed [M, H, W/32] -> Detectron2's (M, H, W) bool."""
        m = packed.shape[0]
        out = torch.empty((m, h, w), dtype=torch.bool, device=self.device)
        _lib.check(self.lib.demia_unpack_masks(_lib.ptr(packed), _lib.ptr(out), m, h, w, self._stream()), "demia_unpack_masks")
        return out

    def area_bbox(self, packed: torch.Tensor, h: int, w: int, hint: Optional[torch.Tensor] = None):
        m = packed.shape[0]
        area = torch.empty((m,), dtype=torch.int32, device=self.device)
        bbox = torch.empty((m, 4), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.demia_mask_area_bbox(_lib.ptr(packed), _lib.ptr(hint), _lib.ptr(area), _lib.ptr(bbox), m, h, w, self._stream()),
                   "demia_mask_area_bbox")
        return area, bbox
