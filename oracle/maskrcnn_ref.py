"""ORACLE (test infrastructure, never shipped, never imported by the product path).

CPU fp32 restatement, in plain torch, of what the reference's
``predictor(image)`` call does (``src/functions/inference.py:1395,1398,1507,1669``
-> ``src/data/models.py:107`` ``DefaultPredictor(cfg)``).  The arithmetic lives in
the un-vendored third-party dependency **detectron2 == 0.6** (+ torchvision
0.11.1 ``nms`` / ``roi_align``, Pillow ``Image.resize``) pinned at
``requirements.txt:14-16,30,32``; none of it is under ``/root/reference`` and none
is installable here, so this file restates the published algorithm of those
versions.  **Parity unpinned** against Detectron2 itself:
the reference has no tests, goldens or checkpoints for this stage.

Where Detectron2 leaves tie order unspecified (``topk``/``sort`` on equal keys)
this oracle fixes it to *stable, lower index first*; the HIP path does the same.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

PIXEL_MEAN = (103.530, 116.280, 123.675)  # BGR, cfg.MODEL.PIXEL_MEAN; PIXEL_STD = 1
RES_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}
ANCHOR_SIZES = (32, 64, 128, 256, 512)
ANCHOR_RATIOS = (0.5, 1.0, 2.0)
STRIDES = (4, 8, 16, 32, 64)
SCALE_CLAMP = math.log(1000.0 / 16)
BN_EPS = 1e-5


# ----------------------------------------------------------------------------
# A.2  DefaultPredictor.__call__: ResizeShortestEdge([800, 800], 1333) via PIL
# ----------------------------------------------------------------------------
def resize_shape(h: int, w: int, short: int = 800, max_size: int = 1333) -> Tuple[int, int]:
    scale = short * 1.0 / min(h, w)
    if h < w:
        newh, neww = short, scale * w
    else:
        newh, neww = scale * h, short
    if max(newh, neww) > max_size:
        s = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * s, neww * s
    return int(newh + 0.5), int(neww + 0.5)


def resize_shortest_edge(img: np.ndarray, short: int = 800, max_size: int = 1333) -> np.ndarray:
    h, w = img.shape[:2]
    newh, neww = resize_shape(h, w, short, max_size)
    pil = Image.fromarray(img)
    pil = pil.resize((neww, newh), Image.BILINEAR)
    return np.asarray(pil)


def pil_bilinear_coeffs(in_size: int, out_size: int):
    """Pillow ``precompute_coeffs`` for the BILINEAR (triangle) filter, 8bpc path.

    Returns (bounds_min[out], ksize, int32 coeffs[out, ksize]) with
    PRECISION_BITS = 32 - 8 - 2 = 22 exactly as ``ImagingResample`` does, so a
    separable integer pass ``clip8((sum(px*k) + (1 << 21)) >> 22)`` reproduces
    ``Image.resize`` bit for bit (checked against PIL in tests)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale  # bilinear support = 1
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    xmins = np.zeros(out_size, dtype=np.int32)
    xsizes = np.zeros(out_size, dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ww = 0.0
        for x in range(xmax):
            arg = (x + xmin - center + 0.5) * ss
            wgt = 1.0 - abs(arg) if abs(arg) < 1.0 else 0.0
            kk[xx, x] = wgt
            ww += wgt
        if ww != 0.0:
            kk[xx, :xmax] /= ww
        xmins[xx] = xmin
        xsizes[xx] = xmax
    prec = 22
    ik = np.where(kk < 0, np.trunc(-0.5 + kk * (1 << prec)), np.trunc(0.5 + kk * (1 << prec))).astype(np.int32)
    return xmins, xsizes, ik


def pil_resize_int(img: np.ndarray, newh: int, neww: int) -> np.ndarray:
    """Integer restatement of Pillow's two-pass 8-bit resize (horizontal then vertical)."""
    h, w = img.shape[:2]
    out = img
    if neww != w:
        xm, xs, ik = pil_bilinear_coeffs(w, neww)
        tmp = np.zeros((h, neww, img.shape[2]), dtype=np.uint8)
        for xx in range(neww):
            seg = out[:, xm[xx] : xm[xx] + xs[xx], :].astype(np.int64)
            acc = (seg * ik[xx, : xs[xx]].astype(np.int64)[None, :, None]).sum(1) + (1 << 21)
            tmp[:, xx, :] = np.clip(acc >> 22, 0, 255).astype(np.uint8)
        out = tmp
    if newh != h:
        ym, ys, ik = pil_bilinear_coeffs(h, newh)
        tmp = np.zeros((newh, out.shape[1], img.shape[2]), dtype=np.uint8)
        for yy in range(newh):
            seg = out[ym[yy] : ym[yy] + ys[yy], :, :].astype(np.int64)
            acc = (seg * ik[yy, : ys[yy]].astype(np.int64)[:, None, None]).sum(0) + (1 << 21)
            tmp[yy] = np.clip(acc >> 22, 0, 255).astype(np.uint8)
        out = tmp
    return out


# ----------------------------------------------------------------------------
# A.4 backbone
# ----------------------------------------------------------------------------
def _frozen_bn(x: torch.Tensor, sd, prefix: str) -> torch.Tensor:
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    scale = w * (rv + BN_EPS).rsqrt()
    bias = b - rm * scale
    return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


# Study switch ("Winograd decision"): which stages evaluate their 3x3 stride-1 convolutions by fp32
# Winograd F(2x2, 3x3) instead of the direct sum -- a set out of {"backbone", "fpn", "rpn", "mask"}; EMPTY (the oracle of
# record) unless ``oracle/winograd_probe.py`` sets it.  Nothing else reads it.
WINOGRAD_STAGES: set = set()
WINOGRAD_F16X2 = False      # ... with the transformed operands U, V carried as two power-of-two-scaled fp16 planes and the product taken
                            # as a_h b_l + a_l b_h + a_h b_h in fp32 (the product path's arithmetic) instead of fp32

_WG = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])
_WBT = torch.tensor([[1.0, 0.0, -1.0, 0.0], [0.0, 1.0, 1.0, 0.0], [0.0, -1.0, 1.0, 0.0], [0.0, 1.0, 0.0, -1.0]])
_WAT = torch.tensor([[1.0, 1.0, 1.0, 0.0], [0.0, 1.0, -1.0, -1.0]])


def conv3x3_winograd(x: torch.Tensor, w: torch.Tensor, b=None) -> torch.Tensor:
    """``F.conv2d(x, w, b, padding=1)`` for a 3x3 stride-1 kernel by Winograd F(2x2, 3x3) (Lavin & Gray 2015), everything in
    fp32: U = G g G^T per (co, ci), V = B^T d B per 4x4 input tile (stride 2), the 16 element-wise positions as 16 GEMMs
    over the channels, Y = A^T M A.  2.25 x fewer multiplications in the contraction; the transforms add and subtract
    neighbouring inputs, so the rounding differs from the direct sum by a few fp32 ulps of the LARGEST term."""
    n, c, h, wd = x.shape
    th, tw = (h + 1) // 2, (wd + 1) // 2
    xp = F.pad(x, (1, 2 * tw - wd + 1, 1, 2 * th - h + 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                         # [n, c, th, tw, 4, 4]
    v = torch.einsum("ai,nctuij,bj->abnctu", _WBT, d, _WBT)        # [4, 4, n, c, th, tw]
    u = torch.einsum("ai,ocij,bj->aboc", _WG, w, _WG)              # [4, 4, co, ci]
    u, v = u.reshape(16, 1, w.shape[0], c), v.reshape(16, n, c, th * tw)
    if WINOGRAD_F16X2:
        def planes(t, dims):
            # one exact power-of-two scale per output channel (U) / per image (V) bringing max |t| under 2^15, then h = half(t s),
            # l = half(t s - h): 22 significand bits; elements far below the maximum keep an absolute error (fp16 denormals)
            amax = t.abs().amax(dim=dims, keepdim=True).clamp(min=1e-30)
            sc = torch.exp2(14 - torch.floor(torch.log2(amax)))
            hi = (t * sc).half().float()
            lo = (t * sc - hi).half().float()
            return hi, lo, sc
        uh, ul, us = planes(u, (0, 1, 3))          # per output channel
        vh, vl, vs = planes(v, (0, 2, 3))          # per image
        m = (torch.matmul(uh, vl) + torch.matmul(ul, vh) + torch.matmul(uh, vh)) / (us * vs)
    else:
        m = torch.matmul(u, v)                                                                # [16, n, co, th * tw]
    m = m.reshape(4, 4, n, w.shape[0], th, tw)
    y = torch.einsum("ia,abnotu,jb->notiuj", _WAT, m, _WAT).reshape(n, w.shape[0], 2 * th, 2 * tw)[:, :, :h, :wd]
    return y if b is None else y + b.reshape(1, -1, 1, 1)


def _conv3x3(x, w, b, stage: str):
    if stage in WINOGRAD_STAGES:
        return conv3x3_winograd(x, w, b)
    return F.conv2d(x, w, b, padding=1)


def _conv_bn(x, sd, prefix, stride=1, padding=0, relu=False):
    if padding == 1 and stride == 1 and "backbone" in WINOGRAD_STAGES and tuple(sd[prefix + ".weight"].shape[2:]) == (3, 3):
        y = conv3x3_winograd(x, sd[prefix + ".weight"])
    else:
        y = F.conv2d(x, sd[prefix + ".weight"], None, stride=stride, padding=padding)
    y = _frozen_bn(y, sd, prefix + ".norm")
    return F.relu(y) if relu else y


def backbone_fpn(x: torch.Tensor, sd, depth: int) -> Dict[str, torch.Tensor]:
    bu = "backbone.bottom_up."
    x = _conv_bn(x, sd, bu + "stem.conv1", stride=2, padding=3, relu=True)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    feats = {}
    for stage, nblk in zip((2, 3, 4, 5), RES_BLOCKS[depth]):
        for i in range(nblk):
            p = f"{bu}res{stage}.{i}."
            stride = 2 if (i == 0 and stage > 2) else 1
            if (p + "shortcut.weight") in sd:
                sc = _conv_bn(x, sd, p + "shortcut", stride=stride)
            else:
                sc = x
            out = _conv_bn(x, sd, p + "conv1", stride=stride, relu=True)  # STRIDE_IN_1X1
            out = _conv_bn(out, sd, p + "conv2", padding=1, relu=True)
            out = _conv_bn(out, sd, p + "conv3")
            x = F.relu(out + sc)
        feats[f"res{stage}"] = x
    results = {}
    prev = None
    for lvl in (5, 4, 3, 2):
        lat = F.conv2d(feats[f"res{lvl}"], sd[f"backbone.fpn_lateral{lvl}.weight"], sd[f"backbone.fpn_lateral{lvl}.bias"])
        if prev is not None:
            lat = lat + F.interpolate(prev, scale_factor=2.0, mode="nearest")
        prev = lat
        results[f"p{lvl}"] = _conv3x3(lat, sd[f"backbone.fpn_output{lvl}.weight"], sd[f"backbone.fpn_output{lvl}.bias"], "fpn")
    results["p6"] = F.max_pool2d(results["p5"], kernel_size=1, stride=2, padding=0)
    results.update(feats)
    return results


# ----------------------------------------------------------------------------
# A.5 RPN, A.6 NMS
# ----------------------------------------------------------------------------
def cell_anchors(size: float) -> torch.Tensor:
    out = []
    area = size ** 2.0
    for r in ANCHOR_RATIOS:
        w = math.sqrt(area / r)
        h = r * w
        out.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return torch.tensor(out, dtype=torch.float32)


def grid_anchors(h: int, w: int, stride: int, size: float) -> torch.Tensor:
    sx = torch.arange(0, w * stride, step=stride, dtype=torch.float32)
    sy = torch.arange(0, h * stride, step=stride, dtype=torch.float32)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
    return (shifts.view(-1, 1, 4) + cell_anchors(size).view(1, -1, 4)).reshape(-1, 4)


def apply_deltas(deltas: torch.Tensor, boxes: torch.Tensor, weights) -> torch.Tensor:
    deltas = deltas.float()
    boxes = boxes.to(deltas.dtype)
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx = deltas[:, 0::4] / wx
    dy = deltas[:, 1::4] / wy
    dw = torch.clamp(deltas[:, 2::4] / ww, max=SCALE_CLAMP)
    dh = torch.clamp(deltas[:, 3::4] / wh, max=SCALE_CLAMP)
    pcx = dx * widths[:, None] + ctr_x[:, None]
    pcy = dy * heights[:, None] + ctr_y[:, None]
    pw = torch.exp(dw) * widths[:, None]
    ph = torch.exp(dh) * heights[:, None]
    x1 = pcx - 0.5 * pw
    y1 = pcy - 0.5 * ph
    x2 = pcx + 0.5 * pw
    y2 = pcy + 0.5 * ph
    return torch.stack((x1, y1, x2, y2), dim=-1).reshape(deltas.shape)


def nms(boxes: torch.Tensor, scores: torch.Tensor, thresh: float) -> torch.Tensor:
    """torchvision 0.11 CPU ``nms_kernel``: fp32, area=(x2-x1)(y2-y1), suppress when
    IoU > thresh; order = stable descending score.  Returns kept indices in that order."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.int64)
    order = torch.sort(scores, descending=True, stable=True)[1]
    b = boxes[order].numpy().astype(np.float32)
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if i + 1 >= n:
            break
        xx1 = np.maximum(x1[i], x1[i + 1 :])
        yy1 = np.maximum(y1[i], y1[i + 1 :])
        xx2 = np.minimum(x2[i], x2[i + 1 :])
        yy2 = np.minimum(y2[i], y2[i + 1 :])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[i + 1 :] - inter)
        suppressed[i + 1 :] |= ovr > np.float32(thresh)
    return order[torch.tensor(keep, dtype=torch.int64)]


def batched_nms(boxes, scores, idxs, thresh, size_rule: bool = False) -> torch.Tensor:
    """torchvision 0.11 ``batched_nms`` (detectron2.layers.nms.batched_nms forwards to it).  ``size_rule`` applies
    torchvision's own choice: more than 4000 box COORDINATES (``boxes.numel() > 4000``) -> ``_batched_nms_vanilla``
    (independent NMS per key), else ``_batched_nms_coordinate_trick``: every box is shifted by ``idx * (boxes.max() + 1)``
    (fp32) and ONE nms runs over the shifted boxes -- the same result unless an IoU sits within fp32 rounding of the
    threshold after the shift.  The ROI heads call it with <= 1000 x K candidates (both branches occur); the RPN call has
    4768 boxes at MIN_SIZE_TEST = 800 and is always the vanilla branch (``size_rule=False`` there).  Result sorted by
    descending score (stable)."""
    if size_rule and boxes.numel() <= 4000:
        if boxes.numel() == 0:
            return torch.zeros((0,), dtype=torch.int64)
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
        return nms(boxes + offsets[:, None], scores, thresh)
    keep_mask = torch.zeros_like(scores, dtype=torch.bool)
    for k in torch.unique(idxs):
        cur = torch.where(idxs == k)[0]
        kk = nms(boxes[cur], scores[cur], thresh)
        keep_mask[cur[kk]] = True
    keep = torch.where(keep_mask)[0]
    return keep[torch.sort(scores[keep], descending=True, stable=True)[1]]


def rpn_forward(feats: Dict[str, torch.Tensor], sd, image_size: Tuple[int, int],
                pre_topk: int = 1000, post_topk: int = 1000, nms_thresh: float = 0.7):
    rp = "proposal_generator.rpn_head."
    all_boxes, all_scores, all_lvl = [], [], []
    per_level = []
    for li, name in enumerate(("p2", "p3", "p4", "p5", "p6")):
        x = feats[name]
        t = F.relu(_conv3x3(x, sd[rp + "conv.weight"], sd[rp + "conv.bias"], "rpn"))
        logits = F.conv2d(t, sd[rp + "objectness_logits.weight"], sd[rp + "objectness_logits.bias"])
        deltas = F.conv2d(t, sd[rp + "anchor_deltas.weight"], sd[rp + "anchor_deltas.bias"])
        n, a, h, w = logits.shape
        logits_f = logits.permute(0, 2, 3, 1).flatten(1)[0]
        deltas_f = deltas.view(n, a, 4, h, w).permute(0, 3, 4, 1, 2).flatten(1, -2)[0]
        anchors = grid_anchors(h, w, STRIDES[li], ANCHOR_SIZES[li])
        k = min(logits_f.numel(), pre_topk)
        order = torch.sort(logits_f, descending=True, stable=True)[1][:k]
        props = apply_deltas(deltas_f[order], anchors[order], (1.0, 1.0, 1.0, 1.0))
        per_level.append(dict(logits=logits_f, deltas=deltas_f, topk_idx=order, topk_boxes=props))
        all_boxes.append(props)
        all_scores.append(logits_f[order])
        all_lvl.append(torch.full((k,), li, dtype=torch.int64))
    boxes = torch.cat(all_boxes)
    scores = torch.cat(all_scores)
    lvl = torch.cat(all_lvl)
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores)
    boxes, scores, lvl = boxes[valid], scores[valid], lvl[valid]
    ih, iw = image_size
    boxes = boxes.clone()
    boxes[:, 0].clamp_(min=0, max=iw)
    boxes[:, 1].clamp_(min=0, max=ih)
    boxes[:, 2].clamp_(min=0, max=iw)
    boxes[:, 3].clamp_(min=0, max=ih)
    nonempty = ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)
    boxes, scores, lvl = boxes[nonempty], scores[nonempty], lvl[nonempty]
    keep = batched_nms(boxes, scores, lvl, nms_thresh)[:post_topk]
    return boxes[keep], scores[keep], dict(per_level=per_level, cand_boxes=boxes, cand_scores=scores, cand_lvl=lvl, keep=keep)


# ----------------------------------------------------------------------------
# A.7 ROIAlignV2 + level assignment
# ----------------------------------------------------------------------------
def assign_levels(boxes: torch.Tensor, min_level=2, max_level=5, canon_size=224, canon_level=4) -> torch.Tensor:
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    sizes = torch.sqrt(area)
    lv = torch.floor(canon_level + torch.log2(sizes / canon_size + 1e-8))
    lv = torch.clamp(lv, min=min_level, max=max_level)
    return lv.to(torch.int64) - min_level


def roi_align_single(feat: torch.Tensor, box: torch.Tensor, scale: float, out: int) -> torch.Tensor:
    """torchvision 0.11 ``roi_align`` CPU kernel, aligned=True, sampling_ratio=0, one ROI.
    feat: (C, H, W) fp32 -> (C, out, out)."""
    c, H, W = feat.shape
    f32 = np.float32
    x1, y1, x2, y2 = [f32(v) for v in box.tolist()]
    sc = f32(scale)
    off = f32(0.5)
    rsw = x1 * sc - off
    rsh = y1 * sc - off
    rew = x2 * sc - off
    reh = y2 * sc - off
    rw = rew - rsw
    rh = reh - rsh
    bh = rh / f32(out)
    bw = rw / f32(out)
    gh = int(math.ceil(float(rh) / out))
    gw = int(math.ceil(float(rw) / out))
    count = f32(max(gh * gw, 1))
    ph = np.arange(out, dtype=np.float32)
    iy = np.arange(max(gh, 0), dtype=np.float32)
    ix = np.arange(max(gw, 0), dtype=np.float32)
    # y[ph, iy] = rsh + ph*bh + (iy + .5) * bh / gh
    ys = (rsh + ph[:, None] * bh + (iy[None, :] + f32(0.5)) * bh / f32(max(gh, 1))).astype(np.float32).reshape(-1)
    xs = (rsw + ph[:, None] * bw + (ix[None, :] + f32(0.5)) * bw / f32(max(gw, 1))).astype(np.float32).reshape(-1)

    def prep(v, size):
        oob = (v < -1.0) | (v > size)
        v = np.where(v <= 0, f32(0), v).astype(np.float32)
        lo = v.astype(np.int32)
        hi_clamp = lo >= size - 1
        lo = np.where(hi_clamp, size - 1, lo)
        hi = np.where(hi_clamp, size - 1, lo + 1)
        v = np.where(hi_clamp, lo.astype(np.float32), v)
        l = (v - lo.astype(np.float32)).astype(np.float32)
        h_ = (f32(1.0) - l).astype(np.float32)
        return oob, lo, hi, l, h_

    oy, ylo, yhi, ly, hy = prep(ys, H)
    ox, xlo, xhi, lx, hx = prep(xs, W)
    ft = feat
    ylo_t, yhi_t = torch.from_numpy(ylo.astype(np.int64)), torch.from_numpy(yhi.astype(np.int64))
    xlo_t, xhi_t = torch.from_numpy(xlo.astype(np.int64)), torch.from_numpy(xhi.astype(np.int64))
    v1 = ft[:, ylo_t][:, :, xlo_t]
    v2 = ft[:, ylo_t][:, :, xhi_t]
    v3 = ft[:, yhi_t][:, :, xlo_t]
    v4 = ft[:, yhi_t][:, :, xhi_t]
    hy_t, ly_t = torch.from_numpy(hy)[:, None], torch.from_numpy(ly)[:, None]
    hx_t, lx_t = torch.from_numpy(hx)[None, :], torch.from_numpy(lx)[None, :]
    w1, w2, w3, w4 = hy_t * hx_t, hy_t * lx_t, ly_t * hx_t, ly_t * lx_t
    val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
    oob = torch.from_numpy(oy)[:, None] | torch.from_numpy(ox)[None, :]
    val = torch.where(oob[None], torch.zeros((), dtype=val.dtype), val)
    if gh <= 0 or gw <= 0:
        return torch.zeros((c, out, out), dtype=feat.dtype)
    val = val.view(c, out, gh, out, gw)
    return val.sum(dim=(2, 4)) / float(count)


def roi_pool(feats: List[torch.Tensor], boxes: torch.Tensor, out: int) -> torch.Tensor:
    """ROIPooler over p2..p5.  feats[i]: (1, C, H, W)."""
    n = boxes.shape[0]
    c = feats[0].shape[1]
    res = torch.zeros((n, c, out, out), dtype=torch.float32)
    if n == 0:
        return res
    lv = assign_levels(boxes)
    for i in range(n):
        l = int(lv[i])
        res[i] = roi_align_single(feats[l][0], boxes[i], 1.0 / STRIDES[l], out)
    return res


# ----------------------------------------------------------------------------
# A.8 box head, A.9 mask head, A.10 post-process + paste
# ----------------------------------------------------------------------------
def box_head(pooled: torch.Tensor, sd) -> Tuple[torch.Tensor, torch.Tensor]:
    bh = "roi_heads.box_head."
    x = pooled.flatten(1)
    x = F.relu(F.linear(x, sd[bh + "fc1.weight"], sd[bh + "fc1.bias"]))
    x = F.relu(F.linear(x, sd[bh + "fc2.weight"], sd[bh + "fc2.bias"]))
    bp = "roi_heads.box_predictor."
    scores = F.linear(x, sd[bp + "cls_score.weight"], sd[bp + "cls_score.bias"])
    deltas = F.linear(x, sd[bp + "bbox_pred.weight"], sd[bp + "bbox_pred.bias"])
    return scores, deltas


def fast_rcnn_inference(boxes, scores, image_size, score_thresh, nms_thresh=0.5, topk=100):
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    if not bool(valid.all()):
        boxes, scores = boxes[valid], scores[valid]
    scores = scores[:, :-1]
    k = boxes.shape[1] // 4
    ih, iw = image_size
    b = boxes.reshape(-1, 4).clone()
    b[:, 0].clamp_(min=0, max=iw)
    b[:, 1].clamp_(min=0, max=ih)
    b[:, 2].clamp_(min=0, max=iw)
    b[:, 3].clamp_(min=0, max=ih)
    b = b.view(-1, k, 4)
    filter_mask = scores > score_thresh
    filter_inds = filter_mask.nonzero()
    b = b[filter_mask]
    s = scores[filter_mask]
    keep = batched_nms(b, s, filter_inds[:, 1], nms_thresh, size_rule=True)
    if topk >= 0:
        keep = keep[:topk]
    return b[keep], s[keep], filter_inds[keep, 1], filter_inds[keep, 0]


def mask_head(pooled: torch.Tensor, classes: torch.Tensor, sd) -> torch.Tensor:
    mh = "roi_heads.mask_head."
    x = pooled
    for i in range(1, 5):
        x = F.relu(_conv3x3(x, sd[f"{mh}mask_fcn{i}.weight"], sd[f"{mh}mask_fcn{i}.bias"], "mask"))
    x = F.relu(F.conv_transpose2d(x, sd[mh + "deconv.weight"], sd[mh + "deconv.bias"], stride=2))
    x = F.conv2d(x, sd[mh + "predictor.weight"], sd[mh + "predictor.bias"])
    n = x.shape[0]
    if n == 0:
        return x[:, :1]
    idx = torch.arange(n)
    return x[idx, classes][:, None].sigmoid()


def paste_masks(masks: torch.Tensor, boxes: torch.Tensor, img_h: int, img_w: int, threshold: float = 0.5, soft: bool = False):
    """``paste_masks_in_image`` CPU path: one instance at a time, skip_empty=True.  ``soft``: the sampled probabilities
    (f32, zero outside the pasted window) instead of their comparison with the threshold -- what the parity tests use to
    show that a pixel on which the product and this path disagree is a threshold tie."""
    n = masks.shape[0]
    out = torch.zeros((n, img_h, img_w), dtype=torch.float32 if soft else torch.bool)
    for i in range(n):
        bx = boxes[i : i + 1]
        x0_int = int(torch.clamp(bx[:, 0].min().floor() - 1, min=0))
        y0_int = int(torch.clamp(bx[:, 1].min().floor() - 1, min=0))
        x1_int = int(torch.clamp(bx[:, 2].max().ceil() + 1, max=img_w))
        y1_int = int(torch.clamp(bx[:, 3].max().ceil() + 1, max=img_h))
        x0, y0, x1, y1 = torch.split(bx, 1, dim=1)
        img_y = torch.arange(y0_int, y1_int, dtype=torch.float32) + 0.5
        img_x = torch.arange(x0_int, x1_int, dtype=torch.float32) + 0.5
        img_y = (img_y - y0) / (y1 - y0) * 2 - 1
        img_x = (img_x - x0) / (x1 - x0) * 2 - 1
        gx = img_x[:, None, :].expand(1, img_y.size(1), img_x.size(1))
        gy = img_y[:, :, None].expand(1, img_y.size(1), img_x.size(1))
        grid = torch.stack([gx, gy], dim=3)
        m = F.grid_sample(masks[i : i + 1, None].float(), grid, align_corners=False)
        out[i, y0_int:y1_int, x0_int:x1_int] = m[0, 0] if soft else (m[0, 0] >= threshold)
    return out


# ----------------------------------------------------------------------------
# whole predictor
# ----------------------------------------------------------------------------
@torch.no_grad()
def predict(image_bgr: np.ndarray, sd: Dict[str, torch.Tensor], depth: int, score_thresh: float,
            return_intermediates: bool = False, min_size_test: int = 800, max_size_test: int = 1333):
    """``DefaultPredictor.__call__`` -> dict(pred_boxes, scores, pred_classes, pred_masks)."""
    h, w = image_bgr.shape[:2]
    resized = resize_shortest_edge(image_bgr, min_size_test, max_size_test)
    newh, neww = resized.shape[:2]
    x = torch.as_tensor(resized.astype("float32").transpose(2, 0, 1))
    mean = torch.tensor(PIXEL_MEAN, dtype=torch.float32).view(3, 1, 1)
    x = (x - mean) / 1.0
    ph = (newh + 31) // 32 * 32
    pw = (neww + 31) // 32 * 32
    xin = torch.zeros((1, 3, ph, pw), dtype=torch.float32)
    xin[0, :, :newh, :neww] = x
    feats = backbone_fpn(xin, sd, depth)
    prop_boxes, prop_scores, rpn_dbg = rpn_forward(feats, sd, (newh, neww))
    pyr = [feats["p2"], feats["p3"], feats["p4"], feats["p5"]]
    pooled = roi_pool(pyr, prop_boxes, 7)
    cls_logits, deltas = box_head(pooled, sd)
    probs = F.softmax(cls_logits, dim=-1)
    pred_boxes = apply_deltas(deltas, prop_boxes, (10.0, 10.0, 5.0, 5.0))
    det_boxes, det_scores, det_classes, det_src = fast_rcnn_inference(pred_boxes, probs, (newh, neww), score_thresh)
    mpooled = roi_pool(pyr, det_boxes, 14)
    mask_probs = mask_head(mpooled, det_classes, sd)
    sx, sy = w / neww, h / newh
    ob = det_boxes.clone()
    ob[:, 0::2] *= sx
    ob[:, 1::2] *= sy
    ob[:, 0].clamp_(min=0, max=w)
    ob[:, 1].clamp_(min=0, max=h)
    ob[:, 2].clamp_(min=0, max=w)
    ob[:, 3].clamp_(min=0, max=h)
    nonempty = ((ob[:, 2] - ob[:, 0]) > 0) & ((ob[:, 3] - ob[:, 1]) > 0)
    ob, sc, cl, mp = ob[nonempty], det_scores[nonempty], det_classes[nonempty], mask_probs[nonempty]
    masks = paste_masks(mp[:, 0], ob, h, w, 0.5)
    out = dict(pred_boxes=ob, scores=sc, pred_classes=cl, pred_masks=masks, mask_probs28=mp[:, 0])
    if return_intermediates:
        out["dbg"] = dict(resized=resized, xin=xin, feats=feats, rpn=rpn_dbg, prop_boxes=prop_boxes,
                          prop_scores=prop_scores, pooled=pooled, cls_logits=cls_logits, deltas=deltas,
                          probs=probs, pred_boxes=pred_boxes, det_boxes=det_boxes, det_scores=det_scores,
                          det_classes=det_classes, det_src=det_src, mpooled=mpooled, mask_probs=mask_probs)
    return out
