"""Deterministic synthetic inputs for the hot path (no network, no datasets).

Two generators, both numpy/torch-CPU only so that the build container and the
GPU box produce identical bytes:

* :func:`em_tile` -- a synthetic electron-microscopy tile (uint8 BGR).
* :func:`random_d2_state_dict` -- seeded random-init Mask R-CNN weights in the
  Detectron2 0.6 checkpoint key layout that ``src/data/models.py:46-50,103-107``
  of the reference expects on disk (``model_final_r{50,101}.pth``).

Nothing here is reference code: the reference ships no sample data and no
weights.
"""
from __future__ import annotations

import hashlib
import math
from typing import Dict

import numpy as np
import torch

RES_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}


def em_tile(index: int = 0, size: int = 2048) -> np.ndarray:
    """Synthetic EM tile ``index`` as (size, size, 3) uint8 BGR.

    seed = 1234 + index (PCG64); background N(90, 12^2) + three random 2-D
    cosines (amplitude 15 in total); 20-60 filled ellipses with semi-axes
    U[15,120] px (scaled by size/2048), angle U[0,180), intensity U[160,220];
    two 3x3 box blurs; clip to uint8; gray replicated to 3 channels.
    """
    rng = np.random.Generator(np.random.PCG64(1234 + int(index)))
    h = w = int(size)
    img = rng.normal(90.0, 12.0, size=(h, w)).astype(np.float32)
    yy, xx = np.meshgrid(
        np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij"
    )
    for _ in range(3):
        fx, fy = rng.uniform(0.5, 2.5, size=2) * (2.0 * math.pi / size)
        ph = rng.uniform(0, 2.0 * math.pi)
        img += np.float32(5.0) * np.cos(fx * xx + fy * yy + ph).astype(np.float32)
    n_ell = int(rng.integers(20, 61))
    s = size / 2048.0
    for _ in range(n_ell):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        a, b = rng.uniform(15, 120, size=2) * s
        th = math.radians(rng.uniform(0, 180))
        val = rng.uniform(160, 220)
        r = int(math.ceil(max(a, b))) + 1
        x0, x1 = max(0, int(cx) - r), min(w, int(cx) + r + 1)
        y0, y1 = max(0, int(cy) - r), min(h, int(cy) + r + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        dx = xx[y0:y1, x0:x1] - np.float32(cx)
        dy = yy[y0:y1, x0:x1] - np.float32(cy)
        c, sn = np.float32(math.cos(th)), np.float32(math.sin(th))
        u = (dx * c + dy * sn) / np.float32(a)
        v = (-dx * sn + dy * c) / np.float32(b)
        inside = (u * u + v * v) <= 1.0
        patch = img[y0:y1, x0:x1]
        patch[inside] = np.float32(val) + (patch[inside] - np.float32(90.0)) * np.float32(0.5)
    for _ in range(2):
        p = np.pad(img, 1, mode="edge")
        acc = np.zeros_like(img)
        for dy_ in range(3):
            for dx_ in range(3):
                acc += p[dy_ : dy_ + h, dx_ : dx_ + w]
        img = acc / np.float32(9.0)
    g = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2))


def em_tiles_device(indices, size: int, device) -> torch.Tensor:
    """``len(indices)`` DISTINCT synthetic EM tiles generated on the device: the recipe of :func:`em_tile` (noise floor,
    three cosines, 20-60 filled ellipses, two 3x3 box blurs) with the pixel noise from the device's own generator, so a
    job of hundreds of tiles (BASELINE configs[4]) costs milliseconds per tile instead of 1.4 s of host numpy.  Seeded per
    tile (1234 + index) and reproducible on one device type; NOT byte-identical to :func:`em_tile` -- the tile a parity
    check looks at must come from there.  Returns [n, size, size, 3] uint8 BGR."""
    import torch.nn.functional as F

    dev = torch.device(device)
    h = w = int(size)
    out = torch.empty((len(indices), h, w, 3), dtype=torch.uint8, device=dev)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32, device=dev), torch.arange(w, dtype=torch.float32, device=dev), indexing="ij")
    s = size / 2048.0
    for k, index in enumerate(indices):
        rng = np.random.Generator(np.random.PCG64(1234 + int(index)))
        g = torch.Generator(device=dev).manual_seed(1234 + int(index))
        img = torch.randn((h, w), generator=g, device=dev, dtype=torch.float32) * 12.0 + 90.0
        for _ in range(3):
            fx, fy = rng.uniform(0.5, 2.5, size=2) * (2.0 * math.pi / size)
            ph = rng.uniform(0, 2.0 * math.pi)
            img += 5.0 * torch.cos(float(fx) * xx + float(fy) * yy + float(ph))
        for _ in range(int(rng.integers(20, 61))):
            cx, cy = rng.uniform(0, w), rng.uniform(0, h)
            a, b = rng.uniform(15, 120, size=2) * s
            th = math.radians(rng.uniform(0, 180))
            val = float(rng.uniform(160, 220))
            r = int(math.ceil(max(a, b))) + 1
            x0, x1 = max(0, int(cx) - r), min(w, int(cx) + r + 1)
            y0, y1 = max(0, int(cy) - r), min(h, int(cy) + r + 1)
            if x0 >= x1 or y0 >= y1:
                continue
            dx, dy = xx[y0:y1, x0:x1] - float(cx), yy[y0:y1, x0:x1] - float(cy)
            c, sn = math.cos(th), math.sin(th)
            u, v = (dx * c + dy * sn) / float(a), (-dx * sn + dy * c) / float(b)
            patch = img[y0:y1, x0:x1]
            img[y0:y1, x0:x1] = torch.where(u * u + v * v <= 1.0, val + (patch - 90.0) * 0.5, patch)
        for _ in range(2):
            img = F.avg_pool2d(F.pad(img[None, None], (1, 1, 1, 1), mode="replicate"), 3, stride=1)[0, 0]
        out[k] = torch.clamp(torch.round(img), 0, 255).to(torch.uint8)[:, :, None]
    return out


def _kaiming_normal_fan_out(shape, gen):
    # Detectron2 weight_init.c2_msra_fill: kaiming_normal_(mode="fan_out", relu)
    fan_out = shape[0] * int(np.prod(shape[2:])) if len(shape) > 2 else shape[0]
    std = math.sqrt(2.0 / fan_out)
    return torch.randn(shape, generator=gen, dtype=torch.float32) * std


def _kaiming_uniform_a1(shape, gen):
    # Detectron2 weight_init.c2_xavier_fill: kaiming_uniform_(a=1) -> bound = sqrt(3/fan_in)
    fan_in = shape[1] * int(np.prod(shape[2:])) if len(shape) > 2 else shape[1]
    bound = math.sqrt(3.0 / fan_in)
    return (torch.rand(shape, generator=gen, dtype=torch.float32) * 2.0 - 1.0) * bound


def random_d2_state_dict(
    depth: int = 101,
    num_classes: int = 2,
    seed: int = 0,
    mask_bias: float = 2.0,
    residual_gain: float = 0.25,
    head_gain: float = 30.0,
    mask_gain: float = 1.0,
) -> Dict[str, torch.Tensor]:
    """Seeded random Mask R-CNN R{depth}-FPN weights in Detectron2 0.6 key layout.

    Init rules follow SURVEY.md Appendix A.11 (MSRA-normal ResNet / mask-head
    convs, Xavier-uniform FPN and FC, N(0, .01) RPN / cls_score, N(0, .001)
    bbox_pred / mask predictor, FrozenBN mean 0 / var 1 / bias 0) with three
    documented deviations that make a *random* network behave like a trained one
    numerically (bounded activations, spread-out scores, solid masks) so that the
    parity tests exercise every stage non-trivially:

    * ``residual_gain``: FrozenBN ``weight`` of each block's ``conv3`` (instead of
      1.0) -- keeps the residual stream O(1) through 33 blocks.
    * ``head_gain``: multiplies the N(0, .01)/N(0, .001) std of RPN logits /
      deltas, ``cls_score`` and ``bbox_pred`` so that logits are not all ~0.
    * ``mask_bias``: ``mask_head.predictor.bias`` (+2 -> solid masks); ``mask_gain`` scales the
      predictor weights (larger -> blobby, non-rectangular masks for the morphology tests).
    * the 1x1 / FC prediction layers (RPN logits and deltas, ``cls_score``, ``bbox_pred``)
      get zero-mean rows, so class scores are balanced and detections saturate at 100.
    """
    assert depth in RES_BLOCKS
    g = torch.Generator().manual_seed(int(seed) * 1000 + depth)
    sd: Dict[str, torch.Tensor] = {}

    def bn(prefix: str, c: int, weight: float = 1.0):
        sd[prefix + ".weight"] = torch.full((c,), float(weight))
        sd[prefix + ".bias"] = 0.02 * torch.randn((c,), generator=g)
        sd[prefix + ".running_mean"] = 0.02 * torch.randn((c,), generator=g)
        sd[prefix + ".running_var"] = 1.0 + 0.1 * torch.rand((c,), generator=g)

    bu = "backbone.bottom_up."
    sd[bu + "stem.conv1.weight"] = _kaiming_normal_fan_out((64, 3, 7, 7), g) * 0.05
    bn(bu + "stem.conv1.norm", 64)
    in_c = 64
    for stage, nblk in zip((2, 3, 4, 5), RES_BLOCKS[depth]):
        mid = 64 * 2 ** (stage - 2)
        out_c = mid * 4
        for i in range(nblk):
            p = f"{bu}res{stage}.{i}."
            if i == 0:
                sd[p + "shortcut.weight"] = _kaiming_normal_fan_out((out_c, in_c, 1, 1), g) * 0.7
                bn(p + "shortcut.norm", out_c)
            sd[p + "conv1.weight"] = _kaiming_normal_fan_out((mid, in_c, 1, 1), g)
            bn(p + "conv1.norm", mid)
            sd[p + "conv2.weight"] = _kaiming_normal_fan_out((mid, mid, 3, 3), g)
            bn(p + "conv2.norm", mid)
            sd[p + "conv3.weight"] = _kaiming_normal_fan_out((out_c, mid, 1, 1), g)
            bn(p + "conv3.norm", out_c, weight=residual_gain)
            in_c = out_c
    for lvl, c in zip((2, 3, 4, 5), (256, 512, 1024, 2048)):
        sd[f"backbone.fpn_lateral{lvl}.weight"] = _kaiming_uniform_a1((256, c, 1, 1), g)
        sd[f"backbone.fpn_lateral{lvl}.bias"] = torch.zeros(256)
        sd[f"backbone.fpn_output{lvl}.weight"] = _kaiming_uniform_a1((256, 256, 3, 3), g)
        sd[f"backbone.fpn_output{lvl}.bias"] = torch.zeros(256)
    def zero_mean_rows(w):
        # post-ReLU features are all positive: rows that sum to zero keep each logit centred on 0
        return w - w.flatten(1).mean(dim=1).view(-1, *([1] * (w.dim() - 1)))

    rp = "proposal_generator.rpn_head."
    sd[rp + "conv.weight"] = torch.randn((256, 256, 3, 3), generator=g) * 0.01
    sd[rp + "conv.bias"] = torch.zeros(256)
    sd[rp + "objectness_logits.weight"] = zero_mean_rows(torch.randn((3, 256, 1, 1), generator=g)) * 0.01 * head_gain
    sd[rp + "objectness_logits.bias"] = torch.zeros(3)
    sd[rp + "anchor_deltas.weight"] = zero_mean_rows(torch.randn((12, 256, 1, 1), generator=g)) * 0.01 * head_gain
    sd[rp + "anchor_deltas.bias"] = torch.zeros(12)
    bh = "roi_heads.box_head."
    sd[bh + "fc1.weight"] = _kaiming_uniform_a1((1024, 256 * 7 * 7), g)
    sd[bh + "fc1.bias"] = torch.zeros(1024)
    sd[bh + "fc2.weight"] = _kaiming_uniform_a1((1024, 1024), g)
    sd[bh + "fc2.bias"] = torch.zeros(1024)
    bp = "roi_heads.box_predictor."
    k = int(num_classes)
    sd[bp + "cls_score.weight"] = zero_mean_rows(torch.randn((k + 1, 1024), generator=g)) * 0.01 * head_gain
    sd[bp + "cls_score.bias"] = torch.zeros(k + 1)
    sd[bp + "bbox_pred.weight"] = zero_mean_rows(torch.randn((4 * k, 1024), generator=g)) * 0.001 * head_gain
    sd[bp + "bbox_pred.bias"] = torch.zeros(4 * k)
    mh = "roi_heads.mask_head."
    for i in range(1, 5):
        sd[f"{mh}mask_fcn{i}.weight"] = _kaiming_normal_fan_out((256, 256, 3, 3), g)
        sd[f"{mh}mask_fcn{i}.bias"] = torch.zeros(256)
    # ConvTranspose2d weight layout is (in, out, kh, kw)
    sd[mh + "deconv.weight"] = _kaiming_normal_fan_out((256, 256, 2, 2), g)
    sd[mh + "deconv.bias"] = torch.zeros(256)
    sd[mh + "predictor.weight"] = zero_mean_rows(torch.randn((k, 256, 1, 1), generator=g)) * 0.001 * head_gain * mask_gain
    sd[mh + "predictor.bias"] = torch.full((k,), float(mask_bias))
    return sd


def save_d2_checkpoint(path: str, state: Dict[str, torch.Tensor]) -> None:
    """``torch.save({"model": state_dict, ...})`` -- the Detectron2 checkpoint
    container that ``DetectionCheckpointer`` writes."""
    torch.save({"model": {k: v.clone() for k, v in state.items()}, "__author__": "deepemia_amd.synth"}, path)


def sha256_of_state(state: Dict[str, torch.Tensor]) -> str:
    h = hashlib.sha256()
    for k in sorted(state):
        h.update(k.encode())
        h.update(state[k].contiguous().numpy().tobytes())
    return h.hexdigest()
