"""CPU-only tests: the oracle against what pins it, the host logic, and the C ABI surface
(library loads and exports every symbol include/deepemia_hip.h declares; no compute calls)."""
import math
import re
from pathlib import Path

import numpy as np
import pytest
import torch
from PIL import Image

ROOT = Path(__file__).resolve().parent.parent


def test_c_abi_exports_every_declared_symbol():
    import ctypes

    from deepemia_amd import _lib

    header = (ROOT / "include" / "deepemia_hip.h").read_text()
    declared = set(re.findall(r"\b(demia_[a-z0-9_]+)\s*\(", header))
    assert declared, "no prototypes found in the header"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.demia_abi_version() == 6
    assert lib.demia_build_arch() == b"gfx950"
    # struct sizes agree with the header's layout (no hidden padding surprises)
    assert ctypes.sizeof(_lib.ConvDesc) == 6 * 8 + 18 * 4 + 2 * 8     # + amax_in, amax_out


def test_product_path_fails_loudly_without_gpu():
    from deepemia_amd import _lib, synth
    from deepemia_amd.engine import MaskRCNNEngine

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.HipExtensionMissing):
        MaskRCNNEngine(synth.random_d2_state_dict(50, 2, 0), 50, 2, 0.3, "cuda:0", "f32")


@pytest.mark.parametrize("hw", [(1024, 1024), (600, 600), (700, 1100), (512, 2000)])
def test_oracle_resize_restatement_equals_pillow(hw):
    from oracle import maskrcnn_ref as R

    h, w = hw
    img = np.random.default_rng(h + w).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    newh, neww = R.resize_shape(h, w)
    a = np.asarray(Image.fromarray(img).resize((neww, newh), Image.BILINEAR))
    b = R.pil_resize_int(img, newh, neww)
    np.testing.assert_array_equal(a, b)


def test_resize_shape_rules():
    from deepemia_amd.engine import resize_shape
    from oracle import maskrcnn_ref as R

    assert resize_shape(2048, 2048) == (800, 800)
    assert resize_shape(1024, 1024) == (800, 800)
    assert resize_shape(480, 640) == (800, 1067)
    assert resize_shape(500, 2000) == (333, 1333)
    for hw in [(2048, 2048), (480, 640), (500, 2000), (3000, 1000), (777, 801)]:
        assert resize_shape(*hw) == R.resize_shape(*hw)


@pytest.mark.parametrize("io", [(2048, 800), (1024, 800), (600, 800), (1333, 800), (777, 800), (800, 800)])
def test_host_resize_tables_equal_oracle(io):
    from deepemia_amd.engine import pil_bilinear_tables
    from oracle.maskrcnn_ref import pil_bilinear_coeffs

    a, b = io
    xm, xs, xk = pil_bilinear_tables(a, b)
    if a == b:
        assert xk.shape == (b, 1) and (xk == 1 << 22).all() and (xm == np.arange(b)).all()
        return
    ym, ys, yk = pil_bilinear_coeffs(a, b)
    np.testing.assert_array_equal(xm, ym)
    np.testing.assert_array_equal(xs, ys)
    np.testing.assert_array_equal(xk, yk)


def test_cell_anchors_match_oracle():
    from deepemia_amd.engine import cell_anchor_table
    from oracle import maskrcnn_ref as R

    t = cell_anchor_table()
    for l, s in enumerate(R.ANCHOR_SIZES):
        np.testing.assert_array_equal(t[l], R.cell_anchors(s).numpy())
    # known answers: 32^2 area, ratio 1 -> (-16, -16, 16, 16); ratio 0.5 -> w = 45.25.., h = 22.62..
    np.testing.assert_allclose(t[0, 1], [-16, -16, 16, 16])
    np.testing.assert_allclose(t[0, 0], [-22.627417, -11.313708, 22.627417, 11.313708], rtol=1e-6)


def test_oracle_nms_known_answers():
    from oracle import maskrcnn_ref as R

    boxes = torch.tensor([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.0]])
    scores = torch.tensor([0.9, 0.8, 0.7, 0.9])
    # IoU(0,1) = 81/119 = 0.68 -> kept at 0.7, suppressed at 0.5; box 3 duplicates box 0 (tie: lower index first)
    assert R.nms(boxes, scores, 0.7).tolist() == [0, 1, 2]
    assert R.nms(boxes, scores, 0.5).tolist() == [0, 2]
    keep = R.batched_nms(boxes, scores, torch.tensor([0, 1, 0, 1]), 0.5)
    assert keep.tolist() == [0, 3, 2]  # key 1: box 3 (0.9) suppresses box 1 (IoU 0.68)


def test_oracle_roi_align_matches_direct_loop():
    """The vectorised oracle ROIAlign against a literal transcription of torchvision's loop."""
    import math

    from oracle import maskrcnn_ref as R

    g = torch.Generator().manual_seed(3)
    feat = torch.randn((4, 13, 17), generator=g)
    for box in ([2.3, 1.1, 40.7, 30.2], [-5.0, -3.0, 12.0, 9.0], [60.0, 45.0, 70.0, 52.0]):
        out = R.roi_align_single(feat, torch.tensor(box), 0.25, 7)
        c, H, W = feat.shape
        exp = torch.zeros((c, 7, 7))
        x1, y1, x2, y2 = [v * 0.25 - 0.5 for v in box]
        rw, rh = x2 - x1, y2 - y1
        bw, bh = rw / 7, rh / 7
        gh, gw = math.ceil(rh / 7), math.ceil(rw / 7)
        cnt = max(gh * gw, 1)
        for ph in range(7):
            for pw in range(7):
                acc = torch.zeros(c)
                for iy in range(gh):
                    y = y1 + ph * bh + (iy + 0.5) * bh / gh
                    for ix in range(gw):
                        x = x1 + pw * bw + (ix + 0.5) * bw / gw
                        if y < -1 or y > H or x < -1 or x > W:
                            continue
                        yy, xx = max(y, 0.0), max(x, 0.0)
                        yl, xl = int(yy), int(xx)
                        if yl >= H - 1:
                            yh = yl = H - 1
                            yy = float(yl)
                        else:
                            yh = yl + 1
                        if xl >= W - 1:
                            xh = xl = W - 1
                            xx = float(xl)
                        else:
                            xh = xl + 1
                        ly, lx = yy - yl, xx - xl
                        hy, hx = 1 - ly, 1 - lx
                        acc += hy * hx * feat[:, yl, xl] + hy * lx * feat[:, yl, xh] + ly * hx * feat[:, yh, xl] + ly * lx * feat[:, yh, xh]
                exp[:, ph, pw] = acc / cnt
        assert float((out - exp).abs().max()) < 1e-5


def test_oracle_paste_known_answer():
    """A constant-1 28x28 map pasted into box (10,20)-(50,60): inside the box every pixel whose
    centre is >= half a source pixel from the border interpolates to >= 0.5."""
    from oracle import maskrcnn_ref as R

    m = torch.ones((1, 28, 28))
    out = R.paste_masks(m, torch.tensor([[10.0, 20.0, 50.0, 60.0]]), 80, 64)
    assert out.shape == (1, 80, 64)
    ys, xs = torch.nonzero(out[0], as_tuple=True)
    assert (int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max())) == (10, 49, 20, 59)
    assert int(out.sum()) == 40 * 40


def test_synth_is_deterministic():
    import hashlib

    from deepemia_amd import synth

    a, b = synth.em_tile(3, 256), synth.em_tile(3, 256)
    assert a.dtype == np.uint8 and a.shape == (256, 256, 3)
    np.testing.assert_array_equal(a, b)
    assert (a[..., 0] == a[..., 1]).all()
    s1 = synth.random_d2_state_dict(50, 2, 0)
    s2 = synth.random_d2_state_dict(50, 2, 0)
    assert synth.sha256_of_state(s1) == synth.sha256_of_state(s2)
    assert s1["roi_heads.box_predictor.cls_score.weight"].shape == (3, 1024)
    assert s1["roi_heads.mask_head.deconv.weight"].shape == (256, 256, 2, 2)
    assert len([k for k in s1 if k.endswith("conv1.weight") and "res4" in k]) == 6


def test_instances_dropin_surface():
    from deepemia_amd.predictor import Instances

    class FakeEngine:
        calls = 0

        def unpack(self, packed, h, w):
            FakeEngine.calls += 1
            return torch.ones((packed.shape[0], h, w), dtype=torch.bool)

    inst = Instances((8, 32), FakeEngine())
    inst.set("scores", torch.tensor([0.9, 0.5]))
    inst.set("pred_classes", torch.tensor([1, 0]))
    inst.set_packed_masks(torch.zeros((2, 8, 1), dtype=torch.int32))
    assert len(inst) == 2 and FakeEngine.calls == 0          # masks are lazy
    assert inst.pred_classes.cpu().numpy().tolist() == [1, 0]  # inference.py:1514 style
    cpu = inst.to("cpu")                                     # inference.py:1401 style
    assert cpu._fields["pred_masks"].numpy().shape == (2, 8, 32)
    assert FakeEngine.calls == 1
    with pytest.raises(AttributeError):
        inst.nope


class _DenseAlgebra:
    """What the greedy host loops ask of DeviceMaskAlgebra, answered from dense numpy masks (no GPU)."""

    def __init__(self, masks):
        m = np.asarray(masks).astype(bool)
        self.n = len(m)
        flat = m.reshape(self.n, -1).astype(np.int64)
        self.I = flat @ flat.T
        self.area = np.diag(self.I).copy()
        self.known = np.ones((self.n, self.n), dtype=bool)
        self.bbox = np.full((self.n, 4), -1, dtype=np.int64)
        for i in range(self.n):
            ys, xs = np.nonzero(m[i])
            if len(ys):
                self.bbox[i] = (ys.min(), xs.min(), ys.max(), xs.max())

    def intersections(self, pi, pj):     # everything is known already
        raise AssertionError("no pair should be missing")


def test_greedy_and_smart_dedup_loops_match_the_dense_oracle():
    """The order-dependent host decisions of a12 / a14 (`_greedy_keep`, `_dedup_smart_order`: bit-set loops over pair
    matrices) against the oracle's literal scalar loops on dense masks, N6 quirks included."""
    from deepemia_amd.functions.inference import InferencePipeline as IP
    from oracle import postproc_ref as P

    rng = np.random.default_rng(12)
    h = w = 48
    yy, xx = np.mgrid[0:h, 0:w]
    for trial in range(40):
        n = int(rng.integers(1, 26))
        masks = []
        for _ in range(n):
            cy, cx, r = rng.uniform(5, h - 5), rng.uniform(5, w - 5), rng.uniform(3, 14)
            masks.append(((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r) & (rng.random((h, w)) > 0.05))
        if trial % 5 == 0 and n > 2:
            masks[2] = masks[0].copy()                         # exact duplicates
            masks[1] = np.zeros((h, w), dtype=bool)            # and an empty mask
        scores = np.round(rng.uniform(0.3, 1.0, n), 2).astype(np.float32)      # rounded: score ties occur
        classes = [int(c) for c in rng.integers(0, 2, n)]
        alg = _DenseAlgebra(masks)
        for thr in (0.3, 0.5, 0.7):
            got = IP._greedy_keep(alg, range(n), thr)
            um, _, _ = P.greedy_dedup(masks, list(scores), 0, thr)
            want = [i for i in range(n) if any(masks[i] is k for k in um)]
            assert got == want, (trial, thr)
            # step 2 of deduplicate_masks_smart on the non-empty masks (step 1 is the contour-based artefact filter)
            k0 = [i for i in range(n) if masks[i].any()]
            bb = [(int(alg.bbox[i, 0]), int(alg.bbox[i, 2]), int(alg.bbox[i, 1]), int(alg.bbox[i, 3])) for i in k0]
            keep = IP._dedup_smart_order(alg, k0, [scores[i] for i in k0], [classes[i] for i in k0], bb, thr)
            sub = [masks[i] for i in k0]
            # the oracle's full function drops low-compactness masks first: compare on its own step-2 input
            bboxes = [P._smart_bbox(m) for m in sub]
            order = np.argsort(np.asarray([scores[i] for i in k0]), kind="stable")[::-1]
            exp, removed = [], set()
            for idx in order:
                if idx in removed:
                    continue
                exp.append(int(idx))
                for other in order[idx + 1:]:
                    if other in removed or classes[k0[other]] != classes[k0[idx]]:
                        continue
                    if not P._bboxes_overlap_literal(bboxes[idx], bboxes[other]):
                        continue
                    if P._calc_iou_literal(sub[idx], sub[other], bboxes[idx], bboxes[other]) > thr:
                        removed.add(other)
            assert keep == exp, (trial, thr)


def test_f16x2_host_split_is_exact_and_the_tiling_is_the_documented_one():
    """Host side of the default conv arithmetic (engine.py): w * 2^e(co) = h + l with both planes in fp16 range, exactly;
    planes tiled [CoutPad/64][K/32][2][64][32] as include/deepemia_hip.h documents."""
    import torch
    from deepemia_amd import engine as E

    g = torch.Generator().manual_seed(0)
    w = torch.randn((128, 3, 3, 64), generator=g) * torch.logspace(-6, 2, 128).view(-1, 1, 1, 1)   # channel maxima over 8 decades
    w[7] = 0.0
    planes, sw = E.split2_f16_scaled(w)
    assert planes.dtype == torch.float16 and planes.shape == (2, 128, 3, 3, 64)
    # exact powers of two, max |w * sw| in [2^14, 2^15] for every non-zero channel
    m, ex = torch.frexp(sw)
    assert torch.all(m == 0.5)
    top = (w * sw.view(-1, 1, 1, 1)).abs().flatten(1).amax(1)
    nz = top > 0
    assert torch.all(top[nz] >= 2.0 ** 14) and torch.all(top[nz] < 2.0 ** 15)
    # h + l reproduces w * sw to <= 2^-22 relative (elements far below the channel maximum: <= 2^-25 absolute of it)
    rec = planes[0].double() + planes[1].double()
    ws = (w * sw.view(-1, 1, 1, 1)).double()
    err = (rec - ws).abs()
    assert torch.all(err <= torch.maximum(ws.abs() * 2.0 ** -22, torch.full_like(ws, 2.0 ** -25)))
    assert torch.isfinite(planes.float()).all()
    # tiling: element (plane p, channel co, k) sits at [co // 64, k // 32, p, co % 64, k % 32]
    t = E.tile_weight_planes(planes, 32)
    flat = planes.reshape(2, 128, -1)
    assert t.shape == (2, 3 * 3 * 64 // 32, 2, 64, 32)
    for (p, co, k) in [(0, 0, 0), (1, 5, 31), (0, 64, 32), (1, 127, 575), (0, 70, 300)]:
        assert t[co // 64, k // 32, p, co % 64, k % 32] == flat[p, co, k]


def test_f16x2_three_product_scheme_has_f32_sized_error():
    """The arithmetic claim of DESIGN.md: a . b from two scaled fp16 planes per operand and the three products
    a1b1 + a1b2 + a2b1, f32 accumulation, is as close to the exact dot product as a plain f32 matmul is -- also for
    tensors whose values sit far below 1 (that is what the power-of-two activation scale is for)."""
    import torch
    from deepemia_amd import engine as E

    g = torch.Generator().manual_seed(1)
    for amp in (3.0, 3.0e-4, 3.0e3):
        a = torch.relu(torch.randn((256, 2304), generator=g)) * amp
        a[:, ::7] *= 1e-3
        w = torch.randn((64, 2304), generator=g) * 0.02
        ref = a.double() @ w.double().t()
        planes, sw = E.split2_f16_scaled(w)
        amax = float(a.abs().max())
        s = 2.0 ** (13 - math.floor(math.log2(amax)))                    # the kernel's 2^(13 - ilogb(amax))
        ah = (a * s).to(torch.float16)
        al = (a * s - ah.float()).to(torch.float16)
        assert torch.isfinite(ah.float()).all()
        bh, bl = planes[0].float(), planes[1].float()
        y = ((ah.float() @ bl.t() + al.float() @ bh.t()) + ah.float() @ bh.t()) / (s * sw)
        err = float((y.double() - ref).abs().max() / ref.abs().max())
        f32 = float(((a @ w.t()).double() - ref).abs().max() / ref.abs().max())
        assert err < 1e-6 and err < 4 * f32 + 1e-7, (amp, err, f32)


def test_contrast_distribution_known_answers_and_host_percentiles():
    """measurements.py:195-215: the oracle restatement (np.histogram on the masked gray levels) on closed-form cases, and the
    product's host half (percentiles from the integer histogram the HIP kernel returns) against it, bit for bit."""
    from deepemia_amd.utils.measurements import contrast_percentiles
    from oracle import postproc_ref as P

    gray = np.full((20, 30), 77, dtype=np.uint8)
    mask = np.zeros((20, 30), dtype=bool)
    mask[3:9, 4:20] = True
    d10, d50, d90 = P.contrast_distribution(gray, mask)
    w = 255.0 / 256.0
    # one populated bin (77): the CDF jumps from 0 to 1 there, np.interp walks the step between the left edges 76 w and 77 w
    assert d10 == pytest.approx((76 + 0.10) * w, rel=1e-12) and d50 == pytest.approx((76 + 0.5) * w, rel=1e-12)
    assert d90 == pytest.approx((76 + 0.9) * w, rel=1e-12)
    assert P.contrast_distribution(gray, np.zeros_like(mask)) == (None, None, None)
    assert contrast_percentiles(np.zeros(256, dtype=np.int64)) == (None, None, None)
    # BGR -> gray: OpenCV's fixed-point weights (white stays 255, pure channels 29 / 150 / 76)
    px = np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255]]], dtype=np.uint8)
    assert P.bgr_to_gray(px).tolist() == [[255, 29, 150, 76]]
    rng = np.random.default_rng(3)
    for t in range(50):
        g = rng.integers(0, 256, size=(33, 41), dtype=np.uint8)
        if t % 4 == 0:
            g = (g // 16 * 16).astype(np.uint8)
        m = rng.random((33, 41)) > 0.5
        want = P.contrast_distribution(g, m)
        got = contrast_percentiles(np.bincount(g[m], minlength=256))
        assert tuple(float(v) for v in want) == got


def test_adaptive_confidence_threshold_follows_the_reference_rules():
    """inference.py:256-362: quality = 0.4 mean / 255 + 0.6 std / 128 (clipped); < 0.3 -> base x 0.7, < 0.5 -> base x 0.85;
    base and mode from the GLOBAL config; product and oracle agree."""
    from deepemia_amd.functions import inference as I
    from oracle import postproc_ref as P

    black = np.zeros((64, 64, 3), dtype=np.uint8)
    white = np.full((64, 64, 3), 255, dtype=np.uint8)
    checker = np.zeros((64, 64, 3), dtype=np.uint8)
    checker[::2, ::2] = 255
    checker[1::2, 1::2] = 255
    assert P.calculate_image_quality_score(black) == 0.0
    assert P.calculate_image_quality_score(white) == pytest.approx(0.4)
    assert P.calculate_image_quality_score(checker) == pytest.approx(0.4 * 0.5 + 0.6 * 127.5 / 128.0)
    cfg = {"inference_settings": {"confidence_mode": "auto", "class_specific_settings": {"class_0": {"confidence_threshold": 0.6}}}}
    small = {1}
    for img, f in ((black, 0.7), (white, 0.85), (checker, 1.0)):
        assert P.get_confidence_threshold(img, 0, small, cfg) == pytest.approx(0.6 * f)
        assert P.get_confidence_threshold(img, 1, small, cfg) == pytest.approx(0.3 * f)      # small-class default
        assert P.get_confidence_threshold(img, 2, small, cfg) == pytest.approx(0.5 * f)      # large-class default
        for c in (0, 1, 2):
            assert I.get_confidence_threshold(img, c, small, cfg) == P.get_confidence_threshold(img, c, small, cfg)
        assert I.calculate_image_quality_score(img) == P.calculate_image_quality_score(img)
    manual = {"inference_settings": {"confidence_mode": "manual", "class_specific_settings": {"class_0": {"confidence_threshold": 0.6}}}}
    assert P.get_confidence_threshold(black, 0, small, manual) == 0.6 == I.get_confidence_threshold(black, 0, small, manual)
    rng = np.random.default_rng(0)
    for _ in range(10):
        img = rng.integers(0, 256, size=(50, 70, 3), dtype=np.uint8)
        assert I.calculate_image_quality_score(img) == P.calculate_image_quality_score(img)


def test_p32_planes_roundtrip_and_weight_tiling():
    """Host side of the f16x2 format: P32 planes carry 22 significand bits of every value behind a 128-byte zero header,
    views keep the bytes, and the weight tiling is the documented [CoutPad / 64][channel group][tap][64][plane][32]."""
    import torch
    from deepemia_amd import engine as E, p32

    g = torch.Generator().manual_seed(4)
    x = torch.randn((3, 5, 7, 64), generator=g) * 37.0
    x[..., :5] *= 1e-4
    t = p32.from_f32(x)
    assert t.buf.dtype == torch.float16 and t.buf.numel() == 64 + 2 * x.numel() and not t.buf[:64].any()
    amax, s = float(t.meta[0, 0]), float(t.meta[0, 1])
    assert amax == float(x.abs().max()) and s == p32.plane_scale(amax) and amax * s < 32768.0 <= 2 * amax * s
    back = p32.to_f32(t)
    err = (back - x).abs()
    big = x.abs() >= amax * 2.0 ** -14                       # both planes normal: 22 significand bits
    assert float((err[big] / x.abs()[big]).max()) < 2.0 ** -21
    assert float(err[~big].max()) <= amax * 2.0 ** -38       # below that the LOW plane is denormal: absolute error, tiny
    v = t.view(3 * 5 * 7 * 2, 32)
    assert v.buf is t.buf and torch.equal(p32.to_f32(v).reshape(-1), back.reshape(3, 5, 7, 2, 32).reshape(-1))
    assert p32.plane_scale(0.0) == 1.0 and p32.plane_scale(1.0) == 2.0 ** 14 and p32.plane_scale(3.9) == 2.0 ** 13
    # scale groups (one per image): every group has the planes, max |x| and s it has as a tensor of its own
    xg = x * torch.tensor([1.0, 1e-3, 50.0]).view(3, 1, 1, 1)
    tg = p32.from_f32(xg, groups=3)
    assert tg.groups == 3 and tuple(tg.meta.shape) == (3, 2) and len(set(tg.meta[:, 1].tolist())) == 3
    per = 5 * 7 * 64 * 2
    for i in range(3):
        ti = p32.from_f32(xg[i:i + 1])
        assert torch.equal(ti.meta[0], tg.meta[i]) and torch.equal(ti.buf[64:], tg.buf[64 + i * per:64 + (i + 1) * per])
    assert torch.equal(p32.to_f32(tg)[1], p32.to_f32(p32.from_f32(xg[1:2]))[0])
    w = torch.randn((128, 3, 3, 64), generator=g)
    planes, sw = E.split2_f16_scaled(w)
    tiled = E.tile_weight_planes_p32(planes)
    assert tuple(tiled.shape) == (2, 2, 9, 64, 2, 32)
    for (p, co, kh, kw, ci) in [(0, 0, 0, 0, 0), (1, 5, 2, 1, 31), (0, 64, 1, 2, 32), (1, 127, 2, 2, 63), (0, 70, 0, 1, 40)]:
        assert tiled[co // 64, ci // 32, kh * 3 + kw, co % 64, p, ci % 32] == planes[p, co, kh, kw, ci]


# ---------------------------------------------------------------------------------------------------------------------
# the scale bar's LINE without OCR (Canny + HoughLinesP + collinear merge, label from the configuration)
def _scalebar_roi_image(seed=0, w=180, h=56, bar=(30, 34, 96, 5), noise=6.0):
    """A dark SEM-like strip with a bright bar (x, y, length, thickness), a blocky 'label' and a faint diagonal scratch."""
    g = np.random.default_rng(seed)
    img = np.full((h, w), 40.0) + g.normal(0.0, noise, (h, w))
    x, y, L, t = bar
    img[y:y + t, x:x + L] = 235.0 + g.normal(0.0, 2.0, (t, L))
    for k, cx in enumerate(range(60, 100, 9)):                      # digits as small bright blocks above the bar
        img[12:22, cx:cx + 5 + (k & 1)] = 225.0
    for k in range(40):                                             # a scratch that must not win
        img[5 + k // 2, 130 + k] = 140.0
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def test_scalebar_canny_and_houghlinesp_numpy_versions_equal_the_literal_loops():
    from deepemia_amd.utils import scalebar as S
    from oracle import scalebar_ref as R

    for seed, noise in ((0, 6.0), (1, 12.0), (2, 0.0)):
        roi = _scalebar_roi_image(seed, noise=noise)
        e_prod, e_ref = S.canny(roi, 50, 150), R.canny(roi, 50, 150)
        np.testing.assert_array_equal(e_prod, e_ref)
        assert 50 < int((e_prod > 0).sum()) < roi.size // 4
        l_prod = S.hough_lines_p(e_prod, 1, np.pi / 180, 50, 20, 10)
        l_ref = R.hough_lines_p(e_ref, 1, np.pi / 180, 50, 20, 10)
        assert l_prod == l_ref and len(l_prod) >= 2                 # same segments in the same order of discovery
    # thresholds given in the wrong order are swapped (canny.cpp), a flat image has no edges, and so no lines
    np.testing.assert_array_equal(S.canny(roi, 150, 50), S.canny(roi, 50, 150))
    flat = np.full((20, 30), 77, dtype=np.uint8)
    assert not S.canny(flat, 50, 150).any() and S.hough_lines_p(S.canny(flat, 50, 150)) == []
    # cv::RNG((uint64)-1): the multiply-with-carry recurrence, first draws written out by hand
    r = S.CvRNG()
    s1 = (0xFFFFFFFF * 4164903690 + 0xFFFFFFFF) & 0xFFFFFFFFFFFFFFFF
    assert r.next() == s1 & 0xFFFFFFFF and r.state == s1
    assert S.CvRNG().uniform(0, 1000) == (s1 & 0xFFFFFFFF) % 1000 and S.CvRNG().uniform(5, 5) == 5


def test_scalebar_line_selection_merge_and_calibration(monkeypatch):
    from deepemia_amd.utils import scalebar as S
    from deepemia_amd.functions import inference as I
    from oracle import scalebar_ref as R

    # merge_collinear_segments: product vs the reference transcription, and a hand case
    segs = [dict(x1=10, y1=20, x2=40, y2=21, length=30.0, intensity=200.0, dist_to_text=12.0, line_idx=0),
            dict(x1=48, y1=22, x2=90, y2=22, length=42.0, intensity=220.0, dist_to_text=30.0, line_idx=1),
            dict(x1=120, y1=22, x2=150, y2=22, length=30.0, intensity=90.0, dist_to_text=70.0, line_idx=2),
            dict(x1=60, y1=40, x2=20, y2=40, length=40.0, intensity=50.0, dist_to_text=5.0, line_idx=3)]
    got, want = S.merge_collinear_segments(segs, 15), R.merge_collinear_segments(segs, 15)
    assert got == want and len(got) == 4                            # sorted by left end: 10 | 20 | 48 | 120; y offsets / gaps split them
    two = S.merge_collinear_segments(segs[:2], 15)
    assert len(two) == 1 and (two[0]["x1"], two[0]["x2"], two[0]["y1"], two[0]["y2"]) == (10, 90, 21, 21)
    assert abs(two[0]["length"] - 80.0) < 1e-12 and abs(two[0]["intensity"] - (200 * 30 + 220 * 42) / 72) < 1e-12

    # the whole detector on a synthetic micrograph: bar of 96 px labelled "500" in the ROI of the default config
    H, W = 400, 600
    img = np.full((H, W, 3), 35, dtype=np.uint8)
    roi_cfg = {"x_start_factor": 0.7, "y_start_factor": 0.05, "width_factor": 1, "height_factor": 0.14}
    x0, y0 = int(W * 0.7), int(H * 0.05)
    roi = _scalebar_roi_image(3, w=W - x0, h=int(H * 0.14))
    img[y0:y0 + roi.shape[0], x0:, :] = roi[:, :, None]
    res = S.find_scale_bar_line(img, roi_cfg, "500 nm", text_center=(80, 17), intensity_threshold=100, proximity_threshold=100)
    assert res["psum"] == "500" and res["line"] is not None
    lx1, ly1, lx2, ly2 = res["line"]
    assert abs(abs(lx2 - lx1) - 96) <= 3 and abs(ly1 - ly2) <= 1 and y0 + 32 <= ly1 <= y0 + 41 and x0 + 27 <= min(lx1, lx2) <= x0 + 33
    assert abs(res["um_pix"] - 500.0 / res["length"]) < 1e-12 and 4.9 < res["um_pix"] < 5.5
    # too strict a brightness threshold or a label without digits: the reference's fallback
    assert S.find_scale_bar_line(img, roi_cfg, "500", (80, 17), intensity_threshold=250)["line"] is None
    assert S.find_scale_bar_line(img, roi_cfg, "nm", (80, 17))["psum"] == "0"

    # detect_scale_bar: configured label -> line detection (+ debug drawing); configured calibration wins; neither -> ("0", 1.0)
    monkeypatch.delenv("DEEPEMIA_UM_PER_PIXEL", raising=False)
    monkeypatch.setattr(I, "get_scalebar_roi_for_dataset", lambda name=None: roi_cfg)
    monkeypatch.setattr(I, "_scale_bar_settings", lambda name: {"label": "500", "text_center": [80, 17]})
    assert I.scale_bar_needs_image("x")
    psum, um = I.detect_scale_bar(img.copy(), dataset_name=None, intensity_threshold=100, proximity_threshold=100)
    assert psum == "500" and abs(um - res["um_pix"]) < 1e-12
    dbg = img.copy()
    I.detect_scale_bar(dbg, intensity_threshold=100, proximity_threshold=100, draw_debug=True)
    assert (dbg != img).any() and tuple(dbg[ly1, (lx1 + lx2) // 2]) == (0, 0, 255)          # the selected line, red in BGR
    monkeypatch.setattr(I, "_scale_bar_settings", lambda name: {"label": "500", "um_per_pixel": 0.25})
    assert I.detect_scale_bar(None) == ("500", 0.25) and not I.scale_bar_needs_image("x")
    monkeypatch.setattr(I, "_scale_bar_settings", lambda name: {})
    assert I.detect_scale_bar(img) == ("0", 1.0)


# ---------------------------------------------------------------------------------------------------------------------
# native decision loops (deepemia_amd/csrc/hostloops.hip) against the Python loops they replace
class _FakeAlgebra:
    """What ``InferencePipeline._greedy_keep`` / ``_dedup_smart_order`` read of a DeviceMaskAlgebra, from plain arrays."""

    def __init__(self, inter, area):
        self.I = inter.astype(np.int64)
        self.area = area.astype(np.int64)
        self.known = np.ones(inter.shape, dtype=bool)
        self.n = len(area)

    def intersections(self, pi, pj):          # pragma: no cover - every pair is known
        raise AssertionError("unexpected device call")


def _random_segments(g, seg_lens, quantised=False):
    """Areas, boxes and a consistent random pair-intersection table for masks in contiguous segments."""
    n = int(sum(seg_lens))
    area = g.integers(40, 4000, n).astype(np.int64)
    if quantised:
        area = (area // 200 + 1) * 200                       # many equal areas: IoUs that sit ON common thresholds
    first = np.repeat(np.concatenate(([0], np.cumsum(seg_lens)[:-1])), seg_lens).astype(np.int32)
    inter = np.zeros((n, n), dtype=np.int64)
    for s0, ln in zip(np.concatenate(([0], np.cumsum(seg_lens)[:-1])), seg_lens):
        for i in range(s0, s0 + ln):
            for j in range(i + 1, s0 + ln):
                if g.random() < 0.35:
                    hi = min(area[i], area[j])
                    v = int(g.integers(0, hi + 1)) if not quantised else int(g.choice([0, hi // 2, hi // 3, hi]))
                    inter[i, j] = inter[j, i] = v
    inter[np.arange(n), np.arange(n)] = area
    y0 = g.integers(0, 200, n)
    x0 = g.integers(0, 200, n)
    bbox = np.stack([y0, x0, y0 + g.integers(1, 120, n), x0 + g.integers(1, 120, n)], axis=1).astype(np.int64)
    return area, bbox, inter, first


def test_native_greedy_keep_and_smart_dedup_equal_the_python_loops():
    import ctypes as C
    from deepemia_amd import _lib
    from deepemia_amd.functions.inference import InferencePipeline as IP

    lib = _lib.load()
    g = np.random.default_rng(11)
    for trial in range(12):
        quant = trial % 2 == 1
        seg_lens = [int(v) for v in g.integers(0, 40, int(g.integers(1, 9)))]
        n = sum(seg_lens)
        if n == 0:
            continue
        area, bbox, inter, first = _random_segments(g, seg_lens, quant)
        ld = max(seg_lens)
        mat = np.zeros((n, ld), dtype=np.int32)              # the device layout: row i, column j - first[i], j > i
        for i in range(n):
            for j in range(i + 1, n):
                if first[j] == first[i]:
                    mat[i, j - first[i]] = inter[i, j]
        alg = _FakeAlgebra(inter, area)
        starts = np.concatenate(([0], np.cumsum(seg_lens))).astype(np.int32)
        # ---- greedy keep: every segment, also with a shortened (truncated) length
        for cut in (0, 3):
            lens = np.asarray([max(0, ln - cut) for ln in seg_lens], dtype=np.int32)
            for thr in (0.5, 0.7, 1.0 / 3.0):
                keep = np.zeros(n, dtype=np.uint8)
                _lib.check(lib.demia_host_greedy_keep(mat.ctypes.data, ld, first.ctypes.data, area.ctypes.data, starts[:-1].copy().ctypes.data,
                                                      lens.ctypes.data, len(seg_lens), float(thr), keep.ctypes.data), "greedy")
                for s, (s0, ln) in enumerate(zip(starts[:-1], lens)):
                    want = IP._greedy_keep(alg, range(int(s0), int(s0) + int(ln)), thr)
                    got = (int(s0) + np.nonzero(keep[s0:s0 + ln])[0]).tolist()
                    assert got == want, (trial, s, thr)
        # ---- smart dedup: per segment a random subset in random class mix, scores with ties
        items, scores, classes, tile_off = [], [], [], [0]
        per_tile = []
        for s0, ln in zip(starts[:-1], seg_lens):
            k0 = [int(i) for i in range(s0, s0 + ln) if g.random() < 0.8]
            sc = np.round(g.uniform(0.3, 1.0, len(k0)), 1 if quant else 6)            # 1 decimal: many ties
            cl = g.integers(0, 2, len(k0)) if not quant else np.zeros(len(k0), dtype=np.int64)
            per_tile.append((k0, sc, cl))
            items += k0
            scores += sc.tolist()
            classes += cl.tolist()
            tile_off.append(len(items))
        if not items:
            continue
        items_a = np.asarray(items, dtype=np.int32)
        sc_a, cl_a = np.asarray(scores, dtype=np.float64), np.asarray(classes, dtype=np.int32)
        off_a = np.asarray(tile_off, dtype=np.int32)
        for thr in (0.7, 0.4, 0.5):
            keep_out = np.zeros(len(items), dtype=np.int32)
            keep_cnt = np.zeros(len(seg_lens), dtype=np.int32)
            _lib.check(lib.demia_host_dedup_smart(mat.ctypes.data, ld, first.ctypes.data, area.ctypes.data, bbox.ctypes.data, items_a.ctypes.data,
                                                  sc_a.ctypes.data, cl_a.ctypes.data, off_a.ctypes.data, len(seg_lens), float(thr),
                                                  keep_out.ctypes.data, keep_cnt.ctypes.data), "dedup")
            for t, (k0, sc, cl) in enumerate(per_tile):
                bb = [(int(bbox[i, 0]), int(bbox[i, 2]), int(bbox[i, 1]), int(bbox[i, 3])) for i in k0]
                want = IP._dedup_smart_order(alg, k0, sc.tolist(), cl.tolist(), bb, thr) if k0 else []
                got = keep_out[off_a[t]:off_a[t] + keep_cnt[t]].tolist()
                assert got == want, (trial, t, thr)


def test_native_csv_text_equals_csv_writer_byte_for_byte():
    """``measurement_csv_text`` (float columns through ``demia_host_repr_rows``) against ``csv.writer`` over
    ``measurement_rows`` -- what ``write_measurements`` writes: random measurement values of every magnitude (random bit
    patterns, integers, tiny / huge exponents, signed zeros, inf / nan), several tiles, contours below the area gate, class
    ids beyond the class list, and a name that needs quoting (falls back to csv.writer)."""
    import csv, io
    from deepemia_amd.functions.inference import measurement_csv_text, measurement_rows

    g = np.random.default_rng(3)
    special = np.array([0.0, -0.0, 1.0, 1e16, 1e15, 9999999999999998.0, 1e-4, 1e-5, 5e-324, 1.7976931348623157e308, 0.1, 1 / 3,
                        float("inf"), float("-inf"), float("nan"), 123456789012345678.0, 1e22, 2.5e-7, 100.0, 0.30000000000000004])

    def recs_for(n_inst):
        out = []
        for _ in range(n_inst):
            cont = []
            for _ in range(int(g.integers(1, 4))):
                kind = g.integers(0, 4)
                if kind == 0:
                    v = g.random(12) * 10.0 ** g.integers(-9, 20, 12)
                elif kind == 1:
                    v = np.frombuffer(g.bytes(96), dtype=np.float64).copy()
                elif kind == 2:
                    v = g.integers(0, 10 ** 7, 12).astype(np.float64)
                else:
                    v = g.choice(special, 12)
                cont.append({"area": float(g.choice([1.0, 3.0, 50.0, 900.0])), "values": v})
            out.append(cont)
        return out

    for names in (["tile0.tif", "em_12.png", "a b.tif"], ['we,ird "name".tif', "x.tif"]):
        tiles = []
        for nm in names:
            n_inst = int(g.integers(0, 40))
            tiles.append((nm, g.integers(0, 4, n_inst).tolist(), recs_for(n_inst)))
        got = measurement_csv_text(tiles, ("pore", "throat"), 5.0, psum="600")
        buf = io.StringIO()
        w = csv.writer(buf)
        rows = 0
        for nm, cl, rc in tiles:
            for r in measurement_rows(nm, cl, rc, ("pore", "throat"), 5.0, None, "600"):
                w.writerow(r)
                rows += 1
        assert got == buf.getvalue()
        assert rows > 10
    assert measurement_csv_text([("t.tif", [], [])], ("a",), 5.0) == ""


def test_native_rle_text_equals_the_reference_pinned_encoding():
    """``demia_host_rle_text`` (host code behind the C ABI: the EncodedPixels column of R50_flip_results.csv for all masks of an
    image in one call, from bbox-cropped packed words) against ``rle_encoding`` -- itself pinned by the reference's goldens
    (tests/golden/mask_utils.npz) -- incl. full-height columns whose runs the column-major scan glues together, an empty mask,
    a single corner pixel, a full frame, and a buffer that is too small (returns -(bytes needed))."""
    import ctypes as C

    import torch

    from deepemia_amd import _lib, parallel
    from deepemia_amd.utils.mask_utils import rle_encoding

    lib = _lib.load()
    rng = np.random.default_rng(0)
    H, W, M = 70, 100, 7
    masks = np.zeros((M, H, W), dtype=bool)
    masks[0, 10:20, 5:9] = True
    masks[1] = rng.random((H, W)) > 0.5
    masks[2, :, 30:41] = True
    masks[3, :, 33] = True
    masks[3, 0, 34] = True
    masks[5, H - 1, W - 1] = True
    masks[6] = True
    wpr = (W + 31) // 32
    m = np.concatenate([masks, np.zeros((M, H, wpr * 32 - W), dtype=bool)], axis=2)
    packed = np.packbits(m.reshape(M, H, wpr, 32), axis=-1, bitorder="little").view(np.uint32).reshape(M, H, wpr)
    bb, area = np.full((M, 4), -1, dtype=np.int32), np.zeros(M, dtype=np.int64)
    for i in range(M):
        ys, xs = np.nonzero(masks[i])
        if len(ys):
            bb[i], area[i] = [ys.min(), xs.min(), ys.max(), xs.max()], len(ys)
    hdr, pay = parallel.encode_instance_table(torch.from_numpy(packed.view(np.int32)), [0.0] * M, [0] * M, [0] * M, bb, area)
    pay = np.ascontiguousarray(pay.numpy())
    offs = np.ascontiguousarray(parallel._offsets(parallel._payload_lengths(hdr.numpy())), dtype=np.int64)
    toff = np.zeros(M + 1, dtype=np.int64)
    small = C.create_string_buffer(8)
    need = lib.demia_host_rle_text(pay.ctypes.data, bb.ctypes.data, offs.ctypes.data, M, H, small, 8, toff.ctypes.data)
    assert need < 0
    out = C.create_string_buffer(-need)
    got = lib.demia_host_rle_text(pay.ctypes.data, bb.ctypes.data, offs.ctypes.data, M, H, out, -need, toff.ctypes.data)
    assert got == -need
    raw = out.raw[:got].decode("ascii")
    for i in range(M):
        assert raw[toff[i]:toff[i + 1]] == " ".join(map(str, rle_encoding(masks[i].astype(np.uint8)))), i


def test_cli_worker_count_rule(tmp_path, monkeypatch):
    """``main.py::worker_processes``: how many processes share one GPU for a run -- an explicit ``DEEPEMIA_WORKERS`` wins (1..6), ``auto``
    looks at the input folder (>= 48 images: 4, >= 24: 3, >= 8: 2, else 1), never starts workers when a GCS download is pending, and is
    bounded by the free device memory it can read from sysfs (~45 GiB per process).  Pure host logic: no GPU call."""
    import types

    import yaml

    import main as cli
    from deepemia_amd.utils import config as C

    cfgdir = tmp_path / "cfg"
    cfgdir.mkdir()
    (cfgdir / "config.yaml").write_text(yaml.safe_dump({"bucket": None, "paths": {"split_dir": str(tmp_path / "s"), "category_json": str(tmp_path / "d.json"),
                                                                                   "local_dataset_root": str(tmp_path)}}))
    inf = tmp_path / "DATASET" / "INFERENCE"
    inf.mkdir(parents=True)
    monkeypatch.setenv("DEEPEMIA_CONFIG_DIR", str(cfgdir))
    monkeypatch.setenv("DEEPEMIA_OFFLINE", "1")
    C.reset_cache()
    args = types.SimpleNamespace(task="inference", dataset_name="x", download=True)
    monkeypatch.setattr(cli, "free_vram_gib", lambda: None)

    def with_images(n):
        for f in inf.glob("*.png"):
            f.unlink()
        for i in range(n):
            (inf / f"im_{i}.png").write_bytes(b"x")
        return cli.worker_processes(args)

    monkeypatch.delenv("DEEPEMIA_WORKERS", raising=False)
    assert with_images(3) == 1 and with_images(8) == 2 and with_images(23) == 2 and with_images(24) == 3 and with_images(47) == 3 and with_images(48) == 4
    monkeypatch.setattr(cli, "free_vram_gib", lambda: 100.0)          # room for two processes only
    assert with_images(30) == 2
    monkeypatch.setattr(cli, "free_vram_gib", lambda: 10.0)
    assert with_images(30) == 1
    monkeypatch.setattr(cli, "free_vram_gib", lambda: None)
    monkeypatch.setenv("DEEPEMIA_WORKERS", "5")
    assert with_images(1) == 5
    monkeypatch.setenv("DEEPEMIA_WORKERS", "99")
    assert cli.worker_processes(args) == 6
    monkeypatch.setenv("DEEPEMIA_WORKERS", "auto")
    assert cli.worker_processes(types.SimpleNamespace(task="train", dataset_name="x", download=True)) == 1
    C.reset_cache()
