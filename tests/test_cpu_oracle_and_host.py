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
    assert lib.demia_abi_version() == 3
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
