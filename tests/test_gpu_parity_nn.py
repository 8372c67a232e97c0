"""GPU parity tests proper: every stage of the HIP predictor against the CPU oracle, called
through the C ABI of libdeepemia_hip.so (ctypes), on the same seeded inputs.

Tolerances (floating point; integer / order / byte work is bit-exact):
  * resize + normalise: bit-exact vs Pillow;
  * f32 convolutions (exact-f32 MFMA): <= 2e-5 of the tensor's max |value|;
  * bf16 convolutions: <= 3e-2 of max |value| (bf16 has 8 significand bits);
  * proposal / detection selection on identical inputs: identical order and classes,
    boxes within 2e-3 px (expf differs by <= 2 ulp between libm and the GPU);
  * pasted masks: IoU >= 0.999 (north_star), here observed 1.0;
  * end to end in f32: same instances in the same order, mask IoU >= 0.999.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from PIL import Image

pytestmark = pytest.mark.gpu

K = 2
THR = 0.3


@pytest.fixture(scope="module")
def env(gpu_device):
    from deepemia_amd import synth
    from deepemia_amd.engine import MaskRCNNEngine
    from oracle import maskrcnn_ref as R

    sd = synth.random_d2_state_dict(50, K, seed=0)
    img = synth.em_tile(0, 1024)
    ref = R.predict(img, sd, 50, THR, return_intermediates=True)
    eng = MaskRCNNEngine(sd, 50, K, THR, gpu_device, "f32")
    from deepemia_amd import _lib
    dev_lib = _lib.is_dev_build()            # f32x3 / f16x2r live in the dev build of the library only (DEEPEMIA_DEV_LIB=1)
    eng3 = MaskRCNNEngine(sd, 50, K, THR, gpu_device, "f32x3") if dev_lib else None   # f32 operands on the bf16 pipe: same parity bar
    eng2 = MaskRCNNEngine(sd, 50, K, THR, gpu_device, "f16x2")   # the default: fp16 pipe, activations as two pre-scaled fp16 planes (P32)
    eng2r = MaskRCNNEngine(sd, 50, K, THR, gpu_device, "f16x2r") if dev_lib else None  # the same arithmetic from f32 activations (round 1's kernel)
    return dict(sd=sd, img=img, ref=ref, eng=eng, f32=eng, f32x3=eng3, f16x2=eng2, f16x2r=eng2r, R=R, synth=synth, dev=gpu_device)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def test_library_is_the_hip_one(env):
    lib = env["eng"].lib
    assert lib.demia_build_arch().decode() == "gfx950"
    assert lib.demia_abi_version() == 6


@pytest.mark.parametrize("hw", [(1024, 1024), (2048, 2048), (600, 600), (700, 1100), (1000, 2000), (601, 1001), (333, 517)])
def test_resize_normalise_bit_exact_vs_pillow(env, hw):
    eng, R = env["eng"], env["R"]
    h, w = hw
    rng = np.random.default_rng(h * 7 + w)
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    x = torch.from_numpy(img)[None].to(env["dev"])
    xin, newh, neww, ph, pw = eng.preprocess(x)
    ref = np.asarray(Image.fromarray(img).resize((neww, newh), Image.BILINEAR)).astype(np.float32)
    ref = ref - np.array(R.PIXEL_MEAN, dtype=np.float32)
    got = xin[0].cpu().numpy()
    assert got.shape == (ph + 6, pw + 8, 4)
    assert (newh, neww) == R.resize_shape(h, w)
    np.testing.assert_array_equal(got[3:3 + newh, 3:3 + neww, :3], ref)
    # border, padding and the 4th channel are zero
    mask = np.ones_like(got, dtype=bool)
    mask[3:3 + newh, 3:3 + neww, :3] = False
    assert not got[mask].any()


@pytest.mark.parametrize("hw", [(512, 512), (300, 500), (2048, 2048)])
def test_stem_on_the_matrix_pipe_equals_the_f32_stem(env, hw):
    """``demia_stem_conv_mfma`` (the stem of the f16x2 path: input and weights as two fp16 planes, three MFMAs per product)
    against the exact-f32 VALU stem on the same zero-bordered input: f32-sized error, ReLU zeros in the same places up to
    that error, and identical results for an image alone and inside a batch."""
    import ctypes as C
    from deepemia_amd import _lib

    eng = env["f16x2"]
    h, w = hw
    rng = np.random.default_rng(h + w)
    imgs = rng.integers(0, 256, size=(2, h, w, 3), dtype=np.uint8)
    imgs[1] = (imgs[1] // 3)                                       # a darker image beside a bright one
    x = torch.from_numpy(imgs).to(env["dev"])
    xin, newh, neww, ph, pw = eng.preprocess(x)
    st = int(torch.cuda.current_stream().cuda_stream)
    a = torch.empty((2, ph // 2, pw // 2, 64), dtype=torch.float32, device=env["dev"])
    b = torch.empty_like(a)
    _lib.check(eng.lib.demia_stem_conv(_lib.ptr(xin), _lib.ptr(eng.stem_w), _lib.ptr(eng.stem_scale), _lib.ptr(eng.stem_bias), _lib.ptr(a),
                                       2, ph, pw, _lib.F32, st), "stem")
    _lib.check(eng.lib.demia_stem_conv_mfma(_lib.ptr(xin), _lib.ptr(eng.stem_planes), _lib.ptr(eng.stem_scale_mfma), _lib.ptr(eng.stem_bias),
                                            _lib.ptr(b), 2, ph, pw, eng.stem_s_in, st), "stem mfma")
    err = float((a - b).abs().max() / a.abs().max())
    assert err < 2e-6, err
    assert float(a.abs().max()) > 1.0
    one = torch.empty((1, ph // 2, pw // 2, 64), dtype=torch.float32, device=env["dev"])
    _lib.check(eng.lib.demia_stem_conv_mfma(_lib.ptr(xin[1:].contiguous()), _lib.ptr(eng.stem_planes), _lib.ptr(eng.stem_scale_mfma),
                                            _lib.ptr(eng.stem_bias), _lib.ptr(one), 1, ph, pw, eng.stem_s_in, st), "stem mfma")
    assert torch.equal(one[0], b[1])
    # stem + max pool in ONE kernel (planes out, the f32 stem output never written) = the two kernels, bit for bit: the same
    # MFMA sequence per conv output, the same maxima, the same plane split and |x| maxima
    from deepemia_amd import p32
    for groups in (1, 2):
        s_out = p32.plane_scale(eng.stem_bound)
        two = p32.alloc((2, ph // 4, pw // 4, 64), env["dev"], groups=groups)
        fused = p32.alloc((2, ph // 4, pw // 4, 64), env["dev"], groups=groups)
        _lib.check(eng.lib.demia_maxpool3x3s2_p32(_lib.ptr(b), _lib.ptr(two.buf), _lib.ptr(two.meta), s_out, 2, ph // 2, pw // 2, 64, groups, 0, st), "pool")
        _lib.check(eng.lib.demia_stem_pool_mfma(_lib.ptr(xin), _lib.ptr(eng.stem_planes), _lib.ptr(eng.stem_scale_mfma), _lib.ptr(eng.stem_bias),
                                                _lib.ptr(fused.buf), _lib.ptr(fused.meta), 2, ph, pw, eng.stem_s_in, s_out, groups, 0, st), "fused")
        assert torch.equal(fused.buf, two.buf) and torch.equal(fused.meta, two.meta), groups
        assert float(fused.meta[:, 0].min()) > 0.0


CONV_CASES = [
    # cin, cout, k, stride, pad, h, w, n, relu, res
    (64, 64, 1, 1, 0, 50, 50, 2, True, 0),
    (64, 256, 1, 1, 0, 37, 41, 1, False, 1),
    (256, 128, 1, 2, 0, 50, 50, 2, True, 0),
    (64, 64, 3, 1, 1, 33, 29, 2, True, 0),
    (128, 128, 3, 1, 1, 25, 25, 3, True, 0),
    (256, 15, 1, 1, 0, 13, 13, 2, False, 0),
    (256, 256, 3, 1, 1, 14, 14, 5, True, 0),
    (512, 256, 1, 1, 0, 50, 50, 1, False, 2),
    (64, 256, 3, 1, 1, 300, 300, 1, True, 1),      # 1408 tiles of 128 x 128: the full tile of the split kernel
    (128, 192, 3, 2, 1, 61, 47, 2, True, 0),       # CoutPad % 128 != 0: the 128 x 64 tile
    (64, 256, 1, 1, 0, 512, 512, 1, True, 1),      # 4096 tiles: the 128 x 128 tile in every split mode (two-pass epilogue, residual)
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_p32_vs_torch(env, case):
    """The default arithmetic (f16x2 on P32 planes, demia_conv2d_p32) at the tolerance of the exact-f32 kernel; the
    activations are scaled from 1e-3 to 1e3 and the epilogue's measured max |out| and plane scale are checked too."""
    from deepemia_amd import engine as E, p32
    from deepemia_amd._lib import ACT_NONE, ACT_RELU, RES_NONE, RES_SAME, RES_UP2

    cin, cout, k, stride, pad, h, w, n, relu, res = case
    if cin % 32:
        pytest.skip("Cin must be a multiple of 32")
    g = torch.Generator().manual_seed(cin * 131 + cout * 7 + k)
    eng, dev = env["f16x2"], env["dev"]
    x = torch.randn((n, cin, h, w), generator=g) * float(10.0 ** ((cin % 7) - 3))
    wt = torch.randn((cout, cin, k, k), generator=g) / (cin * k * k) ** 0.5
    scale = torch.rand((cout,), generator=g) + 0.5
    bias = torch.randn((cout,), generator=g) * 0.1
    y = F.conv2d(x.double(), wt.double(), None, stride=stride, padding=pad) * scale.double().view(1, -1, 1, 1) + bias.double().view(1, -1, 1, 1)
    ho, wo = y.shape[2], y.shape[3]
    residual = None
    if res == 1:
        residual = torch.randn((n, cout, ho, wo), generator=g)
        y = y + residual.double()
    elif res == 2:
        residual = torch.randn((n, cout, (ho + 1) // 2, (wo + 1) // 2), generator=g)
        y = y + F.interpolate(residual.double(), scale_factor=2.0, mode="nearest")[:, :, :ho, :wo]
    if relu:
        y = F.relu(y)
    sd = {"l.weight": wt, "l.bias": bias}
    eng._used = set()
    L = eng._conv(sd, "l", stride=stride, pad=pad, bias=True)
    # a FrozenBN-style scale on top of the bias path: scale3 = scale / 2^e(co), bound rescaled alike
    L.scale3 = (L.scale3 * scale.to(dev)).contiguous()
    L.wbound = float((scale.abs() * wt.abs().flatten(1).sum(1)).max())
    out_f32 = cout % 32 != 0
    xp = p32.from_f32(nhwc(x).to(dev))
    rp = None if residual is None else p32.from_f32(nhwc(residual).to(dev))
    out = eng.conv_p32(xp, L, act=ACT_RELU if relu else ACT_NONE, residual=rp, res_mode=(RES_NONE, RES_SAME, RES_UP2)[res],
                       out_f32=out_f32, out_ld=(cout + 3) // 4 * 4 if out_f32 else 0)
    got = (out[..., :cout] if out_f32 else eng.dense(out)).double().cpu().permute(0, 3, 1, 2)
    err = float((got - y).abs().max() / y.abs().max())
    assert err <= 2e-5, err
    if not out_f32:
        amax, s = float(out.meta[0, 0]), float(out.meta[0, 1])
        assert abs(amax - float(y.abs().max())) <= 1e-5 * float(y.abs().max())
        assert amax * s < 32768.0 and s == 2.0 ** round(np.log2(s))


@pytest.mark.parametrize("case", [(3, 64, 64, 12, 12, 3, 1, 1, 1), (5, 32, 128, 16, 9, 1, 1, 0, 2), (2, 64, 96, 31, 17, 3, 2, 1, 0),
                                  (4, 128, 256, 13, 13, 3, 1, 1, 0)])
def test_conv_p32_scale_groups_equal_the_images_alone(env, case):
    """``demia_conv_p32_desc.groups``: one {max |x|, s} pair per image.  Images of very different amplitude go through one
    launch (tiles straddle the image boundaries: 144, 169 ... rows per image against 128 / 256-row tiles); every image must
    come out with exactly the planes, scale and max |x| it gets when it is convolved alone."""
    from deepemia_amd import p32
    from deepemia_amd._lib import ACT_RELU, RES_NONE, RES_SAME, RES_UP2

    n, cin, cout, h, w, k, stride, pad, res = case
    eng, dev = env["f16x2"], env["dev"]
    g = torch.Generator().manual_seed(sum(case))
    amp = torch.tensor([10.0 ** (2 - 2 * (i % 3)) for i in range(n)]).view(n, 1, 1, 1)
    x = torch.randn((n, h, w, cin), generator=g) * amp
    wt = torch.randn((cout, cin, k, k), generator=g) / (cin * k * k) ** 0.5
    eng._used = set()
    L = eng._conv({"l.weight": wt, "l.bias": torch.randn((cout,), generator=g) * 0.1}, "l", stride=stride, pad=pad, bias=True)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    r = None
    if res == 1:
        r = torch.randn((n, ho, wo, cout), generator=g) * amp
    elif res == 2:
        r = torch.randn((n, (ho + 1) // 2, (wo + 1) // 2, cout), generator=g) * amp
    mode = (RES_NONE, RES_SAME, RES_UP2)[res]
    xp = p32.from_f32(x.to(dev), groups=n)
    rp = None if r is None else p32.from_f32(r.to(dev), groups=n)
    assert xp.meta.shape == (n, 2) and len(set(xp.meta[:, 1].tolist())) > 1
    out = eng.conv_p32(xp, L, act=ACT_RELU, residual=rp, res_mode=mode)
    assert out.groups == n
    per = ho * wo * cout * 2
    for i in range(n):
        xi = p32.from_f32(x[i:i + 1].to(dev))
        ri = None if r is None else p32.from_f32(r[i:i + 1].to(dev))
        oi = eng.conv_p32(xi, L, act=ACT_RELU, residual=ri, res_mode=mode)
        assert torch.equal(oi.meta[0], out.meta[i]), (i, oi.meta, out.meta)
        assert torch.equal(oi.buf[64:], out.buf[64 + i * per:64 + (i + 1) * per]), i


@pytest.mark.parametrize("prec", ["f32", "f32x3", "f16x2r", "bf16x2", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_igemm_vs_torch(env, case, prec):
    from conftest import needs_dev_build
    needs_dev_build(prec)
    from deepemia_amd import engine as E
    from deepemia_amd._lib import ACT_NONE, ACT_RELU, RES_NONE, RES_SAME, RES_UP2

    cin, cout, k, stride, pad, h, w, n, relu, res = case
    g = torch.Generator().manual_seed(cin * 131 + cout * 7 + k)
    eng = (env["f16x2r"] if prec == "f16x2r" else env["eng"]) if prec != "bf16" else E.MaskRCNNEngine.__new__(E.MaskRCNNEngine)
    if prec == "bf16":
        eng.__dict__.update(env["eng"].__dict__)
        eng.dt, eng.tdt, eng.precision = E.BF16, torch.bfloat16, "bf16"
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, k, k), generator=g) / (cin * k * k) ** 0.5
    scale = torch.rand((cout,), generator=g) + 0.5
    bias = torch.randn((cout,), generator=g) * 0.1
    if prec == "bf16":  # compare like with like: reference sees the same rounded operands
        x, wt = x.bfloat16().float(), wt.bfloat16().float()
    y = F.conv2d(x, wt, None, stride=stride, padding=pad) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    ho, wo = y.shape[2], y.shape[3]
    residual = None
    if res == 1:
        residual = torch.randn((n, cout, ho, wo), generator=g)
        y = y + residual
    elif res == 2:
        residual = torch.randn((n, cout, (ho + 1) // 2, (wo + 1) // 2), generator=g)
        y = y + F.interpolate(residual, scale_factor=2.0, mode="nearest")[:, :, :ho, :wo]
    if relu:
        y = F.relu(y)
    cout_pad = (cout + 31) // 32 * 32
    wp = torch.zeros((cout_pad, k, k, cin))
    wp[:cout] = wt.permute(0, 2, 3, 1)
    dev = env["dev"]
    L = E.ConvLayer(wp.to(dev, eng.tdt), scale.to(dev), bias.to(dev), cin, cout, cout_pad, k, k, stride, pad)
    if prec in ("f32x3", "bf16x2"):
        # f32 operands on the bf16 matrix pipe (3-way split, 6 products): same tolerance as the exact-f32 kernel;
        # bf16x2 = 2 planes, 3 products: 16-bit operands, tolerance 2e-4.  Shapes the split kernel does not take stay
        # on the f32 kernel, exactly as the engine packs them
        if cout_pad % 64 or cin % 32:
            pytest.skip("shape stays on the exact-f32 kernel")
        L.w3 = E.tile_weight_planes(E.split3_bf16(wp).to(dev)[: 3 if prec == "f32x3" else 2])
    if prec == "f16x2r":
        # two fp16 planes per operand with exact power-of-two scales (weights per channel here, activations from the
        # |x| bound inside the kernel), three MFMAs per product: same tolerance as the exact-f32 kernel
        if cout_pad % 64 or cin % 32:
            pytest.skip("shape stays on the exact-f32 kernel")
        planes, sw = E.split2_f16_scaled(wp.to(dev))
        L.w3 = E.tile_weight_planes(planes, eng.lib.demia_conv_f16x2_kstep())
        L.scale3 = (scale.to(dev) / sw[:cout]).contiguous()
        x = x * float(10.0 ** ((cin % 7) - 3))          # exercise the activation scale: |x| from 1e-3 to 1e3
        y = F.conv2d(x, wt, None, stride=stride, padding=pad) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
        if res == 1:
            y = y + residual
        elif res == 2:
            y = y + F.interpolate(residual, scale_factor=2.0, mode="nearest")[:, :, :ho, :wo]
        if relu:
            y = F.relu(y)
    rdev = None if residual is None else nhwc(residual).to(dev, eng.tdt)
    if prec == "bf16" and residual is not None:
        # the reference must see the rounded residual as well
        rr = rdev.float().cpu().permute(0, 3, 1, 2)
        y = F.conv2d(x, wt, None, stride=stride, padding=pad) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
        y = y + (rr if res == 1 else F.interpolate(rr, scale_factor=2.0, mode="nearest")[:, :, :ho, :wo])
        y = F.relu(y) if relu else y
    # the residual is read in the OUTPUT dtype (ABI contract), so a bf16 residual means bf16 output
    odt = torch.float32 if (prec != "bf16" or residual is None) else torch.bfloat16
    out = eng.conv(nhwc(x).to(dev, eng.tdt), L, act=ACT_RELU if relu else ACT_NONE, residual=rdev,
                   res_mode=(RES_NONE, RES_SAME, RES_UP2)[res], out_dtype=odt)
    got = out.float().cpu().permute(0, 3, 1, 2)
    tol = {"f32": 2e-5, "f32x3": 2e-5, "f16x2r": 2e-5, "bf16x2": 2e-4, "bf16": 3e-2}[prec]
    if prec == "f16x2r":
        # the epilogue's |out| bound (the next layer's operand scale) is the exact maximum
        assert float(out._amax.item()) == float(out.abs().max().item())
    err = float((got - y).abs().max() / y.abs().max())
    assert err <= tol, err


@pytest.mark.parametrize("prec", ["f32", "f32x3", "f16x2", "f16x2r"])
def test_backbone_fpn_features_f32(env, prec):
    from conftest import needs_dev_build
    needs_dev_build(prec)
    eng, d = env[prec], env["ref"]["dbg"]
    x = torch.from_numpy(env["img"])[None].to(env["dev"])
    xin, newh, neww, ph, pw = eng.preprocess(x)
    feats = eng.backbone(xin, ph, pw)
    for k in ("res2", "res3", "res4", "res5", "p2", "p3", "p4", "p5", "p6"):
        a = eng.dense(feats[k])[0].permute(2, 0, 1).cpu()
        b = d["feats"][k][0]
        assert a.shape == b.shape
        assert float((a - b).abs().max() / b.abs().max()) < 2e-5, k


def _oracle_heads(env):
    """RPN head tensors of the oracle in the [N, H, W, 16] layout of the C ABI."""
    d = env["ref"]["dbg"]
    heads = []
    for li, name in enumerate(("p2", "p3", "p4", "p5", "p6")):
        f = d["feats"][name]
        h, w = f.shape[2], f.shape[3]
        lv = d["rpn"]["per_level"][li]
        hd = torch.zeros((1, h, w, 16))
        hd[0, :, :, :3] = lv["logits"].view(h, w, 3)
        hd[0, :, :, 3:15] = lv["deltas"].view(h, w, 12)
        heads.append(hd.to(env["dev"]))
    return heads


def test_rpn_selection_on_oracle_heads(env):
    from deepemia_amd import _lib, engine as E

    eng, d = env["eng"], env["ref"]["dbg"]
    heads = _oracle_heads(env)
    newh, neww = d["resized"].shape[:2]
    dev = env["dev"]
    boxes = torch.empty((1, 1000, 4), device=dev)
    scores = torch.empty((1, 1000), device=dev)
    count = torch.empty((1,), dtype=torch.int32, device=dev)
    ws = torch.empty((int(eng.lib.demia_rpn_workspace_bytes(1)),), dtype=torch.uint8, device=dev)
    desc = _lib.RpnDesc()
    for i, hd in enumerate(heads):
        desc.head[i] = _lib.ptr(hd)
        desc.H[i], desc.W[i], desc.stride[i] = hd.shape[1], hd.shape[2], E.STRIDES[i]
    cell = E.cell_anchor_table()
    desc.cell_anchors = cell.ctypes.data
    desc.head_ld, desc.N, desc.img_h, desc.img_w = 16, 1, newh, neww
    desc.pre_topk, desc.post_topk, desc.nms_thresh = 1000, 1000, 0.7
    desc.out_boxes, desc.out_scores, desc.out_count, desc.workspace = map(_lib.ptr, (boxes, scores, count, ws))
    _lib.check(eng.lib.demia_rpn_proposals(C.byref(desc), eng._stream()), "rpn")
    torch.cuda.synchronize()
    n = int(count[0])
    assert n == d["prop_boxes"].shape[0]
    # identical inputs -> identical order: scores are the logits themselves, bit-exact
    np.testing.assert_array_equal(scores[0, :n].cpu().numpy(), d["prop_scores"].numpy())
    assert float((boxes[0, :n].cpu() - d["prop_boxes"]).abs().max()) < 2e-3


def test_roi_align_on_oracle_inputs(env):
    eng, d = env["eng"], env["ref"]["dbg"]
    dev = env["dev"]
    feats = {k: nhwc(d["feats"][k]).to(dev) for k in ("p2", "p3", "p4", "p5")}
    for boxes, P, ref in ((d["prop_boxes"], 7, d["pooled"]), (d["det_boxes"], 14, d["mpooled"])):
        n = boxes.shape[0]
        b = boxes[None].contiguous().to(dev)
        cnt = torch.tensor([n], dtype=torch.int32, device=dev)
        out = eng.roi_align(feats, b, cnt, P)[0].permute(0, 3, 1, 2).cpu()
        err = float((out - ref).abs().max() / ref.abs().max())
        assert err < 1e-5, (P, err)
    # rows beyond count are zero
    cnt = torch.tensor([3], dtype=torch.int32, device=dev)
    out = eng.roi_align(feats, d["det_boxes"][None].contiguous().to(dev), cnt, 14)
    assert not out[0, 3:].any()


def test_roi_launch_order_is_a_permutation_and_changes_no_bit(env):
    """``demia_roi_order`` (round 5): workgroup b of ROIAlign pools ROI order[b] -- per image sorted by (FPN level, row band, column)
    and dealt to the XCDs in runs -- while every ROI's output row stays where it is: the order is a permutation of each image's
    ROIs (unused slots last) and the pooled tensors are bit-identical with and without it, for 1000 ROIs per image (8 | R: the
    XCD interleave) and for 100 (plain sorted order)."""
    import ctypes as C

    from deepemia_amd import _lib

    eng, dev = env["eng"], env["dev"]
    g = torch.Generator().manual_seed(11)
    n = 3
    feats = {k: torch.randn(n, h, w, 256, generator=g).to(dev) for k, (h, w) in {"p2": (64, 80), "p3": (32, 40), "p4": (16, 20), "p5": (8, 10)}.items()}
    for r, P in ((1000, 7), (100, 14)):
        cx, cy = torch.rand(n, r, generator=g) * 320, torch.rand(n, r, generator=g) * 256
        sz = torch.exp(torch.rand(n, r, generator=g) * 5.0) * 3.0
        boxes = torch.stack([cx - sz, cy - sz * 0.7, cx + sz, cy + sz * 0.7], dim=2).clamp(min=0).contiguous().to(dev)
        cnt = torch.tensor([r, r - 37, 5], dtype=torch.int32, device=dev)
        order = torch.empty((n * r,), dtype=torch.int32, device=dev)
        _lib.check(eng.lib.demia_roi_order(_lib.ptr(boxes), _lib.ptr(cnt), n, r, _lib.ptr(order), eng._stream()), "demia_roi_order")
        o = order.cpu().numpy().reshape(n, r)
        for i in range(n):
            assert sorted(o[i].tolist()) == list(range(i * r, (i + 1) * r))
        eng.roi_order = True
        a = eng.roi_align(feats, boxes, cnt, P).clone()
        eng.roi_order = False
        b = eng.roi_align(feats, boxes, cnt, P)
        eng.roi_order = True
        assert torch.equal(a, b)


@pytest.mark.parametrize("groups", [1, 2])
def test_roi_align_on_p32_pyramids_equals_the_f32_kernel_on_the_same_values(env, groups):
    """The P32 variant of the ROIAlign kernel (planes in, planes out, per-image scale groups) against the f32 kernel on the
    dequantised pyramids: boxes of every level, slivers, boxes hanging over the image edge, bins with 1 .. 12 samples per
    side (table path and per-sample fallback), rows beyond `count`; one scale per tensor and one scale group per image with
    very different magnitudes."""
    from deepemia_amd import p32

    eng, eng2, dev = env["f32"], env["f16x2"], env["dev"]
    g = torch.Generator().manual_seed(3)
    n = 2
    dims = {"p2": (64, 80), "p3": (32, 40), "p4": (16, 20), "p5": (8, 10)}
    f32, pl = {}, {}
    for k, (h, w) in dims.items():
        x = torch.randn(n, h, w, 256, generator=g) * 3.0
        x[1] *= 37.0                                           # image 1 lives on another scale
        x[..., :5] *= 1e-3
        q = p32.from_f32(x.to(dev), groups=groups)
        pl[k] = q
        f32[k] = p32.to_f32(q).contiguous()
    boxes = torch.zeros(n, 40, 4)
    for i in range(n):
        for r in range(40):
            size = [6, 20, 60, 140, 250, 320][r % 6] * float(torch.empty(1).uniform_(0.6, 1.4, generator=g))
            asp = float(torch.empty(1).uniform_(0.3, 3.0, generator=g))
            cx, cy = float(torch.empty(1).uniform_(-10, 330, generator=g)), float(torch.empty(1).uniform_(-10, 266, generator=g))
            bw, bh = size * asp ** 0.5, size / asp ** 0.5
            boxes[i, r] = torch.tensor([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2])
    boxes[0, 0] = torch.tensor([10.0, 10.0, 10.4, 200.0])        # a sliver
    cnt = torch.tensor([40, 33], dtype=torch.int32, device=dev)
    for P in (7, 14):
        want = eng.roi_align(f32, boxes.to(dev), cnt, P)
        got = eng2.roi_align(pl, boxes.to(dev), cnt, P)
        gv = p32.to_f32(got)
        assert got.groups == groups
        for i in range(n):                                       # per image: its own magnitude
            err = float((gv[i] - want[i]).abs().max() / want[i].abs().max())
            assert err < 2e-6, (P, i, err)
        assert not gv[1, 33:].any()
        # the output's scale covers its values: |x s| < 2^15 in both planes' sum
        assert float((gv.reshape(groups, -1).abs().amax(1) * got.meta[:, 1]).max()) < 32768.0


def test_box_detections_on_oracle_inputs(env):
    eng, d, ref = env["eng"], env["ref"]["dbg"], env["ref"]
    dev = env["dev"]
    r = d["prop_boxes"].shape[0]
    ld = 12
    logits = torch.zeros((1, r, ld))
    logits[0, :, :K + 1] = d["cls_logits"]
    logits[0, :, K + 1:K + 1 + 4 * K] = d["deltas"]
    newh, neww = d["resized"].shape[:2]
    db, ds, dc, dn = eng.detections(logits.to(dev), d["prop_boxes"][None].contiguous().to(dev),
                                    torch.tensor([r], dtype=torch.int32, device=dev), newh, neww)
    n = int(dn[0])
    assert n == d["det_boxes"].shape[0]
    np.testing.assert_array_equal(dc[0, :n].cpu().numpy(), d["det_classes"].numpy())
    assert float((ds[0, :n].cpu() - d["det_scores"]).abs().max()) < 1e-6
    assert float((db[0, :n].cpu() - d["det_boxes"]).abs().max()) < 2e-3


@pytest.mark.parametrize("branch", ["coordinate_trick", "vanilla"])
def test_box_detections_follow_torchvisions_batched_nms_branch_on_threshold_pairs(env, branch):
    """torchvision 0.11 ``batched_nms`` shifts the boxes by class * (boxes.max() + 1) when there are at most 4000 box
    coordinates and runs one NMS per class otherwise -- the same result unless a pair's IoU sits on the threshold, where
    the fp32 rounding of the shifted coordinates decides.  Hundreds of same-class pairs built to have IoU = 0.5 exactly
    (in exact arithmetic) at fractional positions, in both regimes: the kernel keeps exactly what the oracle's
    ``fast_rcnn_inference`` keeps, and the two regimes really differ on such pairs."""
    import torch.nn.functional as F2

    eng, R, dev = env["eng"], env["R"], env["dev"]
    k, newh, neww = 2, 800, 1100
    g = torch.Generator().manual_seed(5)
    pairs = 140 if branch == "coordinate_trick" else 520           # 280 / 2080 candidates: <= / > 1000
    x = torch.rand(pairs, generator=g) * 900 + 20
    y = torch.rand(pairs, generator=g) * 600 + 20
    w = (torch.randint(4, 40, (pairs,), generator=g) * 3).float()  # dx = w / 3 -> IoU = (w - dx) / (w + dx) = 0.5
    h = torch.rand(pairs, generator=g) * 60 + 10
    a = torch.stack([x, y, x + w, y + h], dim=1)
    b = a.clone()
    b[:, 0] += w / 3
    b[:, 2] += w / 3
    props = torch.stack([a, b], dim=1).reshape(-1, 4)
    r = props.shape[0]
    cls_logits = torch.zeros((r, k + 1))
    rank = torch.randperm(r, generator=g).float()                   # distinct scores, gaps far above an ulp of the softmax
    if branch == "coordinate_trick":
        cls_logits[:, 1] = 2.0 + rank * 2e-3                         # one candidate per proposal, all class 1: shifted boxes
    else:
        cls_logits[:, 0] = 0.2 + rank * 2e-5                         # both classes above 0.3: 2 r candidates -> vanilla
        cls_logits[:, 1] = 0.2 + rank * 2e-5 + 1e-5
    deltas = torch.zeros((r, 4 * k))
    probs = F2.softmax(cls_logits, dim=-1)
    pred = R.apply_deltas(deltas, props, (10.0, 10.0, 5.0, 5.0))
    rb, rs, rc, _ = R.fast_rcnn_inference(pred, probs, (newh, neww), 0.3, topk=100000)
    n_cand = int((probs[:, :k] > 0.3).sum())
    assert (4 * n_cand <= 4000) == (branch == "coordinate_trick")
    ld = 12
    logits = torch.zeros((1, r, ld))
    logits[0, :, :k + 1] = cls_logits
    logits[0, :, k + 1:k + 1 + 4 * k] = deltas
    db, ds, dc, dn = eng.detections(logits.to(dev), props[None].contiguous().to(dev), torch.tensor([r], dtype=torch.int32, device=dev),
                                    newh, neww)
    n = int(dn[0])
    m = min(n, 100)
    assert n == min(rb.shape[0], 100)
    np.testing.assert_array_equal(dc[0, :m].cpu().numpy(), rc[:m].numpy())
    assert float((ds[0, :m].cpu() - rs[:m]).abs().max()) < 1e-6 and float((db[0, :m].cpu() - rb[:m]).abs().max()) == 0.0
    # the other regime decides differently on some of these pairs (what the test is sensitive to)
    scores_f = probs[:, :k]
    mask = scores_f > 0.3
    bb = pred.view(-1, k, 4)[mask]
    idx = mask.nonzero()[:, 1]
    other = R.batched_nms(bb, scores_f[mask], idx, 0.5, size_rule=False) if branch == "coordinate_trick" else None
    if other is not None:
        mine = R.batched_nms(bb, scores_f[mask], idx, 0.5, size_rule=True)
        assert len(other) != len(mine) or not torch.equal(other, mine)


def test_box_detections_many_classes_and_a_non_finite_row(env):
    """demia_box_detections with K = 12 classes and 1000 proposals (12 000 (proposal, class) pairs; only those above
    the score threshold take a sort slot) and one proposal whose deltas overflow: same survivors, order and classes as
    the oracle's fast_rcnn_inference."""
    import torch.nn.functional as F2

    eng, R, dev = env["eng"], env["R"], env["dev"]
    k, r, newh, neww = 12, 1000, 800, 800
    g = torch.Generator().manual_seed(12)
    cls_logits = torch.randn((r, k + 1), generator=g) * 2.5
    deltas = torch.randn((r, 4 * k), generator=g) * 0.5
    deltas[17, 6] = 3.0e38                        # exp overflow -> inf coordinate: Detectron2 drops the whole row
    cx, cy = torch.rand(r, generator=g) * 700 + 50, torch.rand(r, generator=g) * 700 + 50
    bw, bh = torch.rand(r, generator=g) * 120 + 8, torch.rand(r, generator=g) * 120 + 8
    props = torch.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], dim=1)
    probs = F2.softmax(cls_logits, dim=-1)
    pred = R.apply_deltas(deltas, props, (10.0, 10.0, 5.0, 5.0))
    rb, rs, rc, _ = R.fast_rcnn_inference(pred, probs, (newh, neww), 0.3)
    ld = 5 * k + 1 + 3
    logits = torch.zeros((1, r, ld))
    logits[0, :, :k + 1] = cls_logits
    logits[0, :, k + 1:k + 1 + 4 * k] = deltas
    saved_k, eng.K = eng.K, k
    try:
        db, ds, dc, dn = eng.detections(logits.to(dev), props[None].contiguous().to(dev), torch.tensor([r], dtype=torch.int32, device=dev),
                                        newh, neww)
    finally:
        eng.K = saved_k
    n = int(dn[0])
    assert n == rb.shape[0] == 100
    np.testing.assert_array_equal(dc[0, :n].cpu().numpy(), rc.numpy())
    assert float((ds[0, :n].cpu() - rs).abs().max()) < 1e-6
    assert float((db[0, :n].cpu() - rb).abs().max()) < 2e-3


def test_paste_on_oracle_inputs(env):
    eng, d, ref = env["eng"], env["ref"]["dbg"], env["ref"]
    dev = env["dev"]
    n = d["det_boxes"].shape[0]
    D = 100
    mp = d["mask_probs"][:, 0]  # [n, 28, 28]
    blocked = torch.zeros((D, 196, 4, 4))
    for dy in range(2):
        for dx in range(2):
            sub = mp[:, dy::2, dx::2].reshape(n, 196)
            for c in range(K):
                blocked[:n, :, dy * 2 + dx, c] = torch.where(d["det_classes"][:, None] == c, sub, torch.full_like(sub, -1.0))
    det_boxes = torch.zeros((1, D, 4))
    det_boxes[0, :n] = d["det_boxes"]
    det_classes = torch.zeros((1, D), dtype=torch.int32)
    det_classes[0, :n] = d["det_classes"].int()
    newh, neww = d["resized"].shape[:2]
    h, w = env["img"].shape[:2]
    ob, valid, packed, hint = eng.paste(blocked.view(D * 196 * 4, 1, 1, 4).to(dev), det_boxes.to(dev), det_classes.to(dev),
                                        torch.tensor([n], dtype=torch.int32, device=dev), newh, neww, h, w)
    masks = eng.unpack(packed[0, :n].contiguous(), h, w).cpu()
    # the paste box handed to the packed-mask kernels as bbox hint contains every set pixel of its mask
    area_h, bbox_h = eng.area_bbox(packed[0, :n].contiguous(), h, w, hint[0, :n].contiguous())
    area_f, bbox_f = eng.area_bbox(packed[0, :n].contiguous(), h, w)
    assert torch.equal(area_h, area_f) and torch.equal(bbox_h, bbox_f)
    assert bool((hint[0, n:] == -1).all())
    assert bool(valid[0, :n].all()) and not bool(valid[0, n:].any())
    assert float((ob[0, :n].cpu() - ref["pred_boxes"]).abs().max()) < 1e-3
    r = ref["pred_masks"]
    diff = int((masks != r).sum())
    inter = (masks & r).sum((1, 2)).float()
    union = (masks | r).sum((1, 2)).float().clamp(min=1)
    assert float((inter / union).min()) >= 0.999
    assert diff <= n  # at most a threshold-tie pixel per instance
    assert not packed[0, n:].any()


def test_paste_of_boxes_clipped_to_a_sliver(env):
    """Boxes that the clip to the image leaves a few 1e-4 px wide (valid: width > 0): the sampling coordinate of their
    columns is tens of thousands of mask cells away, beyond what a 16-bit tap index holds -- widths chosen so that a
    WRAPPED index would land on a valid tap.  Every pixel must equal the CPU path's paste (nothing sampled there)."""
    from oracle import maskrcnn_ref as R

    eng, dev = env["eng"], env["dev"]
    size, D = 128, 100
    ulp = 2.0 ** -17                                       # f32 spacing just below 128
    widths = []
    for k in range(1, 400):                                # width = k ulps at the right image edge
        x0 = np.float32(size) - np.float32(k * ulp)
        wd = np.float32(size) - x0
        for X in (126, 127):
            ix = (((np.float32(X) + np.float32(0.5) - x0) / wd * np.float32(2) - np.float32(1)) + np.float32(1)) * np.float32(28)
            ix = (ix - np.float32(1)) / np.float32(2)
            if abs(float(ix)) < 2 ** 31 and abs(float(ix)) > 40000:
                wrapped = (int(np.floor(ix)) + 32768) % 65536 - 32768
                if -1 <= wrapped <= 27 and k not in widths:
                    widths.append(k)
    assert len(widths) >= 3, widths
    widths = widths[:12]
    n = len(widths)
    boxes = torch.zeros((1, D, 4))
    for i, k in enumerate(widths):
        boxes[0, i] = torch.tensor([float(np.float32(size) - np.float32(k * ulp)), 10.0 + i, 300.0, 60.0 + i])
    classes = torch.zeros((1, D), dtype=torch.int32)
    prob = torch.ones((D * 196 * 4, 1, 1, 4))
    ob, valid, packed, hint = eng.paste(prob.to(dev), boxes.to(dev), classes.to(dev), torch.tensor([n], dtype=torch.int32, device=dev),
                                        size, size, size, size)
    assert bool(valid[0, :n].all())
    got = eng.unpack(packed[0, :n].contiguous(), size, size).cpu()
    want = R.paste_masks(torch.ones((n, 28, 28)), ob[0, :n].cpu(), size, size)
    assert torch.equal(got, want), int((got != want).sum())


def test_incremental_paste_into_reused_planes_equals_the_whole_plane_paste(env):
    """A captured forward pastes into its own planes on every replay and writes only the union of each instance's previous
    and new box (``demia_paste_desc.prev_bbox``).  Three pastes in a row into the same planes -- boxes that move, shrink,
    grow, vanish (fewer detections) and reappear, a non-multiple-of-32 width -- must each leave exactly what the whole-plane
    paste of the same inputs writes, zeros everywhere else."""
    eng, dev = env["eng"], env["dev"]
    g = torch.Generator().manual_seed(9)
    b, D, H, W, size = 2, 100, 300, 333, 320
    planes = torch.zeros((b, D, H, (W + 31) // 32), dtype=torch.int32, device=dev)
    prev = torch.full((b, D, 4), -1, dtype=torch.int32, device=dev)
    for rnd, counts in enumerate(([60, 100], [100, 7], [0, 55])):
        boxes = torch.zeros((b, D, 4))
        cxy = torch.rand((b, D, 2), generator=g) * size
        wh = torch.rand((b, D, 2), generator=g) * (120 if rnd != 1 else 30) + 2
        boxes[..., 0:2] = cxy - wh / 2
        boxes[..., 2:4] = cxy + wh / 2
        prob = torch.rand((b * D * 196 * 4, 1, 1, 4), generator=g)
        classes = torch.randint(0, 2, (b, D), generator=g, dtype=torch.int32)
        cnt = torch.tensor(counts, dtype=torch.int32, device=dev)
        args = (prob.to(dev), boxes.to(dev), classes.to(dev), cnt, size, size, H, W)
        ob0, v0, want, bb0 = eng.paste(*args)
        eng._paste_static = (planes, prev)
        try:
            ob1, v1, got, bb1 = eng.paste(*args)
        finally:
            eng._paste_static = None
        assert got.data_ptr() == planes.data_ptr()
        assert torch.equal(got, want) and torch.equal(bb1, bb0) and torch.equal(v1, v0) and torch.equal(ob1, ob0), rnd
        assert torch.equal(prev, bb1)
        assert int(want.ne(0).sum()) > 0 or rnd == 2


def test_per_shape_caches_are_bounded_over_many_image_sizes(env):
    """Arenas, meta pools and captured graphs are per input shape; a folder of differently sized micrographs must not keep
    one of each per size.  Seven distinct sizes (graphed from their second occurrence, as the pipeline does) through ONE
    engine with two cached shapes: device memory stays bounded by the two largest arenas, shapes are evicted least recently
    used first, and an evicted shape that comes back gives the results a fresh engine gives."""
    from deepemia_amd.engine import MaskRCNNEngine

    sd, synth, dev = env["sd"], env["synth"], env["dev"]
    eng = MaskRCNNEngine(sd, 50, K, THR, dev, "f16x2")
    eng.max_cached_shapes = 2
    sizes = [(256, 256), (320, 256), (384, 320), (448, 256), (512, 384), (256, 384), (300, 500)]
    imgs = {hw: torch.from_numpy(np.ascontiguousarray(synth.em_tile(60 + i, 512)[: hw[0], : hw[1]])).to(dev)[None]
            for i, hw in enumerate(sizes)}

    def run(e, hw, graphed):
        r = (e.forward_graphed if graphed else e.forward)(imgs[hw])
        out = (r.scores.clone(), r.classes.clone(), r.count.clone(), r.packed.clone())
        del r
        return out

    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    arena_bytes, after, first = [], [], {}
    for hw in sizes:
        first[hw] = run(eng, hw, False)
        run(eng, hw, True)                              # second occurrence: captured + replayed
        torch.cuda.synchronize()
        key = (1,) + hw
        assert key in eng._arena and key in eng._graphs and len(eng._arena) <= 2 and len(eng._graphs) <= 2 and len(eng._meta_pools) <= 2
        arena_bytes.append(sum(t.numel() * t.element_size() for t in eng._arena[key]))
        after.append(torch.cuda.memory_allocated() - base)
    assert eng.evictions == len(sizes) - 2
    results = sum(sum(t.numel() * t.element_size() for t in v) for v in first.values())
    big2 = sum(sorted(arena_bytes)[-2:])
    # two arenas + two graphs' private output pools + the kept results; never one arena per size (their sum is ~3.5x big2)
    assert max(after) <= 1.6 * big2 + results + (256 << 20), (after, arena_bytes)
    assert sum(arena_bytes) > 2.5 * big2 * 0.9
    # an evicted shape comes back: same bits as the first time and as a fresh engine
    fresh = MaskRCNNEngine(sd, 50, K, THR, dev, "f16x2")
    for hw in (sizes[0], sizes[3]):
        again, ref = run(eng, hw, False), run(fresh, hw, False)
        for a, b, c in zip(again, first[hw], ref):
            assert torch.equal(a, b) and torch.equal(a, c)
    eng.release_cached_shapes()
    assert not eng._arena and not eng._graphs and not eng._meta_pools


def test_unpack_and_area_bbox_bit_exact(env):
    eng, dev = env["eng"], env["dev"]
    rng = np.random.default_rng(5)
    m, h, w = 7, 96, 128
    masks = np.zeros((m, h, w), dtype=bool)
    for i in range(m - 1):
        y0, x0 = rng.integers(0, h - 10), rng.integers(0, w - 10)
        y1, x1 = rng.integers(y0 + 1, h + 1), rng.integers(x0 + 1, w + 1)
        masks[i, y0:y1, x0:x1] = rng.random((y1 - y0, x1 - x0)) > 0.3
    packed = np.packbits(masks.reshape(m, h, w // 32, 32), axis=-1, bitorder="little").view(np.uint32).reshape(m, h, w // 32)
    p = torch.from_numpy(packed.view(np.int32)).to(dev)
    got = eng.unpack(p, h, w).cpu().numpy()
    np.testing.assert_array_equal(got, masks)
    area, bbox = eng.area_bbox(p, h, w)
    np.testing.assert_array_equal(area.cpu().numpy(), masks.sum((1, 2)))
    for i in range(m):
        ys, xs = np.nonzero(masks[i])
        exp = [-1, -1, -1, -1] if len(ys) == 0 else [ys.min(), xs.min(), ys.max(), xs.max()]
        assert bbox[i].cpu().tolist() == exp


@pytest.mark.parametrize("prec", ["f32", "f32x3", "f16x2", "f16x2r"])
def test_end_to_end_f32_matches_oracle(env, prec):
    from conftest import needs_dev_build
    needs_dev_build(prec)
    from deepemia_amd.predictor import Predictor

    ref = env["ref"]
    inst = Predictor(env[prec])(env["img"])["instances"]
    n = len(inst)
    assert n == ref["scores"].shape[0] == 100
    got = inst.to("cpu")
    np.testing.assert_array_equal(got._fields["pred_classes"].numpy(), ref["pred_classes"].numpy())
    assert got._fields["pred_classes"].dtype == torch.int64
    assert float((got.scores - ref["scores"]).abs().max()) < 1e-4
    assert bool((got.scores[:-1] >= got.scores[1:]).all())
    assert float((got.pred_boxes - ref["pred_boxes"]).abs().max()) < 5e-2
    m, r = got.pred_masks, ref["pred_masks"]
    assert m.dtype == torch.bool and tuple(m.shape) == (n, 1024, 1024)
    iou = (m & r).sum((1, 2)).float() / (m | r).sum((1, 2)).float().clamp(min=1)
    assert float(iou.min()) >= 0.999


def test_full_size_batch_properties(env):
    """BASELINE configs[1] size (2048^2, R101 weights are only needed for speed, not here):
    determinism and batch invariance of the whole path at the full tile size."""
    eng, synth, dev = env["eng"], env["synth"], env["dev"]
    tiles = np.stack([synth.em_tile(i, 2048) for i in range(3)])
    x = torch.from_numpy(tiles).to(dev)
    a = eng.forward(x)
    b = eng.forward(x)
    torch.cuda.synchronize()
    assert torch.equal(a.packed, b.packed) and torch.equal(a.scores, b.scores) and torch.equal(a.boxes, b.boxes)
    single = eng.forward(x[1:2].contiguous())
    assert torch.equal(single.count[0], a.count[1])
    n = int(single.count[0])
    assert torch.equal(single.classes[0, :n], a.classes[1, :n])
    assert torch.equal(single.packed[0, :n], a.packed[1, :n])
    # mask area from the packed popcount equals the unpacked bool sum; masks lie inside their boxes' hull
    area, bbox = eng.area_bbox(a.packed[1, :n].contiguous(), 2048, 2048)
    um = eng.unpack(a.packed[1, :n].contiguous(), 2048, 2048)
    assert torch.equal(area.long(), um.sum((1, 2)))
    bx = a.boxes[1, :n]
    ok = (bbox[:, 1] >= torch.floor(bx[:, 0]).int() - 1) & (bbox[:, 3] <= torch.ceil(bx[:, 2]).int() + 1)
    assert bool(ok[area > 0].all())


def test_bf16_features_within_tolerance(env):
    from conftest import needs_dev_build
    needs_dev_build("bf16")
    from deepemia_amd.engine import MaskRCNNEngine

    eng = MaskRCNNEngine(env["sd"], 50, K, THR, env["dev"], "bf16")
    d = env["ref"]["dbg"]
    x = torch.from_numpy(env["img"])[None].to(env["dev"])
    xin, newh, neww, ph, pw = eng.preprocess(x)
    feats = eng.backbone(xin, ph, pw)
    for k in ("res5", "p2", "p5"):
        a = feats[k][0].permute(2, 0, 1).float().cpu()
        b = d["feats"][k][0]
        assert float((a - b).abs().max() / b.abs().max()) < 5e-2, k
    out = eng.forward(x)
    assert int(out.count[0]) == 100


def test_non_square_image_width_not_multiple_of_32(env):
    """600 x 700 input: resize to 800 x 933 (pad 800 x 960), masks pasted at 600 x 700 (22 words per row)."""
    from deepemia_amd.predictor import Predictor

    R, synth = env["R"], env["synth"]
    img = synth.em_tile(5, 700)[:600]
    ref = R.predict(img, env["sd"], 50, THR)
    inst = Predictor(env["eng"])(img)["instances"].to("cpu")
    n = ref["scores"].shape[0]
    assert len(inst) == n and n > 10
    np.testing.assert_array_equal(inst.pred_classes.numpy(), ref["pred_classes"].numpy())
    m, r = inst.pred_masks, ref["pred_masks"]
    assert tuple(m.shape) == (n, 600, 700)
    iou = (m & r).sum((1, 2)).float() / (m | r).sum((1, 2)).float().clamp(min=1)
    assert float(iou.min()) >= 0.999


def test_native_resolution_mode_matches_oracle_at_the_same_sizes(env):
    """SURVEY 8(f)4, flagged non-parity mode: INPUT.MIN_SIZE_TEST / MAX_SIZE_TEST other than the reference's 800 / 1333
    (here 1024 / 1024 on a 1024 x 1024 tile: the net sees the tile unscaled).  Not the reference's result -- but the HIP
    path and the oracle must still agree with each other when both are given the same sizes."""
    from deepemia_amd.engine import MaskRCNNEngine
    from deepemia_amd.predictor import Predictor

    R, synth = env["R"], env["synth"]
    img = synth.em_tile(3, 1024)
    ref = R.predict(img, env["sd"], 50, THR, min_size_test=1024, max_size_test=1024)
    eng = MaskRCNNEngine(env["sd"], 50, K, THR, env["dev"], "f16x2", min_size_test=1024, max_size_test=1024)
    assert eng._resize_tables(1024, 1024)["newh"] == 1024
    inst = Predictor(eng)(img)["instances"].to("cpu")
    n = ref["scores"].shape[0]
    assert len(inst) == n and n > 10
    np.testing.assert_array_equal(inst.pred_classes.numpy(), ref["pred_classes"].numpy())
    np.testing.assert_allclose(inst.scores.numpy(), ref["scores"].numpy(), atol=2e-5)
    m, r = inst.pred_masks, ref["pred_masks"]
    iou = (m & r).sum((1, 2)).float() / (m | r).sum((1, 2)).float().clamp(min=1)
    assert float(iou.min()) >= 0.999
    # and it is a different answer from the 800-pixel parity mode
    base = R.predict(img, env["sd"], 50, THR)
    assert base["scores"].shape[0] != n or not np.allclose(base["scores"].numpy(), ref["scores"].numpy(), atol=1e-3)


def test_batch_whose_activations_exceed_2gib_is_consistent(env):
    """64 tiles per forward: the p2-level activation tensors are 2.6 GB, beyond a 32-bit byte offset from the tensor
    base -- the split conv kernel's buffer descriptors start at each tile's first image instead.  Every repeated tile
    must come out exactly like its first copy."""
    synth = env["synth"]
    base = np.stack([synth.em_tile(i, 2048) for i in range(16)])
    x = torch.from_numpy(np.concatenate([base] * 4)).to(env["dev"])
    out = env["f16x2"].forward(x)
    cnt = out.count.cpu().numpy()
    assert (cnt > 10).all()
    for g in range(1, 4):
        assert (cnt[:16] == cnt[16 * g:16 * (g + 1)]).all()
        for name in ("boxes", "scores", "classes"):
            t = getattr(out, name)
            assert torch.equal(t[:16], t[16 * g:16 * (g + 1)]), (name, g)
    assert torch.equal(out.packed[:16], out.packed[48:])


def test_rpn_two_pass_selection_equals_the_radix_passes_incl_ties(env, monkeypatch):
    """The RPN's top-k per level: the two-pass selection (12-bit histogram, candidate list, sort) against the five radix /
    compaction passes it replaced (``DEMIA_RPN_SELECT=radix``), bit for bit, on heads that make every branch run: smooth random
    logits, logits quantised to a few dozen values (thousands of ties at the threshold: lowest index first), a level whose
    threshold bin overflows the candidate list (falls back to the radix passes) and an all-equal level."""
    from deepemia_amd import _lib, engine as E

    eng, dev = env["eng"], env["dev"]
    g = torch.Generator().manual_seed(5)
    shapes = [(200, 200), (100, 100), (50, 50), (25, 25), (13, 13)]
    cell = E.cell_anchor_table()

    def run(heads, select):
        if select:
            monkeypatch.setenv("DEMIA_RPN_SELECT", select)
        else:
            monkeypatch.delenv("DEMIA_RPN_SELECT", raising=False)
        n = heads[0].shape[0]
        boxes = torch.zeros((n, 1000, 4), device=dev)
        scores = torch.zeros((n, 1000), device=dev)
        count = torch.zeros((n,), dtype=torch.int32, device=dev)
        ws = torch.empty((int(eng.lib.demia_rpn_workspace_bytes(n)),), dtype=torch.uint8, device=dev)
        desc = _lib.RpnDesc()
        for i, hd in enumerate(heads):
            desc.head[i] = _lib.ptr(hd)
            desc.H[i], desc.W[i], desc.stride[i] = hd.shape[1], hd.shape[2], E.STRIDES[i]
        desc.cell_anchors = cell.ctypes.data
        desc.head_ld, desc.N, desc.img_h, desc.img_w = 16, n, 800, 800
        desc.pre_topk, desc.post_topk, desc.nms_thresh = 1000, 1000, 0.7
        desc.out_boxes, desc.out_scores, desc.out_count, desc.workspace = map(_lib.ptr, (boxes, scores, count, ws))
        _lib.check(eng.lib.demia_rpn_proposals(C.byref(desc), eng._stream()), "rpn")
        torch.cuda.synchronize()
        return boxes.cpu(), scores.cpu(), count.cpu()

    for kind in ("smooth", "quantised", "narrow", "equal"):
        heads = []
        for (h, w) in shapes:
            hd = torch.randn(3, h, w, 16, generator=g)
            hd[..., 3:15] *= 0.3
            if kind == "quantised":
                hd[..., :3] = torch.round(hd[..., :3] * 8) / 8                 # ~50 distinct logits: thousands of ties per value
            elif kind == "narrow":
                hd[..., :3] = 1.0 + hd[..., :3] * 1e-4                         # all logits in one 12-bit bin: list overflow on p2 / p3
            elif kind == "equal":
                hd[..., :3] = -2.5
            heads.append(hd.to(dev).contiguous())
        a = run(heads, None)
        b = run(heads, "radix")
        assert torch.equal(a[2], b[2]) and int(a[2].min()) > 0, kind
        assert torch.equal(a[1], b[1]), kind
        assert torch.equal(a[0], b[0]), kind


@pytest.mark.parametrize("case", [(256, 256, 3, 1, 1, 14, 14, 9, True), (256, 1024, 1, 1, 0, 1, 1, 700, True), (128, 64, 3, 2, 1, 33, 31, 2, False),
                                  (64, 256, 1, 1, 0, 40, 40, 2, True), (96, 64, 3, 1, 1, 20, 20, 1, True)])
def test_single_plane_conv_is_the_high_plane_product(env, case, monkeypatch):
    """``demia_conv_p32_desc.single = 1`` (the mask head's default, engine.DEFAULT_SINGLE_STAGES): ONE fp16 MFMA per product on
    the HIGH planes of both operands, f32 accumulation, output still split into both planes.  Checked against an f64
    convolution of exactly those operands (half(x s) / s and half(w 2^e) / 2^e) -- the only error left is the f32 accumulation
    order (<= 3e-6 of max |out|) -- for the h-only K-step of 64 (Cin % 64 == 0: half the operand stream) and for the plain
    K-step of 32 (Cin = 96, or DEMIA_P32_NO_HK=1 in a subprocess-free way: a channel count that is no multiple of 64)."""
    import torch.nn.functional as F
    from deepemia_amd import engine as E, p32
    from deepemia_amd._lib import ACT_RELU

    cin, cout, k, stride, pad, h, w, n, relu = case
    dev = env["dev"]
    g = torch.Generator().manual_seed(7 * cin + cout + k)
    x = (torch.randn((n, h, w, cin), generator=g) * 2.0).to(dev)
    x[..., :5] *= 1e-3
    wt = torch.randn((cout, cin, k, k), generator=g) / (cin * k * k) ** 0.5
    scale = torch.rand((cout,), generator=g) + 0.5
    bias = torch.randn((cout,), generator=g) * 0.1
    eng = env["f16x2"]
    cout_pad = (cout + 63) // 64 * 64
    wp = torch.zeros((cout_pad, k, k, cin))
    wp[:cout] = wt.permute(0, 2, 3, 1)
    planes, sw = E.split2_f16_scaled(wp.to(dev))
    L = E.ConvLayer(None, scale.to(dev), bias.to(dev), cin, cout, cout_pad, k, k, stride, pad, E.tile_weight_planes_p32(planes),
                    (scale.to(dev) / sw[:cout]).contiguous(), float((scale.abs() * wt.abs().flatten(1).sum(1)).max()), float(bias.abs().max()), single=1)
    xp = p32.from_f32(x)
    out = eng.conv_p32(xp, L, act=ACT_RELU if relu else 0)
    torch.cuda.synchronize()
    got = p32.to_f32(out).double()
    # the operands the kernel multiplies: high planes only
    xh = xp.buf[p32.HEADER_HALFS:].view(-1, cin // 32, 2, 32)[:, :, 0, :].reshape(n, h, w, cin).double() / float(xp.meta[0, 1])
    wh = (planes[0].double() / sw.double().view(-1, 1, 1, 1))[:cout].permute(0, 3, 1, 2)
    ref = F.conv2d(xh.permute(0, 3, 1, 2), wh, None, stride, pad).permute(0, 2, 3, 1) * scale.double().to(dev) + bias.double().to(dev)
    if relu:
        ref = ref.clamp(min=0)
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 3e-6, err
    # and the single-plane product really is an 11-bit one: it differs from the two-plane result at the 1e-4 .. 1e-3 level
    L3 = E.ConvLayer(**{**L.__dict__, "single": 0})
    full = p32.to_f32(eng.conv_p32(xp, L3, act=ACT_RELU if relu else 0)).double()
    d = float((got - full).abs().max() / full.abs().max())
    assert 1e-5 < d < 5e-3, d
    # the measured |out| bound the next layer scales with is exact
    assert abs(float(out.meta[0, 0]) - float(got.abs().max())) <= 1e-5 * float(got.abs().max())


@pytest.mark.parametrize("case", [(256, 1024, 25, 25, 2, True, 1), (64, 256, 31, 17, 3, True, 1), (1024, 256, 40, 40, 1, False, 1),
                                  (32, 128, 20, 20, 2, False, 1), (256, 1024, 50, 50, 4, True, 4), (96, 256, 9, 9, 1, True, 1)])
def test_pingpong_pointwise_kernel_is_bit_identical_to_the_plain_kernel(env, case):
    """Dev build only (tile hint 60): 1x1 layers as a ping-pong of two four-wave groups inside one persistent workgroup -- one group in
    its K loop while the other runs the epilogue of the tile it finished a phase earlier, both executing the same barriers.  Same
    MFMAs per accumulator in the same order, same epilogue code: planes and metas equal the plain kernel's bit for bit, on ragged M,
    one to 32 K-steps, with and without residual, one and several scale groups.  (It is 0.6-0.95 x the plain kernel's speed and
    therefore not used by the product: DESIGN.md section 4.)"""
    from conftest import needs_dev_build
    needs_dev_build("")
    from deepemia_amd import engine as E, p32
    from deepemia_amd._lib import ACT_RELU, RES_NONE, RES_SAME

    cin, cout, h, w, n, res, groups = case
    dev = env["dev"]
    g = torch.Generator().manual_seed(cin + cout + h)
    x = (torch.randn((n, h, w, cin), generator=g) * 2.0).to(dev)
    wt = torch.randn((cout, cin, 1, 1), generator=g) / cin ** 0.5
    scale = torch.rand((cout,), generator=g) + 0.5
    bias = torch.randn((cout,), generator=g) * 0.1
    wp = torch.zeros((cout, 1, 1, cin))
    wp[:] = wt.permute(0, 2, 3, 1)
    planes, sw = E.split2_f16_scaled(wp.to(dev))
    L = E.ConvLayer(None, scale.to(dev), bias.to(dev), cin, cout, cout, 1, 1, 1, 0, E.tile_weight_planes_p32(planes),
                    (scale.to(dev) / sw[:cout]).contiguous(), float((scale.abs() * wt.abs().flatten(1).sum(1)).max()), float(bias.abs().max()))
    eng = env["f16x2"]
    xp = p32.from_f32(x, groups=groups)
    rp = p32.from_f32((torch.randn((n, h, w, cout), generator=g) * 1.5).to(dev), groups=groups) if res else None
    outs = [eng.conv_p32(xp, L, act=ACT_RELU, residual=rp, res_mode=RES_SAME if res else RES_NONE, tile_hint=hint) for hint in (0, 60)]
    torch.cuda.synchronize()
    assert torch.equal(outs[0].buf, outs[1].buf) and torch.equal(outs[0].meta, outs[1].meta)
    assert float(outs[0].meta[:, 0].min()) > 0.0
