"""Dev script (GPU): demia_conv2d_p32 against an f64 convolution on a set of shapes, then its rate on the R101 layers
beside the register-staged f16x2 kernel (demia_conv2d_nhwc) on the same shapes.  usage: gpu_conv_p32_check.py [check|time|all]"""
import ctypes as C
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, '.')
from deepemia_amd import _lib, p32, engine as E        # noqa: E402
from deepemia_amd._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, RES_NONE, RES_SAME, RES_UP2, F16X2, F32   # noqa: E402

import os, pathlib   # noqa: E402
if os.environ.get('AB_LIB'):
    _lib.LIB_PATH = pathlib.Path(os.environ['AB_LIB']).resolve()
dev = torch.device('cuda:0')
lib = _lib.load()
st = lambda: int(torch.cuda.current_stream(dev).cuda_stream)


class Layer:
    def __init__(self, cout, cin, kh, kw, seed=0, bn=True):
        g = torch.Generator().manual_seed(seed)
        w = torch.randn(cout, cin, kh, kw, generator=g) * (2.0 / (cin * kh * kw)) ** 0.5
        self.w = w
        self.cout, self.cin, self.kh, self.kw = cout, cin, kh, kw
        self.cout_pad = (cout + 63) // 64 * 64
        wp = torch.zeros(self.cout_pad, kh, kw, cin)
        wp[:cout] = w.permute(0, 2, 3, 1)
        self.scale = (0.5 + torch.rand(cout, generator=g)) if bn else torch.ones(cout)
        self.bias = torch.randn(cout, generator=g) * 0.1
        planes, sw = E.split2_f16_scaled(wp.to(dev))
        self.planes, self.sw = planes, sw
        self.w_p32 = E.tile_weight_planes_p32(planes)
        self.w_old = E.tile_weight_planes(planes, int(lib.demia_conv_f16x2_kstep()))
        self.scale3 = (self.scale.to(dev) / sw[:cout]).contiguous()
        self.bias_d = self.bias.to(dev).contiguous()
        self.wq = ((planes[0].double() + planes[1].double()) / sw.double().view(-1, 1, 1, 1))[:cout].permute(0, 3, 1, 2).contiguous()
        self.wbound = float((self.scale.abs() * w.abs().flatten(1).sum(1)).max())
        self.bbound = float(self.bias.abs().max())


def conv_p32(x: p32.P32, L: Layer, stride=1, pad=0, act=ACT_NONE, res=None, res_mode=RES_NONE, out_f32=False, out_ld=0, hint=0, single=0):
    n, h, w, cin = x.shape
    ho = (h + 2 * pad - L.kh) // stride + 1
    wo = (w + 2 * pad - L.kw) // stride + 1
    if out_f32:
        ld = out_ld or L.cout
        out = torch.zeros((n, ho, wo, ld), dtype=torch.float32, device=dev)
        optr, ometa = _lib.ptr(out), 0
    else:
        out = p32.alloc((n, ho, wo, L.cout), dev)
        optr, ometa = _lib.ptr(out.buf), _lib.ptr(out.meta)
    d = _lib.ConvP32Desc(_lib.ptr(x.buf), _lib.ptr(x.meta), _lib.ptr(L.w_p32), _lib.ptr(L.scale3), _lib.ptr(L.bias_d),
                         _lib.ptr(res.buf) if res is not None else 0, _lib.ptr(res.meta) if res is not None else 0, optr, ometa,
                         L.wbound, L.bbound, n, h, w, cin, ho, wo, L.cout, L.cout_pad, L.kh, L.kw, stride, pad, act, res_mode,
                         1 if out_f32 else 0, out_ld, hint)
    d.single = single
    _lib.check(lib.demia_conv2d_p32(C.byref(d), st()), 'demia_conv2d_p32')
    return out


def conv_old(x: torch.Tensor, amax: torch.Tensor, L: Layer, stride=1, pad=0, act=ACT_NONE):
    n, h, w, cin = x.shape
    ho = (h + 2 * pad - L.kh) // stride + 1
    wo = (w + 2 * pad - L.kw) // stride + 1
    out = torch.empty((n, ho, wo, L.cout), dtype=torch.float32, device=dev)
    am = torch.zeros(1, dtype=torch.float32, device=dev)
    d = _lib.ConvDesc(_lib.ptr(x), _lib.ptr(L.w_old), _lib.ptr(L.scale3), _lib.ptr(L.bias_d), 0, _lib.ptr(out), n, h, w, cin, ho, wo,
                      L.cout, L.cout_pad, L.kh, L.kw, stride, pad, F16X2, F32, act, RES_NONE, L.cout, 0, _lib.ptr(amax), _lib.ptr(am))
    _lib.check(lib.demia_conv2d_nhwc(C.byref(d), st()), 'demia_conv2d_nhwc')
    return out


def reference(xq, L, stride, pad, act, res=None, res_mode=RES_NONE):
    y = F.conv2d(xq.double().permute(0, 3, 1, 2), L.wq, None, stride, pad).permute(0, 2, 3, 1)
    y = y * L.scale.double().to(dev) + L.bias.double().to(dev)
    if res is not None:
        r = res.double()
        if res_mode == RES_UP2:
            r = r.repeat_interleave(2, 1).repeat_interleave(2, 2)[:, :y.shape[1], :y.shape[2]]
        y = y + r
    if act == ACT_RELU:
        y = y.clamp(min=0)
    if act == ACT_SIGMOID:
        y = torch.sigmoid(y)
    return y


def check():
    cases = [
        # n, h, w, cin, cout, k, stride, pad, act, res_mode, out_f32, hint
        (2, 50, 50, 256, 256, 3, 1, 1, ACT_RELU, RES_NONE, False, 0),
        (1, 37, 29, 64, 64, 3, 1, 1, ACT_RELU, RES_NONE, False, 0),
        (3, 40, 40, 256, 128, 1, 2, 0, ACT_RELU, RES_NONE, False, 0),
        (2, 25, 25, 256, 1024, 1, 1, 0, ACT_RELU, RES_SAME, False, 0),
        (2, 26, 25, 512, 256, 1, 1, 0, ACT_NONE, RES_UP2, False, 0),
        (2, 50, 50, 256, 15, 1, 1, 0, ACT_NONE, RES_NONE, True, 0),
        (1, 1, 1000, 12544, 1024, 1, 1, 0, ACT_RELU, RES_NONE, False, 0),
        (1, 1, 300, 1024, 11, 1, 1, 0, ACT_NONE, RES_NONE, True, 0),
        (4, 14, 14, 256, 256, 3, 1, 1, ACT_RELU, RES_NONE, False, 0),
        (1, 20, 784, 256, 2, 1, 1, 0, ACT_SIGMOID, RES_NONE, True, 0),
        (1, 200, 200, 64, 256, 1, 1, 0, ACT_NONE, RES_NONE, False, 0),
    ] + [(2, 41, 37, 128, 256, 3, 1, 1, ACT_RELU, RES_SAME, False, hh) for hh in list(range(1, 14)) + [21, 22, 26, 31, 34]] + \
        [(3, 50, 50, 256, 256, 3, 1, 1, ACT_RELU, RES_NONE, False, 21), (2, 25, 25, 64, 512, 1, 1, 0, ACT_NONE, RES_SAME, False, 21),
         (1, 1, 1000, 12544, 1024, 1, 1, 0, ACT_RELU, RES_NONE, False, 21), (3, 50, 50, 256, 256, 3, 1, 1, ACT_RELU, RES_NONE, False, 31),
         (2, 25, 25, 64, 512, 1, 1, 0, ACT_NONE, RES_UP2, False, 34), (1, 1, 700, 1024, 1024, 1, 1, 0, ACT_RELU, RES_SAME, False, 31), (5, 40, 40, 256, 128, 1, 2, 0, ACT_RELU, RES_NONE, False, 26)] + \
        [(2, 41, 37, 128, 256, 3, 1, 1, ACT_RELU, RES_SAME, False, hh) for hh in (14, 42, 52, 49)] + \
        [(2, 33, 31, cin, 256, 1, 1, 0, ACT_RELU, RES_NONE, False, hh) for cin in (32, 64, 96, 1024) for hh in (42, 52, 49)]   # 1, 2, 3, 32 K-steps through three stages
    worst = 0.0
    for ci, (n, h, w, cin, cout, k, s, pd, act, rm, of32, hint) in enumerate(cases):
        g = torch.Generator().manual_seed(100 + ci)
        x = (torch.randn(n, h, w, cin, generator=g) * 3.0).to(dev)
        x[..., :7] *= 1e-3                                     # small channels beside large ones
        L = Layer(cout, cin, k, k, seed=ci)
        xp = p32.from_f32(x)
        xq = p32.to_f32(xp)
        ho = (h + 2 * pd - k) // s + 1
        wo = (w + 2 * pd - k) // s + 1
        res = resq = None
        if rm == RES_SAME:
            res = p32.from_f32((torch.randn(n, ho, wo, cout, generator=g) * 2.0).to(dev))
        elif rm == RES_UP2:
            res = p32.from_f32((torch.randn(n, (ho + 1) // 2, (wo + 1) // 2, cout, generator=g) * 2.0).to(dev))
        if res is not None:
            resq = p32.to_f32(res)
        ld = (cout + 3) // 4 * 4 if of32 else 0
        out = conv_p32(xp, L, s, pd, act, res, rm, of32, ld, hint)
        torch.cuda.synchronize()
        ref = reference(xq, L, s, pd, act, resq, rm)
        if of32:
            got = out[..., :cout].double()
        else:
            got = p32.to_f32(out).double()
            amax, sc = float(out.meta[0, 0]), float(out.meta[0, 1])
            true_max = float(ref.abs().max())
            assert abs(amax - true_max) <= 1e-5 * true_max, (amax, true_max)
            assert true_max * sc < 32768.0, (true_max, sc)
        err = float((got - ref).abs().max() / ref.abs().max())
        worst = max(worst, err)
        print(f'case {ci}: n{n} {h}x{w} cin{cin} cout{cout} k{k} s{s} act{act} res{rm} f32out{int(of32)} hint{hint}: rel err {err:.2e}', flush=True)
        assert err < 3e-6, err
    print(f'check ok, worst {worst:.2e}')


HINTS = {60: 'pingpong', 14: '192x256', 42: '128x256s3', 52: '160x256ns3', 49: '256x64s3', 12: '160x256n', 13: '224x256n', 31: '256x256m32', 34: '192x256m32', 21: '256x256pp', 22: '128x256pp', 26: '256x128pp', 1: '256x256', 2: '128x256', 3: '256x256n', 4: '192x256n', 5: '128x256n', 6: '256x128', 7: '128x128', 8: '128x128n',
         9: '256x64', 10: '256x64b', 11: '128x64'}

R101_B16 = [
    # (count, M-shape n h w, cin, cout, k, stride, residual)
    (2, (16, 200, 200), 256, 256, 3, 1, 0),
    (4, (1600, 14, 14), 256, 256, 3, 1, 0),
    (25, (16, 50, 50), 256, 256, 3, 1, 0),
    (23, (16, 50, 50), 256, 1024, 1, 1, 1),
    (23, (16, 50, 50), 1024, 256, 1, 1, 0),
    (1, (1, 1, 16000), 12544, 1024, 1, 1, 0),
    (4, (16, 200, 200), 64, 256, 1, 1, 1),
    (2, (16, 100, 100), 256, 256, 3, 1, 0),
    (1, (1, 1, 313600), 256, 1024, 1, 1, 0),
    (4, (16, 100, 100), 128, 512, 1, 1, 1),
    (4, (16, 100, 100), 128, 128, 3, 1, 0),
    (3, (16, 200, 200), 64, 64, 3, 1, 0),
    (3, (16, 25, 25), 512, 512, 3, 1, 0),
    (1, (16, 200, 200), 256, 256, 1, 1, 1),
    (2, (16, 200, 200), 256, 64, 1, 1, 0),
    (3, (16, 100, 100), 512, 128, 1, 1, 0),
    (3, (16, 25, 25), 512, 2048, 1, 1, 1),
    (2, (16, 25, 25), 2048, 512, 1, 1, 0),
    (1, (16, 200, 200), 256, 15, 1, 1, 0),
    (1, (1, 1, 1254400), 256, 2, 1, 1, 0),
]


def r101_layers(B):
    """The R101 layer shapes at B tiles per forward (R101_B16 scaled in its batch dimension)."""
    out = []
    for cnt, (n, h, w), cin, cout, k, s, rs in R101_B16:
        if n == 16:
            n = B
        elif n == 1600:
            n = 100 * B
        elif n == 1:
            w = w // 16 * B
        out.append((cnt, (n, h, w), cin, cout, k, s, rs))
    return out


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def time_layers(sweep=True, old=True, B=16):
    tot_new = tot_old = tot_fl = tot_auto = 0.0
    for cnt, (n, h, w), cin, cout, k, s, rs in r101_layers(B):
        L = Layer(cout, cin, k, k, seed=1)
        x = torch.randn(n, h, w, cin, device=dev)
        xp = p32.from_f32(x)
        pd = k // 2
        ho, wo = h // s, w // s
        res = p32.from_f32(torch.randn(n, ho, wo, cout, device=dev)) if rs else None
        rm = RES_SAME if rs else RES_NONE
        of32 = cout % 32 != 0
        fl = 2.0 * n * ho * wo * cout * cin * k * k
        hints = [0]
        if sweep:
            hints += [hh for hh in HINTS if ((hh <= 5 or hh in (12, 13, 14, 21, 22, 31, 34, 42, 52)) and L.cout_pad % 256 == 0) or ((6 <= hh <= 8 or hh == 26) and L.cout_pad % 128 == 0) or 9 <= hh <= 11 or hh == 49]
        tt = {}
        for hint in hints:
            tt[hint] = timeit(lambda: conv_p32(xp, L, s, pd, ACT_RELU, res, rm, of32, 16 if of32 else 0, hint))
        t_old = float('nan')
        if old and cin * h * w * 8 < (1 << 30) and not of32 and not rs:
            amax = x.abs().max().reshape(1)
            t_old = timeit(lambda: conv_old(x, amax, L, s, pd, ACT_RELU))
        best = min((v, kk) for kk, v in tt.items() if kk)[1] if sweep else 0
        t_new = tt[best]
        tot_new += cnt * t_new
        tot_auto += cnt * tt[0]
        tot_fl += cnt * fl
        if t_old == t_old:
            tot_old += cnt * t_old
        print(f'M={n*ho*wo:7d} cin={cin:5d} cout={cout:5d} k{k} res{rs} x{cnt:2d}: auto {tt[0]*1e3:7.1f} us {fl/tt[0]/1e9:6.1f} TF/s | best {HINTS.get(best, "auto"):9s} '
              f'{t_new*1e3:7.1f} us {fl/t_new/1e9:6.1f} TF/s | old {t_old*1e3:7.1f} us | ' +
              ' '.join(f'{HINTS[kk]}={v*1e3:.0f}' for kk, v in tt.items() if kk), flush=True)
    print(f'weighted total: auto {tot_auto:.2f} ms = {tot_fl/tot_auto/1e9:.1f} TF/s; best-of-sweep {tot_new:.2f} ms = {tot_fl/tot_new/1e9:.1f} TF/s; '
          f'old (layers it ran) {tot_old:.2f} ms')


def key_layers(hint=1, which='mfma'):
    """A few layers at one tile config: the A/B probe for kernel variants (AB_LIB=... selects the library)."""
    out = []
    layers = {'mfma': [((16, 200, 200), 256, 256, 3, 0), ((16, 50, 50), 256, 256, 3, 0), ((1, 1, 16000), 12544, 1024, 1, 0),
                       ((16, 50, 50), 1024, 256, 1, 0)],
              # the res4 bottleneck tail at the bench's 48 tiles: conv2 (3x3) and conv3 (1x1 + residual) -- the pair a fused kernel would replace
              'res4': [((48, 50, 50), 256, 256, 3, 0), ((48, 50, 50), 256, 1024, 1, 1), ((48, 50, 50), 1024, 256, 1, 0)],
              'mfma48': [((48, 200, 200), 256, 256, 3, 0), ((4800, 14, 14), 256, 256, 3, 0), ((48, 50, 50), 256, 256, 3, 0), ((1, 1, 48000), 12544, 1024, 1, 0),
                         ((48, 50, 50), 1024, 256, 1, 0), ((48, 50, 50), 256, 1024, 1, 1)],
              'hbm48': [((48, 200, 200), 64, 256, 1, 1), ((48, 100, 100), 128, 512, 1, 1), ((48, 200, 200), 256, 64, 1, 0), ((48, 25, 25), 512, 2048, 1, 1)],
              'hbm': [((16, 50, 50), 256, 1024, 1, 1), ((16, 200, 200), 64, 256, 1, 1), ((16, 200, 200), 256, 256, 1, 1),
                      ((16, 100, 100), 128, 512, 1, 1)]}[which]
    for (n, h, w), cin, cout, k, rs in layers:
        L = Layer(cout, cin, k, k, seed=1)
        xp = p32.from_f32(torch.randn(n, h, w, cin, device=dev))
        res = p32.from_f32(torch.randn(n, h, w, cout, device=dev)) if rs else None
        fl = 2.0 * n * h * w * cout * cin * k * k
        gb = (n * h * w * (cin + cout * (2 if rs else 1)) * 4 + cout * cin * k * k * 4) / 1e9
        t = min(timeit(lambda: conv_p32(xp, L, 1, k // 2, ACT_RELU, res, RES_SAME if rs else RES_NONE, hint=hint), reps=8) for _ in range(3))
        out.append(f'M={n*h*w} K={cin*k*k} N={cout}: {t*1e3:.1f} us {fl/t/1e9:.1f} TF/s {gb/t:.2f} TB/s')
    print(os.environ.get('AB_LIB', 'default'), f'hint {hint} |', ' | '.join(out), flush=True)


def single_sweep(B=48):
    """The single-plane layers (mask head) at B tiles over every tile this build instantiates: which tile suits ONE MFMA per product."""
    for name, (n, h, w), cin, cout, k in (('mask_fcn 3x3', (100 * B, 14, 14), 256, 256, 3), ('deconv 1x1', (1, 1, 19600 * B), 256, 1024, 1)):
        L = Layer(cout, cin, k, k, seed=1)
        xp = p32.from_f32(torch.randn(n, h, w, cin, device=dev))
        fl = 2.0 * n * h * w * cout * cin * k * k
        row = []
        for hint in [0, 1, 2, 4, 6, 7, 9, 11, 12, 13] + ([3, 5, 8, 14, 42, 52, 49] if _lib.is_dev_build() else []):
            for single in (1,):
                try:
                    t = min(timeit(lambda: conv_p32(xp, L, 1, k // 2, ACT_RELU, hint=hint, single=single), reps=4) for _ in range(2))
                    row.append(f'{HINTS.get(hint, "auto")}={t*1e3:.0f}us/{fl/t/1e9:.0f}TF')
                except Exception as e:
                    row.append(f'{HINTS.get(hint, "auto")}=ERR')
        t3 = min(timeit(lambda: conv_p32(xp, L, 1, k // 2, ACT_RELU, hint=0, single=0), reps=4) for _ in range(2))
        print(f'{name} M={n*h*w} K={cin*k*k} N={cout}: three-MFMA auto {t3*1e3:.0f}us | single: ' + ' '.join(row), flush=True)


def occupancy_probe():
    """res4 conv3 (1x1, 256 -> 1024, + residual, 256 x 256 tiles) on 64 / 128 / 256 / 512 / 1024 tiles: is a tile's time set by its own
    CU (same time whatever the number of busy CUs) or by what the whole chip asks of HBM at once (faster when fewer CUs run)?"""
    L = Layer(1024, 256, 1, 1, seed=1)
    for tiles in (64, 128, 256, 512, 1024, 1876):
        rows = tiles // 4 * 256
        xp = p32.from_f32(torch.randn(1, 1, rows, 256, device=dev))
        res = p32.from_f32(torch.randn(1, 1, rows, 1024, device=dev))
        t = min(timeit(lambda: conv_p32(xp, L, 1, 0, ACT_RELU, res, RES_SAME, hint=1), reps=10) for _ in range(3))
        t0 = min(timeit(lambda: conv_p32(xp, L, 1, 0, ACT_RELU, None, RES_NONE, hint=1), reps=10) for _ in range(3))
        print(f'{os.environ.get("AB_LIB", "default")} tiles {tiles:5d} ({tiles / 256:.2f} rounds): with residual {t*1e3:7.1f} us, without {t0*1e3:7.1f} us', flush=True)


def pingpong_check(time_too=True):
    """Tile hint 60 (the ping-pong kernel for 1x1 layers) against the plain kernel: bit-identical planes and metas on shapes
    with / without residual, ragged M, one and many K-steps, several scale groups; then its time beside the plain kernel's."""
    shapes = [  # n, h, w, cin, cout, residual, groups
        (2, 25, 25, 256, 1024, 1, 1), (3, 31, 17, 64, 256, 1, 1), (1, 40, 40, 1024, 256, 0, 1), (2, 20, 20, 32, 128, 0, 1),
        (4, 50, 50, 256, 1024, 1, 4), (1, 1, 777, 512, 128, 0, 1), (1, 9, 9, 96, 256, 1, 1), (6, 16, 16, 128, 512, 1, 2),
    ]
    for ci, (n, h, w, cin, cout, rs, groups) in enumerate(shapes):
        g = torch.Generator().manual_seed(500 + ci)
        x = (torch.randn(n, h, w, cin, generator=g) * 2.0).to(dev)
        L = Layer(cout, cin, 1, 1, seed=ci)
        xp = p32.from_f32(x, groups=groups)
        res = p32.from_f32((torch.randn(n, h, w, cout, generator=g) * 1.5).to(dev), groups=groups) if rs else None
        outs = []
        for hint in (0, 60):
            out = p32.alloc((n, h, w, cout), dev, groups=groups)
            d = _lib.ConvP32Desc(_lib.ptr(xp.buf), _lib.ptr(xp.meta), _lib.ptr(L.w_p32), _lib.ptr(L.scale3), _lib.ptr(L.bias_d),
                                 _lib.ptr(res.buf) if rs else 0, _lib.ptr(res.meta) if rs else 0, _lib.ptr(out.buf), _lib.ptr(out.meta),
                                 L.wbound, L.bbound, n, h, w, cin, h, w, cout, L.cout_pad, 1, 1, 1, 0, ACT_RELU, RES_SAME if rs else RES_NONE, 0, 0, hint,
                                 0, 0, 0, 0, 0, 0, groups, (n * h * w) // groups, 0)
            _lib.check(lib.demia_conv2d_p32(C.byref(d), st()), 'demia_conv2d_p32')
            torch.cuda.synchronize()
            outs.append(out)
        same = torch.equal(outs[0].buf, outs[1].buf) and torch.equal(outs[0].meta, outs[1].meta)
        err = float((p32.to_f32(outs[0]) - p32.to_f32(outs[1])).abs().max())
        print(f'pingpong case {ci}: M={n*h*w} K={cin} N={cout} res{rs} groups{groups}: identical={same} max abs diff {err:.3e}', flush=True)
        assert same, ci
    if not time_too:
        return
    for name, (n, h, w), cin, cout, rs in (('res4 conv3', (48, 50, 50), 256, 1024, 1), ('res4 conv1', (48, 50, 50), 1024, 256, 0),
                                            ('res3 conv3', (48, 100, 100), 128, 512, 1), ('res3 conv1', (48, 100, 100), 512, 128, 0),
                                            ('res2 conv3', (48, 200, 200), 64, 256, 1), ('res2 conv1', (48, 200, 200), 256, 64, 0),
                                            ('res5 conv3', (48, 25, 25), 512, 2048, 1), ('res5 conv1', (48, 25, 25), 2048, 512, 0)):
        if cout % 128:
            continue
        L = Layer(cout, cin, 1, 1, seed=1)
        xp = p32.from_f32(torch.randn(n, h, w, cin, device=dev))
        res = p32.from_f32(torch.randn(n, h, w, cout, device=dev)) if rs else None
        tt = {}
        for hint in (0, 60):
            tt[hint] = min(timeit(lambda: conv_p32(xp, L, 1, 0, ACT_RELU, res, RES_SAME if rs else RES_NONE, hint=hint), reps=8) for _ in range(3))
        print(f'{name}: M={n*h*w} K={cin} N={cout} res{rs}: plain {tt[0]*1e3:7.1f} us, ping-pong {tt[60]*1e3:7.1f} us ({tt[0]/tt[60]:.2f}x)', flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'pingpong':
        pingpong_check(len(sys.argv) < 3 or sys.argv[2] != 'notime')
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'pptime':         # timing-only variants (AB_LIB=...): hint 60 on the res4 pointwise layers
        for name, (n, h, w), cin, cout, rs in (('res4 conv3', (48, 50, 50), 256, 1024, 1), ('res4 conv1', (48, 50, 50), 1024, 256, 0)):
            L = Layer(cout, cin, 1, 1, seed=1)
            xp = p32.from_f32(torch.randn(n, h, w, cin, device=dev))
            res = p32.from_f32(torch.randn(n, h, w, cout, device=dev)) if rs else None
            t = min(timeit(lambda: conv_p32(xp, L, 1, 0, ACT_RELU, res, RES_SAME if rs else RES_NONE, hint=60), reps=8) for _ in range(3))
            print(f'{os.environ.get("AB_LIB", "default")} {name}: ping-pong {t*1e3:7.1f} us', flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'occupancy':
        occupancy_probe()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'single':
        single_sweep(int(sys.argv[2]) if len(sys.argv) > 2 else 48)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'key':
        key_layers(int(sys.argv[2]) if len(sys.argv) > 2 else 1, sys.argv[3] if len(sys.argv) > 3 else 'mfma')
        sys.exit(0)
    mode = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if mode in ('check', 'all'):
        check()
    if mode in ('time', 'all'):
        time_layers()
    if mode == 'auto':          # auto tile only, at B tiles: gpu_conv_p32_check.py auto <B>
        time_layers(sweep=False, old=False, B=int(sys.argv[2]) if len(sys.argv) > 2 else 48)
    if mode == 'sweep':         # every tile at B tiles
        time_layers(sweep=True, old=False, B=int(sys.argv[2]) if len(sys.argv) > 2 else 48)
