"""Dev script: one conv case of the parity suite against torch, for an A/B build of the library (AB_LIB)."""
import os, sys, pathlib, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepemia_amd import _lib, synth, engine as E
if os.environ.get("AB_LIB"):
    _lib.LIB_PATH = pathlib.Path(os.environ["AB_LIB"]).resolve()
eng = E.MaskRCNNEngine(synth.random_d2_state_dict(50, 2, 0), 50, 2, 0.3, "cuda:0", "f16x2")
bk = int(eng.lib.demia_conv_f16x2_kstep())
worst = 0
for (cin, cout, k, stride, pad, h, w, n) in [(64, 256, 3, 1, 1, 300, 300, 1), (256, 256, 3, 1, 1, 14, 14, 5), (128, 192, 3, 2, 1, 61, 47, 2),
                                            (64, 64, 1, 1, 0, 50, 50, 2), (512, 256, 1, 1, 0, 50, 50, 1), (32, 64, 3, 1, 1, 20, 20, 1)]:
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn((n, cin, h, w), generator=g); wt = torch.randn((cout, cin, k, k), generator=g) / (cin * k * k) ** 0.5
    y = F.relu(F.conv2d(x, wt, None, stride=stride, padding=pad))
    cp = (cout + 63) // 64 * 64
    wp = torch.zeros((cp, k, k, cin)); wp[:cout] = wt.permute(0, 2, 3, 1)
    L = E.ConvLayer(wp.cuda(), None, None, cin, cout, cp, k, k, stride, pad)
    planes, sw = E.split2_f16_scaled(wp.cuda())
    L.w3 = E.tile_weight_planes(planes, bk); L.scale3 = (1.0 / sw[:cout]).contiguous()
    out = eng.conv(x.permute(0, 2, 3, 1).contiguous().cuda(), L, act=1).cpu().permute(0, 3, 1, 2)
    err = float((out - y).abs().max() / y.abs().max()); worst = max(worst, err)
    print((cin, cout, k, stride, h, w, n), "rel err %.2e" % err)
assert worst < 2e-5, worst
print("OK")
