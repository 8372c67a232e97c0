"""Dev script: the conv layers on ZERO data (weights and image): same instruction stream, far less switching power.
If the achieved TFLOP/s jump, the kernel is running into the clock the chip holds under load (DVFS), not into its structure."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepemia_amd import synth, engine as E
zero = len(sys.argv) > 1 and sys.argv[1] == "zero"
sd = synth.random_d2_state_dict(101, 2, 0)
if zero:
    sd = {k: (v * 0 if v.dtype.is_floating_point and "running_var" not in k else v) for k, v in sd.items()}
eng = E.MaskRCNNEngine(sd, 101, 2, 0.3, 'cuda:0', 'f32x3')
x = torch.from_numpy(np.stack([synth.em_tile(i, 2048) for i in range(8)])).cuda()
if zero:
    x = x * 0
xin, newh, neww, ph, pw = eng.preprocess(x)
if zero:
    xin = xin * 0
log = []
orig = eng.conv
def conv(xx, L, **kw):
    n, h, w, cin = xx.shape
    ho = (h + 2 * L.pad - L.kh) // L.stride + 1; wo = (w + 2 * L.pad - L.kw) // L.stride + 1
    log.append((n * ho * wo, L.cout, L.kh * L.kw * cin))
    return orig(xx, L, **kw)
eng.conv = conv
for _ in range(2): eng.backbone(xin, ph, pw)
torch.cuda.synchronize(); log.clear(); eng.conv_events = []
for _ in range(3): eng.backbone(xin, ph, pw)
torch.cuda.synchronize()
ev = eng.conv_events; n = len(ev) // 3
agg = {}
for i in range(n):
    t = np.mean([ev[i + k * n][0].elapsed_time(ev[i + k * n][1]) for k in range(3)])
    a = agg.setdefault(log[i], [0, 0.0, 0.0]); a[0] += 1; a[1] += t; a[2] += ev[i][2]
tot_t = sum(a[1] for a in agg.values()); tot_f = sum(a[2] for a in agg.values())
print(("ZERO data" if zero else "random data"), f'backbone+FPN conv {tot_t:.2f} ms, {tot_f/tot_t/1e9:.1f} TF/s')
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:5]:
    print(f'  M={key[0]:7d} Cout={key[1]:5d} K={key[2]:6d} x{a[0]:3d} {a[1]:7.3f} ms {a[2]/a[1]/1e9:7.1f} TF/s')
