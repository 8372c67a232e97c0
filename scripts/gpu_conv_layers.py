"""Per-layer-shape conv timings (HIP events around every launch of three forwards, R101, 2048^2, B tiles): achieved
TFLOP/s f32-equivalent and algorithmic GB/s.  usage: gpu_conv_layers.py <precision> <batch> [csv path]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from deepemia_amd import synth, engine as E, _lib
import os, pathlib
if os.environ.get('AB_LIB'):
    _lib.LIB_PATH = pathlib.Path(os.environ['AB_LIB']).resolve()
prec = sys.argv[1]; B = int(sys.argv[2])
sd = synth.random_d2_state_dict(101, 2, 0)
eng = E.MaskRCNNEngine(sd, 101, 2, 0.3, 'cuda:0', prec)
x = torch.from_numpy(np.stack([synth.em_tile(i, 2048) for i in range(min(B, 4))])).cuda()
if B > 4:
    x = torch.cat([x, synth.em_tiles_device(range(50004, 50000 + B), 2048, 'cuda:0')])
log = []
orig = eng.conv_p32 if eng.p32 else eng.conv
def conv(xx, L, *a, **kw):
    n, h, w, cin = xx.shape
    ho = (h + 2 * L.pad - L.kh) // L.stride + 1; wo = (w + 2 * L.pad - L.kw) // L.stride + 1
    log.append((n * ho * wo, L.cout, L.kh * L.kw * cin, L.kh, L.stride))
    return orig(xx, L, *a, **kw)
if eng.p32:
    eng.conv_p32 = conv
else:
    eng.conv = conv
for _ in range(2): eng.forward(x)
torch.cuda.synchronize(); log.clear(); eng.conv_events = []
for _ in range(3): eng.forward(x)
torch.cuda.synchronize()
ev = eng.conv_events; n = len(ev) // 3
agg = {}
for i in range(n):
    t = np.mean([ev[i + k * n][0].elapsed_time(ev[i + k * n][1]) for k in range(3)]); fl = ev[i][2]
    key = log[i]
    a = agg.setdefault(key, [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += t; a[2] += fl; a[3] += ev[i][4]      # [4]: the engine's algorithmic bytes (input + weights + output + residual, 4 B per element)
tot_t = sum(a[1] for a in agg.values()); tot_f = sum(a[2] for a in agg.values())
print(f'total conv {tot_t:.2f} ms, {tot_f/tot_t/1e9:.1f} TF/s')
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, co, K, kh, st = key
    bm, bn = 128, (128 if co >= 128 else (64 if co >= 64 else 32))
    blocks = -(-M // bm) * -(-((co + 31) // 32 * 32) // bn)
    print(f'M={M:7d} Cout={co:5d} K={K:6d} k{kh} s{st} x{a[0]:3d} time={a[1]:7.3f} ms ({a[1]/tot_t*100:4.1f}%) {a[2]/a[1]/1e9:7.1f} TF/s blocks={blocks}')
if len(sys.argv) > 3:
    import csv
    with open(sys.argv[3], 'w', newline='') as f:
        w = csv.writer(f)
        # the roof that BINDS a layer: its FLOPs at 833 TFLOP/s f32-equivalent (fp16 dense peak / 3 MFMAs per product) or its algorithmic
        # bytes (residual included) at 6.3 TB/s -- what a copy kernel reaches on this chip (MI355X_MICROARCH.md) --, whichever takes longer
        w.writerow(['M_rows', 'Cout', 'K', 'kernel', 'stride', 'launches_per_forward', 'us_per_launch', 'share_of_conv_time', 'tflops_f32_equivalent',
                    'frac_of_833_roof', 'algorithmic_GB_per_s', 'us_at_mfma_roof', 'us_at_6p3_TBs', 'binding_roof', 'frac_of_binding_roof'])
        for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            M, co, K, kh, st = key
            us = a[1] / a[0] * 1e3
            t_m, t_h = a[2] / a[0] / 833.33e12 * 1e6, a[3] / a[0] / 6.3e12 * 1e6
            w.writerow([M, co, K, f'{kh}x{kh}', st, a[0], round(us, 1), round(a[1] / tot_t, 4), round(a[2] / a[1] / 1e9, 1),
                        round(a[2] / a[1] / 1e9 / 833.33, 3), round(a[3] / a[1] / 1e6, 0), round(t_m, 1), round(t_h, 1),
                        'hbm' if t_h > t_m else 'mfma', round(max(t_m, t_h) / us, 3)])
        bound = sum(max(a[2] / 833.33e12, a[3] / 6.3e12) for a in agg.values()) * 1e3
        w.writerow(['total', '', '', '', '', sum(a[0] for a in agg.values()), '', 1.0, round(tot_f / tot_t / 1e9, 1), round(tot_f / tot_t / 1e9 / 833.33, 3), '',
                    '', '', '', round(bound / tot_t, 3)])
