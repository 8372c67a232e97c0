"""Dev script (GPU): run-to-run determinism of the f16x2 backbone at a given size -- checksums of every conv output right
after its launch and of the stage outputs after the whole backbone, over several runs."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from deepemia_amd import synth, engine as E, p32

depth, B, size = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sd = synth.random_d2_state_dict(depth, 2, 0)
eng = E.MaskRCNNEngine(sd, depth, 2, 0.3, 'cuda:0', 'f16x2')
x = torch.from_numpy(np.stack([synth.em_tile(i, size) for i in range(B)])).cuda()
orig = eng.conv_p32
log = []


def csum(t):
    b = t.buf if isinstance(t, p32.P32) else t
    v = b.view(torch.int16) if b.dtype == torch.float16 else b.view(torch.int32)
    return int(v.to(torch.int64).sum().item()) , int(b.data_ptr())


def hook(xx, L, *a, **kw):
    out = orig(xx, L, *a, **kw)
    log.append((tuple(xx.shape), L.cout, L.kh, csum(out), csum(xx)))
    return out


eng.conv_p32 = hook
runs, ends = [], []
for r in range(4):
    log.clear()
    xin, newh, neww, ph, pw = eng.preprocess(x)
    f = eng.backbone(xin, ph, pw)
    torch.cuda.synchronize()
    runs.append(list(log))
    ends.append({k: csum(v) for k, v in f.items()})
bad = 0
for i in range(len(runs[0])):
    row = [runs[r][i] for r in range(4)]
    outs = [q[3][0] for q in row]; ins = [q[4][0] for q in row]
    if len(set(outs)) > 1 or len(set(ins)) > 1:
        bad += 1
        print('layer', i, row[0][:3], 'out sums', outs, 'in sums', ins, 'out ptrs', [hex(q[3][1]) for q in row], 'in ptrs', [hex(q[4][1]) for q in row])
print('layers with run-to-run differences:', bad, 'of', len(runs[0]))
for k in ends[0]:
    print('end-of-backbone', k, [ends[r][k][0] for r in range(4)], [hex(ends[r][k][1]) for r in range(4)])
