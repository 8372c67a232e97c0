"""Dev script (GPU): is the f16x2 forward deterministic run to run, and where do a tile alone and the same tile inside a
batch first differ?"""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from deepemia_amd import synth, engine as E

sd = synth.random_d2_state_dict(101, 2, 0)
eng = E.MaskRCNNEngine(sd, 101, 2, 0.3, 'cuda:0', 'f16x2')
tiles = np.stack([synth.em_tile(i, 2048) for i in range(4)])
x = torch.from_numpy(tiles).cuda()


def csum(t):
    return int(t.buf.view(torch.int16).to(torch.int64).sum().item())


def feats_of(xx):
    xin, newh, neww, ph, pw = eng.preprocess(xx)
    f = eng.backbone(xin, ph, pw)
    torch.cuda.synchronize()
    sums = {k: csum(v) for k, v in f.items()}
    out = {k: eng.dense(v).clone() for k, v in f.items()}
    sums2 = {k: csum(v) for k, v in f.items()}
    meta = {k: v.meta.clone() for k, v in f.items()}
    return out, meta, sums, sums2


a, ma, sa, sa2 = feats_of(x)
b, mb, sb, sb2 = feats_of(x)
g, mg, sg, sg2 = feats_of(x)
for k in a:
    print('run-to-run', k, 'dense diff a-b %.3g b-g %.3g' % (float((a[k] - b[k]).abs().max()), float((b[k] - g[k]).abs().max())),
          'buffer sums', sa[k], sb[k], sg[k], 'after dense', sa2[k], sb2[k], sg2[k])
c, mc, _, _ = feats_of(x[1:2].contiguous())
for k in a:
    d = (a[k][1:2] - c[k]).abs()
    d2 = (b[k][1:2] - c[k]).abs()
    print('batch vs single', k, 'a: %.3g  b: %.3g' % (float(d.max()), float(d2.max())), 'meta batch', ma[k].tolist(), 'single', mc[k].tolist())
