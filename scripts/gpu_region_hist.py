"""Dev script (GPU): how large are the mask regions of a real 48-tile forward?  (words of the bbox region grown by 2 pixels:
what decides between the small and the large variant of the per-mask kernels)"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from deepemia_amd import synth
from deepemia_amd.engine import MaskRCNNEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
sd = synth.random_d2_state_dict(101, 2, 0)
eng = MaskRCNNEngine(sd, 101, 2, 0.3, 'cuda:0', 'f16x2')
x = synth.em_tiles_device(range(900, 900 + B), 2048, 'cuda:0')
r = eng.forward(x)
cnt = r.count.cpu().numpy()
bb = r.bbox.cpu().numpy()
words = []
for i in range(B):
    b = bb[i, :cnt[i]]
    b = b[b[:, 0] >= 0]
    rh = np.minimum(b[:, 2] + 2, 2047) - np.maximum(b[:, 0] - 2, 0) + 1
    rw = (np.minimum(b[:, 3] + 2, 2047) >> 5) - (np.maximum(b[:, 1] - 2, 0) >> 5) + 1
    words.append(rh * rw)
w = np.concatenate(words)
print('masks', len(w), 'median words', int(np.median(w)), 'p90', int(np.percentile(w, 90)), 'max', int(w.max()))
for t in (256, 512, 1024, 2048, 4096, 8192):
    print(f'<= {t}: {(w <= t).mean():.3f}')
