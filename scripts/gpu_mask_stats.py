"""Dev script: bbox-region statistics of the masks the bench workload produces."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from deepemia_amd import synth
from deepemia_amd.engine import MaskRCNNEngine
eng = MaskRCNNEngine(synth.random_d2_state_dict(101, 2, 0), 101, 2, 0.3, 'cuda:0', 'f32x3')
x = torch.from_numpy(np.stack([synth.em_tile(i, 2048) for i in range(8)])).cuda()
out = eng.forward(x)
for t in range(8):
    n = int(out.count[t])
    area, bbox = eng.area_bbox(out.packed[t, :n].contiguous(), 2048, 2048)
    bb = bbox.cpu().numpy(); a = area.cpu().numpy()
    rh = bb[:, 2] - bb[:, 0] + 1; rw = (bb[:, 3] >> 5) - (bb[:, 1] >> 5) + 1
    words = rh * rw
    print(t, n, 'area med/max', int(np.median(a)), int(a.max()), 'rh med/max', int(np.median(rh)), int(rh.max()),
          'words med/max', int(np.median(words)), int(words.max()), '>8192:', int((words > 8192).sum()), '>12288:', int((words > 12288).sum()),
          '>32768:', int((words > 32768).sum()))
