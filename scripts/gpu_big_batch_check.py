"""Dev script: a 64-tile batch (activation tensors above 2 GiB: the buffer descriptors of the split conv kernel start
at each tile's first image) must give every repeated tile the same detections as its first copy."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepemia_amd import synth, engine as E
eng = E.MaskRCNNEngine(synth.random_d2_state_dict(101, 2, 0), 101, 2, 0.3, "cuda:0")
base = np.stack([synth.em_tile(i, 2048) for i in range(16)])
x = torch.from_numpy(np.concatenate([base] * 4)).cuda()
out = eng.forward(x)
cnt = out.count.cpu().numpy()
print("counts", cnt[:16], "...")
assert (cnt[:16] == cnt[16:32]).all() and (cnt[:16] == cnt[48:]).all()
for name in ("boxes", "scores", "classes"):
    t = getattr(out, name)
    for g in range(1, 4):
        assert torch.equal(t[:16], t[16 * g:16 * (g + 1)]), (name, g)
assert torch.equal(out.packed[:16], out.packed[48:])
print("64-tile batch: all four copies identical; p2 activation bytes", 64 * 200 * 200 * 256 * 4)
