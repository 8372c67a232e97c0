"""Dev script (GPU box): throughput of the batched forward, R101, 2048^2 tiles."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from deepemia_amd import synth
from deepemia_amd.engine import MaskRCNNEngine
depth = int(sys.argv[1]); prec = sys.argv[2]; B = int(sys.argv[3]); size = int(sys.argv[4]); iters = int(sys.argv[5])
sd = synth.random_d2_state_dict(depth, 2, 0)
eng = MaskRCNNEngine(sd, depth, 2, 0.3, 'cuda:0', prec)
imgs = np.stack([synth.em_tile(i, size) for i in range(B)])
x = torch.from_numpy(imgs).cuda()
for _ in range(2): out = eng.forward(x)
torch.cuda.synchronize()
print('det counts', out.count.tolist())
t = time.time()
for _ in range(iters): out = eng.forward(x)
torch.cuda.synchronize(); dt = (time.time() - t) / iters
print(f'depth {depth} {prec} B={B} size={size}: {dt*1e3:.2f} ms/batch, {B/dt:.2f} tiles/s')
