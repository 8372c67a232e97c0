"""Dev script: time forward-only vs full per-tile pipeline (R101, 2048^2, B tiles)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from deepemia_amd import synth
from deepemia_amd.engine import MaskRCNNEngine
from deepemia_amd.predictor import Predictor
from deepemia_amd.functions.inference import InferencePipeline
depth = int(sys.argv[1]); prec = sys.argv[2]; B = int(sys.argv[3]); size = int(sys.argv[4]); iters = int(sys.argv[5])
sd = synth.random_d2_state_dict(depth, 2, 0)
eng = MaskRCNNEngine(sd, depth, 2, 0.3, 'cuda:0', prec)
pipe = InferencePipeline([Predictor(eng)], 'bench', {}, {})
x = torch.from_numpy(np.stack([synth.em_tile(i, size) for i in range(B)])).cuda()
thr = {0: (0.3, 0.7), 1: (0.3, 0.5)}
for it in range(iters + 1):
    torch.cuda.synchronize(); t0 = time.time()
    pipe.clear_cache()
    raw = pipe._predict_batch(0, 'k', x); torch.cuda.synchronize(); t1 = time.time()
    out = pipe.process_tile_batch('k', x, {1}, thr); torch.cuda.synchronize(); t2 = time.time()
    n = sum(0 if o[0] is None else o[0].shape[0] for o in out); rows = sum(len(r) for o in out for r in o[3])
    print(f'iter {it}: forward {1e3*(t1-t0):.1f} ms, postproc {1e3*(t2-t1):.1f} ms, instances {n}, contours {rows}')
