"""Dev script: throughput of the ensemble (R50 + R101) per-tile path vs the single-model batched path."""
import sys, time, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepemia_amd import synth
from deepemia_amd.engine import MaskRCNNEngine
from deepemia_amd.predictor import Predictor
from deepemia_amd.functions.inference import InferencePipeline
B = 8
preds = [Predictor(MaskRCNNEngine(synth.random_d2_state_dict(d, 2, 0), d, 2, 0.3, 'cuda:0')) for d in (50, 101)]
pipe = InferencePipeline(preds, 'bench', {}, {})
x = torch.from_numpy(np.stack([synth.em_tile(i, 2048) for i in range(B)])).cuda()
thr = {0: (0.3, 0.7), 1: (0.3, 0.5)}
for mids in ((0,), (1,), (0, 1)):
    for rep in range(3):
        pipe.clear_cache(); torch.cuda.synchronize(); t0 = time.perf_counter()
        out = pipe.process_tile_batch(f'k{rep}', x, {1}, thr, model_ids=mids); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    n = sum(0 if r[0] is None else int(r[0].shape[0]) for r in out)
    print(f'models {mids}: {dt*1e3:.1f} ms per {B} tiles = {B/dt:.1f} tiles/s, {n} instances')
