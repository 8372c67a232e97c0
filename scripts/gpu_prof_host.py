import sys, time, cProfile, pstats, numpy as np, torch
sys.path.insert(0, '.')
from deepemia_amd import synth
from deepemia_amd.engine import MaskRCNNEngine
from deepemia_amd.predictor import Predictor
from deepemia_amd.functions.inference import InferencePipeline
sd = synth.random_d2_state_dict(101, 2, 0)
eng = MaskRCNNEngine(sd, 101, 2, 0.3, 'cuda:0', 'f16x2')
pipe = InferencePipeline([Predictor(eng)], 'bench', {}, {})
x = torch.from_numpy(np.stack([synth.em_tile(i, 2048) for i in range(16)])).cuda()
thr = {0: (0.3, 0.7), 1: (0.3, 0.5)}
for _ in range(2):
    pipe.clear_cache(); pipe.process_tile_batch('k', x, {1}, thr)
pipe.clear_cache(); pipe._predict_batch(0, 'k', x); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
pipe.process_tile_batch('k', x, {1}, thr); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
import time
for _ in range(3):
    pipe.clear_cache(); pipe._predict_batch(0, 'k', x); torch.cuda.synchronize()
    t0 = time.perf_counter(); pipe.process_tile_batch('k', x, {1}, thr); torch.cuda.synchronize()
    print('postproc wall ms', (time.perf_counter() - t0) * 1e3)
