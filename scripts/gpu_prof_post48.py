"""Where the host time of the post-processing of one batch goes (alone on the device, no forward beside it)."""
import sys, time, cProfile, pstats, io, numpy as np, torch
sys.path.insert(0, '.')
from deepemia_amd import synth
from deepemia_amd.engine import MaskRCNNEngine
from deepemia_amd.predictor import Predictor
from deepemia_amd.functions.inference import InferencePipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
sd = synth.random_d2_state_dict(101, 2, 0)
eng = MaskRCNNEngine(sd, 101, 2, 0.3, 'cuda:0', 'f16x2')
pipe = InferencePipeline([Predictor(eng)], 'bench', {}, {})
pipe.forward_batch = B
x = synth.em_tiles_device(range(500, 500 + B), 2048, 'cuda:0')
thr = {0: (0.3, 0.7), 1: (0.3, 0.5)}
for _ in range(2):
    pipe.clear_cache(); pipe.process_tile_batch('k', x, {1}, thr)
pipe.clear_cache(); pipe._predict_batch(0, 'k', x); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
pipe.process_tile_batch('k', x, {1}, thr); torch.cuda.synchronize()
pr.disable()
for key in ('cumulative', 'tottime'):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(30); print(s.getvalue()[:6000])
for _ in range(3):
    pipe.clear_cache(); pipe._predict_batch(0, 'k', x); torch.cuda.synchronize()
    t0 = time.perf_counter(); pipe.process_tile_batch('k', x, {1}, thr); torch.cuda.synchronize()
    print(f'B={B} postproc wall ms', (time.perf_counter() - t0) * 1e3)
