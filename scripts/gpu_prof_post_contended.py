"""Dev script (GPU): cProfile of the post-processing of a batch WHILE the next batch's forward runs (the state of the
bench's pipeline), by self time.  usage: gpu_prof_post_contended.py [batch]"""
import sys, time, cProfile, pstats, numpy as np, torch
sys.path.insert(0, '.')
from deepemia_amd import synth
from deepemia_amd.engine import MaskRCNNEngine
from deepemia_amd.predictor import Predictor
from deepemia_amd.functions.inference import InferencePipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sd = synth.random_d2_state_dict(101, 2, 0)
eng = MaskRCNNEngine(sd, 101, 2, 0.3, 'cuda:0', 'f16x2')
pipe = InferencePipeline([Predictor(eng)], 'bench', {}, {})
pipe.use_graphs, pipe.graph_after = True, 1
x = torch.from_numpy(np.stack([synth.em_tile(i % 16, 2048) for i in range(B)])).cuda()
thr = {0: (0.3, 0.7), 1: (0.3, 0.5)}
net, post = torch.cuda.Stream(), torch.cuda.Stream()


def launch():
    with torch.cuda.stream(net):
        return pipe.forward_async(0, x)


def run_post(h):
    with torch.cuda.stream(post):
        dets = pipe.finish_forward(h)
        return pipe.process_tile_batch('k', x, {1}, thr, dets=dets)


h = launch()
for _ in range(3):
    nh = launch(); run_post(h); h = nh
torch.cuda.synchronize()
times = []
pr = cProfile.Profile()
for it in range(6):
    nh = launch()
    t0 = time.perf_counter()
    if it == 5:
        pr.enable()
    run_post(h)
    if it == 5:
        pr.disable()
    times.append((time.perf_counter() - t0) * 1e3)
    h = nh
torch.cuda.synchronize()
print('contended post wall ms per batch:', [round(t, 1) for t in times])
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
