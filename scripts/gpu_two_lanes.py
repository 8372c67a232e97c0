"""Dev script (GPU): does a SECOND pipeline in the same process (its own engine = its own arena and graphs, its own streams and
host thread) add throughput the way a second process does?  usage: gpu_two_lanes.py <lanes> [steps] [batch]"""
import sys, threading, time
import numpy as np, torch
sys.path.insert(0, '.')
from deepemia_amd import synth
from deepemia_amd.engine import MaskRCNNEngine
from deepemia_amd.functions.inference import InferencePipeline, measurement_csv_text
from deepemia_amd.predictor import Predictor

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
B = int(sys.argv[3]) if len(sys.argv) > 3 else 48
dev = 'cuda:0'
sd = synth.random_d2_state_dict(101, 2, seed=0)
x = synth.em_tiles_device(range(900, 900 + B), 2048, dev)
CT, SMALL = {0: (0.3, 0.7), 1: (0.3, 0.5)}, {1}
MIN_AREA = max(5, 2048 * 2048 * 0.000005 * 0.05)
lanes = []
for l in range(L):
    eng = MaskRCNNEngine(sd, 101, 2, 0.3, dev, 'f16x2')
    pipe = InferencePipeline([Predictor(eng)], f'lane{l}', {}, {})
    pipe.use_graphs, pipe.graph_after, pipe.forward_batch = True, 1, B
    pipe.graph_slots, pipe.clone_graph_outputs, pipe.pooled_planes = 2, False, True
    lanes.append(dict(pipe=pipe, net=torch.cuda.Stream(device=dev), post=torch.cuda.Stream(device=dev), n=0))

def run(lane, steps):
    torch.cuda.set_device(0)
    pipe = lane['pipe']
    def launch():
        with torch.cuda.stream(lane['net']):
            return pipe.forward_async(0, x)
    h = launch()
    for i in range(steps):
        nxt = launch() if i + 1 < steps else None
        with torch.cuda.stream(lane['post']):
            dets = pipe.finish_forward(h)
            res = pipe.process_tile_batch(f's{i}', x, SMALL, CT, dets=dets)
            text = measurement_csv_text([(f't{t}.tif', r[2], r[3]) for t, r in enumerate(res)], ('class_0', 'class_1'), MIN_AREA)
            lane['n'] = sum(0 if r[0] is None else int(r[0].shape[0]) for r in res)
        h = nxt

for lane in lanes:       # warm-up (capture) one lane after the other
    run(lane, 3)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    ths = [threading.Thread(target=run, args=(lane, K)) for lane in lanes]
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'lanes {L}: {L * K * B / dt:.1f} tiles/s ({dt / (L * K) * 1e3:.2f} ms per step), instances of a last step {[l["n"] for l in lanes]}', flush=True)
