"""Dev script (GPU): the eight-tile headline parity record of tests/test_gpu_multitile_parity.py for any list of
precisions (f16x2 f32 f32x3 ...), one JSON per precision under gpurun_out/.  usage: gpu_multitile_parity.py <precision> ..."""
import sys
import torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from deepemia_amd import synth
from oracle import maskrcnn_ref
import test_gpu_multitile_parity as T

torch.set_num_threads(16)
sd = synth.random_d2_state_dict(101, 2, seed=0)
tiles = [synth.em_tile(i, 2048) for i in range(T.TILES)]
refs = [maskrcnn_ref.predict(t, sd, 101, T.THR) for t in tiles]
for prec in sys.argv[1:] or ['f16x2', 'f32', 'f32x3']:
    s = T.run_precision(sd, tiles, refs, prec, 'cuda:0')
    print(prec, {k: v for k, v in s.items() if k != 'per_tile'}, flush=True)
    for i, r in enumerate(s['per_tile']):
        print('  tile', i, 'moved', r['moved_positions'], 'differing', r['differing'], flush=True)
