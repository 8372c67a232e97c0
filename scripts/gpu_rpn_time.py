"""Dev script (GPU): time of the RPN proposal kernels on the heads of a real 48-tile forward, both selections."""
import os, sys, torch
sys.path.insert(0, '.')
from deepemia_amd import synth
from deepemia_amd.engine import MaskRCNNEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
sd = synth.random_d2_state_dict(101, 2, 0)
eng = MaskRCNNEngine(sd, 101, 2, 0.3, 'cuda:0', 'f16x2')
x = synth.em_tiles_device(range(900, 900 + B), 2048, 'cuda:0')
r = eng.forward(x, keep_intermediates=True)
feats = r.dbg['feats']
import deepemia_amd.engine as E
orig_conv = eng.conv
heads_cache = {}
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
full = t(lambda: eng.rpn(feats, 800, 800))
os.environ['DEMIA_RPN_SELECT'] = 'radix'
full_r = t(lambda: eng.rpn(feats, 800, 800))
print(f'rpn (convs + proposals) two-pass {full*1e3:.0f} us, radix {full_r*1e3:.0f} us, difference {1e3*(full_r-full):.0f} us')
