"""Dev script (GPU): the 4-pixel vertical resize kernel against the plain one (DEMIA_RESIZE_PLAIN=1):
same bytes on batches of odd sizes, and the time of preprocess() on 48 tiles of 2048^2."""
import os, sys, torch
sys.path.insert(0, '.')
from deepemia_amd import synth, _lib
import pathlib
if os.environ.get('AB_LIB'): _lib.LIB_PATH = pathlib.Path(os.environ['AB_LIB']).resolve()
from deepemia_amd.engine import MaskRCNNEngine
sd = synth.random_d2_state_dict(50, 2, 0)
eng = MaskRCNNEngine(sd, 50, 2, 0.3, 'cuda:0', 'f16x2')
g = torch.Generator().manual_seed(0)
for (b, h, w) in ((3, 2048, 2048), (5, 601, 1001), (2, 1000, 2000), (7, 333, 517), (4, 1024, 1024)):
    x = torch.randint(0, 256, (b, h, w, 3), generator=g, dtype=torch.uint8).cuda()
    os.environ.pop('DEMIA_RESIZE_PLAIN', None)
    a = eng.preprocess(x)[0].clone()
    os.environ['DEMIA_RESIZE_PLAIN'] = '1'
    c = eng.preprocess(x)[0].clone()
    assert torch.equal(a, c), (b, h, w)
    print('equal', b, h, w, flush=True)
x = synth.em_tiles_device(range(900, 948), 2048, 'cuda:0')
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
os.environ.pop('DEMIA_RESIZE_PLAIN', None)
new = t(lambda: eng.preprocess(x))
os.environ['DEMIA_RESIZE_PLAIN'] = '1'
old = t(lambda: eng.preprocess(x))
print(f'preprocess of 48 tiles: {new*1e3:.0f} us, plain vertical kernel {old*1e3:.0f} us')
