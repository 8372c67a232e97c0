"""A/B of the two stem kernels at B tiles of 2048^2 (800 x 800 network input): VALU f32 stem vs the MFMA stem."""
import sys, torch
sys.path.insert(0, '.')
from deepemia_amd import _lib, synth
import os, pathlib
if os.environ.get('AB_LIB'): _lib.LIB_PATH = pathlib.Path(os.environ['AB_LIB']).resolve()
from deepemia_amd.engine import MaskRCNNEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
eng = MaskRCNNEngine(synth.random_d2_state_dict(101, 2, 0), 101, 2, 0.3, 'cuda:0', 'f16x2')
x = synth.em_tiles_device(range(700, 700 + B), 2048, 'cuda:0')
xin, newh, neww, ph, pw = eng.preprocess(x)
eng._arena_key = None
out = torch.empty((B, ph // 2, pw // 2, 64), dtype=torch.float32, device='cuda:0')
st = int(torch.cuda.current_stream().cuda_stream)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
a = t(lambda: _lib.check(eng.lib.demia_stem_conv(_lib.ptr(xin), _lib.ptr(eng.stem_w), _lib.ptr(eng.stem_scale), _lib.ptr(eng.stem_bias), _lib.ptr(out), B, ph, pw, _lib.F32, st), 'a'))
b = t(lambda: _lib.check(eng.lib.demia_stem_conv_mfma(_lib.ptr(xin), _lib.ptr(eng.stem_planes), _lib.ptr(eng.stem_scale_mfma), _lib.ptr(eng.stem_bias), _lib.ptr(out), B, ph, pw, eng.stem_s_in, st), 'b'))
from deepemia_amd import p32
xp = p32.alloc((B, ph // 4, pw // 4, 64), 'cuda:0', groups=B)
s_out = p32.plane_scale(eng.stem_bound)
c = t(lambda: _lib.check(eng.lib.demia_maxpool3x3s2_p32(_lib.ptr(out), _lib.ptr(xp.buf), _lib.ptr(xp.meta), s_out, B, ph // 2, pw // 2, 64, B, 0, st), 'c'))
d = t(lambda: _lib.check(eng.lib.demia_stem_pool_mfma(_lib.ptr(xin), _lib.ptr(eng.stem_planes), _lib.ptr(eng.stem_scale_mfma), _lib.ptr(eng.stem_bias), _lib.ptr(xp.buf), _lib.ptr(xp.meta), B, ph, pw, eng.stem_s_in, s_out, B, 0, st), 'd'))
print(f'max pool (f32 -> P32) {c*1e3:.0f} us; fused stem + pool {d*1e3:.0f} us')
print(f'B={B}: VALU stem {a*1e3:.0f} us, MFMA stem {b*1e3:.0f} us ({out.numel()*4/1e9:.2f} GB written)')
