"""A/B of the ROIAlign kernel on the boxes of a real forward (AB_LIB selects the library): 7x7 over the proposals, 14x14 over
the detections, B tiles.  usage: AB_LIB=build/ab/x.so gpu_roi_ab.py [B]"""
import os, pathlib, sys, torch
sys.path.insert(0, '.')
from deepemia_amd import _lib, synth
if os.environ.get('AB_LIB'):
    _lib.LIB_PATH = pathlib.Path(os.environ['AB_LIB']).resolve()
from deepemia_amd.engine import MaskRCNNEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
sd = synth.random_d2_state_dict(101, 2, 0)
eng = MaskRCNNEngine(sd, 101, 2, 0.3, 'cuda:0', 'f16x2')
x = synth.em_tiles_device(range(900, 900 + B), 2048, 'cuda:0')
r = eng.forward(x, keep_intermediates=True)
d = r.dbg
feats, props, pcount, det_boxes, det_count = d['feats'], d['props'], d['pcount'], d['det_boxes'], r.count
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
eng._arena_key = None
for _ in range(2):
    a = t(lambda: eng.roi_align(feats, props, pcount, 7))
    b = t(lambda: eng.roi_align(feats, det_boxes, det_count, 14))
    print(os.environ.get('AB_LIB', 'default'), f'B={B} 7x7 {a*1e3:.0f} us  14x14 {b*1e3:.0f} us', flush=True)
# checksums of the two outputs (planes as int16 words): the same for two libraries that compute the same bits
for P_, bx, ct in ((7, props, pcount), (14, det_boxes, det_count)):
    o = eng.roi_align(feats, bx, ct, P_).buf
    w = o.view(torch.int16).to(torch.int64)
    print(f'checksum {P_}x{P_}:', int(w.sum()), int((w * (torch.arange(w.numel(), device=w.device) % 8191)).sum()), flush=True)
