"""Dev script: the post-processing kernels in isolation on the bench workload's masks (one 8-tile batch)."""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepemia_amd import synth
from deepemia_amd.engine import MaskRCNNEngine
from deepemia_amd.maskset import MaskOps
eng = MaskRCNNEngine(synth.random_d2_state_dict(101, 2, 0), 101, 2, 0.3, 'cuda:0', 'f32x3')
ops = MaskOps('cuda:0'); ops.set_frame_width(2048)
x = torch.from_numpy(np.stack([synth.em_tile(i, 2048) for i in range(8)])).cuda()
out = eng.forward(x)
packed0 = out.packed.view(-1, 2048, 64).contiguous(); hint = out.bbox.view(-1, 4).contiguous()
M = packed0.shape[0]
def timeit(name, fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(f'{name:40s} {np.median(ts)*1e3:9.1f} us  (M={M})')
p = packed0.clone()
area, bbox = ops.area_bbox(p, hint)
timeit('area_bbox hinted', lambda: ops.area_bbox(p, hint))
timeit('area_bbox full', lambda: ops.area_bbox(p))
timeit('clone', lambda: packed0.clone())
work = packed0.clone()
def prog(st):
    work.copy_(packed0); return None
timeit('copy_ (baseline for programs)', lambda: work.copy_(packed0))
for st in (['fill'], ['dilate', 'erode'], ['fill', 'dilate', 'erode'], ['flag_multi'], ['drop_multi', 'fill', 'erode', 'dilate']):
    timeit('copy_+program ' + ','.join(st), lambda st=st: (work.copy_(packed0), ops.program_(work, st, bbox)))
seg = torch.from_numpy(np.repeat(np.arange(8, dtype=np.int32), M // 8)).cuda()
timeit('copy_+overlap_prefix bbox', lambda: (work.copy_(packed0), ops.overlap_prefix_(work, seg, bbox)))
timeit('column_counts', lambda: ops.column_counts(p, seg, 8, bbox=bbox))
tot = int(area.sum().item())
timeit('trace', lambda: ops.trace(p, max_contours=256, bbox=bbox, total_area=tot))
cs = ops.trace(p, max_contours=256, bbox=bbox, total_area=tot)
cnt, info, red, used = cs.host()
print('contours', int(cnt.sum()), 'points used', used, 'max npts', int(info[:, :, 2].max()), 'masks with >1 contour', int((cnt > 1).sum()))
timeit('measure', lambda: cs.measure(1.0))
