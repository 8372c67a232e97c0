"""Dev script (GPU box): stage-by-stage diff of the HIP engine against the CPU oracle."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from deepemia_amd import synth
from deepemia_amd.engine import MaskRCNNEngine
from oracle import maskrcnn_ref as R

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
prec = sys.argv[2] if len(sys.argv) > 2 else 'f32'
size = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
img = synth.em_tile(0, size)
sd = synth.random_d2_state_dict(depth, 2, 0)
t = time.time(); ref = R.predict(img, sd, depth, 0.3, True); print('oracle s', time.time() - t)
d = ref['dbg']
eng = MaskRCNNEngine(sd, depth, 2, 0.3, 'cuda:0', prec)
print('unmatched', eng.unmatched_keys)
x = torch.from_numpy(img)[None].cuda()
out = eng.forward(x, keep_intermediates=True); torch.cuda.synchronize()
g = out.dbg
def rel(a, b):
    a = a.float().cpu(); b = b.float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12)), float((a - b).abs().mean() / (b.abs().mean() + 1e-12))
xin = g['xin'][0, 3:3 + d['xin'].shape[2], 3:3 + d['xin'].shape[3], :3].permute(2, 0, 1).cpu()
print('xin maxabs diff', float((xin - d['xin'][0]).abs().max()))
for k in ['res2', 'res3', 'res4', 'res5', 'p5', 'p4', 'p3', 'p2', 'p6']:
    print(k, rel(g['feats'][k][0].permute(2, 0, 1), d['feats'][k][0]))
for li in range(5):
    hd = g['heads'][li][0].cpu()  # H,W,16
    lg = hd[..., :3].reshape(-1)
    dl = hd[..., 3:15].reshape(-1, 4)
    print('rpn lvl', li, rel(lg, d['rpn']['per_level'][li]['logits']), rel(dl, d['rpn']['per_level'][li]['deltas']))
pc = int(g['pcount'][0]); print('prop count', pc, d['prop_boxes'].shape[0])
m = min(pc, d['prop_boxes'].shape[0])
pb = g['props'][0, :m].cpu(); ob = d['prop_boxes'][:m]
same = (pb - ob).abs().max(dim=1).values < 1e-2
print('props rows equal', int(same.sum()), '/', m, 'score maxdiff', float((g['pscores'][0, :m].cpu() - d['prop_scores'][:m]).abs().max()))
print('pooled', rel(g['pooled'][0, :m].permute(0, 3, 1, 2), d['pooled'][:m]) if bool(same.all()) else 'skipped (props differ)')
K = 2
lg = g['logits'][0, :m].cpu()
if bool(same.all()):
    print('cls logits', rel(lg[:, :K + 1], d['cls_logits'][:m]), 'deltas', rel(lg[:, K + 1:K + 1 + 4 * K], d['deltas'][:m]))
dc = int(out.count[0]); print('det count', dc, ref['scores'].shape[0])
mm = min(dc, ref['scores'].shape[0])
print('det scores maxdiff', float((out.scores[0, :mm].cpu() - ref['scores'][:mm]).abs().max()),
      'classes equal', int((out.classes[0, :mm].cpu() == ref['pred_classes'][:mm]).sum()), '/', mm)
print('det boxes maxdiff', float((out.boxes[0, :mm].cpu() - ref['pred_boxes'][:mm]).abs().max()))
masks = eng.unpack(out.packed[0, :mm].contiguous(), size, size).cpu()
inter = (masks & ref['pred_masks'][:mm]).sum((1, 2)).float(); uni = (masks | ref['pred_masks'][:mm]).sum((1, 2)).float()
iou = inter / uni.clamp(min=1)
print('mask IoU min/mean', float(iou.min()), float(iou.mean()), 'n<0.999', int((iou < 0.999).sum()))
area, bbox = eng.area_bbox(out.packed[0, :mm].contiguous(), size, size)
print('area match', bool((area.cpu() == masks.sum((1, 2)).int()).all()))
# timing
for _ in range(2): eng.forward(x)
torch.cuda.synchronize(); t = time.time()
for _ in range(5): eng.forward(x)
torch.cuda.synchronize(); print('ms/forward (B=1)', (time.time() - t) / 5 * 1e3)
