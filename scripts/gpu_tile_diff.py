"""Dev script (GPU): predictor(tile) against the fp32 CPU oracle on ONE headline tile, by best match: for every instance whose mask
is not bit-identical, its area, the differing pixels and the oracle's own sampled mask probability on them (a paste-threshold tie
has |p - 0.5| ~ 1e-5).  usage: gpu_tile_diff.py <tile index>"""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from deepemia_amd import synth
from deepemia_amd.engine import MaskRCNNEngine
from deepemia_amd.predictor import Predictor
from oracle import maskrcnn_ref as R

idx = int(sys.argv[1]) if len(sys.argv) > 1 else 7
torch.set_num_threads(16)
sd = synth.random_d2_state_dict(101, 2, seed=0)
img = synth.em_tile(idx, 2048)
raw = R.predict(img, sd, 101, 0.3)
inst = Predictor(MaskRCNNEngine(sd, 101, 2, 0.3, 'cuda:0', 'f16x2'))(img)['instances'].to('cpu')
m, r = inst.pred_masks, raw['pred_masks']
if not torch.is_tensor(inst.pred_boxes):
    inst.pred_boxes = inst.pred_boxes.tensor
n = m.shape[0]
print('instances', n, raw['scores'].shape[0], 'score max abs err at position', float((inst.scores - raw['scores']).abs().max()))
order_diff = [i for i in range(n) if not torch.equal(inst.pred_boxes[i].round(), raw['pred_boxes'][i].round())]
print('positions whose box differs from the oracle box at that position:', order_diff)
for i in order_diff:
    print('  pos', i, 'score gpu %.9f' % float(inst.scores[i]), 'oracle %.9f' % float(raw['scores'][i]), 'class', int(inst.pred_classes[i]), int(raw['pred_classes'][i]))
mf, rf = m.flatten(1).float(), r.flatten(1).float()
inter = mf @ rf.T
union = mf.sum(1)[:, None] + rf.sum(1)[None, :] - inter
iou = inter / union.clamp(min=1)
best = iou.argmax(1)
for i in range(n):
    j = int(best[i])
    if torch.equal(m[i], r[j]):
        continue
    diff = m[i] != r[j]
    soft = R.paste_masks(raw['mask_probs28'][j:j + 1], raw['pred_boxes'][j:j + 1], 2048, 2048, soft=True)[0]
    print(f'instance {i} <-> oracle {j}: IoU {float(iou[i, j]):.6f}, area {int(r[j].sum())}, differing pixels {int(diff.sum())}, '
          f'max |p - 0.5| on them {float((soft[diff] - 0.5).abs().max()):.2e}, score diff {float(inst.scores[i] - raw["scores"][j]):.2e}, '
          f'box diff {float((inst.pred_boxes[i] - raw["pred_boxes"][j]).abs().max()):.2e}')
