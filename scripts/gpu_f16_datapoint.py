"""The flagged single-plane fp16 mode (`--precision f16`: fp16 operands, f32 accumulation, one MFMA per product -- the
arithmetic of the reference's autocast predictor, inference.py:1390-1395) against the fp32 CPU oracle at the headline
configuration (R101-FPN, 2048^2 tiles, threshold 0.3), next to the default f16x2 mode on the same tiles:
instance / order agreement, score error, mask-IoU distribution, and the whole-tile CSV comparison on tile 0.
usage: gpu_f16_datapoint.py <n tiles> <out.json>"""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from deepemia_amd import synth
from deepemia_amd.engine import MaskRCNNEngine
from deepemia_amd.functions.inference import InferencePipeline
from deepemia_amd.predictor import Predictor
from oracle import maskrcnn_ref, tile_parity as TP

n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 6
out_path = sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out/f16_datapoint.json'
THR, CT, SMALL = 0.3, {0: (0.3, 0.7), 1: (0.3, 0.5)}, {1}
torch.set_num_threads(16)
sd = synth.random_d2_state_dict(101, 2, seed=0)
tiles = [synth.em_tile(i, 2048) for i in range(n_tiles)]
t0 = time.time()
refs = [maskrcnn_ref.predict(t, sd, 101, THR) for t in tiles]
ref0 = TP.reference_tile(tiles[0], sd, 101, THR, CT, SMALL)
print('oracle', time.time() - t0, flush=True)
res = {}
for prec in ('f16x2', 'f16'):
    eng = MaskRCNNEngine(sd, 101, 2, THR, 'cuda:0', prec)
    pred = Predictor(eng)
    rows, ious = [], []
    for t, r in zip(tiles, refs):
        inst = pred(t)['instances'].to('cpu')
        n, nr = len(inst), int(r['scores'].shape[0])
        same_n = n == nr
        k = min(n, nr)
        cls_same = int((inst.pred_classes[:k] == r['pred_classes'][:k]).sum())
        m, rm = inst.pred_masks, r['pred_masks']
        if k:
            # each reference instance against the product instance at the same position (order agreement) ...
            un_pos = (m[:k] | rm[:k]).sum((1, 2)).float()
            iou_pos = torch.where(un_pos > 0, (m[:k] & rm[:k]).sum((1, 2)).float() / un_pos.clamp(min=1), torch.ones_like(un_pos)).numpy()   # empty vs empty = 1
            # ... and against its best match anywhere (are the same objects found at all)
            mf, rf = m.flatten(1).float(), rm.flatten(1).float()
            inter = rf @ mf.T
            union = rf.sum(1)[:, None] + mf.sum(1)[None, :] - inter
            iou_best = torch.where(union > 0, inter / union.clamp(min=1), torch.ones_like(union)).max(1).values.numpy()
        else:
            iou_pos = iou_best = np.zeros(0)
        rows.append(dict(instances=n, instances_ref=nr, same_count=same_n, classes_same_at_position=cls_same,
                         score_max_abs_err=float((inst.scores[:k] - r['scores'][:k]).abs().max()) if k else None,
                         iou_at_position_min=float(iou_pos.min()) if k else None,
                         iou_at_position_ge_0999=int((iou_pos >= 0.999).sum()),
                         iou_best_match_min=float(iou_best.min()) if k else None,
                         iou_best_match_ge_0999=int((iou_best >= 0.999).sum()), iou_best_match_ge_099=int((iou_best >= 0.99).sum())))
        ious += iou_best.tolist()
    pipe = InferencePipeline([pred], 'dp', {}, {})
    packed, scores, classes, recs = pipe.process_tile_batch('dp', torch.from_numpy(tiles[0])[None].cuda(), SMALL, CT)[0]
    dense = pipe.ops.to_dense(packed, 2048) if packed is not None else np.zeros((0, 2048, 2048), bool)
    par = TP.compare_tile(ref0, dense, scores, classes, recs)
    q = np.asarray(ious)
    res[prec] = dict(tiles=rows, iou_best_match_quantiles={p: float(np.quantile(q, p)) for p in (0.0, 0.01, 0.05, 0.25, 0.5)},
                     masks=int(len(q)), masks_ge_0999=int((q >= 0.999).sum()), masks_ge_099=int((q >= 0.99).sum()),
                     tiles_with_identical_instance_lists=sum(1 for r_ in rows if r_['same_count'] and r_['classes_same_at_position'] == r_['instances_ref']
                                                             and r_['iou_at_position_ge_0999'] == r_['instances_ref']),
                     whole_tile_parity_tile0={k: par[k] for k in par if k != 'why'} | ({'why': par['why']} if 'why' in par else {}))
    print(prec, json.dumps(res[prec]['iou_best_match_quantiles']), res[prec]['masks_ge_0999'], '/', res[prec]['masks'],
          'identical lists on', res[prec]['tiles_with_identical_instance_lists'], 'of', n_tiles, 'tiles; tile0 ok:', par['ok'], flush=True)
    del eng, pred, pipe
    torch.cuda.empty_cache()
res['config'] = f'R101-FPN, {n_tiles} synthetic 2048^2 tiles (indices 0..{n_tiles - 1}), threshold {THR}, K=2, seeded random Detectron2-layout weights; oracle = fp32 torch-CPU restatement'
json.dump(res, open(out_path, 'w'), indent=1)
