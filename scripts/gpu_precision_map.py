"""Per-stage precision map: the default f16x2 forward with ONE stage group at a time on single-plane
arithmetic (one fp16 MFMA per product instead of three; `MaskRCNNEngine(single_stages=...)`), each against the fp32 CPU
oracle on the eight headline tiles (instance-list agreement, order swaps, mask-IoU distribution) and timed predictor-only at
the bench's batch.  usage: gpu_precision_map.py <out.json> [batch]"""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from deepemia_amd import synth
from deepemia_amd.engine import MaskRCNNEngine
from oracle import maskrcnn_ref
import test_gpu_multitile_parity as T

out_path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/precision_map.json'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 48
torch.set_num_threads(16)
sd = synth.random_d2_state_dict(101, 2, seed=0)
tiles = [synth.em_tile(i, 2048) for i in range(T.TILES)]
refs = [maskrcnn_ref.predict(t, sd, 101, T.THR) for t in tiles]
xb = torch.from_numpy(np.stack([tiles[i % len(tiles)] for i in range(B)])).cuda()
GROUPS = [(), ('rpn_conv',), ('rpn_pred',), ('fpn_output',), ('fpn_lateral',), ('fc1', 'fc2', 'box_pred'), ('mask_fcn', 'deconv'),
          ('res2',), ('res3',), ('res4',), ('res5',)]
res = []
for g in GROUPS:
    tag = 'map_' + ('+'.join(g) if g else 'none')
    s = T.run_precision(sd, tiles, refs, 'f16x2', 'cuda:0', single_stages=g, tag=tag)
    eng = MaskRCNNEngine(sd, 101, 2, T.THR, 'cuda:0', 'f16x2', single_stages=g)
    for _ in range(3):
        eng.forward_graphed(xb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 6
    for _ in range(K):
        eng.forward_graphed(xb)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    del eng
    torch.cuda.empty_cache()
    bad = [r for r in s['per_tile'] if not r['bijection']]
    row = dict(single_stages=list(g), tiles_with_oracle_instance_set=sum(1 for r in s['per_tile'] if r['bijection']),
               tiles_in_oracle_order=s['tiles_in_oracle_order'], order_gap_max=s['order_gap_max'], score_max_abs_err=s['score_max_abs_err'],
               masks=s['masks'], masks_ge_0999=s['masks_ge_0999'], masks_identical=s['masks_identical'], iou_min=s['iou_min'],
               tie_dist_max=s['tie_dist_max'], holds_the_bar=bool(not bad and s['order_gap_max'] <= 2e-6 and s['tie_dist_max'] <= 3e-4 and
                                                                  s['masks_ge_0999'] >= int(np.ceil(0.998 * s['masks']))),
               predictor_only_tiles_per_s=B / dt, forward_ms=dt * 1e3, batch=B)
    res.append(row)
    print(json.dumps(row), flush=True)
    json.dump(dict(config=s['config'], rows=res), open(out_path, 'w'), indent=1)
