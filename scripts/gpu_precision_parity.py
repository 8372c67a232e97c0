"""Dev script: predictor parity of a precision mode against the CPU oracle (R50, 1024^2 tile)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepemia_amd import synth
from deepemia_amd.engine import MaskRCNNEngine
from deepemia_amd.predictor import Predictor
from oracle import maskrcnn_ref as R
torch.set_num_threads(16)
prec = sys.argv[1]
sd = synth.random_d2_state_dict(50, 2, seed=0)
for seed in (0, 3):
    img = synth.em_tile(seed, 1024)
    ref = R.predict(img, sd, 50, 0.3, return_intermediates=True)
    eng = MaskRCNNEngine(sd, 50, 2, 0.3, "cuda:0", prec)
    x = torch.from_numpy(img)[None].cuda()
    xin, newh, neww, ph, pw = eng.preprocess(x)
    feats = eng.backbone(xin, ph, pw)
    for k in ("res2", "res5", "p2", "p6"):
        a = feats[k][0].permute(2, 0, 1).float().cpu(); b = ref["dbg"]["feats"][k][0]
        print(prec, seed, k, "rel err %.2e" % float((a - b).abs().max() / b.abs().max()))
    inst = Predictor(eng)(img)["instances"].to("cpu")
    n, nr = len(inst), ref["scores"].shape[0]
    print("  det count", n, nr)
    if n == nr:
        same_cls = int((inst.pred_classes == ref["pred_classes"]).sum())
        m, r = inst.pred_masks, ref["pred_masks"]
        iou = (m & r).sum((1, 2)).float() / (m | r).sum((1, 2)).float().clamp(min=1)
        print("  classes equal %d / %d; score max diff %.2e; box max diff %.3f; mask IoU min %.5f mean %.6f, < 0.999: %d"
              % (same_cls, n, float((inst.scores - ref["scores"]).abs().max()), float((inst.pred_boxes - ref["pred_boxes"]).abs().max()),
                 float(iou.min()), float(iou.mean()), int((iou < 0.999).sum())))
