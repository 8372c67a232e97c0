import sys, os, tempfile, numpy as np, torch
from pathlib import Path
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_pipeline_e2e as T
from oracle import pipeline_ref as PR, postproc_ref as P
from deepemia_amd import synth
from deepemia_amd.engine import MaskRCNNEngine
from deepemia_amd.predictor import Predictor
from deepemia_amd.functions.inference import InferencePipeline
sds = {d: synth.random_d2_state_dict(d, 2, seed=0, mask_bias=0.5, mask_gain=6.0) for d in (50, 101)}
img = synth.em_tile(41, 512)
gcfg = {"inference_settings": {"ensemble_settings": {"weights": {"R50": 0.6, "R101": 0.4}}}}
ref = PR.RefPipeline(sds, 2, 0.3, {}, gcfg["inference_settings"], True)
preds = [Predictor(MaskRCNNEngine(sds[d], d, 2, 0.3, 'cuda:0', 'f32')) for d in (50, 101)]
pipe = InferencePipeline(preds, 'x', {}, gcfg)
x = torch.from_numpy(img).cuda()
for cls, conf, thr in ((0, 0.3, 0.65), (1, 0.35, 0.6)):
    m, s, c = ref.class_pass([0, 1], ('k', 'full'), img, cls, {0}, conf, thr)
    dets = [pipe._predict_batch(mi, 'k|full', x[None])[0] for mi in (0, 1)]
    gm, gs, gc = pipe._ensemble_class_pass(dets, cls, {0}, conf, thr)
    gd = pipe.ops.to_dense(gm, 512) if gm is not None and not isinstance(gm, str) else np.zeros((0, 512, 512), bool)
    print('class', cls, 'oracle n', len(m), 'gpu n', gd.shape[0])
    for i in range(min(len(m), gd.shape[0], 400)):
        a = np.asarray(m[i]) > 0
        iou = (a & gd[i]).sum() / max((a | gd[i]).sum(), 1)
        if iou < 0.999 or abs(s[i] - gs[i]) > 1e-5:
            print('  first diff at', i, 'iou', iou, 'scores', s[i], gs[i], 'areas', a.sum(), gd[i].sum()); break
    # raw per-model comparison
    for mi in (0, 1):
        om, os_, oc = ref.predict(mi, ('k', 'full'), img)
        d = dets[mi]
        print('   model', mi, 'n', len(os_), len(d.scores), 'score maxdiff', float(np.abs(os_ - d.scores).max()) if len(os_) == len(d.scores) else 'len differs',
              'classes equal', bool((oc == d.classes).all()) if len(oc) == len(d.classes) else None)
print('--- tile pipeline')
allm, alls, allc = [], [], []
gparts, gscores, gclasses = [], [], []
for cls, conf, thr in ((0, 0.3, 0.65), (1, 0.35, 0.6)):
    m, s, c = ref.tile_pipeline([0, 1], 'k', img, cls, {0}, conf, 512, 0.0, 1.0, thr, True)
    gm, gs, gc = pipe.tile_based_inference_pipeline([0, 1], 'k', x, cls, {0}, conf, 512, 0.0, 1.0, thr, True)
    gd = pipe.ops.to_dense(gm, 512) if gm is not None else np.zeros((0, 512, 512), bool)
    print('class', cls, 'oracle n', len(m), 'gpu n', gd.shape[0])
    for i in range(min(len(m), gd.shape[0])):
        a = np.asarray(m[i]) > 0
        iou = (a & gd[i]).sum() / max((a | gd[i]).sum(), 1)
        if iou < 0.999 or abs(s[i] - gs[i]) > 1e-5:
            print('  first diff at', i, 'iou', iou, 'scores', s[i], gs[i], 'areas', a.sum(), gd[i].sum()); break
    allm += list(m); alls += list(s); allc += list(c)
    if gm is not None: gparts.append(gm); gscores += gs; gclasses += gc
fm, fs, fc = P.deduplicate_masks_smart(allm, alls, allc, 0.7)
gp, gs2, gc2 = pipe.deduplicate_masks_smart(torch.cat(gparts), gscores, gclasses, 0.7)
print('cross-class dedup: oracle', len(fm), 'gpu', gp.shape[0])
