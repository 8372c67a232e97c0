"""Dev script: run the e2e case-1 pipeline on the GPU and cross-check contours/measurements of ITS OWN final masks vs the oracle."""
import sys, os, json, tempfile, numpy as np, torch, yaml
from pathlib import Path
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
os.environ['DEEPEMIA_OFFLINE'] = '1'
import test_gpu_pipeline_e2e as T
from oracle import postproc_ref as P
root = Path(tempfile.mkdtemp())
spatial = {"enabled": True, "containment_rules": {1: 0}, "containment_threshold": 0.5, "overlap_rules": {0: {"allow_overlap": False, "max_iou_threshold": 0.3}, 1: {"allow_overlap": False, "max_iou_threshold": 0.5}}}
tile = {"tile_size": 256, "overlap_ratio": 0.125, "upscale_factor": 2.0, "edge_filter_enabled": True}
ds_cfg = {"inference_overrides": {"confidence_mode": "manual", "class_specific_settings": {"class_0": {"confidence_threshold": 0.3, "iou_threshold": 0.6, "min_size": 25}, "class_1": {"confidence_threshold": 0.35, "iou_threshold": 0.5, "min_size": 5}}, "tile_settings": tile, "spatial_constraints": spatial}}
cfgdir, split, sds, images = T._write_tree(root, [50], 0.5, 6.0, 2, 512, ds_cfg)
os.environ['DEEPEMIA_CONFIG_DIR'] = str(cfgdir)
from deepemia_amd.functions.inference import run_inference
from deepemia_amd.maskset import MaskOps
res = run_inference(T.DATASET, str(split), threshold=0.3)
ops = MaskOps('cuda:0')
for name, d in res.items():
    packed = d['masks']; W = d['hw'][1]
    dense = ops.to_dense(packed, W)
    recs = ops.contours(packed, max_contours=256)
    for i in range(dense.shape[0]):
        ref = P.find_external_contours(dense[i])
        if len(ref) != len(recs[i]):
            print(name, i + 1, 'COUNT', len(ref), len(recs[i])); continue
        for rec, c in zip(recs[i], ref):
            if rec['points'].shape != c.shape or not (rec['points'] == c).all():
                print(name, i + 1, 'POINTS differ', len(c), len(rec['points']), c.tolist()[:40], rec['points'].tolist()[:40]); continue
            exp = P.calculate_measurements(c)
            v = rec['values']
            if abs(v[0] - exp['major_axis_length']) > 1e-6 * max(1, exp['major_axis_length']) and not exp['_ellipse_unstable']:
                print(name, i + 1, 'MAJOR', v[0], exp['major_axis_length'], 'n', len(c), 'area', rec['area'], c.tolist())
print('done')
from oracle import pipeline_ref as PR
inf = dict(ds_cfg["inference_overrides"])
glob_inf = {"ensemble_settings": {"enabled": True, "small_classes_only": False, "weights": {"R50": 0.6, "R101": 0.4}}}
ref = PR.RefPipeline(sds, 2, 0.3, inf, glob_inf, True)
names = list(os.listdir(root / "DATASET" / "INFERENCE"))
small = ref.small_classes([(n, images[n]) for n in names])
print('oracle small', small)
for n in names:
    m, s, c = ref.run_image(n, images[n], small, "manual", spatial, True, False)
    d = res[n]
    dense = ops.to_dense(d['masks'], d['hw'][1])
    print(n, 'counts', len(m), dense.shape[0])
    for i in range(min(len(m), dense.shape[0])):
        a = np.asarray(m[i]) > 0; b = dense[i]
        iou = (a & b).sum() / max((a | b).sum(), 1)
        if iou < 0.999 or c[i] != d['classes'][i] or abs(s[i] - d['scores'][i]) > 1e-4:
            ys, xs = np.nonzero(a); yb, xb = np.nonzero(b)
            print('  inst', i + 1, 'iou', iou, 'cls', c[i], d['classes'][i], 'score', s[i], d['scores'][i], 'area', a.sum(), b.sum(),
                  'bbox ref', (ys.min(), xs.min(), ys.max(), xs.max()) if len(ys) else None, 'gpu', (yb.min(), xb.min(), yb.max(), xb.max()) if len(yb) else None)
