import sys, os, tempfile, numpy as np, torch
from pathlib import Path
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
os.environ['DEEPEMIA_OFFLINE'] = '1'
import test_gpu_pipeline_e2e as T
from oracle import pipeline_ref as PR
root = Path(tempfile.mkdtemp())
cfgdir, split, sds, images = T._write_tree(root, [50, 101], 2.0, 1.0, 2, 512, {"inference_overrides": {}})
names = list(os.listdir(root / "DATASET" / "INFERENCE")); print(names)
ref = PR.RefPipeline(sds, 2, 0.3, {}, {}, True)
sizes = {}
for key in names:
    m, s, c = ref.predict(0, (key, 'full'), images[key])
    for mask, cls in zip(m[s >= 0.7], c[s >= 0.7]): sizes.setdefault(int(cls), []).append(int(mask.sum()))
print('oracle', {k: (len(v), float(np.mean(v))) for k, v in sizes.items()}, ref.small_classes([(n, images[n]) for n in names]))
from deepemia_amd.engine import MaskRCNNEngine
from deepemia_amd.predictor import Predictor
from deepemia_amd.functions.inference import InferencePipeline, determine_small_classes
eng = MaskRCNNEngine(sds[50], 50, 2, 0.3, 'cuda:0', 'f32')
pipe = InferencePipeline([Predictor(eng)], 'x', {}, {})
sample = [(n, torch.from_numpy(images[n]).cuda()) for n in names]
avg = pipe.calculate_average_mask_sizes(sample); print('gpu', avg, determine_small_classes(avg))
# via imread
from deepemia_amd.functions.inference import imread_bgr
for n in names:
    a = imread_bgr(str(root / "DATASET" / "INFERENCE" / n)); print(n, 'imread equal', (a == images[n]).all())
