"""Dev script (GPU, run under rocprofv3): launch ONE conv layer shape a few times so that per-dispatch counters can be read.
usage: gpu_profile_layer.py <big3x3|res4c2|res4c3|res4c1|res2c3|lateral|fc> <tile hint> <launches>"""
import os
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
os.chdir(ROOT)
sys.path.insert(0, str(ROOT))
import importlib.util
spec = importlib.util.spec_from_file_location('chk', str(ROOT / 'scripts' / 'gpu_conv_p32_check.py'))
argv = sys.argv
sys.argv = [argv[0], 'none']
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)
LAYERS = {'big3x3': ((16, 200, 200), 256, 256, 3, 0), 'res4c2': ((16, 50, 50), 256, 256, 3, 0), 'res4c3': ((16, 50, 50), 256, 1024, 1, 1),
          'res4c1': ((16, 50, 50), 1024, 256, 1, 0), 'res2c3': ((16, 200, 200), 64, 256, 1, 1), 'lateral': ((16, 200, 200), 256, 256, 1, 1),
          'fc': ((1, 1, 16000), 12544, 1024, 1, 0)}
name, hint, reps = argv[1], int(argv[2]), int(argv[3])
(n, h, w), cin, cout, k, rs = LAYERS[name]
L = chk.Layer(cout, cin, k, k, seed=1)
xp = chk.p32.from_f32(torch.randn(n, h, w, cin, device=chk.dev))
res = chk.p32.from_f32(torch.randn(n, h, w, cout, device=chk.dev)) if rs else None
for _ in range(reps):
    chk.conv_p32(xp, L, 1, k // 2, chk.ACT_RELU, res, chk.RES_SAME if rs else chk.RES_NONE, hint=hint)
torch.cuda.synchronize()
print('done', name, hint, reps)
