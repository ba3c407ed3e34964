"""tools/gpu/quick_one.py: one of bench.py's side configurations, several times (steps 3 and 8), with and without the dense probe."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
import torch, bench
import vgtk.so3conv.functional as L
dev = torch.device('cuda:0')
for mode in ('auto', 'off', 'auto'):
    L.DENSE_MODE = mode
    for steps in (3, 8):
        r = bench.quick_run(dev, 8, 4096, plan_points=512, steps=steps)
        print(mode, steps, round(r['ms_per_step'], 1), r['top_kernels_ms_per_step'], flush=True)
