"""tools/backward_mode_ab.py: the default bench step with the feature-gradient route forced to the re-associated (inverse
lists) or the textbook (dX = W^T dY, transposed grouping) form -- the 'auto' rule was calibrated before the split contraction."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
import vgtk.so3conv.functional as L

dev = torch.device('cuda:0')
for mode in ('auto', 'inverse', 'dx', 'auto'):
    L.BACKWARD_MODE = mode
    r = bench.quick_run(dev, 8, 4096)
    print(mode, json.dumps({k: r[k] for k in ('value', 'ms_per_step', 'top_kernels_ms_per_step')}), flush=True)
