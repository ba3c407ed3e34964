"""tools/gemm_only.py -- the dominant kernel in isolation (L2-layer contraction, B clouds of 4096
points): used for rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE) and tile-shape experiments."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch
from vgtk import _hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
O, CK, PA = 512, 3072, 4096 * 60
dev = torch.device('cuda:0')
W = torch.randn(O, CK, device=dev)
X = torch.randn(B, CK, PA, device=dev)
Y = torch.empty(B, O, PA, device=dev)
reps = 3
for name, fn in (
    ('fwd  Y = W X', lambda: _hip.gemm(0, 0, O, PA, CK, W, CK, 0, X, PA, CK * PA, Y, PA, O * PA, B)),
    ('bwd dX = W^T dY', lambda: _hip.gemm(1, 0, CK, PA, O, W, CK, 0, Y, PA, O * PA, X, PA, CK * PA, B)),
):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'{name}: {ms:.2f} ms  {2.0 * O * CK * PA * B / ms / 1e9:.1f} TFLOP/s', flush=True)
