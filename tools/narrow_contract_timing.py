"""tools/narrow_contract_timing.py: the 1-4-row contraction (csrc/narrow_contract.hip) against the GEMM path at the config-3 shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch
import vgtk.so3conv.functional as L

dev = torch.device('cuda:0')
b, c, n = 16, 256, 4096 * 60
x = torch.randn(b, c, n, device=dev)
for o in (3, 1):
    W = torch.randn(o, c, device=dev)
    g = torch.randn(b, o, n, device=dev)
    for name, fn in (('streaming', L._NarrowContract), ('gemm path', L._Contract)):
        def fwd():
            return fn.apply(W, x)
        Wi, xi = W.clone().requires_grad_(True), x.clone().requires_grad_(True)
        y = fn.apply(Wi, xi)
        def bwd():
            return torch.autograd.grad(y, [Wi, xi], g, retain_graph=True)
        for label, f in (('forward', fwd), ('backward (dW + dx)', bwd)):
            f(); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); f(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            gb = x.numel() * 4 / 1e9 * (1 if label == 'forward' else 2)
            print(f'o={o} {name:10s} {label:20s} {ts[2]:7.3f} ms  ({gb / ts[2] * 1e3:6.0f} GB/s of the x-sized streams)', flush=True)
        del y, Wi, xi
