"""tools/gemm_skinny_time.py -- the backward's GEMMs with a 64-wide side at the bench shapes (8 x 4096): layer 1's dF
(W2 [64, 3072] x Z [3072, 16800]) and dW (Z [3072, 16800] x Fc^T [16800, 64], summed over clouds), layer 0's dW
(dY [64, 245760] x X^T [245760, 24], summed over clouds)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch
from vgtk import _hip
dev = torch.device('cuda:0')
B = 8


def timed(fn, n=7):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[n // 2]


def check(C, ref, what):
    err = float((C.double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, (what, err)
    return err


ra, ok = 16800, 3072
W2 = torch.randn(64, ok, device=dev)
z = torch.randn(B, ok, ra, device=dev)
fc = torch.randn(B, 64, ra, device=dev)
out = torch.empty(B, 64, ra, device=dev)
f1 = lambda: _hip.gemm(0, 0, 64, ra, ok, W2, ok, 0, z, ra, ok * ra, out, ra, 64 * ra, B)
t = timed(f1)
e = check(out[0], W2.double() @ z[0].double(), 'dF')
print(f'layer 1 dF  [64 x 3072] x [3072 x 16800] x {B}: {t:.3f} ms = {2.0 * 64 * ra * ok * B / t / 1e9:.1f} TFLOP/s  (err {e:.1e})', flush=True)
d = torch.empty(ok, 64, device=dev)
f2 = lambda: _hip.gemm_reduce(0, 1, ok, 64, ra, z, ra, ok * ra, fc, ra, 64 * ra, d, 64, B)
t = timed(f2)
e = check(d, torch.einsum('bmk,bnk->mn', z.double(), fc.double()), 'dW layer 1')
print(f'layer 1 dW  sum_b [3072 x 16800] x [16800 x 64]: {t:.3f} ms = {2.0 * 64 * ra * ok * B / t / 1e9:.1f} TFLOP/s  (err {e:.1e})', flush=True)
pa = 245760
gy = torch.randn(B, 64, pa, device=dev)
x = torch.randn(B, 24, pa, device=dev)
gw = torch.empty(64, 24, device=dev)
f3 = lambda: _hip.gemm_reduce(0, 1, 64, 24, pa, gy, pa, 64 * pa, x, pa, 24 * pa, gw, 24, B)
t = timed(f3)
e = check(gw, torch.einsum('bmk,bnk->mn', gy.double(), x.double()), 'dW layer 0')
print(f'layer 0 dW  sum_b [64 x 245760] x [245760 x 24]: {t:.3f} ms = {(gy.numel() + x.numel()) * 4 / t / 1e6:.0f} GB/s of operands  (err {e:.1e})', flush=True)
