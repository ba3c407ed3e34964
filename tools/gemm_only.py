"""tools/gemm_only.py -- the contraction GEMMs in isolation at the benchmark's shapes (B clouds of 4096 points),
every implementation side by side: csrc/gemm_dma_f32.hip (DMA ring), csrc/gemm_f32.hip (register staging) and the
vendor library (torch.matmul -> hipBLASLt) as the yardstick.  Interleaved rounds, median of the rounds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch
from vgtk import _hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda:0')
PA = 4096 * 60


def bench(fns, flops, rounds=5):
    res = {k: [] for k in fns}
    for k, fn in fns.items():
        fn()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, fn in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1))
    for k, v in res.items():
        v.sort()
        print(f'   {k:28s} median {v[len(v) // 2]:8.3f} ms  min {v[0]:8.3f} ms  {flops / v[len(v) // 2] / 1e9:7.1f} TFLOP/s', flush=True)


def old(*a, **k):
    _hip.USE_DMA_GEMM = False
    _hip.SPLIT_BF16_CONTRACTION = False
    try:
        _hip.gemm(*a, **k)
    finally:
        _hip.USE_DMA_GEMM = True
        _hip.SPLIT_BF16_CONTRACTION = True


def old_reduce(*a, **k):
    _hip.USE_DMA_GEMM = False
    try:
        _hip.gemm_reduce(*a, **k)
    finally:
        _hip.USE_DMA_GEMM = True


ONLY_FIRST = len(sys.argv) > 2 and sys.argv[2] == 'first'
for (O, CK) in ((512, 3072),) if ONLY_FIRST else ((512, 3072), (128, 1536)):
    W = torch.randn(O, CK, device=dev)
    XT = torch.randn(B, PA, CK, device=dev)          # the transposed intermediate [P*A, C*K]
    Y = torch.empty(B, O, PA, device=dev)
    print(f'forward contraction Y[{O} x {PA}] = W[{O} x {CK}] . XT^T, batch {B}')
    def fp32_ring():
        _hip.SPLIT_BF16_CONTRACTION = False
        try:
            _hip.gemm(0, 1, O, PA, CK, W, CK, 0, XT, CK, CK * PA, Y, PA, O * PA, B)
        finally:
            _hip.SPLIT_BF16_CONTRACTION = True
    bench({'gemm_bf16x3 (3 x bf16 split)': lambda: _hip.gemm(0, 1, O, PA, CK, W, CK, 0, XT, CK, CK * PA, Y, PA, O * PA, B),
           'gemm_dma_f32 (TN)': fp32_ring,
           'gemm_f32 (TN, reg. staging)': lambda: old(0, 1, O, PA, CK, W, CK, 0, XT, CK, CK * PA, Y, PA, O * PA, B),
           'hipBLASLt (torch.matmul)': lambda: torch.matmul(W, XT.transpose(1, 2), out=Y)}, 2.0 * O * CK * PA * B)
    del XT, Y
if ONLY_FIRST:
    sys.exit(0)
# the re-associated backward's small GEMMs (deepest layer: O = 512, C = 128, R = 136 referenced rows)
O, C, KS, RA = 512, 128, 24, 136 * 60
Z = torch.randn(B, O * KS, RA, device=dev)
W2 = torch.randn(C, O * KS, device=dev)
gF = torch.empty(B, C, RA, device=dev)
print(f'dF rows [{C} x {RA}] = W2[{C} x {O * KS}] . Z, batch {B}')
bench({'gemm_dma_f32 (NN)': lambda: _hip.gemm(0, 0, C, RA, O * KS, W2, O * KS, 0, Z, RA, O * KS * RA, gF, RA, C * RA, B),
       'gemm_f32': lambda: old(0, 0, C, RA, O * KS, W2, O * KS, 0, Z, RA, O * KS * RA, gF, RA, C * RA, B),
       'hipBLASLt': lambda: torch.matmul(W2, Z, out=gF)}, 2.0 * C * RA * O * KS * B)
Fc = torch.randn(B, C, RA, device=dev)
d = torch.empty(O * KS, C, device=dev)
print(f'dW [{O * KS} x {C}] = sum_b Z_b . Fc_b^T (K = {RA}), batch {B}')
bench({'gemm_dma_f32 reduce': lambda: _hip.gemm_reduce(0, 1, O * KS, C, RA, Z, RA, O * KS * RA, Fc, RA, C * RA, d, C, B),
       'gemm_f32 reduce': lambda: old_reduce(0, 1, O * KS, C, RA, Z, RA, O * KS * RA, Fc, RA, C * RA, d, C, B),
       'hipBLASLt + sum': lambda: torch.matmul(Z, Fc.transpose(1, 2)).sum(0)}, 2.0 * C * RA * O * KS * B)
