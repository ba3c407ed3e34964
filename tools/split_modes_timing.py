"""tools/split_modes_timing.py [B]: the split kernel's three B-operand layouts against the fp32-MFMA kernels at config-3 shapes:
the pointwise contraction of the heads (W [256, 512] x [512, P*A]), its dX (W^T through a transposed copy) and the
implicit intra conv at C = O = 512 and 128."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import numpy as np
import torch
from vgtk import _hip
import vgtk.so3conv.functional as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda:0')
P, A = 4096, 60
PA = P * A


def timed(fn, flops, label, rounds=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f'   {label:34s} median {ts[len(ts) // 2]:8.3f} ms  {flops / ts[len(ts) // 2] / 1e9:7.1f} TFLOP/s', flush=True)


def with_split(flag, fn):
    def run():
        _hip.SPLIT_BF16_CONTRACTION = flag
        try:
            return fn()
        finally:
            _hip.SPLIT_BF16_CONTRACTION = True
    return run


for (O, C) in ((256, 512), (512, 256), (512, 512), (128, 128)):
    W = torch.randn(O, C, device=dev) * 0.05
    x = torch.randn(B, C, PA, device=dev)
    y = torch.empty(B, O, PA, device=dev)
    fl = 2.0 * O * C * PA * B
    print(f'pointwise contraction O={O} C={C} B={B}')
    f = lambda: _hip.gemm(0, 0, O, PA, C, W, C, 0, x, PA, C * PA, y, PA, O * PA, B)
    timed(with_split(True, f), fl, 'split (B row-major)')
    _hip.lib.eap_gemm_bf16x3_presplit(0); timed(with_split(True, f), fl, 'split, weights split in the loop'); _hip.lib.eap_gemm_bf16x3_presplit(1)
    timed(with_split(False, f), fl, 'fp32 MFMA')
    del x, y
for (O, CK) in ((512, 3072), (128, 1536)):
    W = torch.randn(O, CK, device=dev) * 0.05
    XT = torch.randn(B, PA, CK, device=dev)
    y = torch.empty(B, O, PA, device=dev)
    fl = 2.0 * O * CK * PA * B
    print(f'forward contraction of the inter conv O={O} CK={CK} B={B}')
    f = lambda: _hip.gemm(0, 1, O, PA, CK, W, CK, 0, XT, CK, PA * CK, y, PA, O * PA, B)
    timed(with_split(True, f), fl, 'split')
    _hip.lib.eap_gemm_bf16x3_presplit(0); timed(with_split(True, f), fl, 'split, weights split in the loop'); _hip.lib.eap_gemm_bf16x3_presplit(1)
    timed(with_split(False, f), fl, 'fp32 MFMA')
    del XT, y
idx = torch.from_numpy(np.ascontiguousarray(L.get_intra_idx())).to(torch.int32).to(dev)
for (O, C) in ((512, 512), (128, 128)):
    W = torch.randn(O, C * 12, device=dev) * 0.05
    feats = torch.randn(B, C, P, A, device=dev)
    fl = 2.0 * O * C * 12 * PA * B
    print(f'intra conv O={O} C={C} B={B}')
    f = lambda: _hip.so3_intra_conv(feats, W, idx)
    timed(with_split(True, f), fl, 'split (implicit gather)')
    _hip.lib.eap_gemm_bf16x3_presplit(0); timed(with_split(True, f), fl, 'split, weights split in the loop'); _hip.lib.eap_gemm_bf16x3_presplit(1)
    timed(with_split(False, f), fl, 'fp32 MFMA (implicit gather)')
    del feats
