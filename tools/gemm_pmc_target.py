"""tools/gemm_pmc_target.py impl -- ONE implementation of the deepest layer's forward contraction (B = 4), three
launches, for rocprofv3 --pmc passes.  impl in {dma, old, lib}."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch
from vgtk import _hip
impl = sys.argv[1]
B, O, CK, PA = 4, 512, 3072, 4096 * 60
dev = torch.device('cuda:0')
W = torch.randn(O, CK, device=dev)
XT = torch.randn(B, PA, CK, device=dev)
Y = torch.empty(B, O, PA, device=dev)
_hip.USE_DMA_GEMM = impl == 'dma'
for _ in range(3):
    if impl == 'lib':
        torch.matmul(W, XT.transpose(1, 2), out=Y)
    else:
        _hip.gemm(0, 1, O, PA, CK, W, CK, 0, XT, CK, CK * PA, Y, PA, O * PA, B)
torch.cuda.synchronize()
