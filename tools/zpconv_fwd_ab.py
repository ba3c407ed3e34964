"""tools/zpconv_fwd_ab.py: the native zpconv forward at the bench shape with the two matrix kernels (eap_inter_zpconv_fwd_kernel)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from vgtk import _hip
dev = torch.device('cuda:0')
for which in (2, 1, 2, 1):
    _hip.lib.eap_inter_zpconv_fwd_kernel(which)
    r = bench.zpconv_roofline(dev, 4096)
    print(f'kernel {which}: forward {r["ms"]:.2f} ms = {r["frac"]:.3f} of the HBM roofline (algorithmic bytes)', flush=True)
_hip.lib.eap_inter_zpconv_fwd_kernel(1)
