import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/equi-articulated-pose_amd'); sys.path.insert(0, '/root/repo/tests')
import torch, numpy as np
import test_gpu_dense as T
from vgtk import _hip
dev = torch.device('cuda:0')
s = T._setup(dev, 2, 512, layer=2)
head, geo, rp = T._geometry(s, dev)
print('rp', rp, 'n_rows', head.n_rows.tolist())
B, o, P, NA = 2, 256, 512, 60
wd = T._dense_weights64(s, head.rows, rp)
gen = torch.Generator(device=dev).manual_seed(5)
for kind in ('randn', 'rowscale', 'const', 'quant'):
    gy = torch.randn(B, o, P, NA, device=dev, generator=gen)
    if kind == 'rowscale':
        gy = gy * torch.exp(2 * torch.randn(B, o, 1, NA, device=dev, generator=gen))
    if kind == 'const':
        gy = torch.ones_like(gy) * 0.7371
    if kind == 'quant':
        gy = gy.half().float()       # l plane zero (up to scale: power of two)
    z = _hip.so3_dense_bwd(gy, geo).view(B, o, 24, NA, rp)
    ref = torch.einsum('bopa,bprak->bokar', gy.double(), wd)
    mag = torch.einsum('bopa,bprak->bokar', gy.double().abs(), wd)
    e = ((z.double() - ref).abs() / mag.clamp(min=1e-30))
    print(kind, 'max rel', float(e.max()), 'frac > 1e-6', float((e > 1e-6).double().mean()), 'median', float(e[mag > 0].median()))
    bad = (e > 1e-4).nonzero()
    print('  bad count', bad.shape[0], 'examples', bad[:8].tolist())
    if bad.shape[0]:
        for d, name in enumerate('bokar'):
            u = torch.unique(bad[:, d])
            print('   ', name, u.numel(), u[:20].tolist())
