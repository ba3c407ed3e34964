"""tools/gpu/pass_bandwidth.py: what a plain pass over the deepest layer's feature map (8 x 512 x 4096 x 60 floats, 503 MB) costs on this
GPU -- torch copy / add / sum -- beside the path's own passes over the same tensor (BatchNorm forward / backward, dense split, re-order)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
from vgtk import _hip
import vgtk.so3conv as sptk
dev = torch.device('cuda:0')
x = torch.randn(8, 512, 4096, 60, device=dev)
y = torch.empty_like(x)
gb = x.numel() * 4 / 1e9


def timed(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, fn, passes in (('torch copy', lambda: y.copy_(x), 2), ('torch add_', lambda: y.add_(x), 3), ('torch sum', lambda: x.sum(), 1),
                         ('torch mul scalar out', lambda: torch.mul(x, 2.0, out=y), 2)):
    ms = timed(fn)
    print(f'{name:28s} {ms:6.2f} ms  {passes * gb / ms:6.2f} TB/s', flush=True)
bn = sptk.BatchNormLeakyReLU(512).to(dev)
ms = timed(lambda: bn(x))
print(f'{"BatchNormLeakyReLU forward (stats + apply: 3 passes)":28s} {ms:6.2f} ms  {3 * gb / ms:6.2f} TB/s', flush=True)
xs = x.detach().clone().requires_grad_(True)
out = bn(xs)
g = torch.randn_like(out)
ms = timed(lambda: torch.autograd.grad(out, xs, g, retain_graph=True))
print(f'{"BatchNormLeakyReLU backward (reduce + apply: 5 passes)":28s} {ms:6.2f} ms  {5 * gb / ms:6.2f} TB/s', flush=True)
ms = timed(lambda: _hip.so3_dense_split(x))
print(f'{"dense split (row maxima + split: 3 passes)":28s} {ms:6.2f} ms  {3 * gb / ms:6.2f} TB/s', flush=True)
