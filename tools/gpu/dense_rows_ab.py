"""tools/gpu/dense_rows_ab.py: the dense product's block variants against each other (eap_so3_dense_block_rows: 0 = default, 128 = 128-row
blocks; 256 = eight waves on 256-row blocks, only with tools/experiments/so3_dense_eight_waves.patch applied): outputs bit-equal?  times of both directions at the bench shape (layer 2, O = 512)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import numpy as np
import torch
import synth_clouds
import vgtk.so3conv.functional as L
import vgtk.cuda.grouping as cuda_nn
from vgtk import _hip
dev = torch.device('cuda:0')
B, P, NN, NA, KS = int(os.environ.get('B', 8)), int(os.environ.get('P', 4096)), 64, 60, 24
layer = int(os.environ.get('LAYER', 2))
c, o, radius, sigma = synth_clouds.backbone_layers(4096)[layer]
o = int(os.environ.get('O', o))
xyz = torch.from_numpy(synth_clouds.laptop_batch(0, B, P)[0]).to(dev).contiguous()
anchors = torch.from_numpy(np.asarray(L.get_anchors(NA), dtype=np.float32)).to(dev)
kernels = torch.from_numpy(L.get_sphereical_kernel_points_from_ply(0.7 * radius, 1)).to(dev)
rk = L.rotated_kernels(anchors, kernels)
idx = cuda_nn.ball_query(xyz, xyz, radius, NN)
head = L._ListHead(idx, P, None, None, dense_probe=(None, None))
rcap, _ = head.decide()
rp = L._dense_rows(rcap, P)
head.wait()
geo = _hip.DenseGeometry(xyz, xyz, head.memb, head.rows, rp, rk, sigma, NN, head.n_rows)
gen = torch.Generator(device=dev).manual_seed(1)
gy = torch.randn(B, o, P, NA, device=dev, generator=gen)
g = torch.randn(B, o, KS, rp * NA, device=dev, generator=gen)


def timed(fn, n=5):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), r


ref = None
for rows in [int(r) for r in os.environ.get('ROWS', '0,128').split(',')]:
    _hip.lib.eap_so3_dense_block_rows(rows)
    tz, z = timed(lambda: _hip.so3_dense_bwd(gy, geo))
    ty, y = timed(lambda: _hip.so3_dense_fwd(g.view(B, o, KS, rp * NA), geo, P))
    if ref is None:
        ref = (z, y)
    print(f'block rows {rows}: backward (split + product) {tz:.2f} ms, forward (split + product + re-order) {ty:.2f} ms; '
          f'Z equal to the first: {torch.equal(z, ref[0])}, max diff {float((z - ref[0]).abs().max()):.3g}; Y equal: {torch.equal(y, ref[1])}, max diff {float((y - ref[1]).abs().max()):.3g}', flush=True)
_hip.lib.eap_so3_dense_block_rows(0)
