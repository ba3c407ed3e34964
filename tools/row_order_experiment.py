"""tools/row_order_experiment.py -- does the ORDER in which the referenced support rows are handed to the backward grouping
kernel matter?  Rows that run at the same time on an XCD share its L2: the lists of spatially close rows name mostly the same
query points.  Times eap_so3_inter_group_inv_f32 (deepest layer, O = 512, 8 x 4096) with the row arrays in: the shipped
order (longest list first), Morton order of the rows' coordinates, a random order."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch
import synth_clouds
import vgtk.cuda.grouping as G
import vgtk.so3conv as sptk
import vgtk.so3conv.functional as L
from vgtk import _hip

B, P, NN, NA, KS = 8, 4096, 64, 60, 24
dev = torch.device('cuda:0')
xyz = torch.from_numpy(synth_clouds.laptop_batch(0, B, P)[0]).to(dev)
for layer in (2, 1):
    c, o, r, s = synth_clouds.backbone_layers(P)[layer]
    conv = sptk.InterSO3PoseConv(c, 8, 1, 1, r, s, NN, kanchor=NA, permute_modes=1).to(dev)
    idx = G.ball_query(xyz, xyz, r, NN)
    gx, nonident = _hip.so3_prep(xyz, xyz, idx, None, None, conv.anchors, 29)
    rk = L.rotated_kernels(conv.anchors, conv.kernels)
    rows, off, cnt, ent_p, ent_gx, rcap, _ = L._inverse_lists(idx, gx, P, 29, nonident)
    gy = torch.randn(B, o, P, NA, device=dev)

    def morton(q):                       # q int64 [b, r] row indices (>= 0) -> key
        pts = torch.gather(xyz, 2, q.unsqueeze(1).expand(-1, 3, -1))                # [b,3,r]
        lo, hi = xyz.amin(2, keepdim=True), xyz.amax(2, keepdim=True)
        u = ((pts - lo) / (hi - lo + 1e-9) * 1023).long().clamp(0, 1023)
        key = torch.zeros_like(q)
        for bit in range(10):
            for ax in range(3):
                key |= ((u[:, ax] >> bit) & 1) << (3 * bit + ax)
        return key

    def reordered(kind):
        valid = rows >= 0
        q = rows.clamp(min=0).long()
        if kind == 'shipped':
            return rows, off, cnt
        if kind == 'morton':
            key = morton(q)
        elif kind == 'x':
            key = (torch.gather(xyz[:, 0], 1, q) * 1e6).long()
        elif kind == 'index':
            key = q.clone()
        elif kind.startswith('bucket'):     # longest lists first in buckets of `width` entries, Morton order inside a bucket
            width = int(kind[6:])
            key = ((1 << 20) - (cnt.long() // width)) * (1 << 31) + morton(q)
        else:
            key = torch.randint(0, 1 << 30, q.shape, device=dev)
        key = torch.where(valid, key, torch.full_like(key, 1 << 40))
        perm = key.argsort(1)
        return (torch.gather(rows, 1, perm).contiguous(), torch.gather(off, 1, perm).contiguous(), torch.gather(cnt, 1, perm).contiguous())

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    ref = None
    fl = 2.0 * B * o * KS * P * NN * NA
    orders = {k: reordered(k) for k in ('shipped', 'bucket32', 'bucket128', 'bucket512', 'morton')}
    res = {k: [] for k in orders}
    for _ in range(6):
        for k, (r_, o_, c_) in orders.items():
            res[k].append(timed(lambda: _hip.so3_inter_group_inv(gy, r_, o_, c_, ent_p, ent_gx, rk, None, s, NN)))
    for k in orders:
        v = sorted(res[k][1:])
        print(f'layer {layer} (O = {o}, {int((rows >= 0).sum(1).max())} rows): rows in {k:10s} order: median {v[2]:7.2f} ms = {fl / v[2] / 1e9 / 157.3:.3f} of peak', flush=True)
