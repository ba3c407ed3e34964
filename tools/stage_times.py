"""tools/stage_times.py -- per-stage wall times of one backbone layer stack (dev utility)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch
import synth_clouds
import vgtk.so3conv as sptk, vgtk.spconv as zptk
from vgtk import _hip
import vgtk.cuda.grouping as G
import vgtk.so3conv.functional as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
bwd = len(sys.argv) > 3 and sys.argv[3] == 'bwd'
dev = torch.device('cuda:0')
xyz, _, pose = synth_clouds.laptop_batch(0, B, P)
xyz, pose = torch.from_numpy(xyz).to(dev), torch.from_numpy(pose).to(dev)


def timed(label, fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f'{label:40s} min {ts[0]:9.2f}  med {ts[len(ts) // 2]:9.2f}  max {ts[-1]:9.2f} ms', flush=True)
    return out


feats = torch.ones(B, 1, P, 60, device=dev)
for li, (c, o, r, s) in enumerate(synth_clouds.backbone_layers(P)):
    conv = sptk.InterSO3PoseConv(c, o, 1, 1, r, s, 64, kanchor=60, permute_modes=1).to(dev)
    if c > 1:
        feats = torch.randn(B, c, P, 60, device=dev)
    idx = timed(f'L{li} ball_query r={r:.3f}', lambda: G.ball_query(xyz, xyz, r, 64))
    mult, ident = L._group_tables(conv.anchors)
    rk = L.rotated_kernels(conv.anchors, conv.kernels)
    gx, nonident = timed(f'L{li} so3_prep', lambda: _hip.so3_prep(xyz, xyz, idx, pose, pose, conv.anchors, ident))
    X = timed(f'L{li} group_fwd C={c}', lambda: _hip.so3_inter_group_fwd(feats, idx, gx, rk, mult, s, nonident))
    Y = timed(f'L{li} gemm {o}x{c*24}', lambda: L.so3_contract(conv.basic_conv.W, X.view(B, c * 24, P * 60)))
    fl = 2.0 * o * c * 24 * P * 60 * B
    if bwd:
        gy = torch.randn_like(Y)
        Xv = X.view(B, c * 24, P * 60)
        gX = torch.empty_like(Xv)
        W = conv.basic_conv.W.detach()
        timed(f'L{li} gemm dX', lambda: _hip.gemm(1, 0, c * 24, P * 60, o, W, c * 24, 0, gy, P * 60, o * P * 60, gX, P * 60, c * 24 * P * 60, B))
        gW = torch.empty_like(W)
        timed(f'L{li} gemm dW', lambda: _hip.gemm_reduce(0, 1, o, c * 24, P * 60, gy, P * 60, o * P * 60, Xv, P * 60, c * 24 * P * 60, gW, c * 24, B))
        if c > 1:
            timed(f'L{li} group_bwd', lambda: _hip.so3_inter_group_bwd(gX.view(B, c, 24, P, 60), idx, gx, rk, mult, s, P, ident))
    del X, Y
