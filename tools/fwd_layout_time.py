"""tools/fwd_layout_time.py -- forward grouping kernel time per output layout + the contraction that follows
(own GEMM on the reference / blocked layouts, library GEMM on the transposed one)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch, synth_clouds
import vgtk.so3conv as sptk
from vgtk import _hip
import vgtk.cuda.grouping as G
import vgtk.so3conv.functional as L
B, P = 8, 4096
dev = torch.device('cuda:0')
xyz, _, pose = synth_clouds.laptop_batch(0, B, P)
xyz, pose = torch.from_numpy(xyz).to(dev), torch.from_numpy(pose).to(dev)
def t_ms(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for li in (1, 2):
    c, o, r, s = synth_clouds.backbone_layers(P)[li]
    conv = sptk.InterSO3PoseConv(c, o, 1, 1, r, s, 64, kanchor=60, permute_modes=1).to(dev)
    feats = torch.randn(B, c, P, 60, device=dev)
    idx = G.ball_query(xyz, xyz, r, 64)
    mult, ident = L._group_tables(conv.anchors)
    rk = L.rotated_kernels(conv.anchors, conv.kernels)
    gx, nonident = _hip.so3_prep(xyz, xyz, idx, pose, pose, conv.anchors, ident)
    W = conv.basic_conv.W.detach()
    ck, pa = c * 24, P * 60
    xs = {}
    for lay in (0, 1, 2):
        ms = t_ms(lambda: xs.__setitem__(lay, _hip.so3_inter_group_fwd(feats, idx, gx, rk, mult, s, nonident, blocked=lay)))
        print(f'layer {li} (C={c}): grouping, layout {lay}: {ms:.2f} ms', flush=True)
    y = torch.empty(B, o, pa, device=dev)
    ms0 = t_ms(lambda: _hip.gemm(0, 0, o, pa, ck, W, ck, 0, xs[0], pa, ck * pa, y, pa, o * pa, B))
    ms1 = t_ms(lambda: _hip.gemm(0, 0, o, pa, ck, W, ck, 0, xs[1], pa, ck * pa, y, pa, o * pa, B, b_blocked=True))
    y1 = y.clone()
    xt = xs[2].view(B, pa, ck)
    ms2 = t_ms(lambda: torch.matmul(W, xt.transpose(1, 2), out=y))
    print(f'   contraction: own GEMM {ms0:.2f} ms, own GEMM blocked {ms1:.2f} ms, library on transposed {ms2:.2f} ms; '
          f'max |y_lib - y_own| / max|y| = {float((y - y1).abs().max() / y1.abs().max()):.2e}', flush=True)
