"""tools/group_fwd_tiles_ab.py [B]: the forward grouping launch (transposed intermediate, layout 2) of the two deep layers with the shipped
two-tile fp32-MFMA kernel (tiles 2) and, in `make EXPERIMENTS=1` builds, the 3 x bf16 (3) and 2 x fp16 plane (4) kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch
import synth_clouds
import vgtk.so3conv as sptk
from vgtk import _hip
import vgtk.cuda.grouping as G
import vgtk.so3conv.functional as L
B, P = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), 4096
dev = torch.device('cuda:0')
xyz, _, pose = synth_clouds.laptop_batch(0, B, P)
xyz, pose = torch.from_numpy(xyz).to(dev), torch.from_numpy(pose).to(dev)
for li in (2, 1):
    c, o, r, s = synth_clouds.backbone_layers(P)[li]
    conv = sptk.InterSO3PoseConv(c, o, 1, 1, r, s, 64, kanchor=60, permute_modes=1).to(dev)
    feats = torch.randn(B, c, P, 60, device=dev)
    idx = G.ball_query(xyz, xyz, r, 64)
    rk = L.rotated_kernels(conv.anchors, conv.kernels)
    gx, nonident = _hip.so3_prep(xyz, xyz, idx, None, None, conv.anchors, 29)
    fl = 2.0 * B * c * 24 * P * 64 * 60
    ref = None
    for tiles in (2, 3, 4):
        if _hip.lib.eap_so3_group_lists_tiles(tiles) != tiles:
            continue
        run = lambda: _hip.so3_inter_group_fwd(feats, idx, gx, rk, None, s, None, blocked=2)
        X = run()
        if ref is None:
            ref = X.double()
        else:
            d = (X.double() - ref).abs()
            print(f'   tiles {tiles}: max |X - X(tiles 2)| / max |X| = {d.max().item() / ref.abs().max().item():.3e}', flush=True)
        del X
        ts = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); X = run(); e1.record(); torch.cuda.synchronize(); del X
            ts.append(e0.elapsed_time(e1))
        ts = sorted(ts[1:])
        print(f'layer {li} C={c} tiles {tiles}: median {ts[len(ts) // 2]:.2f} ms  {fl / ts[len(ts) // 2] / 1e9:.1f} TFLOP/s algorithmic', flush=True)
    _hip.lib.eap_so3_group_lists_tiles(2)
    del ref
