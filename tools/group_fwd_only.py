"""tools/group_fwd_only.py -- the fused grouping forward in isolation (one layer), for PMC passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch
import synth_clouds
import vgtk.so3conv as sptk
from vgtk import _hip
import vgtk.cuda.grouping as G
import vgtk.so3conv.functional as L
B, P = int(sys.argv[1]), 4096
li = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device('cuda:0')
xyz, _, pose = synth_clouds.laptop_batch(0, B, P)
xyz, pose = torch.from_numpy(xyz).to(dev), torch.from_numpy(pose).to(dev)
c, o, r, s = synth_clouds.backbone_layers(P)[li]
conv = sptk.InterSO3PoseConv(c, o, 1, 1, r, s, 64, kanchor=60, permute_modes=1).to(dev)
feats = torch.randn(B, c, P, 60, device=dev)
idx = G.ball_query(xyz, xyz, r, 64)
mult, ident = L._group_tables(conv.anchors)
rk = L.rotated_kernels(conv.anchors, conv.kernels)
gx, nonident = _hip.so3_prep(xyz, xyz, idx, pose, pose, conv.anchors, ident)
for _ in range(3):
    X = _hip.so3_inter_group_fwd(feats, idx, gx, rk, mult, s, nonident)
torch.cuda.synchronize()
