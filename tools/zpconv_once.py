"""tools/zpconv_once.py -- two calls each of the native zpconv forward and backward at the bench workload (8 x 4096, C = 64,
layer-1 radius): what the counter passes of tools/gpu/profile_round.sh run for the zpconv kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch
import synth_clouds
import vgtk.cuda.zpconv as Z
import vgtk.cuda.grouping as G

B, P, A, K, NN, C = 8, 4096, 60, 24, 64, 64
dev = torch.device('cuda:0')
xyz = torch.from_numpy(synth_clouds.laptop_batch(0, B, P)[0]).to(dev)
ball = G.ball_query(xyz, xyz, synth_clouds.backbone_layers(P)[1][2], NN)
idx = ball[:, :, None, None, :].expand(B, P, A, K, NN).contiguous()
w = torch.rand(B, P, A, K, NN, device=dev)
feats = torch.randn(B, C, P, A, device=dev)
g = torch.randn(B, C, K, P, A, device=dev)
for _ in range(2):
    Z.inter_zpconv_forward(idx, w, feats)
    Z.inter_zpconv_backward(idx, w, g, P)
torch.cuda.synchronize()
