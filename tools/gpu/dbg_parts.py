import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import bench
import synth_clouds
import vgtk.so3conv as sptk, vgtk.spconv as zptk
import vgtk.so3conv.functional as L
import vgtk.cuda.grouping as G
from test_gpu_dense import _quat_rot, NA, KS, NN
dev = torch.device('cuda:0')
B, P, c, o = 3, 512, 32, 256
_, _, radius, sigma = synth_clouds.backbone_layers(4096)[2]
xyz_np, lab_np, _ = synth_clouds.laptop_batch(81, B, P)
xyz = torch.from_numpy(xyz_np).to(dev)
rng = np.random.default_rng(5)
R = _quat_rot(rng, 6)
part = np.zeros((B, P), np.int64)
part[0] = lab_np[0] % 2; part[1] = lab_np[1] % 2; part[1, 100:117] = 2
pose_np = np.tile(np.eye(4, dtype=np.float32), (B, P, 1, 1))
pose_np[0, :, :3, :3] = R[0:2][part[0]]; pose_np[1, :, :3, :3] = R[2:5][part[1]]; pose_np[2, :, :3, :3] = R[5]
pose = torch.from_numpy(pose_np).to(dev)
gen = torch.Generator(device=dev).manual_seed(23)
feats0 = torch.randn(B, c, P, NA, device=dev, generator=gen)
W0 = torch.randn(o, c * KS, device=dev, generator=gen) * 0.05
def run(mode):
    L.DENSE_MODE = mode
    torch.manual_seed(2913)
    conv = sptk.InterSO3PoseConv(c, o, 1, 1, radius, sigma, NN, kanchor=NA, permute_modes=1).to(dev)
    with torch.no_grad():
        conv.basic_conv.W.copy_(W0)
        return conv(zptk.SphericalPointCloudPose(xyz, feats0, None, pose))[3].feats
y0, y1 = run('off'), run('force')
d = (y1 - y0).abs()
print('per cloud', [float(d[b].max()) for b in range(B)], 'scale', float(y0.abs().max()))
ball = G.ball_query(xyz, xyz, radius, NN).long()
pt = torch.from_numpy(part).to(dev)
for b in range(2):
    npart = pt[b][ball[b]]                       # [P, NN]
    mixed = (npart != pt[b][:, None]).any(1)
    e = d[b].amax(dim=(0, 2))                    # per point
    print('cloud', b, 'mixed points', int(mixed.sum()), 'err mixed', float(e[mixed].max()) if mixed.any() else None, 'err pure', float(e[~mixed].max()) if (~mixed).any() else None)
    print(' per anchor err', [round(float(d[b][:, :, a].max()), 3) for a in range(0, 60, 6)])
e = d[1].amax(dim=(0, 2))
for k in range(3):
    m = pt[1] == k
    print('cloud 1 part', k, 'size', int(m.sum()), 'max err', float(e[m].max()), 'points wrong', int((e[m] > 1e-3).sum()))
parts = L._pose_parts(pose)
print('sizes', parts.sizes, 'widths', parts.width)
print('labels of cloud1 pts 98..120', parts.labels[1, 98:120].tolist())
