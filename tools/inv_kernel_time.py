"""tools/inv_kernel_time.py B -- HIP-event time of the inverse-list grouping launch alone (layer 3 shapes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch
import synth_clouds
import vgtk.so3conv as sptk, vgtk.spconv as zptk
import vgtk.so3conv.functional as L
from vgtk import _hip
B, P = int(sys.argv[1]), 4096
dev = torch.device('cuda:0')
xyz, _, pose = synth_clouds.laptop_batch(0, B, P)
xyz, pose = torch.from_numpy(xyz).to(dev), torch.from_numpy(pose).to(dev)
c, o, r, s = synth_clouds.backbone_layers(P)[2]
conv = sptk.InterSO3PoseConv(c, o, 1, 1, r, s, 64, kanchor=60, permute_modes=1).to(dev)
f = torch.randn(B, c, P, 60, device=dev).requires_grad_(True)
gy = torch.randn(B, o, P, 60, device=dev)
L.BACKWARD_MODE = 'inverse'
_hip.KERNEL_TIMES = []
y = conv(zptk.SphericalPointCloudPose(xyz, f, None, pose))[3].feats
for _ in range(3):
    g = torch.autograd.grad(y, [f], gy, retain_graph=True)
torch.cuda.synchronize()
for name in ('eap_so3_inter_group_inv_f32', 'eap_so3_inter_group_fwd_f32'):
    ts = [e0.elapsed_time(e1) for n, tag, e0, e1 in _hip.KERNEL_TIMES if n == name]
    print(name, 'min %.2f ms' % min(ts), len(ts), 'launches')
