"""tools/inv_only.py -- timing of the backward feature-gradient strategies for one layer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch
import synth_clouds
import vgtk.so3conv as sptk, vgtk.spconv as zptk
import vgtk.so3conv.functional as L
B, P = int(sys.argv[1]), 4096
dev = torch.device('cuda:0')
xyz, _, pose = synth_clouds.laptop_batch(0, B, P)
xyz, pose = torch.from_numpy(xyz).to(dev), torch.from_numpy(pose).to(dev)
for li in (1, 2):
    c, o, r, s = synth_clouds.backbone_layers(P)[li]
    conv = sptk.InterSO3PoseConv(c, o, 1, 1, r, s, 64, kanchor=60, permute_modes=1).to(dev)
    f0 = torch.randn(B, c, P, 60, device=dev)
    gy = torch.randn(B, o, P, 60, device=dev)
    for mode in ('dx', 'inverse'):
        L.BACKWARD_MODE = mode
        f = f0.clone().requires_grad_(True)
        y = conv(zptk.SphericalPointCloudPose(xyz, f, None, pose))[3].feats
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g = torch.autograd.grad(y, [f], gy, retain_graph=True); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f'L{li} dF via {mode:8s}: {min(ts):8.2f} ms', flush=True)
