"""tools/permuted_pose_time.py -- fwd+bwd time of the deepest backbone layer with identity poses (the entry-list
kernels) and with random per-point poses (anchor permutation per neighbour: csrc/so3_inter_mfma.hip / so3_inter_inv.hip)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import numpy as np, torch, synth_clouds
import vgtk.so3conv as sptk, vgtk.spconv as zptk
dev = torch.device('cuda:0')
B, P = 8, 4096
xyz, _, pose = synth_clouds.laptop_batch(0, B, P)
xyz = torch.from_numpy(xyz).to(dev)
c, o, r, s = synth_clouds.backbone_layers(P)[2]
conv = sptk.InterSO3PoseConv(c, o, 1, 1, r, s, 64, kanchor=60, permute_modes=1).to(dev)
gy = torch.randn(B, o, P, 60, device=dev)
rng = np.random.default_rng(0)
q = rng.standard_normal((B, P, 4)); q /= np.linalg.norm(q, axis=-1, keepdims=True)
w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
              2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(B, P, 3, 3)
rand_pose = np.tile(np.eye(4, dtype=np.float32), (B, P, 1, 1)); rand_pose[..., :3, :3] = R
_, lab, _ = synth_clouds.laptop_batch(0, B, P)
part_pose = np.tile(np.eye(4, dtype=np.float32), (B, P, 1, 1))
for bi in range(B):                                   # one rotation per rigid part (the articulated-object case)
    part_pose[bi, :, :3, :3] = R[bi, :2][lab[bi]]
import vgtk.so3conv.functional as L
from vgtk import _hip
for name, ps in (('identity poses', torch.from_numpy(pose).to(dev)), ('one rotation per rigid part', torch.from_numpy(part_pose).to(dev)),
                 ('random per-point poses', torch.from_numpy(rand_pose).to(dev))):
    for coset in ((2, True, False) if name != 'identity poses' else (True,)):
        # 2 = coset-major operand on the two-tile kernel (csrc/so3_inter_lists2.hip PERM), True = on the whole-row kernel
        # (csrc/so3_inter_inv.hip COSET), False = byte-table lookups
        L.COSET_OPERAND = bool(coset)
        _hip.lib.eap_so3_group_perm_lists2(1 if coset == 2 else 0)
        f = torch.randn(B, c, P, 60, device=dev, requires_grad=True)
        ts = []
        for it in range(4):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            yv = conv(zptk.SphericalPointCloudPose(xyz, f, None, ps))[3].feats
            e1.record()
            torch.autograd.grad(yv, [f, conv.basic_conv.W], gy)
            e2.record(); torch.cuda.synchronize()
            ts.append((e0.elapsed_time(e1), e1.elapsed_time(e2)))
        fw, bw = sorted(t[0] for t in ts[1:])[1], sorted(t[1] for t in ts[1:])[1]
        print(f'{name}{"" if name == "identity poses" else (", two-tile kernel with block moves by DMA" if coset == 2 else ", coset-major operand" if coset else ", byte-table lookups")}: forward {fw:.1f} ms, backward {bw:.1f} ms', flush=True)
L.COSET_OPERAND = True
_hip.lib.eap_so3_group_perm_lists2(1)
