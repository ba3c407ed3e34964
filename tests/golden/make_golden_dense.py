"""tests/golden/make_golden_dense.py -- golden vectors for the layers the DENSE product (csrc/so3_dense.hip) takes, produced by
RUNNING THE REFERENCE on CPU in the build container (through tests/golden/ref_import.py).  Data only.

The other inter-conv fixtures are 64-point clouds with 8 output channels: none of them reaches the dense product
(`eap_so3_dense_supported` wants P % 32 == 0 and O % 128 == 0, the host takes it while a cloud references <= 5 nsample rows and no list
is padded).  These cases are sized so it does, in the DEFAULT mode (no EAP_DENSE=force):

  dense_identity_o128.npz  InterSO3PoseConv 8 -> 128, 2 x 128 points, nsample 16, radius 0.3: every ball holds >= 16 points (no padded
                           list), a cloud's lists name 65-74 of its rows (<= 80); identity poses -> the BACKWARD runs the dense product
  dense_identity_o256.npz  the same at 8 -> 256: forward AND backward run it
  dense_parts_o256.npz     two rigid parts with one rotation each (points alternate), radius 0.35: one product per part
                           (so3conv/functional.py:L1112-1160, L1199-1204: relative rotation + anchor permutation per (part, part))

Reference path: so3conv/modules.py:L222-322 -> so3conv/functional.py:L1025-1261 -> so3conv/modules.py:L48-55, autograd for the
gradients.  To keep the files small the output gradient is rank 2, gy[b,o,p,a] = u1[b,o] v1[b,p,a] + u2[b,o] v2[b,p,a] (the factors
are stored; both sides form it in float32 in this order), and the output is stored on every 16th channel at all points plus on all
channels at every 32nd point.

Re-run:  python tests/golden/make_golden_dense.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (imports the reference through ref_import)
from make_golden import poses, save  # noqa: E402

sptk, L, zptk, synth_clouds = MG.sptk, MG.L, MG.zptk, MG.synth_clouds
B, P, C, NN = 2, 128, 8, 16


def rank2_grad(u1, v1, u2, v2):
    return u1[:, :, None, None] * v1[:, None] + u2[:, :, None, None] * v2[:, None]


def main():
    gen = torch.Generator().manual_seed(60128)
    xyz = torch.from_numpy(synth_clouds.laptop_batch(7, B, P)[0])
    for name, O, radius, sigma, pmode in (('dense_identity_o128', 128, 0.3, 0.045, 'identity'),
                                          ('dense_identity_o256', 256, 0.3, 0.045, 'identity'),
                                          ('dense_parts_o256', 256, 0.35, 0.06, 'parts')):
        torch.manual_seed(2913)
        conv = sptk.InterSO3PoseConv(C, O, 1, 1, radius, sigma, NN, kanchor=60, permute_modes=1)
        pose = poses(gen, B, P, pmode)
        feats = torch.randn(B, C, P, 60, generator=gen).requires_grad_(True)
        # what makes the case a dense one: no padded list, few referenced rows (the reference's ball query = the oracle's C restatement)
        idx = MG.vgtk.cuda.grouping.ball_query(xyz, xyz, radius, NN).numpy()
        d = (xyz[:, :, :, None] - xyz[:, :, None, :]).norm(dim=1)
        assert int((d < radius).sum(-1).min()) >= NN, 'a ball with fewer than nsample points: its list would be padded'
        rows = [len(np.unique(idx[b])) for b in range(B)]
        assert max(rows) <= 5 * NN, rows
        inter_idx, inter_w, sample_idx, y = conv(zptk.SphericalPointCloudPose(xyz, feats, None, pose))
        u1, u2 = torch.randn(B, O, generator=gen), torch.randn(B, O, generator=gen)
        v1, v2 = torch.randn(B, P, 60, generator=gen), torch.randn(B, P, 60, generator=gen)
        gy = rank2_grad(u1, v1, u2, v2)
        gfe, gW = torch.autograd.grad(y.feats, [feats, conv.basic_conv.W], gy)
        save(name + '.npz', xyz=xyz, pose=pose, feats=feats, W=conv.basic_conv.W, anchors=conv.anchors, kernels=conv.kernels,
             radius=np.float32(radius), sigma=np.float32(sigma), nn=np.int32(NN), permute_modes=np.int32(1),
             referenced_rows=np.asarray(rows, np.int32), ball_idx=idx,
             out_channels16=y.feats[:, ::16], out_points32=y.feats[:, :, ::32],
             gy_u1=u1, gy_v1=v1, gy_u2=u2, gy_v2=v2, grad_feats=gfe, grad_W=gW)


if __name__ == '__main__':
    main()
