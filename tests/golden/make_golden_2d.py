"""tests/golden/make_golden_2d.py -- golden vectors of InterSO3PoseConv(use_2d=True) (so3conv/modules.py:L249-255 ->
so3conv/functional.py:L1718-2130, stride-1 branch L1812-2128; the configuration scripts/train/eyeglasses.sh ships), produced by
RUNNING THE REFERENCE on CPU in the build container (tests/golden/ref_import.py).  The anchor axis is 60 x 4: every icosahedral
anchor times four residual rotations about the y axis (get_2D_res_anchors, L29-46); the weights use all 240 rotations, the
per-entry anchor permutation acts on the residual index only.  Data only.

  inter_pose_2d.npz   identity poses (permute_modes 1), one rotation per point (permute_modes 1 and 0): outputs + autograd gradients

Re-run:  python tests/golden/make_golden_2d.py   (the reference calls .cuda() on its residual-rotation table: mapped to a no-op here)"""
import contextlib
import io
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (imports the reference through ref_import)
from make_golden import poses, save  # noqa: E402
from make_golden_extra import synth_clouds  # noqa: E402

vgtk, sptk, L, zptk = MG.vgtk, MG.sptk, MG.L, MG.zptk


def main():
    torch.Tensor.cuda = lambda self, *a, **k: self          # RES_ROT_2D.cuda() (functional.py:L1921, L1934)
    gen = torch.Generator().manual_seed(4242)
    B, P, C, O, NNB = 1, 40, 4, 4, 8
    xyz = torch.from_numpy(synth_clouds.laptop_batch(23, B, P)[0])
    out = {'xyz': xyz, 'res_rot': L.RES_ROT_2D}
    for tag, mode, pm in (('identity_pm1', 'identity', 1), ('random_pm1', 'random', 1), ('random_pm0', 'random', 0)):
        torch.manual_seed(2913)
        conv = sptk.InterSO3PoseConv(C, O, 1, 1, 0.2, 0.02, NNB, kanchor=60, permute_modes=pm, use_2d=True)
        pose = poses(gen, B, P, mode)
        feats = torch.randn(B, C, P, 240, generator=gen).requires_grad_(True)
        with contextlib.redirect_stdout(io.StringIO()):
            inter_idx, inter_w, sample_idx, y = conv(zptk.SphericalPointCloudPose(xyz, feats, None, pose))
        assert tuple(y.feats.shape) == (B, O, P, 240)
        gy = torch.randn(y.feats.shape, generator=gen)
        gfe, gW = torch.autograd.grad(y.feats, [feats, conv.basic_conv.W], gy)
        out.update({f'{tag}_W': conv.basic_conv.W, f'{tag}_pose': pose, f'{tag}_feats': feats, f'{tag}_out': y.feats, f'{tag}_gy': gy,
                    f'{tag}_gfeats': gfe, f'{tag}_gW': gW, f'{tag}_inter_w_sample': inter_w.reshape(B, P, 240, inter_w.shape[-2], NNB)[:, ::8, ::7, ::5]})
        out['anchors'], out['kernels'] = conv.anchors, conv.kernels
    save('inter_pose_2d.npz', **out)


if __name__ == '__main__':
    main()
