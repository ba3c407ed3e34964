"""tests/golden/make_golden_artmode.py -- golden vectors of InterSO3PoseConv(use_art_mode=True) (so3conv/modules.py:L256-262 ->
so3conv/functional.py:L1289-1716, stride-1 branch L1420-1520), produced by RUNNING THE REFERENCE on CPU in the build container
(tests/golden/ref_import.py).  The cloud comes in n_states articulation states (xyz [b, ns, 3, p]); a per-point label picks the
state whose ball query and offsets the point uses; per-point poses select the anchor permutation only.  Data only.

  inter_pose_artmode.npz   two cases: per-part poses with permute_modes = 1, and no permutation; outputs + autograd gradients

Re-run:  python tests/golden/make_golden_artmode.py"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (imports the reference through ref_import)
from make_golden import poses, save  # noqa: E402
from make_golden_extra import synth_clouds  # noqa: E402

vgtk, sptk, L, zptk = MG.vgtk, MG.sptk, MG.L, MG.zptk


def main():
    gen = torch.Generator().manual_seed(909)
    B, NS, P, C, O, NNB = 2, 2, 96, 6, 8, 16
    base, part, _ = synth_clouds.laptop_batch(17, B, P)
    base = torch.from_numpy(base)
    # state 1: the lid points (label 1) moved by a small rigid motion; state 0: the cloud as it is
    moved = base.clone()
    lid = torch.from_numpy(part).bool()
    Rz = torch.tensor([[0.9553, -0.2955, 0.0], [0.2955, 0.9553, 0.0], [0.0, 0.0, 1.0]])
    for b in range(B):
        moved[b][:, lid[b]] = Rz @ base[b][:, lid[b]] + torch.tensor([[0.02], [0.0], [0.01]])
    xyz = torch.stack([base, moved], 1).contiguous()                       # [b, ns, 3, p]
    seg = torch.from_numpy(part).long()                                    # the state every point reads its neighbourhood from
    out = {'xyz': xyz, 'seg': seg}
    for tag, mode, pm in (('parts_pm1', 'parts', 1), ('random_pm0', 'random', 0)):
        torch.manual_seed(2913)
        conv = sptk.InterSO3PoseConv(C, O, 1, 1, 0.2, 0.02, NNB, kanchor=60, permute_modes=pm, use_art_mode=True)
        pose = poses(gen, B, P, mode)
        feats = torch.randn(B, C, P, 60, generator=gen).requires_grad_(True)
        with contextlib.redirect_stdout(io.StringIO()):
            inter_idx, inter_w, sample_idx, y = conv(zptk.SphericalPointCloudPose(xyz, feats, None, pose), seg=seg)
        gy = torch.randn(y.feats.shape, generator=gen)
        gfe, gW = torch.autograd.grad(y.feats, [feats, conv.basic_conv.W], gy)
        out.update({f'{tag}_W': conv.basic_conv.W, f'{tag}_pose': pose, f'{tag}_feats': feats, f'{tag}_out': y.feats, f'{tag}_gy': gy,
                    f'{tag}_gfeats': gfe, f'{tag}_gW': gW, f'{tag}_inter_w_sample': inter_w[:, ::8, ::7, ::5]})
        assert tuple(y.xyz.shape) == tuple(xyz.shape)
    save('inter_pose_artmode.npz', **out)


if __name__ == '__main__':
    main()
