"""tests/golden/make_golden.py -- generate golden fixtures by RUNNING THE REFERENCE.

Build-container only: imports /root/reference/vgtk through tests/golden/ref_import.py
(plyfile/trimesh shims + the CPU oracle standing in for the two CUDA-only
helpers) and records inputs/outputs of the reference's own functions as small
.npz files.  The fixtures are data (inputs + expected outputs); nothing of the
reference's source is stored.  Re-run with:  python tests/golden/make_golden.py

What each fixture pins (reference file:line):
  constants.npz        anchors / intra_idx / kernel points   so3conv/functional.py:L111-121, L2630-2659
  weights.npz          inter_so3conv_grouping_anchor          so3conv/functional.py:L2508-2549
  inter_pose_*.npz     InterSO3PoseConv.forward (+autograd)   so3conv/modules.py:L222-322, functional.py:L1025-1286
  inter_nopose.npz     InterSO3Conv.forward                   so3conv/modules.py:L157-174, functional.py:L144-203
  intra.npz            IntraSO3Conv.forward (+autograd)       so3conv/modules.py:L344-347, functional.py:L2553-2602
  zpconv_naive.npz     inter/intra_zpconv_grouping_naive      spconv/functional.py:L252-272, L375-406
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, '..', '..', 'equi-articulated-pose_amd'))
import ref_import  # noqa: E402

vgtk, sptk, L, zptk = ref_import.import_reference()
import synth_clouds  # noqa: E402  (pure numpy, no vgtk import)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB  ' + ', '.join(f'{k}{list(v.shape)}' for k, v in out.items()))


def rand_rot(gen, *shape):
    q = torch.randn(*shape, 4, generator=gen)
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)
    return R.view(*shape, 3, 3)


def poses(gen, b, p, mode):
    P = torch.eye(4).repeat(b, p, 1, 1)
    if mode == 'random':
        P[:, :, :3, :3] = rand_rot(gen, b, p)
    elif mode == 'parts':  # two rigid parts, each with one rotation (articulated object)
        R = rand_rot(gen, b, 2)
        lab = (torch.arange(p) % 2)[None].expand(b, p)
        P[:, :, :3, :3] = torch.gather(R, 1, lab[..., None, None].expand(b, p, 3, 3))
    return P


def main():
    gen = torch.Generator().manual_seed(2913)
    anchors = L.get_anchors()
    intra_idx = L.get_intra_idx()
    save('constants.npz', anchors=anchors, intra_idx=intra_idx,
         anchors20=L.get_anchors(20), anchors1=L.get_anchors(1),
         kernels_r008=L.get_sphereical_kernel_points_from_ply(0.7 * 0.08, 1),
         kernels_r032=L.get_sphereical_kernel_points_from_ply(0.7 * 0.32, 1),
         kernels30_r01=L.get_sphereical_kernel_points_from_ply(0.7 * 0.1, 2))

    # ---- kernel weights -------------------------------------------------
    A = torch.from_numpy(anchors)
    kern = torch.from_numpy(L.get_sphereical_kernel_points_from_ply(0.7 * 0.08, 1))
    g = (torch.rand(2, 3, 12, 16, generator=gen) - 0.5) * 0.16
    save('weights.npz', grouped_xyz=g, anchors=A, kernels=kern, sigma=np.float32(0.0032),
         inter_w=L.inter_so3conv_grouping_anchor(g, A, kern, 0.0032))

    # ---- InterSO3PoseConv layers ---------------------------------------
    xyz_np, _, _ = synth_clouds.laptop_batch(0, 2, 64)
    xyz = torch.from_numpy(xyz_np)
    cases = [
        # name, C, O, radius, sigma, nn, pose mode, permute_modes
        ('inter_pose_l0_identity', 1, 8, 0.08, 0.0032, 16, 'identity', 1),
        ('inter_pose_identity', 6, 8, 0.16, 0.0128, 16, 'identity', 1),
        ('inter_pose_random_pm1', 6, 8, 0.16, 0.0128, 16, 'random', 1),
        ('inter_pose_parts_pm1', 6, 8, 0.16, 0.0128, 16, 'parts', 1),
        ('inter_pose_random_pm0', 6, 8, 0.16, 0.0128, 16, 'random', 0),
        ('inter_pose_bigball', 4, 8, 0.45, 0.1024, 64, 'identity', 1),
    ]
    for name, C, O, radius, sigma, nn, pmode, pm in cases:
        torch.manual_seed(2913)
        conv = sptk.InterSO3PoseConv(C, O, 1, 1, radius, sigma, nn, kanchor=60, permute_modes=pm)
        pose = poses(gen, 2, 64, pmode)
        if C == 1:
            feats = sptk.get_occupancy_features(xyz.transpose(1, 2), 60, False)
        else:
            feats = torch.randn(2, C, 64, 60, generator=gen)
        feats.requires_grad_(True)
        x = zptk.SphericalPointCloudPose(xyz, feats, None, pose)
        inter_idx, inter_w, sample_idx, y = conv(x)
        gy = torch.randn(y.feats.shape, generator=gen)
        gfe, gW = torch.autograd.grad(y.feats, [feats, conv.basic_conv.W], gy)
        # intermediate: grouped features before the dense contraction
        _, _, _, new_feats, _, _ = L.inter_so3poseconv_grouping_strided(
            xyz, pose, feats.detach(), 1, nn, conv.anchors, conv.kernels, radius, sigma,
            None, None, True, pooling=None, permute_modes=pm)
        save(name + '.npz', xyz=xyz, pose=pose, feats=feats, W=conv.basic_conv.W, anchors=conv.anchors,
             kernels=conv.kernels, radius=np.float32(radius), sigma=np.float32(sigma),
             nn=np.int32(nn), permute_modes=np.int32(pm),
             inter_w_head=inter_w[:, :4], new_feats_head=new_feats[:, :, :, :8], out=y.feats,
             grad_out=gy, grad_feats=gfe, grad_W=gW)

    # ---- InterSO3Conv (pose-free) --------------------------------------
    torch.manual_seed(2913)
    conv = sptk.InterSO3Conv(5, 7, 1, 1, 0.16, 0.0128, 16, kanchor=60)
    feats = torch.randn(2, 5, 64, 60, generator=gen)
    inter_idx, inter_w, sample_idx, y = conv(zptk.SphericalPointCloud(xyz, feats, None))
    save('inter_nopose.npz', xyz=xyz, feats=feats, W=conv.basic_conv.W, anchors=conv.anchors,
         kernels=conv.kernels, radius=np.float32(0.16), sigma=np.float32(0.0128), nn=np.int32(16),
         inter_idx=inter_idx, out=y.feats)

    # ---- IntraSO3Conv ---------------------------------------------------
    torch.manual_seed(2913)
    conv = sptk.IntraSO3Conv(6, 9)
    feats = torch.randn(2, 6, 40, 60, generator=gen, requires_grad=True)
    y = conv(zptk.SphericalPointCloud(xyz[:, :, :40], feats, None))
    gy = torch.randn(y.feats.shape, generator=gen)
    gfe, gW = torch.autograd.grad(y.feats, [feats, conv.basic_conv.W], gy)
    save('intra.npz', feats=feats, W=conv.basic_conv.W, intra_idx=conv.intra_idx, out=y.feats,
         grouped_head=L.intra_so3conv_grouping(conv.intra_idx, feats.detach())[:, :, :, :8],
         grad_out=gy, grad_feats=gfe, grad_W=gW)

    # ---- zpconv naive grouping (shared index across (a,k)) --------------
    b, p, a, k, nn, c, q = 2, 20, 12, 5, 8, 3, 21
    idx = torch.randint(0, q, (b, p, nn), generator=gen)
    w = torch.rand(b, p, a, k, nn, generator=gen)
    feats = torch.randn(b, c, q, a, generator=gen)
    inter = zptk.inter_zpconv_grouping_naive(idx, w, feats)
    iidx = torch.randint(0, a, (a, 4), generator=gen)
    iw = torch.rand(a, k, 4, generator=gen)
    f2 = torch.randn(b, c, p, a, generator=gen)
    intra = zptk.intra_zpconv_grouping_naive(iidx, iw, f2)
    save('zpconv_naive.npz', inter_idx=idx.int(), inter_w=w, inter_feats=feats, inter_out=inter,
         intra_idx=iidx.int(), intra_w=iw, intra_feats=f2, intra_out=intra)


if __name__ == '__main__':
    main()
