"""tests/golden/make_golden_zp.py -- golden vectors for the S^2 ("ZP") layer of vgtk.spconv and the helper exports the
reference package carries beside it, produced by RUNNING THE REFERENCE on CPU in the build container (ref_import.py).
Data only.

  zp_layer.npz   tables: get_anchors(12 / 42), get_kernel_rings_np (int and pair kernel sizes), get_intra_kernels,
                 acos_safe, anchor_knn (3 metrics), get_intra_kernel_weights (with / without suppression),
                 compute_anchor_weights (3 interpolations), anchor_prop, so3conv get_kernel_points_np /
                 get_spherical_kernel_points_np, inter_zpconv_grouping_anchor weights;
                 layers (vgtk/vgtk/spconv/modules.py:L17-161) with seeded parameters on a 2 x 64-point cloud:
                 IntraZPConv, InterZPConv (kernel_size 1, the only size whose einsum the reference can run), AnchorProp,
                 outputs + autograd gradients; vgtk.LearningRateScheduler rate sequences (vgtk/vgtk/utils.py:L33-68)

Re-run:  python tests/golden/make_golden_zp.py"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (imports the reference through ref_import)
from make_golden import save  # noqa: E402
from make_golden_extra import synth_clouds  # noqa: E402

vgtk, sptk, L, zptk = MG.vgtk, MG.sptk, MG.L, MG.zptk
ZL = zptk.functional


def main():
    out = {}
    gen = torch.Generator().manual_seed(77)
    a12, a42 = ZL.get_anchors(12), ZL.get_anchors(42)
    out['anchors12'], out['anchors42'] = a12, a42
    out['rings_int'] = ZL.get_kernel_rings_np(0.2, 1.2, 3, multiplier=2)
    out['rings_pair'] = ZL.get_kernel_rings_np(0.2, 1.2, (2, 3))
    out['intra_kernels'] = ZL.get_intra_kernels(1.2, 4)
    x = torch.cat([torch.linspace(-1.0, 1.0, 41), torch.tensor([0.99985, -0.99995, 0.9999, 1.00002])])
    out['acos_in'], out['acos_out'] = x, ZL.acos_safe(x)
    for metric in ('spherical', 'angular', 'euclidean'):
        val, idx = ZL.anchor_knn(a12, a42, k=4, metric=metric)
        out[f'knn_{metric}_val'], out[f'knn_{metric}_idx'] = val, idx
    for tag, sup in (('plain', False), ('suppressed', True)):
        idx, w = ZL.get_intra_kernel_weights(a12, a42, out['intra_kernels'], 5, 1.2, 0.1, use_suppression=sup)
        out[f'intra_{tag}_idx'], out[f'intra_{tag}_w'] = idx, w
    for interp in ('inv', 'spherical', 'euclidean'):
        idx, w = ZL.compute_anchor_weights(a12, a42, k=3, sigma=0.1, interpolation=interp)
        out[f'aw_{interp}_idx'], out[f'aw_{interp}_w'] = idx, w
    f = torch.randn(2, 3, 5, 12, generator=gen)
    out['prop_in'], out['prop_out'] = f, ZL.anchor_prop(f, out['aw_inv_idx'], out['aw_inv_w'])
    out['kernel_points'] = L.get_kernel_points_np(0.3, 1.0, 3, multiplier=2)
    out['spherical_kernel_points'] = L.get_spherical_kernel_points_np(0.3, 3, multiplier=2)

    xyz = torch.from_numpy(synth_clouds.laptop_batch(11, 2, 64)[0])
    rings = torch.from_numpy(ZL.get_kernel_rings_np(0.25, 1.2, 2, multiplier=2))
    with contextlib.redirect_stdout(io.StringIO()):
        gxyz, ball_idx, cidx, _ = ZL.inter_zpconv_grouping_ball(xyz, 1, 0.25, 12, True)
    _, w = ZL.inter_zpconv_grouping_anchor(gxyz, ball_idx, cidx, a12, rings, 4, 64, 0.25, 1.2, 0.05)
    out['xyz'], out['anchor_w_rings'], out['anchor_w'], out['ball_idx'] = xyz, rings, w, ball_idx

    # layers
    torch.manual_seed(2913)
    intra = zptk.IntraZPConv(6, 9, 3, 1.2, 0.1, 4, 12)
    fi = torch.randn(2, 6, 64, 12, generator=gen).requires_grad_(True)
    y = intra(zptk.SphericalPointCloud(xyz, fi, None)).feats
    gy = torch.randn(y.shape, generator=gen)
    gfi, gW, gb = torch.autograd.grad(y, [fi, intra.basic_conv.W, intra.basic_conv.bias], gy)
    out.update(intra_W=intra.basic_conv.W, intra_bias=intra.basic_conv.bias, intra_idx=intra.intra_idx, intra_w=intra.intra_w,
               intra_in=fi, intra_out=y, intra_gy=gy, intra_gin=gfi, intra_gW=gW, intra_gbias=gb)

    torch.manual_seed(2914)
    inter = zptk.InterZPConv(6, 5, 1, 1, 0.25, 1.2, 0.05, 12, 12, 4)
    fe = torch.randn(2, 6, 64, 12, generator=gen).requires_grad_(True)
    with contextlib.redirect_stdout(io.StringIO()):          # the reference prints the table shapes
        iidx, iw, cloud = inter(zptk.SphericalPointCloud(xyz, fe, None))
    gy = torch.randn(cloud.feats.shape, generator=gen)
    gfe, gW, gb = torch.autograd.grad(cloud.feats, [fe, inter.basic_conv.W, inter.basic_conv.bias], gy)
    out.update(inter_W=inter.basic_conv.W, inter_bias=inter.basic_conv.bias, inter_kernels=inter.kernels, inter_idx=iidx, inter_w=iw,
               inter_in=fe, inter_out=cloud.feats, inter_gy=gy, inter_gin=gfe, inter_gW=gW, inter_gbias=gb)
    # strided (lazy centres), second call re-using the tables of the first
    torch.manual_seed(2915)
    inter2 = zptk.InterZPConv(6, 5, 1, 2, 0.25, 1.2, 0.05, 12, 12, 4)
    with contextlib.redirect_stdout(io.StringIO()):
        i2, w2, c2 = inter2(zptk.SphericalPointCloud(xyz, fe.detach(), None))
    out.update(inter2_W=inter2.basic_conv.W, inter2_bias=inter2.basic_conv.bias, inter2_idx=i2, inter2_w=w2, inter2_xyz=c2.xyz, inter2_out=c2.feats)

    prop = zptk.AnchorProp(12, 42, 0.1)
    out.update(aprop_idx=prop.idx, aprop_w=prop.w, aprop_out=prop(zptk.SphericalPointCloud(xyz, fi.detach(), None)).feats)

    # learning-rate schedules
    for tag, kind, kw in (('const', 'constant', dict(decay_rate=0.5)), ('exp', 'exp_decay', dict(decay_rate=0.7))):
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
        with contextlib.redirect_stdout(io.StringIO()):
            sched = vgtk.LearningRateScheduler(opt, 1e-3, kind, 3, **kw)
            rates = [sched.step() for _ in range(10)]
        out[f'lr_{tag}'] = np.asarray(rates, dtype=np.float64)
        out[f'lr_{tag}_group'] = np.asarray(opt.param_groups[0]['lr'], dtype=np.float64)
    save('zp_layer.npz', **out)


if __name__ == '__main__':
    main()
