"""Host-side tables of the S^2 ("ZP") layer and the helper exports of the package root against fixtures produced by running
the reference (tests/golden/make_golden_zp.py -> zp_layer.npz).  CPU only: these functions build tables with numpy / torch
and launch nothing."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))

import vgtk  # noqa: E402
import vgtk.spconv as zptk  # noqa: E402
import vgtk.so3conv.functional as L  # noqa: E402

G = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'zp_layer.npz')))
T = torch.from_numpy


def close(a, b, tol=1e-6):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=tol, atol=tol)


def test_anchor_sets_and_kernel_tables():
    close(zptk.get_anchors(12), G['anchors12'], 0)
    close(zptk.get_anchors(42), G['anchors42'], 0)
    t = torch.randn(5, 3)
    assert zptk.get_anchors(t) is not None and torch.equal(zptk.get_anchors(t), t)
    close(zptk.get_kernel_rings_np(0.2, 1.2, 3, multiplier=2), G['rings_int'], 0)
    close(zptk.get_kernel_rings_np(0.2, 1.2, (2, 3)), G['rings_pair'], 0)
    close(zptk.get_intra_kernels(1.2, 4), G['intra_kernels'], 0)
    close(L.get_kernel_points_np(0.3, 1.0, 3, multiplier=2), G['kernel_points'], 1e-7)
    close(L.get_spherical_kernel_points_np(0.3, 3, multiplier=2), G['spherical_kernel_points'], 1e-7)


def test_acos_safe_and_anchor_neighbours():
    close(zptk.acos_safe(T(G['acos_in'])), G['acos_out'], 1e-6)
    a12, a42 = T(G['anchors12']), T(G['anchors42'])
    for metric in ('spherical', 'angular', 'euclidean'):
        val, idx = zptk.anchor_knn(a12, a42, k=4, metric=metric)
        close(val, G[f'knn_{metric}_val'], 1e-6)
        # ties between equidistant anchors may come back in either order: compare the neighbour SETS and the values
        assert all(set(r) == set(g) for r, g in zip(idx.tolist(), G[f'knn_{metric}_idx'].tolist())) or \
            np.allclose(np.sort(val.numpy(), 1), np.sort(G[f'knn_{metric}_val'], 1), atol=1e-6)


def test_intra_tables_and_anchor_interpolation():
    a12, a42 = T(G['anchors12']), T(G['anchors42'])
    bins = T(G['intra_kernels'])
    for tag, sup in (('plain', False), ('suppressed', True)):
        idx, w = zptk.get_intra_kernel_weights(a12, a42, bins, 5, 1.2, 0.1, use_suppression=sup)
        assert idx.dtype == torch.int32
        close(w, G[f'intra_{tag}_w'], 1e-6)
    for interp in ('inv', 'spherical', 'euclidean'):
        idx, w = zptk.compute_anchor_weights(a12, a42, k=3, sigma=0.1, interpolation=interp)
        close(w, G[f'aw_{interp}_w'], 2e-6)
    close(zptk.anchor_prop(T(G['prop_in']), T(G['aw_inv_idx']), T(G['aw_inv_w'])), G['prop_out'], 1e-6)


def test_inter_kernel_weights():
    xyz = T(G['xyz'])
    idx = T(G['ball_idx']).long()
    padded = torch.cat([xyz, torch.full((2, 3, 1), 1e4)], 2)
    grouped = torch.gather(padded[:, :, None, :].expand(-1, -1, 64, -1), 3, idx[:, None].expand(-1, 3, -1, -1)) - xyz[:, :, :, None]
    same, w = zptk.inter_zpconv_grouping_anchor(grouped, idx, None, T(G['anchors12']), T(G['anchor_w_rings']), 4, 64, 0.25, 1.2, 0.05)
    assert same is idx
    # the dot product's summation order differs (einsum); near |cos| = 1 the linear continuation of acos_safe has slope 141
    close(w, G['anchor_w'], 1e-5)


def test_learning_rate_scheduler_and_root_exports():
    assert hasattr(vgtk, 'batch_gather') and hasattr(vgtk, 'batch_zip')
    for tag, kind, kw in (('const', 'constant', dict(decay_rate=0.5)), ('exp', 'exp_decay', dict(decay_rate=0.7))):
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
        sched = vgtk.LearningRateScheduler(opt, 1e-3, kind, 3, **kw)
        rates = [sched.step() for _ in range(10)]
        np.testing.assert_allclose(rates, G[f'lr_{tag}'], rtol=1e-15)
        np.testing.assert_allclose(opt.param_groups[0]['lr'], G[f'lr_{tag}_group'], rtol=1e-15)
    try:
        vgtk.batch_zip(None, None, None)
        raise AssertionError('batch_zip should raise')
    except NotImplementedError:
        pass
