"""GPU parity at the BASELINE.json shapes (configs 2-5): the real bench layers (1->64, 64->128,
128->512; radii / sigma of build_model for 4096- and 8192-point clouds,
...pn_38_multi_stage.py:L2115-2126) against the CPU oracle.

The oracle cannot run a whole 4096-point cloud (the reference materialises [B,P,A,K,NN] weights and a
[B,P,NN,A,A,3,3] permutation intermediate), but every op of so3conv/functional.py:L1025-1261 is
independent across QUERY points, so a slab of query points is an exact sub-problem:

  forward   y[b, :, slab, :]                 ==  basic_so3conv(W, _poseconv_slab(slab).new_feats)
  backward  with dY zero outside the slab:   dF (all support rows) and dW == the oracle's autograd of
            the slab -- the GPU runs its FULL-SIZE kernels (inverse lists over all 4096x64 entries, the
            full dX GEMM, ...), only the cotangent is sparse.

Bars (max error / max magnitude, fp32 accumulation over C*K = up to 3072 terms): outputs 2e-5,
feature gradients 2e-5, weight gradients 5e-5.  North-star tolerance: 1e-4 relative on poses.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import native, so3_ref  # noqa: E402  (checker only)

T = torch.from_numpy
NN, NA, KS = 64, 60, 24


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return torch.device('cuda:0')


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _rand_rot(gen, *shape):
    q = torch.randn(*shape, 4, generator=gen)
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)
    return R.view(*shape, 3, 3)


def make_poses(gen, kinds, labels, p):
    """kinds[b] in {'identity', 'random' (one Haar rotation per point), 'parts' (one per rigid part)}."""
    pose = torch.eye(4).repeat(len(kinds), p, 1, 1)
    for b, kind in enumerate(kinds):
        if kind == 'random':
            pose[b, :, :3, :3] = _rand_rot(gen, p)
        elif kind == 'parts':
            R = _rand_rot(gen, 2)
            pose[b, :, :3, :3] = R[torch.from_numpy(labels[b])]
        elif kind == 'parts4':                       # four rigid parts of very different sizes: each of the two labels split 7 : 1 by index
            R = _rand_rot(gen, 4)
            lab4 = 2 * torch.from_numpy(labels[b]) + (torch.arange(p) % 8 == 0).long()
            pose[b, :, :3, :3] = R[lab4]
    return pose


def slab_check(dev, monkeypatch, P, layer, kinds, mode, slabs, q=64, layout='transposed', seed=50,
               plan_points=None, chans=None, tol=(2e-5, 2e-5, 5e-5), partial=False):
    """Run layer `layer` of the P-point backbone on len(kinds) clouds on the GPU and compare the slabs
    `slabs` = [(cloud, first query point)] of `q` query points with the oracle: y, dF, dW."""
    import synth_clouds
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    import vgtk.so3conv.functional as L
    monkeypatch.setattr(L, 'BACKWARD_MODE', mode)
    monkeypatch.setattr(L, 'X_LAYOUT', layout)
    B = len(kinds)
    c, o, radius, sigma = synth_clouds.backbone_layers(plan_points or P)[layer]
    if chans is not None:
        c, o = chans
    xyz_np, lab, _ = synth_clouds.laptop_batch(seed, B, P, partial=partial)
    gen = torch.Generator().manual_seed(seed)
    pose = make_poses(gen, kinds, lab, P)
    xyz = T(xyz_np)
    torch.manual_seed(2913)
    conv = sptk.InterSO3PoseConv(c, o, 1, 1, radius, sigma, NN, kanchor=NA, permute_modes=1)
    W_cpu = conv.basic_conv.W.detach().clone()
    anchors, kernels = conv.anchors.clone(), conv.kernels.clone()
    conv = conv.to(dev)
    dgen = torch.Generator(device=dev).manual_seed(seed)
    if c == 1:
        feats = torch.ones(B, 1, P, NA, device=dev)
    else:
        feats = torch.randn(B, c, P, NA, device=dev, generator=dgen)
    feats.requires_grad_(True)
    y = conv(zptk.SphericalPointCloudPose(xyz.to(dev), feats, None, pose.to(dev)))[3].feats
    assert y.shape == (B, o, P, NA)
    gy = torch.zeros_like(y)
    gy_slabs = []
    for (b, s) in slabs:
        g = torch.randn(1, o, q, NA, generator=gen)
        gy_slabs.append(g)
        gy[b, :, s:s + q] = g[0].to(dev)
    gF, gW = torch.autograd.grad(y, [feats, conv.basic_conv.W], gy)
    y_c, gF_c, gW_c = y.detach(), gF, gW.cpu()
    # oracle, slab by slab (weight gradient: sum over the slabs)
    gW_ref = torch.zeros_like(W_cpu)
    for (b, s), g in zip(slabs, gy_slabs):
        f_b = feats.detach()[b:b + 1].cpu().requires_grad_(True)
        W = W_cpu.clone().requires_grad_(True)
        fs = so3_ref.add_shadow_feature(f_b)
        res = so3_ref._poseconv_slab(xyz[b:b + 1, :, s:s + q].contiguous(), pose[b:b + 1, s:s + q].contiguous(),
                                     xyz[b:b + 1], pose[b:b + 1], fs, NN, anchors, kernels, radius, sigma, 1,
                                     kinds[b] == 'identity')       # identity poses: the 60x60 search returns arange
        y_ref = so3_ref.basic_so3conv(W, res[3])
        assert rel_err(y_c[b:b + 1, :, s:s + q].cpu().numpy(), y_ref.detach().numpy()) < tol[0], (b, s, 'forward')
        gf_ref, gw = torch.autograd.grad(y_ref, [f_b, W], g)
        gW_ref += gw
        # clouds are independent: the feature gradient of cloud b is this slab's (one slab per cloud)
        assert sum(1 for (bb, _) in slabs if bb == b) == 1
        assert rel_err(gF_c[b:b + 1].cpu().numpy(), gf_ref.numpy()) < tol[1], (b, s, 'dF')
    for b in range(B):            # clouds without a slab receive no gradient at all
        if all(bb != b for (bb, _) in slabs):
            assert float(gF_c[b].abs().max()) == 0.0
    assert rel_err(gW_c.numpy(), gW_ref.numpy()) < tol[2], 'dW'


# ---- config 2 / 3 shapes: 4096-point clouds, identity poses (what the shipped model feeds) --------------
@pytest.mark.parametrize('mode', ['inverse', 'dx'])
@pytest.mark.parametrize('layer', [0, 1, 2])
def test_4096_layers_identity_pose(dev, monkeypatch, layer, mode):
    slab_check(dev, monkeypatch, 4096, layer, ['identity', 'identity'], mode, [(0, 1000), (1, 3777)])


# ---- mixed batch: identity cloud + per-point random rotations + per-part rotations ----------------------
# (the lists kernel and the permuted-anchor MFMA kernels launched side by side, each skipping the other's
#  clouds by the nonident flag; the permuted clouds take so3_inter_mfma.hip / so3_inter_inv.hip)
@pytest.mark.parametrize('mode', ['inverse', 'dx'])
@pytest.mark.parametrize('layer', [1, 2])
def test_4096_layers_mixed_poses(dev, monkeypatch, layer, mode):
    slab_check(dev, monkeypatch, 4096, layer, ['identity', 'random', 'parts'], mode,
               [(0, 64), (1, 2048), (2, 4032)], q=32)


# ---- articulated input: one rotation per rigid part -> the dense product, one launch per part (vgtk/so3conv/functional.py
#      _PartsDense: row rotations in the k-side table, per-row anchor permutation of the stored operand), straight against the oracle
@pytest.mark.parametrize('layer', [1, 2])
def test_4096_layers_one_rotation_per_rigid_part(dev, monkeypatch, layer):
    import vgtk.so3conv.functional as L
    monkeypatch.setattr(L, 'BACKWARD_LOG', [])
    slab_check(dev, monkeypatch, 4096, layer, ['parts', 'identity', 'parts'], 'inverse', [(0, 64), (1, 2048), (2, 4032)], q=32)
    assert [r['regime'] for r in L.BACKWARD_LOG] == ['dense rows'] and L.BACKWARD_LOG[0].get('parts') == 2


def test_4096_deepest_layer_four_rigid_parts(dev, monkeypatch):
    """four parts per cloud (sizes ~ 7 : 1 : 7 : 1), three clouds: four launches of the product per direction, against the oracle"""
    import vgtk.so3conv.functional as L
    monkeypatch.setattr(L, 'BACKWARD_LOG', [])
    slab_check(dev, monkeypatch, 4096, 2, ['parts4', 'parts4', 'parts'], 'inverse', [(0, 64), (1, 2048), (2, 4032)], q=32)
    assert [r['regime'] for r in L.BACKWARD_LOG] == ['dense rows'] and L.BACKWARD_LOG[0].get('parts') == 4


# ---- config 4: 16 clouds of 4096 points per GPU --------------------------------------------------------
def test_4096_batch16_deepest_layer(dev, monkeypatch):
    slab_check(dev, monkeypatch, 4096, 2, ['identity'] * 16, 'auto', [(0, 0), (15, 4032)], seed=60)


# ---- config 5: 8192-point PARTIAL clouds (MotionHOIDatasetPartial shape: only the surfaces facing the camera, at twice
# the density of a complete cloud) and, for comparison, complete clouds of the same size ---------------------------------
@pytest.mark.parametrize('partial', [True, False])
@pytest.mark.parametrize('layer', [0, 1, 2])
def test_8192_layers(dev, monkeypatch, layer, partial):
    slab_check(dev, monkeypatch, 8192, layer, ['identity', 'parts'], 'auto', [(0, 5000), (1, 8128)], q=32, seed=70, partial=partial)


# ---- the 'dx' regime of the benchmark (512-point radii: more than a quarter of the rows referenced) -----
@pytest.mark.parametrize('layer', [1, 2])
def test_4096_points_with_512pt_radii_takes_dx_path(dev, monkeypatch, layer):
    slab_check(dev, monkeypatch, 4096, layer, ['identity', 'random'], 'auto', [(0, 100), (1, 3000)], q=32,
               plan_points=512, seed=80)


# ---- the permuted-pose production path with >= 16 channels, every layout x strategy -----------
@pytest.mark.parametrize('layout', ['transposed', 'blocked', 'reference'])
@pytest.mark.parametrize('mode', ['inverse', 'dx'])
def test_permuted_pose_path_16plus_channels_all_layouts(dev, monkeypatch, layout, mode):
    """C = 32 -> 64 at P = 256 with one identity-pose cloud and two permuted clouds in one batch; two slabs of
    64 query points per cloud (first and last quarter)."""
    import vgtk.so3conv.functional as L
    from vgtk import _hip
    if layout != 'reference':
        assert _hip.so3_inter_group_fwd_can_block(32, 256, 60, 24, True, True)      # the sweep is not vacuous
    for s in (0, 192):
        slab_check(dev, monkeypatch, 256, 1, ['identity', 'random', 'parts'], mode,
                   [(0, s), (1, s), (2, s)], q=64, layout=layout, plan_points=512, chans=(32, 64), seed=90)


def test_gemm_blocked_b_operand(dev):
    """eap_gemm_f32_xb / eap_gemm_f32_reduce_xb: B 'blocked by 4' along its long axis (include/eap_hip.h)."""
    from vgtk import _hip
    gen = torch.Generator().manual_seed(12)
    for (M, N, K, batch) in [(64, 240, 48, 2), (128, 1920, 384, 1), (40, 60, 24, 3)]:
        A = torch.randn(M, K, generator=gen)
        B = torch.randn(batch, K, N, generator=gen)
        ref = torch.matmul(A.double(), B.double())
        # element (row r, position x) at (x >> 2) * K * 4 + r * 4 + (x & 3)
        Bb = B.view(batch, K, N // 4, 4).permute(0, 2, 1, 3).contiguous()
        C = torch.empty(batch, M, N, device=dev)
        _hip.gemm(0, 0, M, N, K, A.to(dev), K, 0, Bb.to(dev), N, K * N, C, N, M * N, batch, b_blocked=True)
        assert rel_err(C.cpu().numpy(), ref.numpy()) < 1e-5
        # weight-gradient shape: sum_b G_b [M, N] X_b^T with X_b [K, N] blocked along N (transB = 1: rows = N?)
        G = torch.randn(batch, M, N, generator=gen)
        ref2 = torch.einsum('bmn,bkn->mk', G.double(), B.double())
        C2 = torch.empty(M, K, device=dev)
        _hip.gemm_reduce(0, 1, M, K, N, G.to(dev), N, M * N, Bb.to(dev), N, K * N, C2, K, batch, b_blocked=True)
        assert rel_err(C2.cpu().numpy(), ref2.numpy()) < 1e-5


# ---- config 3: chamfer at its shape ---------------------------------------------------------------------
def test_chamfer_config3_shape(dev):
    """[16,4096,3] <-> [16,4096,3] (...pn_38_multi_stage.py:L1744-1746): distances and first-min indices
    bit-exact, gradients 1e-5, against the oracle's C restatement of chamfer.cu."""
    import chamfer
    from extensions.chamfer_dist import ChamferFunction
    import synth_clouds
    a = np.ascontiguousarray(synth_clouds.laptop_batch(100, 16, 4096)[0].transpose(0, 2, 1))
    rng = np.random.default_rng(3)
    b = (a[:, rng.permutation(4096)] + rng.normal(0, 0.01, a.shape)).astype(np.float32)
    d1, d2, i1, i2 = native.chamfer_forward(a, b)
    t1 = T(a).to(dev).requires_grad_(True); t2 = T(b).to(dev).requires_grad_(True)
    o1, o2, j1, j2 = chamfer.forward(t1.detach(), t2.detach())
    np.testing.assert_array_equal(j1.cpu().numpy(), i1)
    np.testing.assert_array_equal(j2.cpu().numpy(), i2)
    np.testing.assert_array_equal(o1.cpu().numpy(), d1)
    np.testing.assert_array_equal(o2.cpu().numpy(), d2)
    g1 = rng.standard_normal(d1.shape).astype(np.float32); g2 = rng.standard_normal(d2.shape).astype(np.float32)
    u, v = ChamferFunction.apply(t1, t2)
    (u * T(g1).to(dev)).sum().add((v * T(g2).to(dev)).sum()).backward()
    r1, r2 = native.chamfer_backward(a, b, i1, i2, g1, g2)
    assert rel_err(t1.grad.cpu().numpy(), r1) < 1e-5 and rel_err(t2.grad.cpu().numpy(), r2) < 1e-5


@pytest.mark.parametrize('P,partial', [(4096, False), (8192, True)])
def test_bench_configuration_properties(dev, P, partial):
    """The benchmark configuration ITSELF (BASELINE config 1 at scale: 8 x 4096-point clouds, the 3-block inter backbone
    1 -> 64 -> 128 -> 512 with its fused BatchNorm + leaky_relu epilogues, exactly bench.py's model) through size-independent
    properties -- the oracle cannot run this size:
      * equivariance: rotating every cloud by anchor A_j permutes the anchor axis of the final feature map
        (out'[..., a] = out[..., pi(a)], A_pi(a) = A_j^T A_a), through all three layers and their batch statistics;
      * the backward against central differences of the forward: directional derivatives of a random linear functional of
        the output along random directions in the first and the last layer's weights (float64 accumulation of the loss);
      * run-to-run bit reproducibility of forward and gradients (no atomics on the path)."""
    import bench
    import synth_clouds
    import vgtk.so3conv.functional as L
    from vgtk.functional import anchor_group_tables
    B = 8                    # (8 x 4096 complete clouds: configs 2-4 per GPU; 8 x 8192 partial clouds: config 5 per GPU)
    xyz_np, _, pose_np = synth_clouds.laptop_batch(0, B, P, partial=partial)
    xyz, pose = T(xyz_np).to(dev), T(pose_np).to(dev)
    torch.manual_seed(2913)
    model = bench.Backbone(P).to(dev)
    anchors = torch.from_numpy(np.ascontiguousarray(L.get_anchors()))
    mult, inv = anchor_group_tables(anchors.numpy())

    with torch.no_grad():
        base = model(xyz, pose)
        assert tuple(base.shape) == (B, 512, P, 60) and torch.isfinite(base).all()
        j = 17
        rot = torch.einsum('ij,bjn->bin', anchors[j].to(dev), xyz).contiguous()
        moved = model(rot, pose)
        pi = torch.from_numpy(mult[inv[j]].astype(np.int64)).to(dev)
        scale = float(base.abs().max())
        worst = 0.0
        for b in range(B):                       # cloud by cloud: the permuted copy of the 4 GB map is never materialised
            worst = max(worst, float((moved[b] - base[b][..., pi]).abs().max()))
        assert worst < 1e-4 * scale, (worst, scale)
        del moved, rot

    gen = torch.Generator().manual_seed(5)
    probe = (torch.randn(B, 512, 1, 60, generator=gen) / (512 * 60)).to(dev)      # one weight per (cloud, channel, anchor), broadcast over points

    def loss_of():
        out = model(xyz, pose)
        return (out.double() * probe.double()).sum() / P, out

    grads = []
    L.BACKWARD_LOG = []
    for trial in range(2):
        model.zero_grad(set_to_none=True)
        loss, out = loss_of()
        loss.backward()
        grads.append([None if p.grad is None else p.grad.clone() for p in model.parameters()] + [out.detach().clone() if trial == 0 else out.detach()])
        del out
    for a, b in zip(grads[0], grads[1]):          # (the stand-in pose head's layer is not reached by this functional: no gradient)
        assert (a is None and b is None) or torch.equal(a, b), 'forward / backward of the benchmark configuration is not bit-reproducible'
    del grads[1]
    regimes, L.BACKWARD_LOG = L.BACKWARD_LOG, None
    print(f'backward regimes at {B} x {P}' + (' partial' if partial else '') + ': ' +
          '; '.join(f"{r['channels']}: {r['referenced_rows_max']} of {r['support_rows']} rows -> {r['regime']}" for r in regimes[:3]))
    assert len(regimes) >= 4 and all(r['referenced_rows_max'] <= r['support_rows'] for r in regimes)
    # the two deep layers reference a few hundred rows: they must take a re-associated backward (inverse lists, or the dense
    # product over the referenced rows where the width fills its blocks), not the
    # textbook dX route with its full-size GEMM and re-run grouping (a 2 x slower step when this regresses)
    deep = [r for r in regimes if r['channels'][0] >= 64]
    assert deep and all(r['regime'] in ('inverse lists', 'dense rows') and 0 < r['referenced_rows_max'] <= r['support_rows'] // 4 for r in deep), regimes
    params = dict(model.named_parameters())
    for name in ('convs.0.basic_conv.W', 'convs.2.basic_conv.W'):
        w = params[name]
        g = grads[0][list(params).index(name)]
        d = torch.randn(w.shape, generator=gen).to(dev)
        d = d / d.norm() * w.detach().norm()
        analytic = float((g.double() * d.double()).sum())
        eps = 2e-3
        vals = []
        with torch.no_grad():
            for sgn in (1.0, -1.0):
                w.add_(d, alpha=sgn * eps)
                vals.append(float(loss_of()[0]))
                w.add_(d, alpha=-sgn * eps)
        fd = (vals[0] - vals[1]) / (2 * eps)
        assert abs(fd - analytic) < 3e-2 * max(abs(analytic), 1e-12), (name, fd, analytic)
