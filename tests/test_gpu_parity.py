"""GPU parity tests: the HIP path (through the C ABI, via the vgtk operator layer) against the
CPU oracle and the golden fixtures generated from the reference.  Run with `-m gpu` on an MI355X.

Bars: integer outputs (neighbour lists, permutation indices) bit-exact; floating point within
the tolerance written next to each assert (the north star asks for 1e-4 relative on poses; the
operator-level bars here are tighter)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import native, so3_ref  # noqa: E402  (checker only)

T = torch.from_numpy
INTER_CASES = ['inter_pose_l0_identity', 'inter_pose_identity', 'inter_pose_random_pm1',
               'inter_pose_parts_pm1', 'inter_pose_random_pm0', 'inter_pose_bigball']


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def vg():
    import vgtk
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    import vgtk.so3conv.functional as L
    return vgtk, sptk, zptk, L


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


# ------------------------------------------------------------------------------------------------
# native ops (boundary B2)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('n,m,radius,nsample,dtype', [
    (512, 512, 0.08, 64, np.float32), (512, 512, 0.16, 64, np.float32), (700, 333, 0.3, 16, np.float32),
    (4096, 4096, 0.08, 64, np.float32), (4096, 4096, 0.45, 64, np.float32), (256, 256, 0.2, 32, np.float64),
    (1500, 40, 0.05, 8, np.float32)])
def test_ball_query_exact(dev, vg, n, m, radius, nsample, dtype):
    import synth_clouds
    import vgtk.cuda.grouping as G
    xyz = synth_clouds.laptop_batch(0, 2, n)[0].astype(dtype)
    q = xyz[:, :, :m].copy() if m <= n else xyz
    if m == 40:   # queries off the surface: some balls are empty, some hold exactly a few points
        q = q + np.asarray([0.03, 0.0, 0.0], dtype)[None, :, None]
    ref = native.ball_query(q, xyz, radius, nsample)
    got = G.ball_query(T(q).to(dev), T(xyz).to(dev), radius, nsample)
    assert got.dtype == torch.int32
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


def test_ball_query_known_answers(dev):
    """Hand-built cases of grouping_cuda_kernel.cu:L68-113 semantics."""
    import vgtk.cuda.grouping as G
    xyz = np.zeros((1, 3, 8), np.float32)
    xyz[0, 0] = [0, 1, 2, 3, 4, 5, 6, 7]
    q = np.zeros((1, 3, 3), np.float32)
    q[0, 0] = [0, 3, 100]
    # radius 1.5 around x=0 -> {0,1}; around x=3 -> {2,3,4}; around 100 -> {}
    got = G.ball_query(T(q).to(dev), T(xyz).to(dev), 1.5, 4).cpu().numpy()[0]
    np.testing.assert_array_equal(got[0], [0, 1, 0, 1])      # cnt=2 < 3: cyclic repeat
    np.testing.assert_array_equal(got[1], [2, 3, 4, 0])      # cnt=3 == nsample-1: last slot stays 0
    np.testing.assert_array_equal(got[2], [0, 0, 0, 0])      # empty ball
    # strict '<': a point at distance exactly r is excluded; more hits than nsample -> first in index order
    got = G.ball_query(T(q).to(dev), T(xyz).to(dev), 1.0, 2).cpu().numpy()[0]
    np.testing.assert_array_equal(got[0], [0, 0])            # only x=0 (x=1 is at d == r); cnt=1 == nsample-1
    got = G.ball_query(T(q).to(dev), T(xyz).to(dev), 10.0, 3).cpu().numpy()[0]
    np.testing.assert_array_equal(got[1], [0, 1, 2])


def test_gather_points(dev):
    import vgtk.cuda.gathering as GA
    rng = np.random.default_rng(0)
    pts = rng.standard_normal((3, 5, 97)).astype(np.float32)
    idx = rng.integers(0, 97, (3, 211)).astype(np.int32)
    out = GA.gather_points_forward(T(pts).to(dev), T(idx).to(dev)).cpu().numpy()
    np.testing.assert_array_equal(out, native.gather_points_forward(pts, idx))
    for dt, tol in ((np.float32, 1e-5), (np.float64, 1e-12)):
        g = rng.standard_normal((3, 5, 211)).astype(dt)
        got = GA.gather_points_backward(T(g).to(dev), T(idx).to(dev), 97).cpu().numpy()
        np.testing.assert_allclose(got, native.gather_points_backward(g, idx, 97), rtol=tol, atol=tol)


@pytest.mark.parametrize('dtype,tol', [(np.float32, 2e-6), (np.float64, 1e-13)])
@pytest.mark.parametrize('shape', [(2, 9, 11, 12, 5, 8, 3), (1, 17, 20, 60, 24, 64, 10), (2, 4, 4, 1, 3, 7, 2)])
def test_inter_zpconv(dev, dtype, tol, shape):
    import vgtk.cuda.zpconv as Z
    b, p, q, a, k, ann, c = shape
    rng = np.random.default_rng(1)
    idx = rng.integers(0, q, (b, p, a, k, ann)).astype(np.int32)
    w = rng.random((b, p, a, k, ann)).astype(dtype)
    feats = rng.standard_normal((b, c, q, a)).astype(dtype)
    out = Z.inter_zpconv_forward(T(idx).to(dev), T(w).to(dev), T(feats).to(dev)).cpu().numpy()
    ref = native.inter_zpconv_forward(idx, w, feats)
    assert rel_err(out, ref) < tol
    g = rng.standard_normal(ref.shape).astype(dtype)
    got = Z.inter_zpconv_backward(T(idx).to(dev), T(w).to(dev), T(g).to(dev), q).cpu().numpy()
    assert rel_err(got, native.inter_zpconv_backward(idx, w, g, q)) < 10 * tol


@pytest.mark.parametrize('case', ['shared', 'irregular', 'mixed', 'split'])
@pytest.mark.parametrize('shape', [(2, 9, 50, 60, 24, 64, 40), (1, 5, 33, 28, 17, 32, 16), (1, 3, 20, 60, 24, 64, 130),
                                   (3, 21, 40, 60, 24, 16, 64), (2, 6, 17, 12, 24, 24, 33)])
def test_inter_zpconv_row_kernel(dev, case, shape):
    """>= 8 channels: csrc/zpconv_mfma.hip (index check + matrix-core kernel for clouds whose index is one
    neighbour list per point) and csrc/zpconv_rows.hip (every other cloud).  'shared' = the index the Python
    layer builds (one neighbour list per point broadcast over anchors and kernel points);
    'irregular' = an arbitrary 5-D index (the workgroup falls back to the gather loop);
    'mixed' = some points of each kind; 'split' = the first cloud shared, the others irregular (both paths in
    one call, chosen per cloud on the device)."""
    import vgtk.cuda.zpconv as Z
    b, p, q, a, k, ann, c = shape
    rng = np.random.default_rng(7)
    shared = np.broadcast_to(rng.integers(0, q, (b, p, 1, 1, ann)), (b, p, a, k, ann)).astype(np.int32)
    random = rng.integers(0, q, (b, p, a, k, ann)).astype(np.int32)
    if case == 'shared':
        idx = shared.copy()
    elif case == 'irregular':
        idx = random
    elif case == 'split':
        idx = np.where((np.arange(b) == 0)[:, None, None, None, None], shared, random).astype(np.int32)
        if b > 1:
            idx[1] = shared[1]
            idx[1, p - 1, a - 1, k - 1, ann - 1] = (idx[1, p - 1, 0, 0, ann - 1] + 1) % q   # cloud 1: one deviating entry at the very end
    else:
        idx = np.where((np.arange(p) % 2 == 0)[None, :, None, None, None], shared, random).astype(np.int32)
        idx[0, 1, a - 1, k - 1, ann - 1] = (idx[0, 1, 0, 0, ann - 1] + 1) % q      # a single deviating entry in the last row
    w = rng.random((b, p, a, k, ann)).astype(np.float32)
    feats = rng.standard_normal((b, c, q, a)).astype(np.float32)
    out = Z.inter_zpconv_forward(T(idx).to(dev), T(w).to(dev), T(feats).to(dev)).cpu().numpy()
    ref = native.inter_zpconv_forward(idx, w, feats)
    assert rel_err(out, ref) < 2e-6
    # backward: csrc/zpconv_bwd.hip (products in forward order on the matrix cores + sorted sums) for the shared-index
    # clouds, the scatter kernel for the others; twice -> the atomics-free part is bit-reproducible
    g = rng.standard_normal(ref.shape).astype(np.float32)
    got = Z.inter_zpconv_backward(T(idx).to(dev), T(w).to(dev), T(g).to(dev), q)
    assert rel_err(got.cpu().numpy(), native.inter_zpconv_backward(idx, w, g, q)) < 1e-5
    if case == 'shared' and c % 4 == 0 and c >= 16:
        again = Z.inter_zpconv_backward(T(idx).to(dev), T(w).to(dev), T(g).to(dev), q)
        assert torch.equal(got, again)


@pytest.mark.parametrize('dtype,tol', [(np.float32, 2e-6), (np.float64, 1e-13)])
def test_intra_zpconv(dev, dtype, tol):
    import vgtk.cuda.zpconv as Z
    rng = np.random.default_rng(2)
    b, c, p, a_in, a_out, k, ann = 2, 5, 33, 60, 60, 4, 12
    idx = rng.integers(0, a_in, (a_out, ann)).astype(np.int32)
    w = rng.random((a_out, k, ann)).astype(dtype)
    feats = rng.standard_normal((b, c, p, a_in)).astype(dtype)
    out = Z.intra_zpconv_forward(T(idx).to(dev), T(w).to(dev), T(feats).to(dev)).cpu().numpy()
    ref = native.intra_zpconv_forward(idx, w, feats)
    assert rel_err(out, ref) < tol
    g = rng.standard_normal(ref.shape).astype(dtype)
    got = Z.intra_zpconv_backward(T(idx).to(dev), T(w).to(dev), T(g).to(dev), a_in).cpu().numpy()
    assert rel_err(got, native.intra_zpconv_backward(idx, w, g, a_in)) < 10 * tol


def test_native_ops_reject_host_tensors(vg):
    import vgtk.cuda.zpconv as Z
    with pytest.raises(RuntimeError):
        Z.inter_zpconv_forward(torch.zeros(1, 1, 1, 1, 1, dtype=torch.int32), torch.zeros(1, 1, 1, 1, 1),
                               torch.zeros(1, 1, 1, 1))


def test_zpconv_naive_golden(dev, vg, golden):
    _, _, zptk, _ = vg
    g = golden('zpconv_naive.npz')
    out = zptk.inter_zpconv_grouping_naive(T(g['inter_idx']).to(dev), T(g['inter_w']).to(dev), T(g['inter_feats']).to(dev))
    assert rel_err(out.cpu().numpy(), g['inter_out']) < 2e-6
    out = zptk.intra_zpconv_grouping_naive(T(g['intra_idx']).to(dev), T(g['intra_w']).to(dev), T(g['intra_feats']).to(dev))
    assert rel_err(out.cpu().numpy(), g['intra_out']) < 2e-6


# ------------------------------------------------------------------------------------------------
# dense contraction (fp32 MFMA GEMM)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K,batch', [(128, 256, 64, 1), (64, 128, 24, 2), (8, 3840, 144, 2), (512, 1920, 3072, 1),
                                        (100, 333, 50, 3), (7, 60, 5, 1), (130, 129, 17, 2)])
def test_gemm_variants(dev, M, N, K, batch):
    from vgtk import _hip
    gen = torch.Generator().manual_seed(3)
    A = torch.randn(M, K, generator=gen); B = torch.randn(batch, K, N, generator=gen)
    ref = torch.matmul(A.double(), B.double())
    tol = 2e-6 if K <= 256 else 1e-5   # sequential fp32 accumulation: error grows ~ sqrt(K) ulp
    Ad, Bd = A.to(dev), B.to(dev)
    C = torch.empty(batch, M, N, device=dev)
    _hip.gemm(0, 0, M, N, K, Ad, K, 0, Bd, N, K * N, C, N, M * N, batch)
    assert rel_err(C.cpu().numpy(), ref.numpy()) < tol
    # A^T variant: A stored [K, M]
    At = A.t().contiguous().to(dev)
    C.zero_()
    _hip.gemm(1, 0, M, N, K, At, M, 0, Bd, N, K * N, C, N, M * N, batch)
    assert rel_err(C.cpu().numpy(), ref.numpy()) < tol
    # B^T variant: B stored [N, K]
    Bt = B.transpose(1, 2).contiguous().to(dev)
    C.zero_()
    _hip.gemm(0, 1, M, N, K, Ad, K, 0, Bt, K, K * N, C, N, M * N, batch)
    assert rel_err(C.cpu().numpy(), ref.numpy()) < tol
    # batch-reduced (weight-gradient shape): sum_b A_b B_b^T with a long K
    A2 = torch.randn(batch, M, N, generator=gen); B2 = torch.randn(batch, K, N, generator=gen)
    ref2 = torch.einsum('bmn,bkn->mk', A2.double(), B2.double())
    C2 = torch.empty(M, K, device=dev)
    _hip.gemm_reduce(0, 1, M, K, N, A2.to(dev), N, M * N, B2.to(dev), N, K * N, C2, K, batch)
    assert rel_err(C2.cpu().numpy(), ref2.numpy()) < 3e-6


def test_gemm_is_transpose_detecting(dev):
    """A = I with an asymmetric B: catches a swapped C/D fragment layout."""
    from vgtk import _hip
    n = 128
    A = torch.eye(n, device=dev)
    B = (torch.arange(n * n, dtype=torch.float32).view(1, n, n) % 251).to(dev)
    C = torch.empty(1, n, n, device=dev)
    _hip.gemm(0, 0, n, n, n, A, n, 0, B, n, n * n, C, n, n * n, 1)
    assert torch.equal(C, B)


# ------------------------------------------------------------------------------------------------
# SO(3) conv stages against golden fixtures (generated by running the reference)
# ------------------------------------------------------------------------------------------------
def test_kernel_weights_golden(dev, vg, golden):
    _, _, _, L = vg
    g = golden('weights.npz')
    w = L.inter_so3conv_grouping_anchor(T(g['grouped_xyz']).to(dev), T(g['anchors']).to(dev),
                                        T(g['kernels']).to(dev), float(g['sigma']))
    # same operation order as the reference (sub, square, ordered sum, divide, relu): tight bar
    np.testing.assert_allclose(w.cpu().numpy(), g['inter_w'], rtol=0, atol=2e-6)


def _run_layer(vg, dev, g):
    _, sptk, zptk, L = vg
    C, O = g['feats'].shape[1], g['W'].shape[0]
    conv = sptk.InterSO3PoseConv(C, O, 1, 1, float(g['radius']), float(g['sigma']), int(g['nn']), kanchor=60,
                                 permute_modes=int(g['permute_modes'])).to(dev)
    with torch.no_grad():
        conv.basic_conv.W.copy_(T(g['W']))
    # the module's own constants must BE the reference's (a drift has to show here; tests/test_constants.py
    # pins them bit for bit).  The fixture stores the radius as float32, so kernel points rebuilt from it
    # may differ from the reference's (built from the Python float) in the last bit: checked to 1 ulp, then
    # the fixture's own values are used so the layer runs on exactly the reference's inputs.
    np.testing.assert_array_equal(conv.anchors.cpu().numpy(), g['anchors'])
    np.testing.assert_allclose(conv.kernels.cpu().numpy(), g['kernels'], rtol=3e-7, atol=0)
    with torch.no_grad():
        conv.kernels.copy_(T(g['kernels']))
    feats = T(g['feats']).to(dev).requires_grad_(True)
    x = zptk.SphericalPointCloudPose(T(g['xyz']).to(dev), feats, None, T(g['pose']).to(dev))
    return conv, feats, conv(x)


@pytest.mark.parametrize('blocked', ['transposed', 'blocked', 'reference'])
@pytest.mark.parametrize('mode', ['dx', 'inverse'])
@pytest.mark.parametrize('name', INTER_CASES)
def test_inter_pose_layer_golden(dev, vg, golden, name, mode, blocked, monkeypatch):
    """Forward + both feature-gradient strategies (dX + transposed grouping / re-associated
    inverse-list grouping of dY) against the reference's autograd; with the fused conv's
    intermediate in the blocked layout and in the reference layout."""
    _, _, _, L = vg
    monkeypatch.setattr(L, 'BACKWARD_MODE', mode)
    monkeypatch.setattr(L, 'X_LAYOUT', blocked)
    g = golden(name + '.npz')
    conv, feats, (inter_idx, inter_w, sample_idx, y) = _run_layer(vg, dev, g)
    assert inter_idx is None and sample_idx is None            # reference stride-1 return values
    assert tuple(inter_w.shape) == (2, 64, 60, 24, int(g['nn']))
    w = inter_w.materialize()
    np.testing.assert_allclose(w[:, :4].cpu().numpy(), g['inter_w_head'], rtol=0, atol=5e-6)
    assert y.feats.shape == g['out'].shape and y.pose.shape == g['pose'].shape
    assert rel_err(y.feats.detach().cpu().numpy(), g['out']) < 1e-5
    gf, gW = torch.autograd.grad(y.feats, [feats, conv.basic_conv.W], T(g['grad_out']).to(dev))
    assert rel_err(gf.cpu().numpy(), g['grad_feats']) < 1e-5
    assert rel_err(gW.cpu().numpy(), g['grad_W']) < 2e-5


@pytest.mark.parametrize('name', INTER_CASES)
def test_inter_group_stage_golden(dev, vg, golden, name):
    _, _, _, L = vg
    g = golden(name + '.npz')
    _, w, _, nf, _, _ = L.inter_so3poseconv_grouping_strided(
        T(g['xyz']).to(dev), T(g['pose']).to(dev), T(g['feats']).to(dev), 1, int(g['nn']),
        T(g['anchors']).to(dev), T(g['kernels']).to(dev), float(g['radius']), float(g['sigma']),
        permute_modes=int(g['permute_modes']))
    assert rel_err(nf[:, :, :, :8].cpu().numpy(), g['new_feats_head']) < 5e-6


def test_anchor_permutation_matches_reference_search(dev, vg, golden):
    """mult[r][a] against the oracle's 60x60 trace arg-max (so3conv/functional.py:L1199-1204)."""
    _, _, _, L = vg
    for name in ('inter_pose_random_pm1', 'inter_pose_parts_pm1', 'inter_pose_identity'):
        g = golden(name + '.npz')
        xyz, pose, A = T(g['xyz']), T(g['pose']), T(g['anchors'])
        perm = L.anchor_permutation_index(xyz.to(dev), pose.to(dev), int(g['nn']), A.to(dev), float(g['radius']))
        ref = so3_ref.inter_so3poseconv_grouping_strided(xyz, pose, T(g['feats']), int(g['nn']), A, T(g['kernels']),
                                                         float(g['radius']), float(g['sigma']), permute_modes=1)
        np.testing.assert_array_equal(perm.cpu().numpy(), ref['rotated_anchor_idx'].numpy())


def test_inter_nopose_golden(dev, vg, golden):
    _, sptk, zptk, _ = vg
    g = golden('inter_nopose.npz')
    conv = sptk.InterSO3Conv(5, 7, 1, 1, float(g['radius']), float(g['sigma']), int(g['nn']), kanchor=60).to(dev)
    with torch.no_grad():
        conv.basic_conv.W.copy_(T(g['W']))
    inter_idx, inter_w, sample_idx, y = conv(zptk.SphericalPointCloud(T(g['xyz']).to(dev), T(g['feats']).to(dev), None))
    np.testing.assert_array_equal(inter_idx.cpu().numpy(), g['inter_idx'])
    assert rel_err(y.feats.detach().cpu().numpy(), g['out']) < 1e-5


def test_intra_golden(dev, vg, golden):
    _, sptk, zptk, L = vg
    g = golden('intra.npz')
    conv = sptk.IntraSO3Conv(6, 9).to(dev)
    with torch.no_grad():
        conv.basic_conv.W.copy_(T(g['W']))
    assert np.array_equal(conv.intra_idx.cpu().numpy(), g['intra_idx'])
    feats = T(g['feats']).to(dev).requires_grad_(True)
    grouped = L.intra_so3conv_grouping(conv.intra_idx, feats)
    np.testing.assert_array_equal(grouped[:, :, :, :8].detach().cpu().numpy(), g['grouped_head'])
    y = conv(zptk.SphericalPointCloud(None, feats, None))
    assert rel_err(y.feats.detach().cpu().numpy(), g['out']) < 1e-5
    gf, gW = torch.autograd.grad(y.feats, [feats, conv.basic_conv.W], T(g['grad_out']).to(dev))
    assert rel_err(gf.cpu().numpy(), g['grad_feats']) < 1e-5
    assert rel_err(gW.cpu().numpy(), g['grad_W']) < 2e-5


# ------------------------------------------------------------------------------------------------
# larger shapes against the oracle (sizes the oracle finishes in seconds)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('P,C,O,layer', [(512, 1, 64, 0), (512, 64, 128, 1), (384, 128, 64, 2)])
def test_layer_vs_oracle_512(dev, vg, P, C, O, layer):
    import synth_clouds
    _, sptk, zptk, L = vg
    _, _, radius, sigma = synth_clouds.backbone_layers(512)[layer]
    xyz, _, pose = synth_clouds.laptop_batch(3, 2, P)
    torch.manual_seed(2913)
    conv = sptk.InterSO3PoseConv(C, O, 1, 1, radius, sigma, 64, kanchor=60, permute_modes=1)
    feats = torch.ones(2, 1, P, 60) if C == 1 else torch.randn(2, C, P, 60)
    ref = so3_ref.inter_so3poseconv_layer(T(xyz), T(pose), feats, conv.basic_conv.W.detach(), conv.anchors,
                                          conv.kernels, radius, sigma, 64, permute_modes=1, chunk=64,
                                          skip_perm_search=True)
    conv = conv.to(dev)
    y = conv(zptk.SphericalPointCloudPose(T(xyz).to(dev), feats.to(dev), None, T(pose).to(dev)))[3].feats
    assert rel_err(y.detach().cpu().numpy(), ref.numpy()) < 1e-5


def test_equivariance_property(dev, vg):
    """Rotating the cloud by anchor A_j permutes the anchor axis: out'[..., a] = out[..., pi(a)]
    with A_pi(a) = A_j^T A_a (probed on the reference)."""
    import synth_clouds
    _, sptk, zptk, L = vg
    from vgtk.functional import anchor_group_tables
    xyz = T(synth_clouds.laptop_batch(7, 1, 256)[0])
    anchors = torch.from_numpy(L.get_anchors())
    mult, inv = anchor_group_tables(anchors.numpy())
    torch.manual_seed(0)
    inter = sptk.InterSO3PoseConv(1, 16, 1, 1, 0.16, 0.0128, 32, kanchor=60, permute_modes=1).to(dev)
    intra = sptk.IntraSO3Conv(16, 8).to(dev)

    def run(x):
        pose = torch.eye(4).repeat(1, x.shape[2], 1, 1).to(dev)
        f = torch.ones(1, 1, x.shape[2], 60, device=dev)
        y = inter(zptk.SphericalPointCloudPose(x.to(dev).contiguous(), f, None, pose))[3]
        return y.feats, intra(y).feats

    a0, b0 = run(xyz)
    for j in (3, 17, 44):
        a1, b1 = run(torch.einsum('ij,bjn->bin', anchors[j], xyz))
        pi = torch.from_numpy(mult[inv[j]].astype(np.int64)).to(dev)     # A_pi(a) = A_j^T A_a
        assert rel_err(a1.detach().cpu().numpy(), a0[..., pi].detach().cpu().numpy()) < 2e-5
        assert rel_err(b1.detach().cpu().numpy(), b0[..., pi].detach().cpu().numpy()) < 2e-5


# ------------------------------------------------------------------------------------------------
# remaining native entry points
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('n,m', [(64, 128), (512, 4096), (1500, 777)])
def test_chamfer(dev, vg, n, m):
    from extensions.chamfer_dist import ChamferDistance, ChamferFunction
    rng = np.random.default_rng(5)
    x1 = rng.standard_normal((3, n, 3)).astype(np.float32)
    x2 = rng.standard_normal((3, m, 3)).astype(np.float32)
    d1, d2, i1, i2 = native.chamfer_forward(x1, x2)
    t1 = T(x1).to(dev).requires_grad_(True); t2 = T(x2).to(dev).requires_grad_(True)
    import chamfer
    o1, o2, j1, j2 = chamfer.forward(t1.detach(), t2.detach())
    np.testing.assert_array_equal(j1.cpu().numpy(), i1)
    np.testing.assert_array_equal(j2.cpu().numpy(), i2)
    np.testing.assert_array_equal(o1.cpu().numpy(), d1)
    np.testing.assert_array_equal(o2.cpu().numpy(), d2)
    g1 = rng.standard_normal(d1.shape).astype(np.float32); g2 = rng.standard_normal(d2.shape).astype(np.float32)
    a, b = ChamferFunction.apply(t1, t2)
    (a * T(g1).to(dev)).sum().add((b * T(g2).to(dev)).sum()).backward()
    r1, r2 = native.chamfer_backward(x1, x2, i1, i2, g1, g2)
    assert rel_err(t1.grad.cpu().numpy(), r1) < 1e-5 and rel_err(t2.grad.cpu().numpy(), r2) < 1e-5
    loss = ChamferDistance()(t1, t2)
    assert abs(loss.item() - (d1.mean() + d2.mean())) < 1e-5


@pytest.mark.parametrize('n,m', [(100, 17), (1024, 256), (4096, 64), (333, 333)])
def test_furthest_point_sampling(dev, n, m):
    import synth_clouds
    import vgtk.cuda.grouping as G
    xyz = synth_clouds.laptop_batch(11, 2, n)[0]
    xyz[:, :, 5] = 0.0    # a point at the origin is skipped by the reference (|x|^2 <= 1e-3)
    got = G.furthest_point_sampling(T(xyz).to(dev), m).cpu().numpy()
    np.testing.assert_array_equal(got, native.furthest_point_sampling(xyz, m))


def test_anchor_queries(dev):
    import vgtk.cuda.grouping as G
    rng = np.random.default_rng(6)
    gx = (rng.random((2, 3, 9, 8)).astype(np.float32) - 0.5)
    anchors = rng.standard_normal((12, 3)).astype(np.float32)
    anchors /= np.linalg.norm(anchors, axis=1, keepdims=True)
    kp = rng.random((5, 2)).astype(np.float32)
    z = torch.zeros(2, 9, dtype=torch.int32, device=dev)
    got = G.anchor_query(z, torch.zeros(2, 9, 8, dtype=torch.int32, device=dev), T(gx).to(dev), T(anchors).to(dev), T(kp).to(dev), 10)[0]
    ref = native.anchor_query(None, None, gx, anchors, kp, 10)[0]
    assert rel_err(got.cpu().numpy(), ref) < 1e-5
    centers = (rng.random((2, 3, 6)).astype(np.float32) - 0.5)
    frag = (rng.random((200, 3)).astype(np.float32) - 0.5)
    kpts = (rng.random((4, 7, 3)).astype(np.float32) - 0.5) * 0.3
    w, cnt = G.initial_anchor_query(T(centers).to(dev), T(frag).to(dev), T(kpts).to(dev), 0.4, 0.05)
    rw, rc = native.initial_anchor_query(centers, frag, kpts, 0.4, 0.05)
    np.testing.assert_array_equal(cnt.cpu().numpy(), rc)
    assert rel_err(w.cpu().numpy(), rw) < 1e-5


# ------------------------------------------------------------------------------------------------
# BASELINE.json sizes (4096 points): size-independent properties instead of an oracle run
# ------------------------------------------------------------------------------------------------
def _raw_group(dev, B, P, C, layer, valu=False):
    import ctypes
    import synth_clouds
    import vgtk.so3conv as sptk
    import vgtk.so3conv.functional as L
    import vgtk.cuda.grouping as G
    from vgtk import _hip
    c, o, r, s = synth_clouds.backbone_layers(P)[layer]
    xyz, _, pose = synth_clouds.laptop_batch(20, B, P)
    xyz, pose = T(xyz).to(dev), T(pose).to(dev)
    conv = sptk.InterSO3PoseConv(C, 8, 1, 1, r, s, 64, kanchor=60, permute_modes=1).to(dev)
    idx = G.ball_query(xyz, xyz, r, 64)
    mult, ident = L._group_tables(conv.anchors)
    rk = L.rotated_kernels(conv.anchors, conv.kernels)
    gx, nonident = _hip.so3_prep(xyz, xyz, idx, pose, pose, conv.anchors, ident)
    return idx, gx, rk, mult, s, ident


def test_full_size_group_mfma_equals_valu_and_is_linear(dev, vg):
    """4096-point clouds, C=64 (BASELINE config 2 shapes): the matrix-core grouping against the
    independent VALU kernel, linearity in the features, and run-to-run bit reproducibility."""
    import ctypes
    from vgtk import _hip
    B, P, C = 2, 4096, 64
    idx, gx, rk, mult, sigma, ident = _raw_group(dev, B, P, C, 1)
    gen = torch.Generator().manual_seed(9)
    f1 = torch.randn(B, C, P, 60, generator=gen).to(dev)
    f2 = torch.randn(B, C, P, 60, generator=gen).to(dev)
    x1 = _hip.so3_inter_group_fwd(f1, idx, gx, rk, mult, sigma)
    x1b = _hip.so3_inter_group_fwd(f1, idx, gx, rk, mult, sigma)
    assert torch.equal(x1, x1b)                                    # deterministic (no atomics)
    ref = torch.empty_like(x1)
    _hip.call('eap_so3_inter_group_fwd_valu_f32', ref, B, C, P, P, 64, 60, 24, ctypes.c_float(sigma), _hip._ptr(f1),
              _hip._ptr(idx), _hip._ptr(gx), _hip._ptr(rk), _hip._ptr(mult), _hip._ptr(ref))
    scale = ref.abs().max().item()
    assert (x1 - ref).abs().max().item() < 2e-5 * scale
    x2 = _hip.so3_inter_group_fwd(f2, idx, gx, rk, mult, sigma)
    x12 = _hip.so3_inter_group_fwd(2.0 * f1 - 0.5 * f2, idx, gx, rk, mult, sigma)
    assert (x12 - (2.0 * x1 - 0.5 * x2)).abs().max().item() < 2e-5 * scale


def test_full_size_layer_gradients_agree_between_strategies(dev, vg, monkeypatch):
    """4096 points, L2-layer radius (the hot-row regime): dF and dW from the re-associated
    (inverse-list) path equal dF from dX + transposed grouping and dW from dY X^T."""
    import synth_clouds
    _, sptk, zptk, L = vg
    P = 4096
    _, _, r, s = synth_clouds.backbone_layers(P)[2]
    xyz, _, pose = synth_clouds.laptop_batch(40, 2, P)
    torch.manual_seed(4)
    conv = sptk.InterSO3PoseConv(32, 64, 1, 1, r, s, 64, kanchor=60, permute_modes=1).to(dev)
    f0 = torch.randn(2, 32, P, 60, device=dev)
    gy = torch.randn(2, 64, P, 60, device=dev)
    grads = {}
    for mode in ('dx', 'inverse', 'auto'):
        monkeypatch.setattr(L, 'BACKWARD_MODE', mode)
        f = f0.clone().requires_grad_(True)
        y = conv(zptk.SphericalPointCloudPose(T(xyz).to(dev), f, None, T(pose).to(dev)))[3].feats
        grads[mode] = torch.autograd.grad(y, [f, conv.basic_conv.W], gy)
    scale = grads['dx'][0].abs().max().item()
    assert (grads['dx'][0] - grads['inverse'][0]).abs().max().item() < 2e-5 * scale
    assert torch.equal(grads['inverse'][0], grads['auto'][0])        # auto picks the inverse path here
    wscale = grads['dx'][1].abs().max().item()                      # dW: reduction over referenced rows vs over all points
    assert (grads['dx'][1] - grads['inverse'][1]).abs().max().item() < 5e-5 * wscale


def test_full_size_backward_is_the_adjoint(dev, vg):
    """<G(f), g> == <f, G^T(g)> at 4096 points for both backward implementations (slab / atomics)."""
    from vgtk import _hip
    B, P, C = 1, 4096, 32
    idx, gx, rk, mult, sigma, ident = _raw_group(dev, B, P, C, 2)
    gen = torch.Generator().manual_seed(10)
    f = torch.randn(B, C, P, 60, generator=gen).to(dev)
    x = _hip.so3_inter_group_fwd(f, idx, gx, rk, mult, sigma)
    g = torch.randn(x.shape, generator=gen).to(dev)
    lhs = (x.double() * g.double()).sum().item()
    for force_atomic in (False, True):
        _hip.FORCE_ATOMIC_BWD = force_atomic
        try:
            gf = _hip.so3_inter_group_bwd(g, idx, gx, rk, mult, sigma, P, ident)
        finally:
            _hip.FORCE_ATOMIC_BWD = False
        rhs = (f.double() * gf.double()).sum().item()
        assert abs(lhs - rhs) < 1e-5 * max(abs(lhs), 1.0), (force_atomic, lhs, rhs)
    a = _hip.so3_inter_group_bwd(g, idx, gx, rk, mult, sigma, P, ident)
    b = _hip.so3_inter_group_bwd(g, idx, gx, rk, mult, sigma, P, ident)
    assert torch.equal(a, b)                                       # slab backward is bit-reproducible


def test_full_size_ball_query_properties(dev):
    """4096 points, all three radii: every returned neighbour is inside the ball, lists are sorted
    until the padding starts, and the query itself is among its neighbours when it fits."""
    import synth_clouds
    import vgtk.cuda.grouping as G
    xyz = T(synth_clouds.laptop_batch(30, 4, 4096)[0]).to(dev)
    for (_, _, r, _) in synth_clouds.backbone_layers(4096):
        idx = G.ball_query(xyz, xyz, r, 64).long()
        nb = torch.gather(xyz.unsqueeze(2).expand(-1, -1, 4096, -1), 3, idx.unsqueeze(1).expand(-1, 3, -1, -1))
        d2 = ((nb - xyz.unsqueeze(3)) ** 2).sum(1)
        outside = d2 >= r * r * (1 + 1e-5)
        # the reference's padding quirk: with exactly nsample-1 hits the last slot is index 0,
        # which need not be inside the ball (grouping_cuda_kernel.cu:L100-105)
        assert not outside[:, :, :-1].any()
        assert (idx[:, :, -1][outside[:, :, -1]] == 0).all()
        first = idx[:, :, 0]
        assert (first <= torch.arange(4096, device=dev)).all()      # index order: first hit <= self


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(3, 5, 7, 12), (2, 64, 33, 60), (1, 3, 2051, 4)])
@pytest.mark.parametrize('training', [True, False])
def test_batchnorm_leaky_relu_block_epilogue(dev, vg, shape, training):
    """csrc/bn_act.hip against torch's BatchNorm2d + leaky_relu (fp32 torch reference of the same
    op on the same device; base_so3poseconv.py:L214-221): outputs, input / affine gradients and
    running statistics."""
    import vgtk.so3conv as sptk
    torch.manual_seed(3)
    b, c, p, a = shape
    x = (torch.randn(b, c, p, a, device=dev) * 2.0 + 3.0)
    g = torch.randn(b, c, p, a, device=dev)
    ref = torch.nn.BatchNorm2d(c).to(dev)
    fused = sptk.BatchNormLeakyReLU(c, negative_slope=0.01).to(dev)
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5); ref.bias.uniform_(-1, 1)
        ref.running_mean.uniform_(2.5, 3.5); ref.running_var.uniform_(3.0, 5.0)
    fused.load_state_dict(ref.state_dict())
    ref.train(training); fused.train(training)
    xr = x.clone().requires_grad_(True); xf = x.clone().requires_grad_(True)
    yr = torch.nn.functional.leaky_relu(ref(xr), 0.01)
    yf = fused(xf)
    yr.backward(g); yf.backward(g)
    def close(u, v, tol):
        assert float((u - v).detach().abs().max()) <= tol * float(v.detach().abs().max() + 1e-12)
    close(yf, yr, 2e-6)
    close(xf.grad, xr.grad, 2e-5)
    close(fused.weight.grad, ref.weight.grad, 2e-5)
    close(fused.bias.grad, ref.bias.grad, 2e-5)
    close(fused.running_mean, ref.running_mean, 1e-6)
    close(fused.running_var, ref.running_var, 1e-5)
    assert int(fused.num_batches_tracked) == int(ref.num_batches_tracked)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [
    # b, c, p, n_sup, nn, na, ks
    (2, 40, 37, 50, 20, 12, 24),      # one anchor group of 12, channel tail, nn not a multiple of the chunk
    (1, 33, 19, 19, 64, 28, 17),      # 28 anchors: one group, odd piece count, 17 kernel points
    (1, 64, 21, 30, 8, 64, 32),       # 64 anchors: two groups of 32, rotated LDS image, 32 kernel points
    (2, 17, 70, 70, 24, 60, 24),      # 60 anchors: groups 32 + 28; 3 chunks per row, rows streamed 8 at a time
    (1, 96, 9, 40, 16, 60, 24),       # fewer rows than a run
])
def test_entry_list_grouping_kernel_shapes(dev, vg, shape):
    """csrc/so3_inter_lists.hip (forward dispatcher for >= 16 channels, and the backward's Z) against
    the VALU kernel / a dense torch evaluation with materialised weights, on anchor counts, kernel
    sizes and list lengths the golden layers do not cover; shadow rows (idx == n_sup) included."""
    from vgtk import _hip
    b, c, p, n, nn, na, ks = shape
    torch.manual_seed(5)
    feats = torch.randn(b, c, n, na, device=dev)
    idx = torch.randint(0, n + 1, (b, p, nn), device=dev, dtype=torch.int32)      # n = shadow row
    gx = torch.zeros(b, p, nn, 4, device=dev)
    gx[..., :3] = torch.randn(b, p, nn, 3, device=dev) * 0.05
    rk = torch.randn(na, ks, 3, device=dev) * 0.05
    sigma = 0.01
    got = _hip.so3_inter_group_fwd(feats, idx, gx, rk, None, sigma)
    ref = torch.empty_like(got)
    _hip.call('eap_so3_inter_group_fwd_valu_f32', ref, b, c, p, n, nn, na, ks, _hip._F32(sigma), _hip._ptr(feats),
              _hip._ptr(idx), _hip._ptr(gx), _hip._ptr(rk), _hip._ptr(None), _hip._ptr(ref))
    assert rel_err(got.cpu().numpy(), ref.cpu().numpy()) < 5e-6

    # backward: Z over inverse lists vs dense scatter of dY * w
    import vgtk.so3conv.functional as L
    o = c
    gy = torch.randn(b, o, p, na, device=dev)
    idx_v = idx.clamp(max=n - 1)                                                   # lists never hold shadow rows
    rows, off, cnt, ent_p, ent_gx, rcap, all_ident = L._inverse_lists(idx_v, gx, n, 0, None)
    z = _hip.so3_inter_group_inv(gy, rows, off, cnt, ent_p, ent_gx, rk, None, sigma, nn)
    w = _hip.so3_inter_weights(gx, rk, sigma)                                      # [b,p,na,ks,nn]
    dense = torch.zeros(b, o, ks, n, na, device=dev, dtype=torch.float64)
    contrib = torch.einsum('bopa,bpakn->bokpna', gy.double(), w.double())          # [b,o,ks,p,nn,na]
    flat_q = idx_v.long().reshape(b, 1, 1, p * nn, 1).expand(b, o, ks, p * nn, na)
    dense.scatter_add_(3, flat_q, contrib.reshape(b, o, ks, p * nn, na))
    for bi in range(b):
        for ri in range(rcap):
            q = int(rows[bi, ri])
            want = dense[bi, :, :, q] if q >= 0 else torch.zeros_like(dense[bi, :, :, 0])
            assert rel_err(z[bi, :, :, ri].double().cpu().numpy(), want.cpu().numpy() + 0.0) < 1e-5 or float(want.abs().max()) == 0.0


@pytest.mark.gpu
def test_instancenorm_leaky_relu_block_epilogue(dev, vg):
    """The intra blocks' InstanceNorm2d(affine=False) + leaky_relu (base_so3poseconv.py:L88) through
    the same kernels, against torch."""
    import vgtk.so3conv as sptk
    torch.manual_seed(4)
    x = torch.randn(3, 7, 21, 60, device=dev) * 1.5 - 2.0
    g = torch.randn_like(x)
    xr = x.clone().requires_grad_(True); xf = x.clone().requires_grad_(True)
    yr = torch.nn.functional.leaky_relu(torch.nn.InstanceNorm2d(7, affine=False)(xr), 0.01)
    yf = sptk.InstanceNormLeakyReLU(7)(xf)
    yr.backward(g); yf.backward(g)
    assert float((yf - yr).detach().abs().max()) <= 2e-6 * float(yr.detach().abs().max())
    assert float((xf.grad - xr.grad).abs().max()) <= 2e-5 * float(xr.grad.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(2, 24, 40, 37), (1, 130, 8, 300), (1, 64, 512, 64)])
def test_intra_conv_implicit_gemm(dev, vg, shape):
    """IntraSO3Conv = implicit GEMM in both directions (eap_so3_intra_conv_f32 forward and feature gradient, channel-
    sliced weight gradient: no [B,C,12,P,A] tensor) against the materialised intra_so3conv_grouping + BasicSO3Conv
    formulation (so3conv/functional.py:L2553-2602, modules.py:L48-55): output, dF, dW."""
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    import vgtk.so3conv.functional as L
    b, c, o, p = shape
    torch.manual_seed(11)
    conv = sptk.IntraSO3Conv(c, o).to(dev)
    feats = torch.randn(b, c, p, 60, device=dev)
    g = torch.randn(b, o, p, 60, device=dev)
    f1 = feats.clone().requires_grad_(True)
    fast = conv(zptk.SphericalPointCloud(torch.zeros(b, 3, p, device=dev), f1, conv.anchors)).feats
    gf1, gw1 = torch.autograd.grad(fast, [f1, conv.basic_conv.W], g)
    f2 = feats.clone().requires_grad_(True)
    slow = conv.basic_conv(L.intra_so3conv_grouping(conv.intra_idx, f2))          # materialised path
    gf2, gw2 = torch.autograd.grad(slow, [f2, conv.basic_conv.W], g)
    assert rel_err(fast.detach().cpu().numpy(), slow.detach().cpu().numpy()) < 1e-5
    assert rel_err(gf1.cpu().numpy(), gf2.cpu().numpy()) < 1e-5
    assert rel_err(gw1.cpu().numpy(), gw2.cpu().numpy()) < 2e-5
    with torch.no_grad():
        assert torch.equal(conv(zptk.SphericalPointCloud(None, feats, conv.anchors)).feats, fast.detach())


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(2, 3, 5, 7, 33), (1, 2, 60, 64, 300)])
def test_orbit_reconstruction_distances(dev, vg, shape):
    """the orbit-selection distances (…pn_38_multi_stage.py:L1341-1361) as a
    60-way batched chamfer on the chamfer kernels, against the reference's materialised
    [B,S,A,M,N] formulation (oracle/orbit_ref.py): values and the gradient reaching the
    reconstructions.  One slot is left empty (its masked minima are the 99999 constant)."""
    from extensions.chamfer_dist import orbit_reconstruction_distances
    from oracle import orbit_ref
    b, s, a, m, n = shape
    torch.manual_seed(9)
    recon = torch.randn(b, s, a, m, 3) * 0.3
    ori = torch.randn(b, 3, n) * 0.3
    slot = torch.randint(0, s - 1 if s > 2 else s, (b, n))          # for s > 2 the last slot stays empty
    labels = torch.nn.functional.one_hot(slot, s).float()           # [B,N,S]
    rc = recon.clone().requires_grad_(True)
    want = orbit_ref.orbit_reconstruction_distances(rc, ori, labels)
    rg = recon.clone().to(dev).requires_grad_(True)
    got = orbit_reconstruction_distances(rg, ori.to(dev), labels.to(dev))
    for u, v in zip(got, want):
        assert u.shape == v.shape
        np.testing.assert_allclose(u.detach().cpu().numpy(), v.detach().numpy(), rtol=1e-6, atol=1e-7)
    wts = [torch.randn_like(v) for v in want]
    finite = [torch.where(v.detach() < 9e4, w, torch.zeros_like(w)) for v, w in zip(want, wts)]   # constants carry no gradient
    sum((v * w).sum() for v, w in zip(want, finite)).backward()
    sum((u * w.to(dev)).sum() for u, w in zip(got, finite)).backward()
    np.testing.assert_allclose(rg.grad.cpu().numpy(), rc.grad.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_three_block_backbone_end_to_end_vs_oracle(dev, vg):
    """The whole hot path composed as bench.py runs it -- 3 x (InterSO3PoseConv -> BatchNorm2d ->
    leaky_relu) with the fused grouping, the transposed intermediate, the library contraction and the
    fused block epilogue -- against the torch-CPU oracle layer by layer (256-point clouds, channel
    plan 1 -> 16 -> 32 -> 24): the feature maps the pose head would read, within 1e-4 of their
    magnitude, and the gradients of the first two layers' weights."""
    import synth_clouds
    _, sptk, zptk, L = vg
    P = 256
    xyz, _, pose = synth_clouds.laptop_batch(11, 2, P)
    plan = [(1, 16), (16, 32), (32, 24)]
    params = synth_clouds.backbone_layers(512)
    torch.manual_seed(5)
    convs = [sptk.InterSO3PoseConv(c, o, 1, 1, params[i][2], params[i][3], 64, kanchor=60, permute_modes=1) for i, (c, o) in enumerate(plan)]
    bns = [torch.nn.BatchNorm2d(o) for _, o in plan]
    for bn in bns:
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.2, 0.2)
    # oracle: CPU, reference formulation
    Ws = [cv.basic_conv.W.detach().clone().requires_grad_(True) for cv in convs]
    f = torch.ones(2, 1, P, 60)
    outs_ref = []
    for i, cv in enumerate(convs):
        y = so3_ref.inter_so3poseconv_layer(T(xyz), T(pose), f, Ws[i], cv.anchors, cv.kernels, params[i][2], params[i][3], 64,
                                            permute_modes=1, chunk=64, skip_perm_search=True)
        f = torch.nn.functional.leaky_relu(bns[i](y), 0.01)
        outs_ref.append(f)
    g_out = torch.randn_like(f)
    gW_ref = torch.autograd.grad(f, Ws[:2], g_out)
    # build under test
    fused = []
    for i, (_, o) in enumerate(plan):
        m = sptk.BatchNormLeakyReLU(o, negative_slope=0.01)
        bn_fresh = torch.nn.BatchNorm2d(o)
        bn_fresh.load_state_dict({k: (v if 'running' not in k and 'num_batches' not in k else bn_fresh.state_dict()[k]) for k, v in bns[i].state_dict().items()})
        m.load_state_dict(bn_fresh.state_dict())
        fused.append(m.to(dev))
    convs = [cv.to(dev) for cv in convs]
    x = zptk.SphericalPointCloudPose(T(xyz).to(dev), torch.ones(2, 1, P, 60, device=dev), None, T(pose).to(dev))
    outs = []
    for cv, m in zip(convs, fused):
        _, _, _, x = cv(x)
        x = zptk.SphericalPointCloudPose(x.xyz, m(x.feats), x.anchors, x.pose)
        outs.append(x.feats)
    for got, want in zip(outs, outs_ref):
        assert rel_err(got.detach().cpu().numpy(), want.detach().numpy()) < 1e-4
    gW = torch.autograd.grad(outs[-1], [convs[0].basic_conv.W, convs[1].basic_conv.W], g_out.to(dev))
    for got, want in zip(gW, gW_ref):
        assert rel_err(got.cpu().numpy(), want.numpy()) < 2e-4


@pytest.mark.gpu
def test_inter_zpconv_matrix_path_edges(dev):
    """csrc/zpconv_mfma.hip / zpconv_bwd.hip at the edges of their shape range: a single point, fewer points than a
    run of 8, a channel count that is not a multiple of 64, every neighbour the same support point (one inverse list
    holding every entry), and an empty batch."""
    import vgtk.cuda.zpconv as Z
    from vgtk import _hip as _h
    rng = np.random.default_rng(11)
    # (in a `make EXPERIMENTS=1` library the same cases also run on the 32-neighbour re-cut of tools/experiments/kernels/zpconv_mfma2.hip)
    experiments = hasattr(_h.lib, 'eap_inter_zpconv_fwd_kernel')
    for which_fwd in ((1, 2) if experiments else (1,)):
        if experiments:
            _h.lib.eap_inter_zpconv_fwd_kernel(which_fwd)
        for (b, p, q, a, k, ann, c) in ((1, 1, 5, 60, 24, 64, 64), (2, 5, 9, 60, 24, 16, 80), (1, 11, 3, 28, 24, 32, 16),
                                      (1, 1, 1, 4, 24, 4, 16), (3, 3, 5, 60, 24, 12, 16),       # scratch chunks far from 256-byte multiples
                                      # csrc/zpconv_mfma2.hip (64 / 128 neighbours): runs of 8 + 8 + 3 points with a partial second
                                      # channel slice; 28 anchors (a wave without anchors) at 128 neighbours; 17 kernel points
                                      (2, 19, 40, 60, 24, 64, 80), (1, 9, 30, 28, 24, 128, 16), (2, 8, 50, 60, 17, 64, 64)):
            idx = np.broadcast_to(rng.integers(0, q, (b, p, 1, 1, ann)), (b, p, a, k, ann)).astype(np.int32).copy()
            if p == 5:
                idx[:] = 2                                          # all entries reference support point 2
            w = rng.random((b, p, a, k, ann)).astype(np.float32)
            feats = rng.standard_normal((b, c, q, a)).astype(np.float32)
            out = Z.inter_zpconv_forward(T(idx).to(dev), T(w).to(dev), T(feats).to(dev)).cpu().numpy()
            ref = native.inter_zpconv_forward(idx, w, feats)
            assert rel_err(out, ref) < 2e-6, (b, p, q, a, k, ann, c)
            g = rng.standard_normal(ref.shape).astype(np.float32)
            got = Z.inter_zpconv_backward(T(idx).to(dev), T(w).to(dev), T(g).to(dev), q).cpu().numpy()
            assert rel_err(got, native.inter_zpconv_backward(idx, w, g, q)) < 1e-5, (b, p, q, a, k, ann, c)
    if experiments:
        _h.lib.eap_inter_zpconv_fwd_kernel(1)
    # a benchmark-like shape, and a batch that mixes a broadcast-index cloud with an arbitrary 5-D index (served by
    # csrc/zpconv_rows.hip)
    from vgtk import _hip
    b, p, q, a, k, ann, c = 2, 64, 64, 60, 24, 64, 128
    idx = np.broadcast_to(rng.integers(0, q, (b, p, 1, 1, ann)), (b, p, a, k, ann)).astype(np.int32).copy()
    w = rng.random((b, p, a, k, ann)).astype(np.float32)
    feats = rng.standard_normal((b, c, q, a)).astype(np.float32)
    outs = {}
    for which in ((2, 1) if experiments else (1,)):
        was = _hip.lib.eap_inter_zpconv_fwd_kernel(which) if experiments else 1
        try:
            outs[which] = Z.inter_zpconv_forward(T(idx).to(dev), T(w).to(dev), T(feats).to(dev)).cpu().numpy()
        finally:
            if experiments:
                _hip.lib.eap_inter_zpconv_fwd_kernel(was)
    ref = native.inter_zpconv_forward(idx, w, feats)
    for got in outs.values():
        assert rel_err(got, ref) < 2e-6
    idx[1, 3, 7, 5, :] = (idx[1, 3, 7, 5, :] + 1) % q                 # cloud 1: one (a, k) row differs -> arbitrary-index path
    out = Z.inter_zpconv_forward(T(idx).to(dev), T(w).to(dev), T(feats).to(dev)).cpu().numpy()
    assert rel_err(out, native.inter_zpconv_forward(idx, w, feats)) < 2e-6
    e = Z.inter_zpconv_forward(torch.zeros(0, 4, 60, 24, 64, dtype=torch.int32, device=dev), torch.zeros(0, 4, 60, 24, 64, device=dev),
                               torch.zeros(0, 16, 4, 60, device=dev))
    assert tuple(e.shape) == (0, 16, 24, 4, 60)


@pytest.mark.gpu
def test_native_known_answers(dev):
    """The hand-derived known-answer cases of tests/test_oracle_native.py (ball query padding rules and strict radius,
    gather with repeated indices, furthest-point-sampling tie / skip rules, zpconv with per-(a,k) indices, chamfer
    first-minimum ties within and across the reference's 512-point tiles) asserted on the HIP kernels themselves."""
    import chamfer
    import vgtk.cuda.gathering as GA
    import vgtk.cuda.grouping as G
    import vgtk.cuda.zpconv as Z
    import test_oracle_native as K

    def on_gpu(fn):
        def run(*args):
            conv = [T(np.ascontiguousarray(a)).to(dev) if isinstance(a, np.ndarray) else a for a in args]
            out = fn(*conv)
            return [o.cpu().numpy() for o in out] if isinstance(out, (list, tuple)) else out.cpu().numpy()
        return run

    K.known_ball_query(on_gpu(G.ball_query))
    K.known_gather(on_gpu(GA.gather_points_forward), on_gpu(GA.gather_points_backward))
    K.known_fps(on_gpu(G.furthest_point_sampling))
    K.known_zpconv(on_gpu(Z.inter_zpconv_forward), on_gpu(Z.inter_zpconv_backward), on_gpu(Z.intra_zpconv_forward), on_gpu(Z.intra_zpconv_backward))
    K.known_chamfer(on_gpu(chamfer.forward), on_gpu(chamfer.backward))


def ref_valu(feats, idx, gx, rk, sigma, b, c, p, n, nn, na, ks, dev):
    from vgtk import _hip
    ref = torch.empty(b, c, ks, p, na, device=dev)
    _hip.call('eap_so3_inter_group_fwd_valu_f32', ref, b, c, p, n, nn, na, ks, _hip._F32(sigma), _hip._ptr(feats),
              _hip._ptr(idx), _hip._ptr(gx), _hip._ptr(rk), _hip._ptr(None), _hip._ptr(ref))
    return ref


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [
    (2, 64, 37, 37, 64, 60, 24),      # the shipped geometry: 64 channels = one full block, anchor groups 16+16+16+12
    (1, 128, 70, 50, 24, 60, 24),     # two blocks, 3 chunks per row, rows streamed 8 at a time, shadow rows
    (1, 112, 19, 19, 20, 28, 17),     # 112 = 64 + 48: a partial second tile; 28 anchors (16 + 12); nn not a multiple of 8; 17 kernel points
    (1, 192, 9, 40, 16, 64, 32),      # 64 anchors (4 full groups), 32 kernel points, fewer rows than a run
    (3, 64, 33, 33, 8, 16, 24),       # 16 anchors: one group; batch 3 (slice count not a multiple of 8)
])
def test_two_tile_grouping_kernel_equals_one_tile_kernel(dev, vg, shape):
    """csrc/so3_inter_lists2.hip (two channel tiles per wave, 64-channel blocks) against csrc/so3_inter_lists.hip and the
    VALU kernel: forward in the reference and the transposed layout, and the backward's Z over inverse lists.  Every output
    element accumulates the same products in the same order in both matrix kernels, so they must agree BIT FOR BIT."""
    import vgtk.so3conv.functional as L
    from vgtk import _hip
    b, c, p, n, nn, na, ks = shape
    torch.manual_seed(7)
    feats = torch.randn(b, c, n, na, device=dev)
    idx = torch.randint(0, n + 1, (b, p, nn), device=dev, dtype=torch.int32)      # n = shadow row
    gx = torch.zeros(b, p, nn, 4, device=dev)
    gx[..., :3] = torch.randn(b, p, nn, 3, device=dev) * 0.05
    rk = torch.randn(na, ks, 3, device=dev) * 0.05
    sigma = 0.01
    gy = torch.randn(b, c, p, na, device=dev)
    idx_v = idx.clamp(max=n - 1)
    rows, off, cnt, ent_p, ent_gx, rcap, _ = L._inverse_lists(idx_v, gx, n, 0, None)
    out = {}
    assert _hip.lib.eap_so3_group_lists_tiles(0) == 2
    try:
        modes = (1, 2, 3) if _hip.lib.eap_so3_group_lists_tiles(3) == 3 else (1, 2)     # 3: `make EXPERIMENTS=1` libraries only
        for tiles in modes:
            assert _hip.lib.eap_so3_group_lists_tiles(tiles) == tiles
            out[tiles] = (_hip.so3_inter_group_fwd(feats, idx, gx, rk, None, sigma),
                          _hip.so3_inter_group_fwd(feats, idx, gx, rk, None, sigma, blocked=2),
                          _hip.so3_inter_group_inv(gy, rows, off, cnt, ent_p, ent_gx, rk, None, sigma, nn))
    finally:
        _hip.lib.eap_so3_group_lists_tiles(2)
    for a, bb, what in zip(out[1], out[2], ('forward', 'forward, transposed', 'backward Z')):
        assert torch.equal(a, bb), what
    # the transposed forward with its row end as dword stores (MFMA operands not exchanged): the same bits
    was = _hip.lib.eap_so3_group_lists_store16(0)
    try:
        narrow = _hip.so3_inter_group_fwd(feats, idx, gx, rk, None, sigma, blocked=2)
    finally:
        _hip.lib.eap_so3_group_lists_store16(was)
    assert was == 1 and torch.equal(narrow, out[2][1]), 'forward, transposed: 16-byte row-end stores differ from dword stores'
    # mode 3 (tools/experiments/kernels/so3_inter_lists3.hip: the same products on the bf16 matrix cores from exact 3 x bf16 splits of the fp32
    # operands, fp32 accumulation) agrees with the fp32-MFMA kernels to fp32 rounding
    if 3 in out:
        for a, bb, what in zip(out[2], out[3], ('forward', 'forward, transposed', 'backward Z')):
            assert rel_err(bb.cpu().numpy(), a.cpu().numpy()) < 1e-6, what
        assert rel_err(out[3][0].cpu().numpy(), ref_valu(feats, idx, gx, rk, sigma, b, c, p, n, nn, na, ks, dev).cpu().numpy()) < 5e-6
    ref = torch.empty_like(out[2][0])
    _hip.call('eap_so3_inter_group_fwd_valu_f32', ref, b, c, p, n, nn, na, ks, _hip._F32(sigma), _hip._ptr(feats),
              _hip._ptr(idx), _hip._ptr(gx), _hip._ptr(rk), _hip._ptr(None), _hip._ptr(ref))
    assert rel_err(out[2][0].cpu().numpy(), ref.cpu().numpy()) < 5e-6
    # transposed layout holds the same numbers: [b, p*na + a, c*ks + k]
    xt = out[2][1].view(b, p * na, c * ks)
    assert torch.equal(xt.view(b, p, na, c, ks).permute(0, 3, 4, 1, 2), out[2][0])
    # ... and with its columns in the kernel's store order (LAYOUT 4: 1 KB store runs) the same bits at the positions
    # eap_so3_group_fwd_tp_columns names
    if _hip.so3_group_fwd_tp_takes(c, na, ks):
        tp = _hip.so3_inter_group_fwd(feats, idx, gx, rk, None, sigma, blocked=2, store_order=True).view(b, p * na, c * ks)
        cols = _hip.so3_group_fwd_tp_columns(c, ks, dev)
        assert sorted(cols.tolist()) == list(range(c * ks))
        assert torch.equal(tp, xt.index_select(2, cols))
    else:
        assert c % 64 != 0 or ks % 8 != 0


@pytest.mark.gpu
def test_native_zpconv_at_the_bench_shape(dev):
    """vgtk.cuda.zpconv.inter_zpconv_forward / _backward at the shape bench.py's `zpconv_roofline` times (8 x 4096 points,
    C = 64, 60 anchors, 24 kernel points, 64 neighbours: 12 GB of indices, 12 GB of weights) -- the oracle cannot run this
    size, so: a slab of points of two clouds against the C oracle (every output row depends on its own point's rows only),
    the adjoint identity <fwd(f), g> = <f, bwd(g)> over the whole batch (forward and backward are different kernels), and
    run-to-run bit equality of both."""
    import synth_clouds
    import vgtk.cuda.grouping as G
    import vgtk.cuda.zpconv as Z
    B, P, C, A, K, NN = 8, 4096, 64, 60, 24, 64
    xyz = T(synth_clouds.laptop_batch(0, B, P)[0]).to(dev)
    ball = G.ball_query(xyz, xyz, synth_clouds.backbone_layers(P)[1][2], NN)
    idx = ball[:, :, None, None, :].expand(B, P, A, K, NN).contiguous()
    gen = torch.Generator(device=dev).manual_seed(4)
    w = torch.rand(B, P, A, K, NN, device=dev, generator=gen)
    feats = torch.randn(B, C, P, A, device=dev, generator=gen)
    out = Z.inter_zpconv_forward(idx, w, feats)
    assert tuple(out.shape) == (B, C, K, P, A)
    assert torch.equal(out, Z.inter_zpconv_forward(idx, w, feats))
    for b, p0 in ((0, 0), (5, 2011)):
        sl = slice(p0, p0 + 48)
        ref = native.inter_zpconv_forward(idx[b:b + 1, sl].cpu().numpy(), w[b:b + 1, sl].cpu().numpy(), feats[b:b + 1].cpu().numpy())
        assert rel_err(out[b:b + 1, :, :, sl].cpu().numpy(), ref) < 2e-6, (b, p0)
    g = torch.randn(B, C, K, P, A, device=dev, generator=gen)
    gf = Z.inter_zpconv_backward(idx, w, g, P)
    assert tuple(gf.shape) == (B, C, P, A)
    assert torch.equal(gf, Z.inter_zpconv_backward(idx, w, g, P))
    lhs = float((out.double() * g.double()).sum())
    rhs = float((feats.double() * gf.double()).sum())
    assert abs(lhs - rhs) < 1e-6 * max(abs(lhs), abs(rhs), float(out.double().abs().sum()) * 1e-3), (lhs, rhs)


def test_native_zpconv_backward_with_rows_on_chip(dev):
    """csrc/zpconv_bwd_hot.hip (scatter target in LDS, no per-(point, neighbour) intermediate) behind
    vgtk.cuda.zpconv.inter_zpconv_backward: against the C oracle (zpconv_cuda_kernel.cu:L77-116 restated) at a size it
    runs, against the product pipeline (csrc/zpconv_bwd.hip) at the bench shape with the batch whole (one point range per
    cloud) and in a slice of two (four point ranges + the fixed-order partial sum), with a cloud whose lists reference more
    rows than fit and a cloud whose 5-D index is not one list per point in the same batch (both reported and handed to the
    other path), and bit equality run to run."""
    import synth_clouds
    import vgtk.cuda.grouping as G
    import vgtk.cuda.zpconv as Z
    from vgtk import _hip
    A, K, NN = 60, 24, 64
    gen = torch.Generator(device=dev).manual_seed(9)
    cap = int(_hip.lib.eap_inter_zpconv_bwd_hot_rows())

    def status_of(idx, w, g, nq):
        b, p = idx.shape[:2]
        c = g.shape[1]
        nbytes = int(_hip.lib.eap_inter_zpconv_bwd_hot_workspace(b, p, nq, A, K, NN, c))
        assert nbytes > 0
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=dev)
        st = torch.empty(b, dtype=torch.int32, device=dev)
        out = torch.empty(b, c, nq, A, device=dev)
        _hip.call('eap_inter_zpconv_bwd_hot_f32', out, b, p, nq, A, K, NN, c, _hip._ptr(idx), _hip._ptr(w), _hip._ptr(g), _hip._ptr(out),
                  _hip._ptr(ws), _hip._ptr(st))
        return st.tolist(), out

    def products(idx, w, g, nq):
        Z.ON_CHIP_BACKWARD = False
        try:
            return Z.inter_zpconv_backward(idx, w, g, nq)
        finally:
            Z.ON_CHIP_BACKWARD = True

    # ---- 1. against the oracle: 2 x 256 points, C = 32, a ball that takes in most of the cloud (few referenced rows),
    #         a shadow row (nq = P + 1)
    B, P, C = 2, 256, 32
    xyz = T(synth_clouds.laptop_batch(30, B, P)[0]).to(dev)
    ball = G.ball_query(xyz, xyz, 0.45, NN)
    rows = [len(torch.unique(ball[i])) for i in range(B)]
    assert max(rows) <= cap, rows
    idx = ball[:, :, None, None, :].expand(B, P, A, K, NN).contiguous()
    w = torch.rand(B, P, A, K, NN, device=dev, generator=gen)
    g = torch.randn(B, C, K, P, A, device=dev, generator=gen)
    st, _ = status_of(idx, w, g, P + 1)
    assert st == [0, 0]
    got = Z.inter_zpconv_backward(idx, w, g, P + 1)
    ref = native.inter_zpconv_backward(idx.cpu().numpy(), w.cpu().numpy(), g.cpu().numpy(), P + 1)
    assert rel_err(got.cpu().numpy(), ref) < 1e-5            # (sums of ~16 k products per element in another order; the bar of the other zpconv tests)
    assert torch.equal(got, Z.inter_zpconv_backward(idx, w, g, P + 1))

    # ---- 2. the bench shape, whole batch and a slice of two clouds, against the product pipeline
    B, P, C = 8, 4096, 64
    xyz = T(synth_clouds.laptop_batch(0, B, P)[0]).to(dev)
    ball = G.ball_query(xyz, xyz, synth_clouds.backbone_layers(P)[1][2], NN)
    idx = ball[:, :, None, None, :].expand(B, P, A, K, NN).contiguous()
    w = torch.rand(B, P, A, K, NN, device=dev, generator=gen)
    g = torch.randn(B, C, K, P, A, device=dev, generator=gen)
    st, _ = status_of(idx, w, g, P)
    assert st == [0] * B, (st, [len(torch.unique(ball[i])) for i in range(B)], cap)
    got = Z.inter_zpconv_backward(idx, w, g, P)
    assert torch.equal(got, Z.inter_zpconv_backward(idx, w, g, P))
    want = products(idx, w, g, P)
    assert rel_err(got.cpu().numpy(), want.cpu().numpy()) < 1e-5             # (sums of ~1000 x 24 products per element in two different orders)
    two = Z.inter_zpconv_backward(idx[2:4], w[2:4], g[2:4], P)
    assert rel_err(two.cpu().numpy(), want[2:4].cpu().numpy()) < 1e-5
    assert torch.equal(two, Z.inter_zpconv_backward(idx[2:4], w[2:4], g[2:4], P))
    del got, want, two

    # ---- 3. one batch, three kinds of cloud: few rows (on chip), a small ball (more rows than fit), an index that differs
    #         between kernel points (not one list per point)
    B, P, C = 3, 1024, 32
    xyz = T(synth_clouds.laptop_batch(40, B, P)[0]).to(dev)
    big, small = G.ball_query(xyz, xyz, 0.6, NN), G.ball_query(xyz, xyz, 0.05, NN)
    assert len(torch.unique(big[0])) <= cap < len(torch.unique(small[1]))
    idx = torch.stack([big[0], small[1], big[2]])[:, :, None, None, :].expand(B, P, A, K, NN).contiguous()
    idx[2, 17, 3, 5, :] = idx[2, 17, 3, 5, :].flip(0)
    w = torch.rand(B, P, A, K, NN, device=dev, generator=gen)
    g = torch.randn(B, C, K, P, A, device=dev, generator=gen)
    st, _ = status_of(idx, w, g, P)
    assert st == [0, 1, 1], st
    got = Z.inter_zpconv_backward(idx, w, g, P)
    ref = native.inter_zpconv_backward(idx.cpu().numpy(), w.cpu().numpy(), g.cpu().numpy(), P)
    # (cloud 2 -- an arbitrary 5-D index -- runs the scatter kernel with float atomics, as the reference's kernel does: the order the
    # atomics land in moves the last digit from run to run.  Seen between 0.8e-5 and 1.09e-5 over twenty runs: the bar of the
    # atomics-free paths, 1e-5, is not one this path can promise)
    assert rel_err(got[:2].cpu().numpy(), ref[:2]) < 1e-5
    assert rel_err(got.cpu().numpy(), ref) < 2e-5


@pytest.mark.gpu
def test_inter_zpconv_backward_remembers_the_verdict_per_index(dev):
    """vgtk.cuda.zpconv.inter_zpconv_backward (zpconv_cuda.cpp:L58-75) with the SAME index tensor again: which clouds the on-chip
    kernel leaves to the product pipeline depends on the index alone and is remembered (round-4 advisor finding: a batch of padded
    lists paid the prelude and a host read on every call).  Clouds 1, 2 and 4 of 5 name a row twice (the reference's padding):
    they go to the product pipeline as the runs [1, 2] and [4]; results equal the oracle's on the first and on the remembered
    call, and after an in-place change of the index the verdict is taken again."""
    import vgtk.cuda.zpconv as Z
    rng = np.random.default_rng(23)
    b, p, q, a, k, ann, c = 5, 8, 80, 60, 24, 64, 32                                                 # (the on-chip kernel takes 64 neighbours, 24 kernel points)
    idx = np.stack([np.stack([rng.permutation(q)[:ann] for _ in range(p)]) for _ in range(b)])         # distinct rows per list
    for cloud in (1, 2, 4):
        idx[cloud, :, ann // 2:] = idx[cloud, :, :1]                                                 # short lists padded with their first hit
    idx = np.broadcast_to(idx[:, :, None, None, :], (b, p, a, k, ann)).astype(np.int32).copy()
    w = rng.random((b, p, a, k, ann)).astype(np.float32)
    g = rng.standard_normal((b, c, k, p, a)).astype(np.float32)
    ref = native.inter_zpconv_backward(idx, w, g, q)
    d_idx, d_w, d_g = T(idx).to(dev), T(w).to(dev), T(g).to(dev)
    Z._HOT_VERDICTS.clear()
    first = Z.inter_zpconv_backward(d_idx, d_w, d_g, q)
    assert rel_err(first.cpu().numpy(), ref) < 1e-5
    assert len(Z._HOT_VERDICTS) == 1 and list(Z._HOT_VERDICTS.values())[0][1] == (1, 2, 4)
    reads = []
    orig = torch.Tensor.tolist
    torch.Tensor.tolist = lambda self: (reads.append(1), orig(self))[1]
    try:
        again = Z.inter_zpconv_backward(d_idx, d_w, d_g, q)
    finally:
        torch.Tensor.tolist = orig
    assert torch.equal(again, first) and not reads                                                   # no host read on the remembered call
    # all clouds padded: the second call does not even launch the on-chip kernel's prelude
    idx2 = idx.copy()
    idx2[:, :, :, :, ann // 2:] = idx2[:, :, :, :, :1]
    d_idx.copy_(T(idx2).to(dev))                                                                     # in place: the version counter moves
    ref2 = native.inter_zpconv_backward(idx2, w, g, q)
    for _ in range(2):
        assert rel_err(Z.inter_zpconv_backward(d_idx, d_w, d_g, q).cpu().numpy(), ref2) < 1e-5
    assert any(v[1] == (0, 1, 2, 3, 4) for v in Z._HOT_VERDICTS.values())


@pytest.mark.gpu
def test_inter_zpconv_backward_notices_an_index_changed_behind_its_version_counter(dev):
    """Round-5 advisor finding: the remembered verdict is keyed on (pointer, version, shape); `idx.data.copy_` changes the contents
    without a version bump.  The remembered call still runs without a host read, but the kernel's status comes back asynchronously
    and the NEXT call into the module compares it: the mismatch raises and the verdict is forgotten, after which the same tensor
    gives the oracle's result again."""
    import vgtk.cuda.zpconv as Z
    rng = np.random.default_rng(29)
    b, p, q, a, k, ann, c = 4, 8, 80, 60, 24, 64, 32
    idx = np.stack([np.stack([rng.permutation(q)[:ann] for _ in range(p)]) for _ in range(b)])
    idx5 = np.broadcast_to(idx[:, :, None, None, :], (b, p, a, k, ann)).astype(np.int32).copy()
    w = rng.random((b, p, a, k, ann)).astype(np.float32)
    g = rng.standard_normal((b, c, k, p, a)).astype(np.float32)
    d_idx, d_w, d_g = T(idx5).to(dev), T(w).to(dev), T(g).to(dev)
    Z._HOT_VERDICTS.clear()
    del Z._HOT_PENDING[:]
    first = Z.inter_zpconv_backward(d_idx, d_w, d_g, q)
    assert rel_err(first.cpu().numpy(), native.inter_zpconv_backward(idx5, w, g, q)) < 1e-5
    assert list(Z._HOT_VERDICTS.values())[0][1] == ()                       # every cloud on chip
    changed = idx5.copy()
    changed[2, :, :, :, ann // 2:] = changed[2, :, :, :, :1]                  # cloud 2 now names rows twice: the on-chip kernel rejects it
    version = d_idx._version
    d_idx.data.copy_(T(changed).to(dev))
    assert d_idx._version == version                                        # (the case: contents moved, counter did not)
    Z.inter_zpconv_backward(d_idx, d_w, d_g, q)                             # trusted the stale verdict: cloud 2 is not redone ...
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match='without a version bump'):       # ... and the next call says so
        Z.inter_zpconv_backward(d_idx, d_w, d_g, q)
    assert not Z._HOT_VERDICTS and not Z._HOT_PENDING
    again = Z.inter_zpconv_backward(d_idx, d_w, d_g, q)                     # verdict taken afresh
    assert rel_err(again.cpu().numpy(), native.inter_zpconv_backward(changed, w, g, q)) < 1e-5
    Z._check_pending_verdicts(block=True)
