"""GPU: the device-side inverse neighbour lists (csrc/inv_lists.hip) and the B1 modules added in round 2
(KernelPropagation, IntraSO3Conv2D, the zpconv wrappers) against plain torch / the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import native, so3_ref  # noqa: E402  (checker only)

T = torch.from_numpy


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return torch.device('cuda:0')


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _torch_inverse_lists(idx, gx, n):
    """Round 1's construction (stable sort by support row, rows by descending count): the reference the
    kernels must reproduce bit for bit."""
    b, p, nn = idx.shape
    keys = idx.reshape(b, p * nn).long()
    skeys, order = torch.sort(keys, dim=1, stable=True)
    counts = torch.zeros(b, n + 1, dtype=torch.int64)
    counts.scatter_add_(1, keys.clamp(max=n), torch.ones_like(keys))
    counts = counts[:, :n]
    offs = torch.cumsum(counts, 1) - counts
    n_rows = (counts > 0).sum(1)
    rcap = int(n_rows.max())
    rows = torch.argsort(counts, dim=1, descending=True, stable=True)[:, :rcap]
    valid = torch.arange(rcap)[None, :] < n_rows[:, None]
    cnt_c = torch.gather(counts, 1, rows) * valid
    rows_c = torch.where(valid, rows, torch.full_like(rows, -1))
    ent_p = torch.div(order, nn, rounding_mode='floor')
    ent_gx = torch.gather(gx.reshape(b, p * nn, 4), 1, order[..., None].expand(-1, -1, 4))
    return rows_c, torch.gather(offs, 1, rows), cnt_c, ent_p, ent_gx, rcap, skeys


@pytest.mark.parametrize('case', ['hot_rows_4096', 'uniform', 'ragged_small', 'with_shadow'])
def test_inverse_lists_on_device(dev, case):
    import synth_clouds
    import vgtk.cuda.grouping as G
    import vgtk.so3conv.functional as L
    gen = torch.Generator().manual_seed(17)
    if case == 'hot_rows_4096':      # the benchmark regime: first-64-in-index-order lists of a big ball
        xyz = T(synth_clouds.laptop_batch(5, 3, 4096)[0])
        idx = G.ball_query(xyz.to(dev), xyz.to(dev), synth_clouds.backbone_layers(4096)[2][2], 64).cpu()
        n = 4096
    elif case == 'uniform':
        n, idx = 500, torch.randint(0, 500, (2, 333, 16), generator=gen, dtype=torch.int32)
    elif case == 'ragged_small':     # clouds with different numbers of referenced rows, some rows empty
        n = 40
        idx = torch.randint(0, 40, (3, 9, 8), generator=gen, dtype=torch.int32)
        idx[1] = idx[1] % 5
        idx[2] = 7
    else:                            # shadow entries (== n) are ignored
        n = 64
        idx = torch.randint(0, 65, (2, 50, 12), generator=gen, dtype=torch.int32)
    b, p, nn = idx.shape
    gx = torch.randn(b, p, nn, 4, generator=gen)
    rows, off, cnt, ent_p, ent_gx, rcap, all_ident = L._inverse_lists(idx.to(dev), gx.to(dev), n, 0, torch.zeros(b, dtype=torch.int32, device=dev))
    r_rows, r_off, r_cnt, r_ent_p, r_ent_gx, r_rcap, skeys = _torch_inverse_lists(idx, gx, n)
    assert rcap == r_rcap and all_ident
    np.testing.assert_array_equal(rows.cpu().numpy(), r_rows.numpy())
    np.testing.assert_array_equal(cnt.cpu().numpy(), r_cnt.numpy())
    # entry lists: row r's entries are the same (p, gx) sequence; offsets may differ (here rows are stored
    # in slot order, round 1 stored them in support-index order)
    ent_p, ent_gx, off = ent_p.cpu(), ent_gx.cpu(), off.cpu()
    for bi in range(b):
        for r in range(rcap):
            c = int(r_cnt[bi, r])
            if c == 0:
                continue
            a0, b0 = int(off[bi, r]), int(r_off[bi, r])
            np.testing.assert_array_equal(ent_p[bi, a0:a0 + c].numpy(), r_ent_p[bi, b0:b0 + c].numpy())
            np.testing.assert_array_equal(ent_gx[bi, a0:a0 + c].numpy(), r_ent_gx[bi, b0:b0 + c].numpy())


def test_rows_gather_scatter(dev):
    from vgtk import _hip
    gen = torch.Generator().manual_seed(3)
    b, c, n, na, rcap = 2, 5, 37, 60, 9
    src = torch.randn(b, c, n, na, generator=gen)
    rows = torch.stack([torch.randperm(n, generator=gen)[:rcap], torch.randperm(n, generator=gen)[:rcap]]).int()
    rows[1, 6:] = -1
    rows_full = torch.full((b, n), -1, dtype=torch.int32)
    rows_full[:, :rcap] = rows                                       # leading dimension n, as the list kernels write it
    g = _hip.rows_gather(src.to(dev), rows_full.to(dev), rcap).cpu()
    want = torch.zeros(b, c, rcap, na)
    for bi in range(b):
        for r in range(rcap):
            if rows[bi, r] >= 0:
                want[bi, :, r] = src[bi, :, rows[bi, r]]
    assert torch.equal(g, want)
    s = _hip.rows_scatter(want.to(dev), rows_full.to(dev), n).cpu()
    back = torch.zeros(b, c, n, na)
    for bi in range(b):
        for r in range(rcap):
            if rows[bi, r] >= 0:
                back[bi, :, rows[bi, r]] = want[bi, :, r]
    assert torch.equal(s, back)


def test_kernel_propagation(dev):
    """KernelPropagation.forward (so3conv/modules.py:L57-119) = initial_anchor_query / (cnt + 1) -> BasicSO3Conv."""
    import synth_clouds
    import vgtk.so3conv as sptk
    torch.manual_seed(1)
    prop = sptk.KernelPropagation(1, 8, 32, 1, 0.2, 0.02, kanchor=60)
    xyz = synth_clouds.laptop_batch(2, 2, 32)[0]
    frag = np.ascontiguousarray(synth_clouds.laptop_batch(9, 1, 300)[0][0].T)
    w, cnt = native.initial_anchor_query(xyz, frag, prop.kernels.numpy(), 0.2, 0.02)
    ref = so3_ref.basic_so3conv(prop.basic_conv.W.detach(), T(w / (cnt + 1.0)).unsqueeze(1))
    prop = prop.to(dev)
    out = prop(T(frag).to(dev), T(xyz).to(dev))
    assert out.feats.shape == (2, 8, 32, 60) and out.xyz.shape == (2, 3, 32)
    assert rel_err(out.feats.detach().cpu().numpy(), ref.numpy()) < 1e-5


def test_intra_so3conv_2d(dev):
    """IntraSO3Conv2D (so3conv/modules.py:L350-373) against the index_select formulation in plain torch."""
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    torch.manual_seed(2)
    conv = sptk.IntraSO3Conv2D(5, 7)
    f = torch.randn(2, 5, 11, 240)
    fi = f.view(2, 5, 11, 60, 4).index_select(3, conv.intra_idx.view(-1)).view(2, 5, 11, 60, 12, 4)
    grouped = fi.permute(0, 1, 4, 2, 3, 5).reshape(2, 5, 12, 11, 240)
    ref = torch.matmul(conv.basic_conv.W.detach(), grouped.reshape(2, 60, 11 * 240)).view(2, 7, 11, 240)
    conv = conv.to(dev)
    out = conv(zptk.SphericalPointCloud(torch.zeros(2, 3, 11, device=dev), f.to(dev), None)).feats
    assert rel_err(out.detach().cpu().numpy(), ref.numpy()) < 1e-5


def test_zpconv_wrappers_pooling_and_index_helpers(dev):
    """the autograd wrappers around the native zpconv ops (spconv/functional.py:L102-129,
    L211-238) through the "naive" shared-index entry points, values and gradients against the reference's einsum
    formulation on the CPU (oracle/so3_ref.py, pinned by zpconv_naive.npz); the pooling / blurring blends and the two
    index helpers the SO(3) layer shares with it."""
    import vgtk.spconv as zptk
    gen = torch.Generator().manual_seed(4)
    b, p, a, k, nn, c, q = 2, 20, 12, 5, 8, 3, 21
    idx = torch.randint(0, q, (b, p, nn), generator=gen)
    w = torch.rand(b, p, a, k, nn, generator=gen)
    f = torch.randn(b, c, q, a, generator=gen)
    fr = f.clone().requires_grad_(True)
    ref = so3_ref.inter_zpconv_grouping_naive(idx, w, fr)
    g = torch.randn(ref.shape, generator=gen)
    fd = f.to(dev).requires_grad_(True)
    out = zptk.inter_zpconv_grouping_naive(idx.to(dev), w.to(dev), fd)
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < 2e-6
    assert rel_err(torch.autograd.grad(out, fd, g.to(dev))[0].cpu().numpy(), torch.autograd.grad(ref, fr, g)[0].numpy()) < 2e-6
    iidx = torch.randint(0, a, (a, 4), generator=gen)
    iw = torch.rand(a, k, 4, generator=gen)
    f2 = torch.randn(b, c, p, a, generator=gen)
    f2r = f2.clone().requires_grad_(True)
    ref = so3_ref.intra_zpconv_grouping_naive(iidx, iw, f2r)
    g = torch.randn(ref.shape, generator=gen)
    f2d = f2.to(dev).requires_grad_(True)
    out = zptk.intra_zpconv_grouping_naive(iidx.to(dev), iw.to(dev), f2d)
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < 2e-6
    assert rel_err(torch.autograd.grad(out, f2d, g.to(dev))[0].cpu().numpy(), torch.autograd.grad(ref, f2r, g)[0].numpy()) < 2e-6
    # pooling / blurring (spconv/functional.py:L274-311): alpha * centre + (1 - alpha) * neighbourhood mean, shadow index = zeros
    idx_s = torch.randint(0, q + 1, (b, q, nn), generator=gen)                     # q = shadow row
    pad = torch.cat([f, torch.zeros(b, c, 1, a)], 2)
    mean = torch.stack([pad[i][:, idx_s[i]] for i in range(b)]).mean(3)            # [b,c,q,a]
    assert rel_err(zptk.inter_blurring_naive(idx_s.to(dev), f.to(dev)).cpu().numpy(), (0.5 * f + 0.5 * mean).numpy()) < 1e-6
    sample = torch.randint(0, q, (b, 7), generator=gen)
    want = 0.25 * torch.stack([f[i][:, sample[i]] for i in range(b)]) + 0.75 * torch.stack([pad[i][:, idx_s[i, :7]] for i in range(b)]).mean(3)
    assert rel_err(zptk.inter_pooling_naive(idx_s[:, :7].to(dev), sample.to(dev), f.to(dev), alpha=0.25).cpu().numpy(), want.numpy()) < 1e-6
    # index helpers against the oracle's restatement of the reference's (pinned through the layer fixtures)
    x = torch.randn(3, 5, 7, 4, generator=gen)
    for dim, n in ((1, 5), (2, 7), (3, 4)):
        i = torch.randint(0, n, (3, 9), generator=gen)
        assert torch.equal(zptk.batched_index_select(x, dim, i), so3_ref.batched_index_select(x, dim, i))
    v = torch.randn(2, 11, 3, 3, generator=gen)
    ii = torch.randint(0, 11, (2, 4, 6), generator=gen)
    assert torch.equal(zptk.batched_index_select_other(v, ii, 1), so3_ref.batched_index_select_other(v, ii, 1))
    v2 = torch.randn(2, 4, 6, 60, 5, generator=gen)
    i2 = torch.randint(0, 60, (2, 4, 6, 60), generator=gen)
    assert torch.equal(zptk.batched_index_select_other(v2, i2, 3), so3_ref.batched_index_select_other(v2, i2, 3))


@pytest.mark.parametrize('M,N,K,batch', [(512, 1920, 3072, 1), (128, 2040, 1536, 2), (256, 128, 32, 1), (130, 260, 48, 2),
                                        (64, 4, 16, 1), (300, 1000, 160, 3), (12288, 128, 960, 2),
                                        (64, 1680, 384, 2), (52, 1000, 96, 1), (1536, 64, 320, 2), (700, 60, 160, 1)])   # 64 x 512 / 512 x 64 tiles
def test_gemm_dma_all_operand_layouts(dev, M, N, K, batch):
    """csrc/gemm_dma_f32.hip (DMA-fed ring) in its four operand layouts + the k-split batch reduction, against
    float64 matmul; ragged M / N tiles, a single k-tile, K = 2 k-tiles (ring shorter than its depth)."""
    from vgtk import _hip
    gen = torch.Generator().manual_seed(21)
    A = torch.randn(M, K, generator=gen)
    B = torch.randn(batch, K, N, generator=gen)
    ref = torch.matmul(A.double(), B.double()).numpy()
    tol = 2e-6 if K <= 256 else 1e-5
    Ad, Bd = A.to(dev), B.to(dev)
    At = A.t().contiguous().to(dev)                       # stored [K, M]
    Bt = B.transpose(1, 2).contiguous().to(dev)           # stored [N, K]
    for ta, tb, a, lda, bm, ldb in ((0, 0, Ad, K, Bd, N), (0, 1, Ad, K, Bt, K), (1, 0, At, M, Bd, N), (1, 1, At, M, Bt, K)):
        if (ta and M % 4) or (not tb and N % 4):
            assert not _hip._dma_ok(ta, tb, M, N, K, a, lda, 0, bm, ldb, K * N)
            continue
        assert _hip._dma_ok(ta, tb, M, N, K, a, lda, 0, bm, ldb, K * N)
        C = torch.full((batch, M, N), float('nan'), device=dev)
        _hip.call('eap_gemm_dma_f32', C, ta, tb, M, N, K, _hip._ptr(a), _hip._I64(lda), _hip._I64(0), _hip._ptr(bm), _hip._I64(ldb),
                  _hip._I64(K * N), _hip._ptr(C), _hip._I64(N), _hip._I64(M * N), batch)
        assert rel_err(C.cpu().numpy(), ref) < tol, (ta, tb)
    # batch-reduced weight-gradient shape: C[M, K'] = sum_b G_b[M, N] X_b[K', N]^T, contraction over N (long)
    if N % 16 == 0:
        G = torch.randn(batch, M, N, generator=gen)
        X = torch.randn(batch, 40, N, generator=gen)
        ref2 = torch.einsum('bmn,bkn->mk', G.double(), X.double()).numpy()
        C2 = torch.full((M, 40), float('nan'), device=dev)
        _hip.gemm_reduce(0, 1, M, 40, N, G.to(dev), N, M * N, X.to(dev), N, 40 * N, C2, 40, batch)
        assert rel_err(C2.cpu().numpy(), ref2) < 1e-5


@pytest.mark.gpu
def test_gemm_dma_wide_tile_for_few_rows(dev):
    """M <= 128 with enough column tiles takes the 128 x 512 block tile (ten staged pieces per thread): all four
    operand layouts, ragged M and N, against float64."""
    from vgtk import _hip
    gen = torch.Generator().manual_seed(77)
    M, N, K, batch = 100, 65536 + 24, 48, 8
    A = torch.randn(M, K, generator=gen)
    B = torch.randn(batch, K, N, generator=gen)
    ref = torch.matmul(A.double(), B.double()).numpy()
    Ad, Bd = A.to(dev), B.to(dev)
    At = A.t().contiguous().to(dev)
    Bt = B.transpose(1, 2).contiguous().to(dev)
    for ta, tb, a, lda, bm, ldb in ((0, 0, Ad, K, Bd, N), (0, 1, Ad, K, Bt, K), (1, 0, At, M, Bd, N), (1, 1, At, M, Bt, K)):
        C = torch.full((batch, M, N), float('nan'), device=dev)
        _hip.call('eap_gemm_dma_f32', C, ta, tb, M, N, K, _hip._ptr(a), _hip._I64(lda), _hip._I64(0), _hip._ptr(bm), _hip._I64(ldb),
                  _hip._I64(K * N), _hip._ptr(C), _hip._I64(N), _hip._I64(M * N), batch)
        assert rel_err(C.cpu().numpy(), ref) < 2e-6, (ta, tb)


def test_gemm_dma_is_transpose_detecting_and_deterministic(dev):
    from vgtk import _hip
    n = 256
    A = torch.eye(n, device=dev)
    B = (torch.arange(n * n, dtype=torch.float32).view(1, n, n) % 251).to(dev)
    C = torch.empty(1, n, n, device=dev)
    _hip.call('eap_gemm_dma_f32', C, 0, 0, n, n, n, _hip._ptr(A), _hip._I64(n), _hip._I64(0), _hip._ptr(B), _hip._I64(n), _hip._I64(n * n),
              _hip._ptr(C), _hip._I64(n), _hip._I64(n * n), 1)
    assert torch.equal(C, B)
    gen = torch.Generator().manual_seed(5)
    X = torch.randn(512, 3072, generator=gen).to(dev); Y = torch.randn(1, 1024, 3072, generator=gen).to(dev)
    outs = []
    for _ in range(3):
        C = torch.empty(1, 512, 1024, device=dev)
        _hip.gemm(0, 1, 512, 1024, 3072, X, 3072, 0, Y, 3072, 1024 * 3072, C, 1024, 512 * 1024, 1)
        outs.append(C)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])


@pytest.mark.parametrize('lazy', [False, True])
@pytest.mark.parametrize('chans,kind', [((1, 16), 'identity'), ((16, 32), 'random')])
def test_strided_pose_conv(dev, chans, kind, lazy):
    """InterSO3PoseConv with stride 2 -- furthest-point sampled (or lazily taken) centres,
    gathered poses, centres -> all points ball query (so3conv/functional.py:L931-1013) -- against the oracle:
    sample indices exact, outputs, dF, dW."""
    import synth_clouds
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    from test_gpu_bench_shapes import make_poses
    c, o = chans
    P, B, radius, sigma, nn = 256, 2, 0.16, 0.0128, 32
    xyz_np, lab, _ = synth_clouds.laptop_batch(33, B, P)
    gen = torch.Generator().manual_seed(8)
    pose = make_poses(gen, [kind] * B, lab, P)
    xyz = T(xyz_np)
    torch.manual_seed(6)
    conv = sptk.InterSO3PoseConv(c, o, 1, 2, radius, sigma, nn, lazy_sample=lazy, kanchor=60, permute_modes=1)
    f = torch.ones(B, 1, P, 60) if c == 1 else torch.randn(B, c, P, 60, generator=gen)
    fr = f.clone().requires_grad_(True)
    Wr = conv.basic_conv.W.detach().clone().requires_grad_(True)
    _, w_ref, new_xyz, nf, sidx, spose = so3_ref.inter_so3poseconv_grouping_strided_sampled(
        xyz, pose, fr, 2, nn, conv.anchors, conv.kernels, radius, sigma, permute_modes=1, lazy_sample=lazy,
        skip_perm_search=(kind == 'identity'))
    y_ref = so3_ref.basic_so3conv(Wr, nf)
    gy = torch.randn(y_ref.shape, generator=gen)
    gf_ref, gw_ref = torch.autograd.grad(y_ref, [fr, Wr], gy)
    conv = conv.to(dev)
    fd = f.to(dev).requires_grad_(True)
    inter_idx, w, sample_idx, out = conv(zptk.SphericalPointCloudPose(xyz.to(dev), fd, None, pose.to(dev)))
    assert inter_idx is None
    np.testing.assert_array_equal(sample_idx.cpu().numpy(), sidx.numpy())
    assert out.feats.shape == (B, o, P // 2, 60)
    np.testing.assert_array_equal(out.xyz.cpu().numpy(), new_xyz.numpy())
    np.testing.assert_array_equal(out.pose.cpu().numpy(), spose.numpy())
    assert rel_err(w.materialize().cpu().numpy(), w_ref.numpy()) < 5e-6
    assert rel_err(out.feats.detach().cpu().numpy(), y_ref.detach().numpy()) < 1e-5
    gF, gW = torch.autograd.grad(out.feats, [fd, conv.basic_conv.W], gy.to(dev))
    assert rel_err(gF.cpu().numpy(), gf_ref.numpy()) < 1e-5
    assert rel_err(gW.cpu().numpy(), gw_ref.numpy()) < 2e-5


def test_strided_pose_free_conv(dev):
    """InterSO3Conv with stride 2 (so3conv/functional.py:L144-203): sampled centres, xyz and sample_idx returned."""
    import synth_clouds
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    xyz = T(synth_clouds.laptop_batch(34, 2, 128)[0])
    torch.manual_seed(7)
    conv = sptk.InterSO3Conv(4, 6, 1, 2, 0.2, 0.02, 16, lazy_sample=False, kanchor=60)
    f = torch.randn(2, 4, 128, 60)
    sidx = T(native.furthest_point_sampling(xyz.numpy(), 64))
    cx = so3_ref.group_nd(xyz, sidx)
    idx, gxyz = so3_ref.ball_query(cx, xyz, 0.2, 16)
    wref = so3_ref.inter_so3conv_grouping_anchor(gxyz - cx.unsqueeze(3), conv.anchors, conv.kernels, 0.02)
    ref = so3_ref.basic_so3conv(conv.basic_conv.W.detach(), so3_ref.inter_zpconv_grouping_naive(idx, wref, so3_ref.add_shadow_feature(f)))
    conv = conv.to(dev)
    inter_idx, inter_w, sample_idx, out = conv(zptk.SphericalPointCloud(xyz.to(dev), f.to(dev), None))
    np.testing.assert_array_equal(sample_idx.cpu().numpy(), sidx.numpy())
    np.testing.assert_array_equal(inter_idx.cpu().numpy(), idx.numpy())
    assert rel_err(out.feats.detach().cpu().numpy(), ref.numpy()) < 1e-5


def test_anchor_attention_pool_and_invariant_head(dev):
    """InvPPOutBlockOurs (base_so3conv.py:L842-917) with attention pooling -- the fused
    anchor softmax + weighted sum against the reference's torch formulation, outputs and all gradients."""
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    torch.manual_seed(12)
    params = {'dim_in': 24, 'mlp': [16, 12], 'fc': [12], 'k': 12, 'kanchor': 60, 'temperature': 3.0}
    head = sptk.InvPPOutBlockOurs(params, pooling_method='attention').to(dev)
    x = torch.randn(2, 24, 37, 60, device=dev, requires_grad=True)
    out, conf = head(zptk.SphericalPointCloud(None, x, None))
    # reference formulation in plain torch on the same device
    xr = x.detach().clone().requires_grad_(True)
    h = xr
    for lid, lin in enumerate(head.linear):
        h = F_relu(head.norm[lid](lin(h)))
    logit = head.attention_layer(h)
    cref = torch.softmax(logit * 3.0, dim=-1)
    oref = (h * cref).sum(-1)
    assert out.shape == (2, 12, 37) and conf.shape == (2, 37, 60)
    assert rel_err(out.detach().cpu().numpy(), oref.detach().cpu().numpy()) < 2e-6
    assert rel_err(conf.detach().cpu().numpy(), cref.squeeze(1).detach().cpu().numpy()) < 2e-6
    g = torch.randn_like(out)
    names = [n for n, _ in head.named_parameters()]
    grads = torch.autograd.grad(out, [x] + list(head.parameters()), g, retain_graph=True)
    grefs = torch.autograd.grad(oref, [xr] + list(head.parameters()), g)
    top = max(float(b.abs().max()) for b in grefs)
    for nme, a, b in zip(['x'] + names, grads, grefs):
        # conv biases in front of a training-mode BatchNorm have an exactly-zero gradient (rounding noise on both
        # sides, ~1e-6): the bar is relative to the tensor's own scale, floored at 1 % of the largest gradient
        if nme == 'attention_layer.bias':      # the softmax ignores a constant logit shift: analytically zero, noise on both sides
            assert float(a.abs().max()) < 1e-3 * top and float(b.abs().max()) < 1e-3 * top, nme
            continue
        scale = max(float(b.abs().max()), 1e-2 * top)
        assert float((a - b).abs().max()) < 2e-5 * scale, nme
    # the other pooling modes are plain reductions
    for mode in ('max', 'mean'):
        h2 = sptk.InvPPOutBlockOurs(params, pooling_method=mode).to(dev)
        assert h2(zptk.SphericalPointCloud(None, x.detach(), None)).shape == (2, 12, 37)


def F_relu(t):
    return torch.nn.functional.relu(t)


def test_slot_masked_mean_and_orbit_selection(dev):
    """the per-slot masked point averages of the pose head, every slot in one pass, against
    torch; and the orbit arg-min of ...pn_38_multi_stage.py:L1381-1399."""
    import vgtk.so3conv as sptk
    torch.manual_seed(13)
    b, c, n, na, ns = 2, 19, 333, 60, 3
    x = torch.randn(b, c, n, na, device=dev, requires_grad=True)
    mask = (torch.rand(b, ns, n, device=dev) > 0.6).float()
    mask[0, 2] = 0.0                                               # an empty slot: clamp(min=1e-8) -> zeros
    out = sptk.slot_masked_mean(x, mask)
    xr = x.detach().clone().requires_grad_(True)
    ref = torch.einsum('bcna,bsn->bsca', xr, mask) / mask.sum(-1).clamp(min=1e-8)[:, :, None, None]
    assert rel_err(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) < 2e-6
    g = torch.randn_like(out)
    assert rel_err(torch.autograd.grad(out, x, g)[0].cpu().numpy(), torch.autograd.grad(ref, xr, g)[0].cpu().numpy()) < 2e-6
    soft = torch.rand(b, ns, n, device=dev)
    assert rel_err(sptk.slot_masked_mean(x.detach(), soft).cpu().numpy(),
                   (torch.einsum('bcna,bsn->bsca', x.detach(), soft) / soft.sum(-1)[:, :, None, None]).cpu().numpy()) < 2e-6
    d1, d2 = torch.rand(b, ns, 60, device=dev), torch.rand(b, ns, 60, device=dev)
    dist, orbit = sptk.orbit_selection(d1, d2)
    assert torch.equal(orbit, (d1 + d2).argmin(-1)) and torch.equal(dist, (d1 + d2).min(-1)[0])
    dist, orbit = sptk.orbit_selection(d1, d2, slot_single_cd=1, slot_single_mode=1)
    assert torch.equal(orbit[:, 0], d1.sum(1).argmin(-1)) and orbit.shape == (b, ns)
    ang = torch.rand(5, 7, device=dev) * 6.0
    ax = torch.nn.functional.normalize(torch.randn(5, 7, 3, device=dev), dim=-1)
    R = sptk.rotation_from_angle_axis(ang, ax)
    assert rel_err((R @ R.transpose(-1, -2)).cpu().numpy(), torch.eye(3).expand(5, 7, 3, 3).numpy()) < 1e-5
    assert rel_err((R @ ax.unsqueeze(-1)).squeeze(-1).cpu().numpy(), ax.cpu().numpy()) < 1e-5


@pytest.mark.parametrize('training', [True, False])
def test_block_epilogue_with_skip_add(dev, training):
    """`x.feats + relu(norm(skip))` (base_so3poseconv.py:L319-328) in the epilogue pass."""
    import vgtk.so3conv as sptk
    torch.manual_seed(21)
    x = torch.randn(2, 9, 13, 60, device=dev) * 1.5 + 0.5
    res = torch.randn(2, 9, 13, 60, device=dev)
    g = torch.randn(2, 9, 13, 60, device=dev)
    ref = torch.nn.BatchNorm2d(9).to(dev)
    fused = sptk.BatchNormLeakyReLU(9, negative_slope=0.01).to(dev)
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5); ref.bias.uniform_(-1, 1); ref.running_var.uniform_(0.5, 2.0)
    fused.load_state_dict(ref.state_dict())
    ref.train(training); fused.train(training)
    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    xf, rf = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    yr = torch.nn.functional.leaky_relu(ref(xr), 0.01) + rr
    yf = fused(xf, residual=rf)
    yr.backward(g); yf.backward(g)
    assert float((yf - yr).abs().max()) <= 2e-6 * float(yr.abs().max())
    assert float((xf.grad - xr.grad).abs().max()) <= 2e-5 * float(xr.grad.abs().max())
    assert torch.equal(rf.grad, rr.grad)


def test_pose_head_matches_reference_golden(dev, golden):
    """vgtk.so3conv.SO3OutBlockRTWithMaskSep (section 8(f) row 3) against tests/golden/pose_head.npz, produced by the
    reference class (SPConvNets/models/model_utils.py:L363-677) on CPU with the same state_dict and inputs: the
    whole output dictionary in training mode (batch statistics; running statistics after the step) and in eval mode."""
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    G = golden('pose_head.npz')
    head = sptk.SO3OutBlockRTWithMaskSep({'dim_in': 16, 'mlp': [32, 24], 'kanchor': 60, 'temperature': 3.0}, norm=1, pooling_method='max',
                                         pred_axis=True, pred_pv_points=True, pred_central_points=True, num_heads=1, representation='quat')
    state = {k[len('state_'):]: torch.from_numpy(np.asarray(v)) for k, v in G.items() if k.startswith('state_')}
    assert set(state) == set(head.state_dict().keys())
    T = lambda k: torch.from_numpy(np.asarray(G[k])).to(dev)
    for mode in ('train', 'eval'):
        head.load_state_dict(state)
        head = head.to(dev)
        head.train(mode == 'train')
        x = zptk.SphericalPointCloud(T('xyz'), T('feats'), None)
        with torch.no_grad():
            res = head(x, T('mask'), T('trans_feats'), trans_xyz=T('xyz'), anchors=T('anchors'))
        for k in ('R', 'T', 'axis', 'pv_points', 'central_points'):
            ref = np.asarray(G[f'{mode}_{k}'])
            assert res[k].shape == ref.shape, k
            assert rel_err(res[k].cpu().numpy(), ref) < 2e-5, (mode, k)
        if mode == 'train':
            for k, v in head.state_dict().items():
                if 'running' in k:
                    assert rel_err(v.cpu().numpy(), np.asarray(G[f'after_{k}'])) < 1e-5, k


@pytest.mark.parametrize('shape,slope', [((3, 8, 40, 60), 0.0), ((2, 5, 700, 60), 0.2), ((4, 16, 33, 12), 0.01)])
def test_subset_batchnorm_act_matches_torch(dev, shape, slope):
    """csrc/bn_act.hip eap_bn_act_cloud_*: leaky_relu(BatchNorm2d(y + bias)) with per-cloud statistics over a point subset
    against plain torch fp32 ops (per-cloud nn.functional.batch_norm on the gathered subset statistics), training and
    eval mode, forward, all gradients, running statistics."""
    import vgtk.so3conv.heads as H
    torch.manual_seed(5)
    B, C, P, A = shape
    y0 = (torch.randn(B, C, P, A, device=dev) * 1.7 + 0.4)
    bias0 = torch.randn(C, device=dev)
    member = torch.rand(B, P, device=dev) > 0.4
    member[:, 0] = True
    member[0] = False; member[0, :3] = True
    mask = member.float()
    probe = torch.randn(B, C, P, A, device=dev) * mask.view(B, 1, P, 1) + 0.1 * torch.randn(B, C, P, A, device=dev)
    for training in (True, False):
        bns = [torch.nn.BatchNorm2d(C).to(dev).train(training) for _ in range(2)]
        for bn in bns:
            with torch.no_grad():
                bn.weight.copy_(torch.linspace(0.5, 1.5, C)); bn.bias.copy_(torch.linspace(-0.3, 0.3, C))
                bn.running_mean.copy_(torch.linspace(-0.2, 0.6, C)); bn.running_var.copy_(torch.linspace(0.7, 2.0, C))
        res = []
        for which, bn in enumerate(bns):
            y = y0.clone().requires_grad_(True)
            bias = bias0.clone().requires_grad_(True)
            if which == 0:                       # reference: one batch-1 BatchNorm call per cloud; statistics of the members
                outs = []
                for b in range(B):
                    sel = member[b].nonzero().squeeze(1)
                    z = y[b:b + 1] + bias.view(1, C, 1, 1)
                    if training:
                        sub = z[:, :, sel]
                        mean = sub.mean((0, 2, 3)); var = sub.var((0, 2, 3), unbiased=False)
                        cnt = sub.numel() / C
                        with torch.no_grad():
                            bn.running_mean.mul_(1 - bn.momentum).add_(mean, alpha=bn.momentum)
                            bn.running_var.mul_(1 - bn.momentum).add_(var * cnt / max(cnt - 1, 1), alpha=bn.momentum)
                    else:
                        mean, var = bn.running_mean, bn.running_var
                    zn = (z - mean.view(1, C, 1, 1)) * torch.rsqrt(var.view(1, C, 1, 1) + bn.eps) * bn.weight.view(1, C, 1, 1) + bn.bias.view(1, C, 1, 1)
                    outs.append(torch.nn.functional.leaky_relu(zn, slope))
                out = torch.cat(outs, 0)
            else:
                out = H._subset_batchnorm_act(y, bias, mask, bn, slope)
            (out * probe).sum().backward()
            res.append((out.detach(), y.grad, bias.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone()))
        names = ('out', 'dy', 'dbias', 'dgamma', 'dbeta', 'running_mean', 'running_var')
        for name, r, g in zip(names, res[0], res[1]):
            r, g = r.cpu().numpy(), g.cpu().numpy()
            if name == 'dbias' and training:     # absorbed by the batch mean: rounding noise in torch, exactly zero here
                assert np.abs(g).max() == 0.0 and np.abs(r).max() < 1e-3 * np.abs(res[0][1].cpu().numpy()).max() * P * A
                continue
            assert rel_err(g, r) < 2e-5, (training, name, rel_err(g, r))


@pytest.mark.parametrize('pooling', ['max', 'mean'])
def test_pose_head_over_subsets_equals_the_per_cloud_loop(dev, pooling, A=60):
    """vgtk.so3conv.pose_head_over_subsets (one call per slot) against the model's inner loop: the head called once per
    cloud, batch 1, on the gathered point subset with mask=None (...pn_38_multi_stage.py:L706-830) -- outputs and
    the BatchNorm running statistics after the pass, training and eval mode."""
    import copy
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    import vgtk.so3conv.functional as L
    torch.manual_seed(3)
    B, C, N = 3, 16, 40
    head = sptk.SO3OutBlockRTWithMaskSep({'dim_in': C, 'mlp': [32, 32], 'kanchor': A, 'temperature': 3.0}, norm=1, pooling_method=pooling,
                                         pred_axis=True, pred_pv_points=True, pred_central_points=True).to(dev)
    feats = torch.randn(B, C, N, A, device=dev)
    xyz = torch.randn(B, 3, N, device=dev) * 0.3
    member = torch.rand(B, N, device=dev) > 0.5
    member[:, 0] = True
    member[1, 4:] = False; member[1, :4] = True                    # a 4-point subset
    anchors = torch.from_numpy(np.ascontiguousarray(L.get_anchors(A))).to(dev)
    for training in (True, False):
        ref_head, fast_head = copy.deepcopy(head).train(training), copy.deepcopy(head).train(training)
        outs = []
        with torch.no_grad():
            for b in range(B):
                sel = member[b].nonzero().squeeze(1)
                fx, px = feats[b:b + 1, :, sel].contiguous(), xyz[b:b + 1, :, sel].contiguous()
                outs.append(ref_head(zptk.SphericalPointCloud(px, fx, None), None, fx, trans_xyz=px, anchors=anchors.unsqueeze(0)))
            got = sptk.pose_head_over_subsets(fast_head, feats, xyz, member, anchors)
        for k in ('R', 'T', 'axis', 'pv_points', 'central_points'):
            ref = torch.cat([o[k] for o in outs], 0)
            assert got[k].shape == ref.shape, k
            assert rel_err(got[k].cpu().numpy(), ref.cpu().numpy()) < 2e-5, (training, k)
        for (k, v), (_, w) in zip(ref_head.state_dict().items(), fast_head.state_dict().items()):
            if 'running' in k:
                assert rel_err(w.cpu().numpy(), v.cpu().numpy()) < 1e-5, k
        # gradients of a random linear functional of every output: the fused per-cloud BatchNorm + activation backward
        # (csrc/bn_act.hip, eap_bn_act_cloud_*) and the masked pooling against autograd through the per-cloud loop
        ref_head, fast_head = copy.deepcopy(head).train(training), copy.deepcopy(head).train(training)
        keys = ('R', 'T', 'axis', 'pv_points', 'central_points')
        gen = torch.Generator().manual_seed(11)
        f_ref = feats.clone().requires_grad_(True)
        outs = []
        for b in range(B):
            sel = member[b].nonzero().squeeze(1)
            fx, px = f_ref[b:b + 1, :, sel].contiguous(), xyz[b:b + 1, :, sel].contiguous()
            outs.append(ref_head(zptk.SphericalPointCloud(px, fx, None), None, fx, trans_xyz=px, anchors=anchors.unsqueeze(0)))
        probes = {k: torch.randn(torch.cat([o[k] for o in outs], 0).shape, generator=gen).to(dev) for k in keys}
        sum((torch.cat([o[k] for o in outs], 0) * probes[k]).sum() for k in keys).backward()
        f_fast = feats.clone().requires_grad_(True)
        got = sptk.pose_head_over_subsets(fast_head, f_fast, xyz, member, anchors)
        sum((got[k] * probes[k]).sum() for k in keys).backward()
        assert rel_err(f_fast.grad.cpu().numpy(), f_ref.grad.cpu().numpy()) < 5e-5, training
        assert (f_fast.grad.permute(0, 2, 1, 3)[~member] == 0).all(), 'points outside the subset must receive no gradient'
        for (k, v), (_, w) in zip(ref_head.named_parameters(), fast_head.named_parameters()):
            if v.grad is None:
                assert w.grad is None or w.grad.abs().max().item() == 0.0, k
                continue
            ref_g = v.grad.cpu().numpy()
            pre_bn_bias = training and k.endswith('.bias') and (k.startswith('linear.') or k.startswith('trans_linear.') or k.startswith('regressor_dense_layer.0.'))
            if pre_bn_bias:        # a bias in front of a training-mode BatchNorm is absorbed by the batch mean: rounding noise in
                                   # autograd's sum, exactly zero in the fused backward
                top = max(float(p_.grad.abs().max()) for p_ in ref_head.parameters() if p_.grad is not None)
                # (the single-anchor case runs the same expressions as device torch ops: rounding noise there too)
                noise = 0.0 if A % 4 == 0 else 1e-6 * top
                assert np.abs(ref_g).max() < 1e-4 * top and (w.grad is None or np.abs(w.grad.cpu().numpy()).max() <= noise), k
            else:
                assert rel_err(w.grad.cpu().numpy(), ref_g) < 1e-4, (training, k)


# ------------------------------------------------------------------------------------------------------------------
# SURVEY.md 8(f) rows 2 and 4 against fixtures produced by RUNNING THE REFERENCE (tests/golden/make_golden_extra.py)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('tag,lazy', [('lazy', True), ('fps', False)])
def test_strided_pose_conv_matches_reference_golden(dev, golden, tag, lazy):
    """InterSO3PoseConv(stride=2), random per-point poses, anchor permutation on: sample indices / centres / gathered
    poses exact, materialised kernel weights, outputs, dF, dW against the reference's own forward + autograd."""
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    g = golden('strided_pose.npz')
    conv = sptk.InterSO3PoseConv(6, 8, 1, int(g['stride']), float(g['radius']), float(g['sigma']), int(g['nn']), lazy_sample=lazy,
                                 kanchor=60, permute_modes=1)
    np.testing.assert_array_equal(conv.anchors.numpy(), g['anchors'])
    np.testing.assert_array_equal(conv.kernels.numpy(), g['kernels'])
    with torch.no_grad():
        conv.basic_conv.W.copy_(T(g[f'{tag}_W']))
    conv = conv.to(dev)
    fd = T(g[f'{tag}_feats']).to(dev).requires_grad_(True)
    inter_idx, w, sample_idx, out = conv(zptk.SphericalPointCloudPose(T(g['xyz']).to(dev), fd, None, T(g[f'{tag}_pose']).to(dev)))
    assert inter_idx is None
    assert sample_idx.dtype == torch.int64                                      # spconv/functional.py:L476 `idx.long()`
    np.testing.assert_array_equal(sample_idx.cpu().numpy(), g[f'{tag}_sample_idx'])
    np.testing.assert_array_equal(out.xyz.cpu().numpy(), g[f'{tag}_new_xyz'])
    np.testing.assert_array_equal(out.pose.cpu().numpy(), g[f'{tag}_new_pose'])
    assert rel_err(w.materialize()[:, :4].cpu().numpy(), g[f'{tag}_inter_w_head']) < 5e-6
    assert rel_err(out.feats.detach().cpu().numpy(), g[f'{tag}_out']) < 1e-5
    gF, gW = torch.autograd.grad(out.feats, [fd, conv.basic_conv.W], T(g[f'{tag}_grad_out']).to(dev))
    assert rel_err(gF.cpu().numpy(), g[f'{tag}_grad_feats']) < 1e-5
    assert rel_err(gW.cpu().numpy(), g[f'{tag}_grad_W']) < 2e-5


@pytest.mark.parametrize('mode', ['attention', 'max', 'mean'])
def test_invariant_head_matches_reference_golden(dev, golden, mode):
    """InvPPOutBlockOurs against the reference class (base_so3conv.py:L842-917) run on CPU: outputs in training and
    eval mode, anchor confidences, running-statistic updates, and for attention pooling every gradient."""
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    g = golden('inv_head.npz')
    params = {'dim_in': 24, 'mlp': [16, 12], 'fc': [12], 'k': 12, 'kanchor': 60, 'temperature': 3.0}
    head = sptk.InvPPOutBlockOurs(params, norm=1, pooling_method=mode)
    pre = f'{mode}_state_'
    state = {k[len(pre):]: T(v) for k, v in g.items() if k.startswith(pre)}
    assert set(state) == set(head.state_dict()), 'state_dict names differ from the reference class'
    head = head.to(dev)
    for phase in ('train', 'eval'):
        head.load_state_dict(state)
        head.train(phase == 'train')
        x = T(g['x']).to(dev).requires_grad_(True)
        res = head(zptk.SphericalPointCloud(None, x, None))
        y = res[0] if mode == 'attention' else res
        assert rel_err(y.detach().cpu().numpy(), g[f'{mode}_{phase}_out']) < 5e-6
        if mode == 'attention':
            assert rel_err(res[1].detach().cpu().numpy(), g[f'{mode}_{phase}_conf']) < 5e-6
            names = [n for n, _ in head.named_parameters()]
            grads = torch.autograd.grad(y, [x] + list(head.parameters()), T(g[f'{mode}_{phase}_grad_out']).to(dev))
            top = max(float(np.abs(g[f'{mode}_{phase}_grad_{n}']).max()) for n in names)
            for n, got in zip(['x'] + names, grads):
                want = g[f'{mode}_{phase}_grad_{n}']
                # a conv bias in front of a training-mode BatchNorm has an exactly-zero gradient: rounding noise on both sides
                # (the reference's CPU value is ~1e-5 where the weight gradients are ~20), hence the floor
                scale = max(float(np.abs(want).max()), 5e-2 * top)
                assert float(np.abs(got.cpu().numpy() - want).max()) < 5e-5 * scale, (phase, n)
        if phase == 'train':
            for k, v in head.state_dict().items():
                if 'running' in k:
                    assert rel_err(v.cpu().numpy(), g[f'{mode}_after_{k}']) < 1e-5, k


@pytest.mark.parametrize('tag,cd,single', [('cd0_multi', 0, 0), ('cd1_single', 1, 1)])
def test_orbit_selection_matches_reference_statements(dev, golden, tag, cd, single):
    """The 60-way batched chamfer (extensions.chamfer_dist.orbit_reconstruction_distances) + the point-weighted
    aggregation (vgtk.so3conv.orbit_slot_distances) + the arg-min (orbit_selection) against the values the reference's
    own lines ...pn_38_multi_stage.py:L1341-1399 produced on the same inputs (an empty slot included)."""
    import vgtk.so3conv as sptk
    from extensions.chamfer_dist import orbit_reconstruction_distances
    g = golden('orbit.npz')
    hard, attn = T(g[f'{tag}_hard_one_hot_labels']).to(dev), T(g[f'{tag}_attn_ori']).to(dev)
    o2r_all, r2o_all, r2o, o2r = orbit_reconstruction_distances(T(g[f'{tag}_transformed_pts']).to(dev), T(g[f'{tag}_ori_pts']).to(dev), hard)
    d_hard, d_soft, d_all = sptk.orbit_slot_distances(o2r_all, o2r, hard, attn)
    for got, key in ((r2o_all, 'minn_dist_recon_to_ori_all_pts'), (r2o, 'minn_dist_recon_to_ori'), (d_hard, 'minn_dist_ori_to_recon_hard'),
                     (d_soft, 'minn_dist_ori_to_recon'), (d_all, 'minn_dist_ori_to_recon_all_pts')):
        np.testing.assert_allclose(got.cpu().numpy(), g[f'{tag}_{key}'], rtol=2e-5, atol=1e-7, err_msg=key)
    dist, orbit = sptk.orbit_selection(d_soft, r2o, slot_single_cd=cd, slot_single_mode=single)
    np.testing.assert_array_equal(orbit.cpu().numpy(), g[f'{tag}_slot_orbits'])
    np.testing.assert_allclose(dist.cpu().numpy(), g[f'{tag}_slot_dist'], rtol=2e-5)


@pytest.mark.parametrize('M,N,K,batch', [(512, 2048, 3072, 2), (512, 1024, 6144, 1), (128, 1920, 1536, 2), (384, 768, 48, 3), (256, 256, 16, 1), (512, 384, 32, 2)])
def test_split_bf16_contraction_is_fp32_accurate(dev, M, N, K, batch):
    """csrc/gemm_bf16x3.hip: C = A B^T with fp32 operands on the bf16 matrix cores (3 x bf16 split, six partial products,
    fp32 accumulation) against fp64 -- its error must be of the size of the fp32-MFMA kernel's own (an fmaf chain), on
    operands like the path's (weights ~N(0, s), grouped features non-negative with a wide dynamic range) and on edge tiles."""
    from vgtk import _hip
    gen = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=gen) * 0.05).to(dev)
    B = (torch.randn(batch, N, K, generator=gen).abs() * torch.exp(torch.randn(batch, N, 1, generator=gen) * 2.0)).to(dev)
    ref = torch.matmul(A.double().cpu(), B.double().cpu().transpose(1, 2))                      # [batch, M, N]
    out = {}
    for split in (True, False):
        _hip.SPLIT_BF16_CONTRACTION = split
        try:
            C = torch.full((batch, M, N), float('nan'), device=dev)
            _hip.gemm(0, 1, M, N, K, A, K, 0, B, K, N * K, C, N, M * N, batch)
        finally:
            _hip.SPLIT_BF16_CONTRACTION = True
        out[split] = C.double().cpu()
    scale = ref.abs().max().item()
    err_split = (out[True] - ref).abs().max().item() / scale
    err_fp32 = (out[False] - ref).abs().max().item() / scale
    # per-element bound: |a|.|b| accumulated in fp32
    bound = torch.matmul(A.abs().double().cpu(), B.abs().double().cpu().transpose(1, 2))
    rel_el = ((out[True] - ref).abs() / bound.clamp_min(1e-300)).max().item()
    print(f'\nGEMM {M}x{N}x{K}: split {err_split:.2e}, fp32 MFMA {err_fp32:.2e} (max error / max |C|); split per-element {rel_el:.2e} of sum |a||b|')
    # both kernels accumulate K products in fp32: their errors are of the same size (the split adds ~2^-23 per product).
    # K = 6144 is the intra conv's C * 12 at C = 512.
    rms = lambda x: float((x - ref).pow(2).mean().sqrt()) / scale
    assert err_split < 2 * err_fp32 + 1e-7, (err_split, err_fp32)
    assert rms(out[True]) < 2 * rms(out[False]) + 1e-8, (rms(out[True]), rms(out[False]))
    assert rel_el < 1e-6, rel_el


@pytest.mark.parametrize('M,N,K,batch', [(512, 2048, 512, 2), (256, 1920, 256, 2), (128, 768, 48, 3), (384, 256, 16, 1)])
def test_split_bf16_pointwise_contraction_is_fp32_accurate(dev, M, N, K, batch):
    """The same kernel reading B_z row-major [K, N] (so3_contract: W [O, C] times x_z [C, P*A]) against fp64, with the
    fp32-MFMA kernel's error as the yardstick; then so3_contract forward + both gradients with the split on and off."""
    from vgtk import _hip
    import vgtk.so3conv.functional as L
    gen = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=gen) * 0.05).to(dev)
    B = (torch.randn(batch, K, N, generator=gen).abs() * torch.exp(torch.randn(batch, 1, N, generator=gen) * 2.0)).to(dev)
    assert _hip.lib.eap_gemm_bf16x3_nn_f32_supported(M, N, K, _hip._ptr(A), _hip._I64(K), _hip._ptr(B), _hip._I64(N), _hip._I64(K * N))
    ref = torch.matmul(A.double().cpu(), B.double().cpu())                                       # [batch, M, N]
    out = {}
    for split in (True, False):
        _hip.SPLIT_BF16_CONTRACTION = split
        try:
            C = torch.full((batch, M, N), float('nan'), device=dev)
            _hip.gemm(0, 0, M, N, K, A, K, 0, B, N, K * N, C, N, M * N, batch)
        finally:
            _hip.SPLIT_BF16_CONTRACTION = True
        out[split] = C.double().cpu()
    scale = ref.abs().max().item()
    err_split = (out[True] - ref).abs().max().item() / scale
    err_fp32 = (out[False] - ref).abs().max().item() / scale
    bound = torch.matmul(A.abs().double().cpu(), B.abs().double().cpu())
    rel_el = ((out[True] - ref).abs() / bound.clamp_min(1e-300)).max().item()
    print(f'\nNN GEMM {M}x{N}x{K}: split {err_split:.2e}, fp32 MFMA {err_fp32:.2e}; split per-element {rel_el:.2e} of sum |a||b|')
    assert err_split < 3 * err_fp32 + 2e-7, (err_split, err_fp32)
    assert rel_el < 1e-6, rel_el
    # the autograd op on top of it (dX through the transposed weights on the same kernel)
    res = {}
    gy = torch.randn(batch, M, N, generator=gen).to(dev)
    for split in (True, False):
        _hip.SPLIT_BF16_CONTRACTION = split
        try:
            W = A.clone().requires_grad_(True)
            x = B.clone().requires_grad_(True)
            y = L.so3_contract(W, x)
            gW, gx = torch.autograd.grad(y, [W, x], gy)
        finally:
            _hip.SPLIT_BF16_CONTRACTION = True
        res[split] = (y.detach(), gW, gx)
    for name, a, b in zip(('y', 'dW', 'dx'), res[True], res[False]):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5, name


@pytest.mark.parametrize('B,C,O,P', [(2, 64, 128, 64), (1, 128, 256, 32), (2, 16, 128, 128)])
def test_split_bf16_intra_conv_matches_fp32_kernel_and_fp64(dev, B, C, O, P):
    """eap_so3_intra_conv_bf16x3_f32 (implicit gather on the split kernel) against the gathered fp64 einsum and the fp32-MFMA
    implicit kernel; forward and dF (the same kernel with the inverse table)."""
    from vgtk import _hip
    import vgtk.so3conv.functional as L
    A_ = 60
    gen = torch.Generator().manual_seed(B + C + O + P)
    idx = torch.from_numpy(np.ascontiguousarray(L.get_intra_idx())).long().to(dev)                # [60, 12]
    nt = idx.shape[1]
    feats = (torch.randn(B, C, P, A_, generator=gen).abs() * torch.exp(torch.randn(B, 1, P, 1, generator=gen))).to(dev)
    W = (torch.randn(O, C * nt, generator=gen) * 0.05).to(dev)
    assert _hip.lib.eap_so3_intra_conv_bf16x3_f32_supported(B, O, C, P, A_, nt)
    g = feats.double().cpu()[:, :, :, idx.cpu()]                                                 # [B, C, P, A, T]
    ref = torch.einsum('oct,bcpat->bopa', W.double().cpu().view(O, C, nt), g)
    out = {}
    gy = torch.randn(B, O, P, A_, generator=gen).to(dev)
    for split in (True, False):
        _hip.SPLIT_BF16_CONTRACTION = split
        try:
            f = feats.clone().requires_grad_(True)
            y = L.intra_so3conv(f, W, idx)
            gF, = torch.autograd.grad(y, [f], gy)
        finally:
            _hip.SPLIT_BF16_CONTRACTION = True
        out[split] = (y.detach().double().cpu(), gF.double().cpu())
    scale = ref.abs().max().item()
    err_split = (out[True][0] - ref).abs().max().item() / scale
    err_fp32 = (out[False][0] - ref).abs().max().item() / scale
    print(f'\nintra conv B{B} C{C} O{O} P{P}: split {err_split:.2e}, fp32 MFMA {err_fp32:.2e}')
    assert err_split < 3 * err_fp32 + 2e-7, (err_split, err_fp32)
    assert rel_err(out[True][1].numpy(), out[False][1].numpy()) < 1e-5       # two fp32 accumulations of K = 12 O products against each other


def test_split_contraction_presplit_weights_is_bit_identical(dev):
    """eap_gemm_bf16x3_presplit: the shared operand split once per call (scratch from hipMallocAsync) or inside the k-loop --
    the same roundings, so the three layouts must agree bit for bit."""
    from vgtk import _hip
    import vgtk.so3conv.functional as L
    gen = torch.Generator().manual_seed(77)
    idx = torch.from_numpy(np.ascontiguousarray(L.get_intra_idx())).to(torch.int32).to(dev)
    W = (torch.randn(256, 768, generator=gen) * 0.05).to(dev)
    Bt = torch.randn(2, 1920, 768, generator=gen).to(dev)            # k-contiguous rows
    Bn = torch.randn(2, 768, 1920, generator=gen).to(dev)            # row-major [K, N]
    feats = torch.randn(2, 64, 32, 60, generator=gen).to(dev)        # intra: K = 64 * 12 = 768
    outs = {}
    for on in (2, 0):
        was = _hip.lib.eap_gemm_bf16x3_presplit(on)
        try:
            c1 = torch.empty(2, 256, 1920, device=dev); c2 = torch.empty(2, 256, 1920, device=dev)
            _hip.gemm(0, 1, 256, 1920, 768, W, 768, 0, Bt, 768, 1920 * 768, c1, 1920, 256 * 1920, 2)
            _hip.gemm(0, 0, 256, 1920, 768, W, 768, 0, Bn, 1920, 768 * 1920, c2, 1920, 256 * 1920, 2)
            c3 = _hip.so3_intra_conv(feats, W, idx)
            outs[on] = (c1.cpu(), c2.cpu(), c3.cpu())
        finally:
            _hip.lib.eap_gemm_bf16x3_presplit(was)
    assert _hip.lib.eap_gemm_bf16x3_presplit(-1) == 1, 'pre-split weights are the default'
    for a, b in zip(outs[2], outs[0]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('c,o', [(1, 64), (256, 3), (24, 1), (64, 128)])
def test_pointwise_conv_equals_conv2d(dev, c, o):
    """vgtk.so3conv.pointwise_conv (the contraction kernels behind every 1x1 conv of blocks and heads, narrow and single-channel
    shapes included) against nn.Conv2d: output and all three gradients; with the bias folded into BatchNormLeakyReLU
    (pre_bias) against conv -> BatchNorm2d -> leaky_relu."""
    import vgtk.so3conv as sptk
    torch.manual_seed(c * 7 + o)
    conv = torch.nn.Conv2d(c, o, 1).to(dev)
    x = torch.randn(2, c, 37, 60, device=dev)
    g = torch.randn(2, o, 37, 60, device=dev)
    res = []
    for fn in (lambda t: conv(t), lambda t: sptk.pointwise_conv(conv, t)):
        xi = x.clone().requires_grad_(True)
        y = fn(xi)
        grads = torch.autograd.grad(y, [xi, conv.weight, conv.bias], g)
        res.append((y.detach(),) + grads)
    for name, a, b in zip(('y', 'dx', 'dW', 'dbias'), res[1], res[0]):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5, name
    # bias folded into the fused BatchNorm + leaky_relu
    bn_ref = torch.nn.BatchNorm2d(o).to(dev)
    bn = sptk.BatchNormLeakyReLU(o, negative_slope=0.01).to(dev)
    with torch.no_grad():
        bn_ref.weight.uniform_(0.5, 1.5); bn_ref.bias.uniform_(-0.3, 0.3)
    bn.load_state_dict(bn_ref.state_dict())
    for training in (True, False):
        bn_ref.train(training); bn.train(training)
        xi = x.clone().requires_grad_(True)
        ref = torch.nn.functional.leaky_relu(bn_ref(conv(xi)), 0.01)
        gr = torch.autograd.grad(ref, [xi, conv.weight], g)
        xj = x.clone().requires_grad_(True)
        got = bn(sptk.pointwise_conv(conv, xj, add_bias=False), pre_bias=conv.bias)
        gg = torch.autograd.grad(got, [xj, conv.weight], g)
        # c = 1: a channel is w * x + bias with |w| down to 1e-3 -- adding the bias first (the reference order) rounds the values
        # at 6e-8 while their spread is 1e-3, i.e. the REFERENCE order carries ~1e-4 of noise into the normalised output
        tol = 3e-4 if c == 1 else 1e-5
        assert rel_err(got.detach().cpu().numpy(), ref.detach().cpu().numpy()) < tol, training
        for name, a, b in zip(('dx', 'dW'), gg, gr):
            if c == 1 and training and name == 'dW':
                continue        # BatchNorm is invariant to the scale of its input channel: with one input channel dW is zero up to eps, all noise
            assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 5 * tol, (training, name)
        assert rel_err(bn.running_mean.cpu().numpy(), bn_ref.running_mean.cpu().numpy()) < 1e-5
        assert rel_err(bn.running_var.cpu().numpy(), bn_ref.running_var.cpu().numpy()) < 1e-5


@pytest.mark.parametrize('b,o,c,n', [(2, 3, 40, 42000), (1, 4, 17, 8196), (3, 1, 256, 2220), (2, 2, 5, 4)])
def test_narrow_contraction_equals_the_gemm_path(dev, b, o, c, n):
    """csrc/narrow_contract.hip (1-4 output channels, streaming) against the general contraction and fp64: forward, dX, dW over
    several column slabs, a ragged channel group, a single 4-column row."""
    import vgtk.so3conv.functional as L
    gen = torch.Generator().manual_seed(b + o + c + n)
    W = torch.randn(o, c, generator=gen).to(dev)
    x = torch.randn(b, c, n, generator=gen).to(dev)
    g = torch.randn(b, o, n, generator=gen).to(dev)
    res = []
    for fn in (L._NarrowContract, L._Contract):
        Wi, xi = W.clone().requires_grad_(True), x.clone().requires_grad_(True)
        y = fn.apply(Wi, xi)
        gW, gx = torch.autograd.grad(y, [Wi, xi], g)
        res.append((y.detach(), gW, gx))
    ref = (torch.matmul(W.double().cpu(), x.double().cpu()),
           torch.einsum('bon,bcn->oc', g.double().cpu(), x.double().cpu()),
           torch.matmul(W.double().cpu().t(), g.double().cpu()))
    for name, a, bb, r in zip(('y', 'dW', 'dx'), res[0], res[1], ref):
        assert rel_err(a.cpu().numpy(), r.numpy()) < 3e-6, name
        assert rel_err(a.cpu().numpy(), bb.cpu().numpy()) < 5e-6, name
    y = L.so3_contract(W, x)
    assert torch.equal(y, res[0][0]), 'so3_contract must take the streaming kernels for <= 4 output channels'


def test_pose_heads_over_slot_groups_equal_the_masked_form(dev):
    """vgtk.so3conv.pose_head_over_slot_groups (every slot's member points compacted: work of P points per cloud) against
    pose_head_over_subsets on the full clouds with the membership mask (slots x P): outputs, gradients of the feature map and
    of the head parameters, running statistics; one cloud leaves a slot empty (fallback to the whole cloud), capacities are
    padded past the cloud size."""
    import copy
    import vgtk.so3conv as sptk
    import vgtk.so3conv.functional as L
    torch.manual_seed(21)
    B, C, N, A, S = 3, 16, 70, 60, 2
    heads = [sptk.SO3OutBlockRTWithMaskSep({'dim_in': C, 'mlp': [32, 32], 'kanchor': A, 'temperature': 3.0}, norm=1, pooling_method='mean',
                                           pred_axis=True, pred_central_points=True).to(dev) for _ in range(S)]
    feats = torch.randn(B, C, N, A, device=dev)
    xyz = torch.randn(B, 3, N, device=dev) * 0.3
    labels = torch.randint(0, S, (B, N), device=dev)
    labels[1] = 0                                                     # cloud 1: nobody chose slot 1
    anchors = torch.from_numpy(np.ascontiguousarray(L.get_anchors(A))).to(dev)
    groups = sptk.slot_point_groups(labels, S)
    assert [tuple(g.shape) for g in groups] == [(B, 96), (B, 96)] and all(g.dtype == torch.int32 for g in groups)
    for s_, g in enumerate(groups):
        for b in range(B):
            want = (labels[b] == s_).nonzero().squeeze(1)
            want = want if want.numel() else torch.arange(N, device=dev)
            got = g[b][g[b] >= 0].long()
            assert torch.equal(got, want) and (g[b][want.numel():] == -1).all()
    ref_heads, new_heads = copy.deepcopy(heads), copy.deepcopy(heads)
    f_ref, f_new = feats.clone().requires_grad_(True), feats.clone().requires_grad_(True)
    outs_ref = []
    for s_, head in enumerate(ref_heads):
        member = labels == s_
        member = member | (member.sum(1, keepdim=True) == 0)
        outs_ref.append(sptk.pose_head_over_subsets(head, f_ref, xyz, member, anchors))
    outs_new = sptk.pose_head_over_slot_groups(new_heads, f_new, xyz, labels, anchors)
    keys = ('R', 'T', 'axis', 'central_points')
    gen = torch.Generator().manual_seed(3)
    loss_ref = loss_new = 0.0
    for o_r, o_n in zip(outs_ref, outs_new):
        for k in keys:
            assert rel_err(o_n[k].detach().cpu().numpy(), o_r[k].detach().cpu().numpy()) < 1e-5, k
            probe = torch.randn(o_r[k].shape, generator=gen).to(dev)
            loss_ref = loss_ref + (o_r[k] * probe).sum()
            loss_new = loss_new + (o_n[k] * probe).sum()
    loss_ref.backward(); loss_new.backward()
    assert rel_err(f_new.grad.cpu().numpy(), f_ref.grad.cpu().numpy()) < 2e-5
    for hr, hn in zip(ref_heads, new_heads):
        for (k, v), (_, w) in zip(hr.named_parameters(), hn.named_parameters()):
            if v.grad is not None and float(v.grad.abs().max()) > 0:
                assert rel_err(w.grad.cpu().numpy(), v.grad.cpu().numpy()) < 1e-4, k
        for (k, v), (_, w) in zip(hr.state_dict().items(), hn.state_dict().items()):
            if 'running' in k:
                assert rel_err(w.cpu().numpy(), v.cpu().numpy()) < 1e-5, k


@pytest.mark.parametrize('shape', [(3, 8, 40, 60), (2, 5, 700, 60), (1, 4, 1, 12), (2, 3, 17, 64)])
def test_masked_max_equals_torch(dev, shape):
    """vgtk.so3conv.masked_max (csrc/heads.hip) against (x * mask).max(2): values (non-members contribute their zeros, so a
    subset of negative values pools to 0), arg-max gradient, a cloud with a single member."""
    import vgtk.so3conv as sptk
    torch.manual_seed(sum(shape))
    B, C, N, A = shape
    x = torch.randn(B, C, N, A, device=dev)
    x[0, 0] = -x[0, 0].abs() - 0.1                                    # channel 0 of cloud 0: all negative
    member = torch.rand(B, N, device=dev) > 0.5
    member[:, 0] = True
    if N > 2:
        member[-1] = False; member[-1, N // 2] = True                  # one member only
    m = member.float()
    g = torch.randn(B, C, A, device=dev)
    xr = x.clone().requires_grad_(True)
    ref = (xr * m.view(B, 1, N, 1)).max(2)[0]
    gr, = torch.autograd.grad(ref, [xr], g)
    xn = x.clone().requires_grad_(True)
    got = sptk.masked_max(xn, member)
    gn, = torch.autograd.grad(got, [xn], g)
    assert torch.equal(got, ref)
    if not bool(member.all()):
        assert float(got[0, 0].max()) == 0.0
    # where the maximum is attained by a member the gradients agree; where it is one of the non-members' zeros both are zero
    assert torch.equal(gn, gr) or rel_err(gn.cpu().numpy(), gr.cpu().numpy()) == 0.0


def test_zp_layers_match_the_reference(dev, golden):
    """vgtk.spconv.modules (IntraZPConv, InterZPConv incl. a strided call, AnchorProp) on the HIP zpconv kernels + the
    contraction GEMM against outputs and autograd gradients of the reference's own classes run on CPU
    (tests/golden/make_golden_zp.py -> zp_layer.npz; reference vgtk/vgtk/spconv/modules.py:L17-161)."""
    import vgtk.spconv as zptk
    g = golden('zp_layer.npz')
    xyz = T(g['xyz']).to(dev)

    intra = zptk.IntraZPConv(6, 9, 3, 1.2, 0.1, 4, 12)
    np.testing.assert_array_equal(intra.intra_idx.numpy(), g['intra_idx'])
    assert rel_err(intra.intra_w.numpy(), g['intra_w']) < 1e-6
    intra.load_state_dict({'basic_conv.W': T(g['intra_W']), 'basic_conv.bias': T(g['intra_bias'])}, strict=False)
    intra = intra.to(dev)
    fi = T(g['intra_in']).to(dev).requires_grad_(True)
    y = intra(zptk.SphericalPointCloud(xyz, fi, None)).feats
    gin, gW, gb = torch.autograd.grad(y, [fi, intra.basic_conv.W, intra.basic_conv.bias], T(g['intra_gy']).to(dev))
    for got, want in ((y, 'intra_out'), (gin, 'intra_gin'), (gW, 'intra_gW'), (gb, 'intra_gbias')):
        assert rel_err(got.detach().cpu().numpy(), g[want]) < 1e-5, want

    inter = zptk.InterZPConv(6, 5, 1, 1, 0.25, 1.2, 0.05, 12, 12, 4)
    assert rel_err(inter.kernels.numpy(), g['inter_kernels']) == 0
    inter.load_state_dict({'basic_conv.W': T(g['inter_W']), 'basic_conv.bias': T(g['inter_bias'])}, strict=False)
    inter = inter.to(dev)
    fe = T(g['inter_in']).to(dev).requires_grad_(True)
    iidx, iw, cloud = inter(zptk.SphericalPointCloud(xyz, fe, None))
    np.testing.assert_array_equal(iidx.cpu().numpy(), g['inter_idx'])
    assert tuple(iw.shape) == g['inter_w'].shape and np.abs(iw.cpu().numpy() - g['inter_w']).max() < 1e-5
    gin, gW, gb = torch.autograd.grad(cloud.feats, [fe, inter.basic_conv.W, inter.basic_conv.bias], T(g['inter_gy']).to(dev))
    for got, want in ((cloud.feats, 'inter_out'), (gin, 'inter_gin'), (gW, 'inter_gW'), (gb, 'inter_gbias')):
        assert rel_err(got.detach().cpu().numpy(), g[want]) < 2e-5, want
    # tables handed back in: the second call skips ball query and weights
    _, _, again = inter(zptk.SphericalPointCloud(xyz, fe.detach(), None), iidx, iw)
    assert torch.equal(again.feats, cloud.feats)

    inter2 = zptk.InterZPConv(6, 5, 1, 2, 0.25, 1.2, 0.05, 12, 12, 4)
    inter2.load_state_dict({'basic_conv.W': T(g['inter2_W']), 'basic_conv.bias': T(g['inter2_bias'])}, strict=False)
    i2, w2, c2 = inter2.to(dev)(zptk.SphericalPointCloud(xyz, fe.detach(), None))
    np.testing.assert_array_equal(i2.cpu().numpy(), g['inter2_idx'])
    np.testing.assert_array_equal(c2.xyz.cpu().numpy(), g['inter2_xyz'])
    assert rel_err(c2.feats.detach().cpu().numpy(), g['inter2_out']) < 2e-5

    prop = zptk.AnchorProp(12, 42, 0.1).to(dev)
    got = prop(zptk.SphericalPointCloud(xyz, fi.detach(), None)).feats
    assert got.shape == (2, 6, 64, 42) and rel_err(got.cpu().numpy(), g['aprop_out']) < 1e-6


@pytest.mark.parametrize('pooling', ['mean', 'max'])
def test_pose_head_over_subsets_with_a_single_anchor(dev, pooling):
    """kanchor = 1 (one anchor per point: rows of 1 float, which the 16-byte head kernels do not take): the batched head
    falls back to the same expressions as device torch ops and still equals the per-cloud loop."""
    test_pose_head_over_subsets_equals_the_per_cloud_loop(dev, pooling, A=1)


def test_streaming_kernels_take_operands_at_any_4_byte_offset(dev):
    """The narrow contraction and the masked max move 16-byte words: a contiguous VIEW that starts 4 bytes into its storage
    (what autograd can hand a backward as its gradient) must give the same results as an aligned copy."""
    import vgtk.so3conv as sptk
    import vgtk.so3conv.functional as L
    gen = torch.Generator().manual_seed(5)
    b, o, c, n = 2, 3, 24, 240
    W = torch.randn(o, c, generator=gen).to(dev)
    store_x = torch.randn(1 + b * c * n, generator=gen).to(dev)
    store_g = torch.randn(1 + b * o * n, generator=gen).to(dev)
    x_off, g_off = store_x[1:].view(b, c, n), store_g[1:].view(b, o, n)
    assert x_off.is_contiguous() and x_off.data_ptr() % 16 == 4 and g_off.data_ptr() % 16 == 4
    res = []
    for x, g in ((x_off, g_off), (x_off.clone(), g_off.clone())):
        Wi, xi = W.clone().requires_grad_(True), x.detach().requires_grad_(True)      # detach() keeps the view's storage offset
        assert xi.data_ptr() == x.data_ptr()
        y = L.so3_contract(Wi, xi)
        gW, gx = torch.autograd.grad(y, [Wi, xi], g)
        res.append((y.detach(), gW, gx))
    for a, bb in zip(res[0], res[1]):
        assert torch.equal(a, bb)
    f = torch.randn(1 + 2 * 8 * 16 * 60, generator=gen).to(dev)
    mask = (torch.rand(2, 16, generator=gen) > 0.4).float().to(dev)
    mask[:, 0] = 1.0
    v = f[1:].view(2, 8, 16, 60)
    assert torch.equal(sptk.masked_max(v, mask), sptk.masked_max(v.clone(), mask))


def test_inference_mode_batchnorm_folds_into_the_contraction(dev, monkeypatch):
    """SURVEY.md 8(f) row 1, second half: in eval mode under no_grad the block layer's BatchNorm2d + leaky_relu (and the
    separable block's skip branch with its sum, base_so3poseconv.py:L214-221, L319-328) ride in the epilogue of the
    contraction that produces their input (csrc/gemm_bf16x3.hip).  Against conv + eval-mode norm as separate passes, and
    the norm against nn.BatchNorm2d + F.leaky_relu; a layer too small for the split kernel takes the separate passes."""
    import synth_clouds
    import torch.nn.functional as F
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    import vgtk.so3conv.functional as L
    from vgtk import _hip
    monkeypatch.setattr(L, 'DENSE_MODE', 'off')            # the list kernels' contraction first; the dense forward's re-ordering pass below
    torch.manual_seed(21)
    B, P = 2, 512
    xyz, _, pose = synth_clouds.laptop_batch(3, B, P)
    xyz, pose = T(xyz).to(dev), T(pose).to(dev)
    c, o = 64, 128
    conv = sptk.InterSO3PoseConv(c, o, 1, 1, 0.3, 0.05, 64, kanchor=60, permute_modes=1).to(dev)
    norm = sptk.BatchNormLeakyReLU(o, negative_slope=0.01).to(dev)
    skip, skip_norm = torch.nn.Conv2d(c, o, 1).to(dev), sptk.BatchNormLeakyReLU(o, negative_slope=0.01).to(dev)
    with torch.no_grad():
        for n_ in (norm, skip_norm):
            n_.weight.uniform_(0.5, 1.5); n_.bias.uniform_(-0.3, 0.3); n_.running_mean.uniform_(-0.5, 0.5); n_.running_var.uniform_(0.5, 2.0)
    feats = torch.randn(B, c, P, 60, device=dev)
    x = zptk.SphericalPointCloudPose(xyz, feats, None, pose)
    for m in (conv, norm, skip, skip_norm):
        m.eval()
    launched = []
    _hip.KERNEL_TIMES = launched
    try:
        with torch.no_grad():
            _, _, _, fused = sptk.conv_norm_act(conv, norm, x)
            raw = conv(x)[3].feats
            separate = norm(raw)
            ref_bn = torch.nn.BatchNorm2d(o).to(dev).eval()
            ref_bn.load_state_dict({k: v for k, v in norm.state_dict().items()})
            torch_ref = F.leaky_relu(ref_bn(raw), 0.01)
            res = torch.randn(B, o, P, 60, device=dev)
            fused_skip = sptk.pointwise_norm_act(skip, skip_norm, feats, residual=res)
            separate_skip = skip_norm(sptk.pointwise_conv(skip, feats, add_bias=False), residual=res, pre_bias=skip.bias)
    finally:
        _hip.KERNEL_TIMES = None
    names = [n for n, *_ in launched]
    # both contractions took a kernel with the epilogue: the inter conv's with two fp16 planes (it knows a bound on its operand), the
    # 128-row pointwise one with three bf16 planes (a pass over its operand would cost more than it saves, vgtk/_hip.py _planes2)
    # (eap_gemm_f16x2_f32 twice: conv_norm_act with the epilogue, and the plain conv(x) this test compares it with)
    assert [n for n in names if 'gemm' in n] == ['eap_gemm_f16x2_f32', 'eap_gemm_f16x2_f32', 'eap_gemm_bf16x3_ep_f32', 'eap_gemm_bf16x3_nn_f32'], names
    assert rel_err(separate.cpu().numpy(), torch_ref.cpu().numpy()) < 2e-6
    assert rel_err(fused.feats.cpu().numpy(), separate.cpu().numpy()) < 2e-6
    assert rel_err(fused_skip.cpu().numpy(), separate_skip.cpu().numpy()) < 2e-6
    # the same layer on the dense product (the default where few rows are referenced): the folded norm rides in the re-ordering pass
    monkeypatch.setattr(L, 'DENSE_MODE', 'auto')
    launched = []
    _hip.KERNEL_TIMES = launched
    try:
        with torch.no_grad():
            L.FORWARD_LOG = []
            _, _, _, fused_d = sptk.conv_norm_act(conv, norm, x)
            flog, L.FORWARD_LOG = L.FORWARD_LOG, None
    finally:
        _hip.KERNEL_TIMES = None
    names = [n for n, *_ in launched]
    assert flog[0]['dense'] and 'eap_so3_dense_untranspose_bnact_f32' in names and not any(n.startswith('eap_bn_act') for n in names), names
    assert rel_err(fused_d.feats.cpu().numpy(), separate.cpu().numpy()) < 1e-5
    # training mode or gradients on: nothing is folded, same API
    norm.train()
    _, _, _, tr = sptk.conv_norm_act(conv, norm, x)
    assert tr.feats.requires_grad
    # a first layer (1 -> 64 channels, K = 24): below the split kernel's tile -> conv and norm as separate passes, same result
    conv0 = sptk.InterSO3PoseConv(1, 64, 1, 1, 0.3, 0.05, 64, kanchor=60, permute_modes=1).to(dev).eval()
    norm0 = sptk.BatchNormLeakyReLU(64, negative_slope=0.01).to(dev).eval()
    x0 = zptk.SphericalPointCloudPose(xyz, torch.ones(B, 1, P, 60, device=dev), None, pose)
    with torch.no_grad():
        a = sptk.conv_norm_act(conv0, norm0, x0)[3].feats
        bsep = norm0(conv0(x0)[3].feats)
    assert torch.equal(a, bsep)


@pytest.mark.parametrize('tag,pm', [('parts_pm1', 1), ('random_pm0', 0)])
def test_art_mode_conv_matches_the_reference(dev, golden, tag, pm):
    """InterSO3PoseConv(use_art_mode=True) (so3conv/modules.py:L256-262, functional.py:L1420-1520): the cloud in n_states
    articulation states, a per-point label picking the state whose ball query and (unrotated) offsets the point uses, poses
    selecting the anchor permutation only -- output, feature and weight gradients against the reference's own layer run on CPU
    (tests/golden/make_golden_artmode.py)."""
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    g = golden('inter_pose_artmode.npz')
    xyz, seg = T(g['xyz']).to(dev), T(g['seg']).to(dev)
    conv = sptk.InterSO3PoseConv(6, 8, 1, 1, 0.2, 0.02, 16, kanchor=60, permute_modes=pm, use_art_mode=True)
    conv.basic_conv.W.data.copy_(T(g[f'{tag}_W']))
    conv = conv.to(dev)
    feats = T(g[f'{tag}_feats']).to(dev).requires_grad_(True)
    pose = T(g[f'{tag}_pose']).to(dev)
    inter_idx, inter_w, sample_idx, y = conv(zptk.SphericalPointCloudPose(xyz, feats, None, pose), seg=seg)
    assert inter_idx is None and sample_idx is None and tuple(y.xyz.shape) == tuple(xyz.shape)
    w = inter_w.materialize() if hasattr(inter_w, 'materialize') else inter_w
    assert np.abs(w[:, ::8, ::7, ::5].cpu().numpy() - g[f'{tag}_inter_w_sample']).max() < 2e-6
    gf, gW = torch.autograd.grad(y.feats, [feats, conv.basic_conv.W], T(g[f'{tag}_gy']).to(dev))
    assert rel_err(y.feats.detach().cpu().numpy(), g[f'{tag}_out']) < 1e-5
    assert rel_err(gf.cpu().numpy(), g[f'{tag}_gfeats']) < 1e-5
    assert rel_err(gW.cpu().numpy(), g[f'{tag}_gW']) < 2e-5


def _random_rotations(rng, shape):
    q = rng.standard_normal(shape + (4,))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                     2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(shape + (3, 3)).astype(np.float32)


@pytest.mark.parametrize('kind', ['parts', 'random', 'mixed'])
def test_permuted_clouds_on_the_two_tile_kernel(dev, monkeypatch, kind):
    """Clouds WITH anchor permutations on csrc/so3_inter_lists2.hip (PERM: coset-major anchor axis, block moves by the DMA source
    addresses, per-entry words from eap_so3_perm_entries_f32) against the whole-row kernels of csrc/so3_inter_inv.hip, which the
    golden layers from the reference pin (tests/test_gpu_parity.py): output, feature gradient and weight gradient of a 64 -> 128
    layer; 'mixed' = one cloud without rotations beside two with (both kernels launched, each skipping the other's clouds).
    Bars: 2e-6 of the tensor's scale (the same products, summed in another order across anchors groups), 4e-6 for the output."""
    import synth_clouds
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    import vgtk.so3conv.functional as L
    from vgtk import _hip
    torch.manual_seed(5)
    monkeypatch.setattr(L, 'BACKWARD_MODE', 'inverse')          # the re-associated backward whatever the number of referenced rows
    monkeypatch.setattr(L, 'DENSE_MODE', 'off')                 # (one rotation per rigid part would take the per-part dense product: this test is about the list kernels)
    B, P, c, o = 3, 640, 64, 128
    xyz, lab, pose = synth_clouds.laptop_batch(11, B, P)
    rng = np.random.default_rng(4)
    pose = pose.copy()
    if kind == 'parts':
        R = _random_rotations(rng, (B, 2))
        for bi in range(B):
            pose[bi, :, :3, :3] = R[bi][lab[bi]]
    else:
        pose[:, :, :3, :3] = _random_rotations(rng, (B, P))
        if kind == 'mixed':
            pose[1] = np.eye(4, dtype=np.float32)
    xyz, pose = T(xyz).to(dev), T(pose).to(dev)
    conv = sptk.InterSO3PoseConv(c, o, 1, 1, 0.3, 0.05, 64, kanchor=60, permute_modes=1).to(dev)
    assert _hip.so3_group_perm_lists2_takes(c, 60, 24, P) and _hip.so3_group_perm_lists2_takes(o, 60, 24, P)
    gy = torch.randn(B, o, P, 60, device=dev)
    res = {}
    for two_tile in (1, 0):
        was = _hip.lib.eap_so3_group_perm_lists2(two_tile)
        launched = []
        _hip.KERNEL_TIMES = launched
        try:
            f = torch.randn(B, c, P, 60, device=dev, generator=torch.Generator(device=dev).manual_seed(9), requires_grad=True)
            y = conv(zptk.SphericalPointCloudPose(xyz, f, None, pose))[3].feats
            gf, gw = torch.autograd.grad(y, [f, conv.basic_conv.W], gy)
        finally:
            _hip.KERNEL_TIMES = None
            _hip.lib.eap_so3_group_perm_lists2(was)
        res[two_tile] = (y.detach(), gf, gw, [n for n, *_ in launched])
    names = res[1][3]
    assert 'eap_so3_inter_group_fwd_perm2_t_f32' in names and 'eap_so3_perm_entries_f32' in names, names
    assert 'eap_so3_inter_group_inv_perm2_f32' in names, names                 # (the re-associated backward took the lists)
    assert not any('perm2' in n for n in res[0][3])
    # (the output's contraction sums its 1536 columns in another order when the intermediate is in store order: 4e-6)
    for a, b_, what, bar in zip(res[1][:3], res[0][:3], ('output', 'feature gradient', 'weight gradient'), (4e-6, 2e-6, 2e-6)):
        assert rel_err(a.cpu().numpy(), b_.cpu().numpy()) < bar, what


def test_perm_entries_words(dev):
    """eap_so3_perm_entries_f32 against the tables it packs (include/eap_hip.h): byte offsets = point row + 16 * source block of
    each piece, XOR bits, rotated / dead offset vectors, skipped clouds."""
    import vgtk.so3conv.functional as L
    from vgtk import _hip
    A = torch.from_numpy(np.ascontiguousarray(L.get_anchors())).to(dev)
    mult, ident = L._group_tables(A)
    order, code, pos = L._coset_tables(L._group_tables_inverse(mult), ident)
    rng = np.random.default_rng(3)
    b, per, n = 3, 500, 777
    ent_p = rng.integers(0, n + 40, (b, per)).astype(np.int32)             # some past the last row: shadow neighbours
    g = rng.standard_normal((b, per, 4)).astype(np.float32)
    r = rng.integers(0, 60, (b, per)).astype(np.int32)
    g[..., 3] = r.view(np.float32)
    flags = torch.tensor([1, 0, 1], dtype=torch.int32, device=dev)
    for anchors in (A, None):
        pc, g2 = _hip.so3_perm_entries(T(ent_p).to(dev), T(g).to(dev), code, anchors, ident, 60, n, flags)
        pc, g2 = pc.cpu().numpy().astype(np.int64), g2.cpu().numpy()
        cd = code.cpu().numpy().astype(np.int64)
        An = A.cpu().numpy()
        for bi in (0, 2):
            shadow = ent_p[bi] >= n
            row = np.where(shadow, 0, ent_p[bi].astype(np.int64) * 240)
            sig = cd[r[bi]] & 15                                              # [per,16]
            want = (row[:, None] + 16 * sig).reshape(per, 4, 4)
            live = np.ones((4, 4), bool); live[3, 3] = False                  # block 15 does not exist
            assert np.array_equal(pc[bi][:, live], want[:, live])
            xb = g2[bi, :, 3].view(np.uint32).astype(np.int64)
            for blk in range(15):
                assert np.array_equal((xb >> (2 * blk)) & 3, cd[r[bi], blk] >> 4)
            vec = g[bi, :, :3]
            if anchors is not None:
                vec = np.einsum('eij,ej->ei', An[r[bi]], vec)
            assert np.allclose(g2[bi, ~shadow, :3], vec[~shadow], atol=1e-6)
            assert (g2[bi, shadow, :3] > 1e17).all()


@pytest.mark.parametrize('poses', ['none', 'identity', 'parts'])
def test_store_order_columns_of_the_streamed_intermediate(dev, monkeypatch, poses):
    """The streamed forward keeps the transposed intermediate's columns in the grouping kernel's store order and hands the
    contraction W[:, columns] (csrc/so3_inter_lists2.hip LAYOUT 4): same layer output as with plain columns up to the summation
    order of the contraction (4e-6 of the scale: 1536 fp32 terms), the gradients (which never see that intermediate) bit-equal."""
    import synth_clouds
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    import vgtk.so3conv.functional as L
    from vgtk import _hip
    torch.manual_seed(8)
    monkeypatch.setattr(L, 'BACKWARD_MODE', 'inverse')
    B, P, c, o = 2, 600, 64, 128
    xyz, lab, pose = synth_clouds.laptop_batch(21, B, P)
    if poses == 'parts':
        R = _random_rotations(np.random.default_rng(6), (B, 2))
        pose = pose.copy()
        for bi in range(B):
            pose[bi, :, :3, :3] = R[bi][lab[bi]]
    xyz, pose = T(xyz).to(dev), T(pose).to(dev)
    if poses == 'none':             # the layer without poses (so3conv/modules.py:L125-174): no permutation table at all
        conv = sptk.InterSO3Conv(c, o, 1, 1, 0.3, 0.05, 64, kanchor=60).to(dev)
    else:
        conv = sptk.InterSO3PoseConv(c, o, 1, 1, 0.3, 0.05, 64, kanchor=60, permute_modes=1).to(dev)
    assert _hip.so3_group_fwd_tp_takes(c, 60, 24)
    gy = torch.randn(B, o, P, 60, device=dev)
    res = {}
    for on in (True, False):
        monkeypatch.setattr(L, 'STORE_ORDER_COLUMNS', on)
        launched = []
        _hip.KERNEL_TIMES = launched
        try:
            f = torch.randn(B, c, P, 60, device=dev, generator=torch.Generator(device=dev).manual_seed(3), requires_grad=True)
            if poses == 'none':
                y = conv(zptk.SphericalPointCloud(xyz, f, None))[-1].feats
            else:
                y = conv(zptk.SphericalPointCloudPose(xyz, f, None, pose))[3].feats
            gf, gw = torch.autograd.grad(y, [f, conv.basic_conv.W], gy)
        finally:
            _hip.KERNEL_TIMES = None
        res[on] = (y.detach(), gf, gw, [(n, t) for n, t, *_ in launched])
    shapes = [t['shape'][0] for n, t in res[True][3] if t is not None and 'shape' in t]
    # (with a pose tensor the layer hands over the permutation table and per-cloud flags: one entry launches both kernels)
    assert ('group_fwd_tp' in shapes) if poses == 'none' else ('group_fwd_perm2' in shapes), shapes
    assert not torch.equal(res[True][0], res[False][0]), 'the switch changed nothing: the store-order path did not run'

    assert rel_err(res[True][0].cpu().numpy(), res[False][0].cpu().numpy()) < 4e-6
    assert torch.equal(res[True][1], res[False][1]) and torch.equal(res[True][2], res[False][2])


@pytest.mark.parametrize('M,N,K,batch', [(64, 24, 245760, 2), (64, 24, 20004, 3), (40, 17, 9000, 1), (32, 32, 4096, 5), (7, 3, 5000, 2)])
def test_gemm_skinny_reduce(dev, M, N, K, batch):
    """csrc/gemm_skinny.hip: C[M,N] = sum_b A_b[M,K] B_b[N,K]^T for a small output over a long contraction (the first layer's
    weight gradient at the bench shape first) against float64; ragged M, N and K tails; run-to-run bit equality; and
    vgtk._hip.gemm_reduce routes such operands there."""
    from vgtk import _hip
    gen = torch.Generator().manual_seed(5)
    A = torch.randn(batch, M, K, generator=gen).to(dev)
    B = torch.randn(batch, N, K, generator=gen).to(dev)
    ref = torch.einsum('bmk,bnk->mn', A.double(), B.double())
    assert _hip.lib.eap_gemm_skinny_reduce_f32_supported(M, N, K, _hip._ptr(A), _hip._I64(K), _hip._I64(M * K), _hip._ptr(B), _hip._I64(K), _hip._I64(N * K))
    outs = []
    for _ in range(2):
        C = torch.full((M, N + 3), float('nan'), device=dev)                # a row pitch wider than N
        ws = torch.empty(int(_hip.lib.eap_gemm_skinny_reduce_workspace(M, N, K, batch)), device=dev)
        _hip.call('eap_gemm_skinny_reduce_f32', C, M, N, K, _hip._ptr(A), _hip._I64(K), _hip._I64(M * K), _hip._ptr(B), _hip._I64(K), _hip._I64(N * K),
                  _hip._ptr(C), _hip._I64(N + 3), batch, _hip._ptr(ws))
        outs.append(C)
    assert torch.equal(outs[0][:, :N], outs[1][:, :N]) and torch.isnan(outs[0][:, N:]).all()
    scale = float((A.double().abs().unsqueeze(2) * B.double().abs().unsqueeze(1)).sum((0, 3)).max())
    assert float((outs[0][:, :N].double() - ref).abs().max()) < 1e-6 * scale
    launched = []
    _hip.KERNEL_TIMES = launched
    try:
        C2 = torch.empty(M, N, device=dev)
        _hip.gemm_reduce(0, 1, M, N, K, A, K, M * K, B, K, N * K, C2, N, batch)
    finally:
        _hip.KERNEL_TIMES = None
    assert [n for n, *_ in launched] == ['eap_gemm_skinny_reduce_f32'] and torch.equal(C2, outs[0][:, :N].contiguous())
