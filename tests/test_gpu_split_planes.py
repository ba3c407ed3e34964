"""GPU: the contractions with two fp16 planes per operand (csrc/gemm_bf16x3.hip, gemm_f16x2_kernel; include/eap_hip.h
eap_gemm_f16x2_f32) -- three matrix instructions per k-tile instead of the six of the three-bf16-plane split.  The bar
is the arithmetic the reference runs: torch.matmul on fp32 operands (vgtk/vgtk/so3conv/modules.py:L48-55), here the
fp32-MFMA kernel of this library; the two-plane kernel's error against fp64 must not exceed that kernel's."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return torch.device('cuda:0')


class mode:
    """with mode('f16x2' | 'bf16x3' | 'fp32'): which kernel vgtk._hip.gemm & co. take."""

    def __init__(self, name, scan_rows=0):
        self.want = {'f16x2': (2, True), 'bf16x3': (3, True), 'fp32': (3, False)}[name] + (scan_rows,)

    def __enter__(self):
        from vgtk import _hip
        self.was = (_hip.SPLIT_PLANES, _hip.SPLIT_BF16_CONTRACTION, _hip.SPLIT_PLANES_SCAN_ROWS)
        _hip.SPLIT_PLANES, _hip.SPLIT_BF16_CONTRACTION, _hip.SPLIT_PLANES_SCAN_ROWS = self.want

    def __exit__(self, *exc):
        from vgtk import _hip
        _hip.SPLIT_PLANES, _hip.SPLIT_BF16_CONTRACTION, _hip.SPLIT_PLANES_SCAN_ROWS = self.was


def launched(fn):
    """-> (result of fn(), names of the C-ABI entries it launched)"""
    from vgtk import _hip
    rec = []
    _hip.KERNEL_TIMES = rec
    try:
        out = fn()
    finally:
        _hip.KERNEL_TIMES = None
    return out, [n for n, *_ in rec]


def test_default_is_two_planes():
    from vgtk import _hip
    assert _hip.SPLIT_PLANES == 2 and _hip.SPLIT_BF16_CONTRACTION


@pytest.mark.parametrize('shape', [(1, 7, 64, 64, 0), (3, 5, 12, 20, 400), (2, 128, 1920, 1920, 128 * 1920 + 64)])
def test_operand_magnitudes(dev, shape):
    """eap_absmax_rows_f32 / eap_absmax_colgroups_f32 over strided [batch][rows][cols] views: bit patterns of the largest
    magnitudes, nothing outside the view; views that are not whole aligned 16-byte pieces are refused (-> three planes)."""
    from vgtk import _hip
    batch, rows, cols, ld, stride = shape
    stride = stride or rows * ld
    gen = torch.Generator().manual_seed(rows + cols)
    buf = torch.randn(batch * stride + 64, generator=gen).to(dev)
    view = torch.as_strided(buf, (batch, rows, cols), (stride, ld, 1))
    if ld > cols:                                      # something larger just outside every row: must be ignored
        torch.as_strided(buf, (batch, rows, ld - cols), (stride, ld, 1), cols).fill_(1e9)
    view[batch - 1, rows // 2, cols - 1] = -37.5
    got = _hip.absmax_rows(buf, batch, rows, cols, ld, stride).view(torch.float32)
    assert torch.equal(got, view.abs().amax(2))
    for grp in (4, cols) if cols % 8 else (4, 8, cols):
        got = _hip.absmax_colgroups(buf, batch, rows, cols, ld, stride, grp).view(torch.float32)
        assert torch.equal(got, view.abs().amax(1).view(batch, cols // grp, grp).amax(2)), grp
    z = torch.zeros(4, 8, device=dev)
    assert not _hip.absmax_rows(z, 1, 4, 8, 8, 0).any() and not _hip.absmax_colgroups(z, 1, 4, 8, 8, 0, 4).any()
    assert _hip.absmax_rows(buf[1:], 1, 4, 8, 8, 0) is None and _hip.absmax_rows(buf, 1, 4, 6, 8, 0) is None
    assert _hip.absmax_colgroups(buf, 1, 4, 8, 8, 0, 6) is None


def test_grouped_bound_bounds_the_grouped_tensor(dev):
    """eap_so3_grouped_bound_f32: sum over a point's neighbours of the neighbour's largest feature magnitude -- against torch, and
    against the grouped tensor the kernels actually produce (every |X[., ., p, .]| below its point's bound; clouds with anchor
    permutations and shadow neighbours included)."""
    import synth_clouds
    import vgtk.cuda.grouping as G
    from vgtk import _hip
    torch.manual_seed(11)
    B, P, c, na, nn = 2, 256, 64, 60, 32
    xyz = torch.from_numpy(synth_clouds.laptop_batch(2, B, P)[0]).to(dev)
    idx = G.ball_query(xyz, xyz, 0.25, nn)
    idx[0, 5, 7:] = P                                       # shadow neighbours
    feats = torch.randn(B, c, P, na, device=dev) * torch.exp(torch.randn(B, 1, P, 1, device=dev) * 2)
    got = _hip.so3_grouped_bound(feats, idx).view(torch.float32)
    pm = torch.cat([feats.abs().amax((1, 3)), torch.zeros(B, 1, device=dev)], 1)          # [B, P + 1], shadow row = 0
    ref = torch.gather(pm, 1, idx.long().clamp(max=P).view(B, -1)).view(B, P, nn).double().sum(2)
    assert ((got.double() - ref).abs() <= 1e-5 * ref).all()


def _errors(C, ref, bound):
    C = C.double().cpu()
    err = (C - ref).abs()
    s = ref.abs().max().item()
    return err.max().item() / s, err.pow(2).mean().sqrt().item() / s, (err / bound.clamp_min(1e-300)).max().item()


@pytest.mark.parametrize('trans_b', [1, 0])
@pytest.mark.parametrize('M,N,K,batch', [(512, 2048, 3072, 2), (512, 1024, 6144, 1), (128, 1920, 1536, 2), (384, 768, 48, 3), (256, 256, 16, 1)])
def test_two_plane_contraction_against_fp64(dev, trans_b, M, N, K, batch):
    """C = A B with fp32 operands: two fp16 planes (three products), three bf16 planes (six products) and the fp32 matrix
    pipe against fp64, operands like the path's (weights ~ N(0, 0.05), non-negative features whose rows differ by e^+-6).
    The two-plane kernel must be at least as accurate as the fp32 pipe (measured: rms 0.6 of it at K = 3072 -- both are
    dominated by the fp32 accumulation, of which the two-plane kernel does less) and within 1e-6 of sum |a||b| per element."""
    from vgtk import _hip
    gen = torch.Generator().manual_seed(M + N + K + trans_b)
    A = (torch.randn(M, K, generator=gen) * 0.05).to(dev)
    if trans_b:
        B = (torch.randn(batch, N, K, generator=gen).abs() * torch.exp(torch.randn(batch, N, 1, generator=gen) * 2.0)).to(dev)
        Bm = B.double().cpu().transpose(1, 2)
    else:
        B = (torch.randn(batch, K, N, generator=gen).abs() * torch.exp(torch.randn(batch, 1, N, generator=gen) * 2.0)).to(dev)
        Bm = B.double().cpu()
    ref = torch.matmul(A.double().cpu(), Bm)
    bound = torch.matmul(A.abs().double().cpu(), Bm.abs())
    res = {}
    for name in ('f16x2', 'bf16x3', 'fp32'):
        C = torch.full((batch, M, N), float('nan'), device=dev)
        with mode(name):
            _, names = launched(lambda: _hip.gemm(0, trans_b, M, N, K, A, K, 0, B, K if trans_b else N, N * K, C, N, M * N, batch))
        assert names[-1] == {'f16x2': 'eap_gemm_f16x2_f32', 'bf16x3': 'eap_gemm_bf16x3_f32' if trans_b else 'eap_gemm_bf16x3_nn_f32'}.get(name, names[-1]), names
        res[name] = _errors(C, ref, bound)
    print(f'\n{"NT" if trans_b else "NN"} {M}x{N}x{K}: (max, rms, per-element) two planes {res["f16x2"]}, three planes {res["bf16x3"]}, fp32 pipe {res["fp32"]}')
    assert res['f16x2'][0] <= 1.25 * res['fp32'][0] + 1e-7
    assert res['f16x2'][1] <= 1.1 * res['fp32'][1] + 1e-8
    assert res['f16x2'][2] < 1e-6
    assert res['bf16x3'][0] < 2 * res['fp32'][0] + 1e-7 and res['bf16x3'][2] < 1e-6        # (the three-plane kernel's bar, as in round 3)


@pytest.mark.parametrize('magnitude', [1e-30, 1.0, 1e30])
def test_two_planes_at_any_magnitude_and_with_a_loose_bound(dev, magnitude):
    """The operand scales are taken from the operands' rows: the same relative error for operands of size 1e-30 and 1e+30 (the
    product stays inside fp32), with a bound 64 x too large (what the inter conv's per-point bound may lose against the largest
    grouped value) and with one 2^20 too large; a bound too SMALL is the caller's error and not exercised."""
    from vgtk import _hip
    M, N, K = 512, 768, 512
    gen = torch.Generator().manual_seed(5)
    A = (torch.randn(M, K, generator=gen) * 0.05).to(dev)
    B = (torch.randn(1, N, K, generator=gen) * magnitude).to(dev)
    ref = torch.matmul(A.double().cpu(), B.double().cpu().transpose(1, 2))
    bound = torch.matmul(A.abs().double().cpu(), B.abs().double().cpu().transpose(1, 2))
    word = _hip.absmax_rows(B, 1, N, K, K, N * K)
    base = None
    for mult in (None, 1.0, 64.0, 2.0 ** 20):
        C = torch.full((1, M, N), float('nan'), device=dev)
        with mode('f16x2'):
            _hip.gemm(0, 1, M, N, K, A, K, 0, B, K, N * K, C, N, M * N, 1, b_bound=None if mult is None else (word, 1, mult))
        e = _errors(C, ref, bound)
        print(f'\nmagnitude {magnitude:g}, bound x {mult}: {e}')
        assert e[2] < (1e-6 if (mult or 1) <= 64 else 2e-4), (mult, e)      # 2^20 of head room lost: l is a subnormal for most elements
        if mult in (None, 1.0):
            base = base if base is not None else C.clone()
            assert torch.equal(C, base)                                      # the maximum itself, from the caller or from the entry's own pass


def test_two_planes_wide_dynamic_range_and_zeros(dev):
    """Elements far below the largest magnitude of their operand ROW (down to 2^-40 of it) keep an ABSOLUTE error of 2^-40 of that
    magnitude: the error bound is  2^-23 sum |a||b|  +  2^-39 K max|a_row| max|b_row|; all-zero operands give zeros (scale 1), not NaNs."""
    from vgtk import _hip
    M, N, K = 384, 512, 256
    gen = torch.Generator().manual_seed(9)
    A = (torch.randn(M, K, generator=gen) * 0.05).to(dev)
    B = torch.randn(1, N, K, generator=gen) * torch.exp2(-torch.randint(0, 40, (1, N, K), generator=gen).float())
    B[0, 0, 0] = 3.0
    B = B.to(dev)
    ref = torch.matmul(A.double().cpu(), B.double().cpu().transpose(1, 2))
    bound = torch.matmul(A.abs().double().cpu(), B.abs().double().cpu().transpose(1, 2))
    C = torch.empty(1, M, N, device=dev)
    with mode('f16x2'):
        _hip.gemm(0, 1, M, N, K, A, K, 0, B, K, N * K, C, N, M * N, 1)
    err = (C.double().cpu() - ref).abs()
    floor = 2.0 ** -39 * K * A.abs().amax(1).double().cpu()[:, None] * B.abs().amax(2).double().cpu()[0][None, :]
    # representation 2 x 2^-23 + dropped l l' 2^-22 per product, worst case, + the floor; the fp32 accumulation of K terms on top
    # (the fp32-MFMA kernel has it too: measured below)
    C32 = torch.empty(1, M, N, device=dev)
    with mode('fp32'):
        _hip.gemm(0, 1, M, N, K, A, K, 0, B, K, N * K, C32, N, M * N, 1)
    err32 = (C32.double().cpu() - ref).abs()
    print(f'\nwide range: two planes max {err.max().item():.2e} rms {err.pow(2).mean().sqrt().item():.2e}; fp32 pipe max {err32.max().item():.2e} rms {err32.pow(2).mean().sqrt().item():.2e}')
    assert (err <= (2.0 ** -21 + K * 2.0 ** -24) * bound + floor).all()
    assert err.max() <= 1.25 * err32.max() + 1e-9 and err.pow(2).mean().sqrt() <= 1.1 * err32.pow(2).mean().sqrt() + 1e-10
    for zero_a, zero_b in ((True, False), (False, True), (True, True)):
        C.fill_(float('nan'))
        with mode('f16x2'):
            _hip.gemm(0, 1, M, N, K, torch.zeros_like(A) if zero_a else A, K, 0, torch.zeros_like(B) if zero_b else B, K, N * K, C, N, M * N, 1)
        assert torch.equal(C, torch.zeros_like(C))


def test_two_planes_presplit_and_reruns_are_bit_identical(dev):
    """The weights split once per call (scratch) or inside the k-loop: the same roundings -- and two runs of either agree."""
    from vgtk import _hip
    gen = torch.Generator().manual_seed(77)
    W = (torch.randn(512, 1024, generator=gen) * 0.05).to(dev)
    Bt = torch.randn(2, 1920, 1024, generator=gen).to(dev)
    outs = []
    for on in (2, 0, 2):
        was = _hip.lib.eap_gemm_bf16x3_presplit(on)
        try:
            c = torch.empty(2, 512, 1920, device=dev)
            with mode('f16x2'):
                _hip.gemm(0, 1, 512, 1920, 1024, W, 1024, 0, Bt, 1024, 1920 * 1024, c, 1920, 512 * 1920, 2)
            outs.append(c.cpu())
        finally:
            _hip.lib.eap_gemm_bf16x3_presplit(was)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize('B,C,O,P', [(2, 64, 128, 64), (1, 128, 256, 32)])
def test_two_plane_intra_conv_against_fp64(dev, B, C, O, P):
    """eap_so3_intra_conv_f16x2_f32 (implicit gather, so3conv/functional.py:L2553-2602) against the gathered fp64 einsum."""
    from vgtk import _hip
    import vgtk.so3conv.functional as L
    gen = torch.Generator().manual_seed(B + C + O + P)
    idx = torch.from_numpy(np.ascontiguousarray(L.get_intra_idx())).long().to(dev)
    nt = idx.shape[1]
    feats = (torch.randn(B, C, P, 60, generator=gen).abs() * torch.exp(torch.randn(B, 1, P, 1, generator=gen))).to(dev)
    W = (torch.randn(O, C * nt, generator=gen) * 0.05).to(dev)
    ref = torch.einsum('oct,bcpat->bopa', W.double().cpu().view(O, C, nt), feats.double().cpu()[:, :, :, idx.cpu()])
    err = {}
    for name in ('f16x2', 'fp32'):
        with mode(name):
            y, names = launched(lambda: L.intra_so3conv(feats, W, idx))
        if name == 'f16x2':
            assert 'eap_so3_intra_conv_f16x2_f32' in names, names
        err[name] = (y.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f'\nintra conv B{B} C{C} O{O} P{P}: two planes {err["f16x2"]:.2e}, fp32 pipe {err["fp32"]:.2e}')
    assert err['f16x2'] <= 1.25 * err['fp32'] + 1e-7


def test_inter_conv_layer_with_two_planes(dev, monkeypatch):
    """A whole InterSO3PoseConv layer (forward, dF, dW) with the contraction on two planes against three planes and against the
    fp32 pipe; the forward contraction takes its operand bound from the features (nn x max|feats|), not from a pass over X.
    (The list-kernel forward: at this width the dense product would take it since round 6 -- switched off here.)"""
    import synth_clouds
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    import vgtk.so3conv.functional as L
    monkeypatch.setattr(L, 'DENSE_FWD_NARROW', False)
    torch.manual_seed(3)
    B, P, c, o = 2, 512, 64, 128
    xyz, _, pose = synth_clouds.laptop_batch(1, B, P)
    xyz, pose = torch.from_numpy(xyz).to(dev), torch.from_numpy(pose).to(dev)
    conv = sptk.InterSO3PoseConv(c, o, 1, 1, 0.3, 0.05, 64, kanchor=60, permute_modes=1).to(dev)
    feats0 = torch.randn(B, c, P, 60, device=dev)
    gy = torch.randn(B, o, P, 60, device=dev)
    res = {}
    for name in ('f16x2', 'bf16x3', 'fp32'):
        f = feats0.clone().requires_grad_(True)
        with mode(name, scan_rows=384):
            (y, gF, gW), names = launched(lambda: (lambda y: (y.detach(),) + torch.autograd.grad(y, [f, conv.basic_conv.W], gy))(
                conv(zptk.SphericalPointCloudPose(xyz, f, None, pose))[3].feats))
        if name == 'f16x2':
            assert 'eap_gemm_f16x2_f32' in names and 'eap_so3_grouped_bound_f32' in names, names
        res[name] = (y.double().cpu(), gF.double().cpu(), gW.double().cpu())
    for i, what in enumerate(('y', 'dF', 'dW')):
        s = res['fp32'][i].abs().max().item()
        d2 = (res['f16x2'][i] - res['fp32'][i]).abs().max().item() / s
        d3 = (res['bf16x3'][i] - res['fp32'][i]).abs().max().item() / s
        print(f'\n{what}: two planes vs fp32 pipe {d2:.2e}, three planes vs fp32 pipe {d3:.2e}')
        assert d2 < 5e-6, (what, d2)
