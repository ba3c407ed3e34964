"""GPU: the inter conv re-associated over its referenced rows as a dense product (csrc/so3_dense.hip, round 5).

Oracle here = float64 torch expressions of the reference's formulas on the same inputs (the weights
relu(1 - |x_n - x_p - A_a kappa_k|^2 / sigma) of so3conv/functional.py:L2508-2549, the einsum of L1261 and its autograd
transpose) and, for whole layers, the list kernels the golden fixtures pin (DENSE_MODE 'off'); the slab comparisons against
oracle/so3_ref.py at the bench shapes run the dense path too (tests/test_gpu_bench_shapes.py, DENSE_MODE 'auto').

Bars: stored-operand split 2^-21 of a row's maximum; generated weights 5e-7 absolute (the bar of the list kernels' weights: 2e-6);
Z and y 1e-5 of their scale against float64; gradients as tests/test_gpu_bench_shapes.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NN, NA, KS = 64, 60, 24


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return torch.device('cuda:0')


def _setup(dev, B, P, layer=2, plan=4096, seed=11, radius=None):
    import synth_clouds
    import vgtk.so3conv as sptk
    import vgtk.so3conv.functional as L
    import vgtk.cuda.grouping as cuda_nn
    from vgtk import _hip
    c, o, r, sigma = synth_clouds.backbone_layers(plan)[layer]
    radius = r if radius is None else radius
    xyz_np, _, _ = synth_clouds.laptop_batch(seed, B, P)
    xyz = torch.from_numpy(xyz_np).to(dev).contiguous()
    anchors = torch.from_numpy(np.asarray(L.get_anchors(NA), dtype=np.float32)).to(dev)
    kernels = torch.from_numpy(L.get_sphereical_kernel_points_from_ply(0.7 * radius, 1)).to(dev)
    rk = L.rotated_kernels(anchors, kernels)
    idx = cuda_nn.ball_query(xyz, xyz, radius, NN)
    return dict(xyz=xyz, idx=idx, rk=rk, sigma=float(sigma), c=c, o=o, radius=radius, anchors=anchors, kernels=kernels)


def _geometry(s, dev):
    import vgtk.so3conv.functional as L
    from vgtk import _hip
    xyz, idx = s['xyz'], s['idx']
    n = xyz.shape[2]
    head = L._ListHead(idx, n, None, None, dense_probe=(None, None))
    rcap, _ = head.decide()
    assert head.dense_possible()
    rp = L._dense_rows(rcap, n)
    head.wait()
    geo = _hip.DenseGeometry(xyz, xyz, head.memb, head.rows, rp, s['rk'], s['sigma'], NN, head.n_rows)
    return head, geo, rp


def _dense_weights64(s, rows, rp):
    """Wd[b,p,k,r,a] in float64 from the reference's formula + the membership of row r in p's list."""
    xyz, idx, rk = s['xyz'].double(), s['idx'].long(), s['rk'].double()
    B, _, P = xyz.shape
    out = []
    for b in range(B):
        rw = rows[b, :rp].long()
        ok = rw >= 0
        xr = xyz[b][:, rw.clamp(min=0)]                                   # [3, rp]
        g = xr[:, None, :] - xyz[b][:, :, None]                           # [3, P, rp]  x_r - x_p
        d = g.permute(1, 2, 0)[:, :, None, None, :] - rk[None, None]      # [P, rp, A, K, 3]
        w = torch.relu(1.0 - (d * d).sum(-1) / s['sigma'])                # [P, rp, A, K]
        member = (idx[b][:, :, None] == rw[None, None, :]).any(1) & ok[None, :]      # [P, rp]
        out.append(w * member[:, :, None, None].double())
    return torch.stack(out)                                               # [B, P, rp, A, K]


def _member(s, rows, rp):
    idx = s['idx'].long()
    return torch.stack([((idx[b][:, :, None] == rows[b, :rp].long()[None, None, :]).any(1) & (rows[b, :rp] >= 0)[None, :]).double()
                        for b in range(idx.shape[0])])                    # [B, P, rp]


def test_split_planes_reconstruct_the_operand(dev):
    """dense_split_kernel: (h + l) / scale == x to 2^-21 of the row maximum (two fp16 planes after the row's power-of-two
    scale; an element 2^-17 below the maximum keeps the relative bound), planes in the fragment order the product reads."""
    from vgtk import _hip
    gen = torch.Generator(device=dev).manual_seed(3)
    b, m, l, na = 2, 64, 80, 60
    x = torch.randn(b, m, l, na, device=dev, generator=gen) * torch.exp(3 * torch.randn(b, m, 1, na, device=dev, generator=gen))
    x[0, 3] = 0.0
    scale, planes = _hip.so3_dense_split(x)
    assert torch.equal(scale[1].view(b, m, na), scale[0].permute(0, 2, 1))
    scale = scale[0]
    lp = (l + 31) // 32 * 32
    pl = planes.view(torch.float16).view(b, na, lp // 16, m // 32, 2, 64, 8).float()       # [b,a,kb,mt,plane,lane,e]
    v = pl[:, :, :, :, 0] + pl[:, :, :, :, 1]                                              # [b,a,kb,mt,lane,e]
    v = v.view(b, na, lp // 16, m // 32, 2, 32, 8)                                         # lane = 32 kg + i
    v = v.permute(0, 1, 3, 5, 2, 4, 6).reshape(b, na, m, lp)                               # [b,a,row,k] with k = 16 kb + 8 kg + e
    rec = (v / scale[:, :, :, None])[:, :, :, :l].permute(0, 2, 3, 1)
    rowmax = x.abs().amax(dim=2, keepdim=True)
    assert float(((rec - x).abs() / rowmax.clamp(min=1e-30)).max()) < 2.0 ** -21
    assert float(v[:, :, :, l:].abs().max()) == 0.0                                         # the padding past l
    s = scale[rowmax[:, :, 0].permute(0, 2, 1) > 0]
    top = (rowmax[:, :, 0].permute(0, 2, 1)[rowmax[:, :, 0].permute(0, 2, 1) > 0] * s)
    assert float(top.min()) >= 2.0 ** 14 and float(top.max()) < 2.0 ** 15
    assert float(scale[0, :, 3].min()) == 1.0 and float(scale[0, :, 3].max()) == 1.0       # an all-zero row


@pytest.mark.parametrize('P,layer', [(512, 2), (1024, 1), (288, 2)])
def test_generated_weights_and_backward_product(dev, P, layer):
    """Z[b,o,k,a,r] = sum_p dY[b,o,p,a] Wd[p,(k,r),a]: with dY = one-hot rows the product IS the generated operand -> the bar on
    the weights; with random dY against the float64 sum."""
    from vgtk import _hip
    s = _setup(dev, 2, P, layer=layer)
    head, geo, rp = _geometry(s, dev)
    B, o = 2, (256 if layer == 2 else 128)                                     # 256-row and 128-row blocks of the product kernel
    wd = _dense_weights64(s, head.rows, rp)                                   # [B,P,rp,A,K]
    # one-hot: dY[b,o,p,a] = 1 iff p == p0 + o  ->  Z[b,o,k,a,r] = Wd[p0 + o,(k,r),a]
    for p0 in (0, P - o):
        gy = torch.zeros(B, o, P, NA, device=dev)
        gy[:, torch.arange(o), p0 + torch.arange(o), :] = 1.0
        z = _hip.so3_dense_bwd(gy, geo).view(B, o, KS, NA, rp)               # [B,o,K,A,rp]
        ref = wd[:, p0:p0 + o].permute(0, 1, 4, 3, 2)                         # [B,o,K,A,rp]
        assert z.shape == ref.shape
        err = float((z.double() - ref).abs().max())
        assert err < 5e-7, (p0, err)                                           # (bar of the list kernels' weights: 2e-6)
        assert float(ref.max()) > 0.5                                          # not vacuous
    gen = torch.Generator(device=dev).manual_seed(5)
    gy = torch.randn(B, o, P, NA, device=dev, generator=gen) * torch.exp(2 * torch.randn(B, o, 1, NA, device=dev, generator=gen))
    z = _hip.so3_dense_bwd(gy, geo).view(B, o, KS, NA, rp)
    ref = torch.einsum('bopa,bprak->bokar', gy.double(), wd)
    # bound: the products (1e-6 of sum |dY| w) + the weights' own evaluation error (5e-7 absolute, 4 x below the bar above) on
    # every list member
    mag = torch.einsum('bopa,bprak->bokar', gy.double().abs(), wd)
    magm = torch.einsum('bopa,bpr->boar', gy.double().abs(), _member(s, head.rows, rp))[:, :, None]
    assert float(((z.double() - ref).abs() / (1e-6 * mag + 5e-7 * magm).clamp(min=1e-30)).max()) < 1.0
    z2 = _hip.so3_dense_bwd(gy, geo).view(B, o, KS, NA, rp)
    assert torch.equal(z, z2)                                                  # bit-reproducible


def test_backward_product_matches_the_list_kernel(dev):
    """Same Z as eap_so3_inter_group_inv_f32 (rows in the same order, anchor and row axes exchanged)."""
    import vgtk.so3conv.functional as L
    from vgtk import _hip
    s = _setup(dev, 2, 512)
    head, geo, rp = _geometry(s, dev)
    xyz, idx = s['xyz'], s['idx']
    gx, _ = _hip.so3_prep(xyz, xyz, idx, None, None, s['anchors'].contiguous(), 0)
    ent_p, ent_gx = _hip.inv_lists_fill(idx, gx, head.rows, head.off, rp)
    gen = torch.Generator(device=dev).manual_seed(7)
    gy = torch.randn(2, 256, 512, NA, device=dev, generator=gen)
    z_list = _hip.so3_inter_group_inv(gy, head.rows[:, :rp].contiguous(), head.off[:, :rp].contiguous(), head.cnt[:, :rp].contiguous(),
                                      ent_p, ent_gx, s['rk'], None, s['sigma'], NN)                  # [b,o,ks,rp,na]
    z = _hip.so3_dense_bwd(gy, geo).view(2, 256, KS, NA, rp)
    scale = float(z_list.abs().max())
    assert float((z.transpose(3, 4) - z_list).abs().max()) < 1e-5 * scale
    # rows padded to a pitch: the same numbers, the padding untouched
    ld = NA * rp + 40
    zp = torch.full((2, 256, KS, ld), 7.0, device=dev)
    from vgtk._hip import _ptr, _I64, _F32
    sc, pl = _hip.so3_dense_split(gy, colmap=geo.columns())          # (the geometry's point order: occupancy-sorted)
    _hip.call('eap_so3_dense_product_f32', gy, 0, 2, 256, 512, NA, KS, rp, _I64(ld), _F32(geo.sigma), _ptr(geo.n_rows), _ptr(pl), _ptr(sc), _ptr(geo.pt), _ptr(geo.kr),
              _ptr(geo.mask(0)), _ptr(zp))
    assert torch.equal(zp[..., :NA * rp].reshape(2, 256, KS, NA, rp), z) and float((zp[..., NA * rp:] - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize('P,layer', [(512, 2), (1024, 1), (288, 2)])
def test_forward_product(dev, P, layer):
    """Yt[b,a,o,p] = sum_(k,r) G[b,o,(k,r),a] Wd[p,(k,r),a] against the float64 sum, and back in the reference layout."""
    from vgtk import _hip
    s = _setup(dev, 2, P, layer=layer)
    head, geo, rp = _geometry(s, dev)
    B, o = 2, (256 if layer == 2 else 128)
    wd = _dense_weights64(s, head.rows, rp)                                   # [B,P,rp,A,K]
    gen = torch.Generator(device=dev).manual_seed(9)
    g = torch.randn(B, o, KS, rp, NA, device=dev, generator=gen) * torch.exp(2 * torch.randn(B, o, 1, 1, NA, device=dev, generator=gen))
    y = _hip.so3_dense_fwd(g.view(B, o, KS, rp * NA), geo, P)
    assert y.shape == (B, o, P, NA)
    # the same operand with padded rows [o, k, pitch] (what the small GEMM in front leaves): bit-equal
    ld = rp * NA + 56
    gp = torch.full((B, o, KS, ld), float('nan'), device=dev)
    gp[..., :rp * NA] = g.view(B, o, KS, rp * NA)
    assert torch.equal(_hip.so3_dense_fwd(gp, geo, P, ldg=ld), y)
    ref = torch.einsum('bokra,bprak->bopa', g.double(), wd)
    mag = torch.einsum('bokra,bprak->bopa', g.double().abs(), wd)
    magm = torch.einsum('bokra,bpr->bopa', g.double().abs(), _member(s, head.rows, rp))
    assert float(((y.double() - ref).abs() / (1e-6 * mag + 5e-7 * magm).clamp(min=1e-30)).max()) < 1.0
    assert torch.equal(y, _hip.so3_dense_fwd(g.view(B, o, KS, rp * NA), geo, P))


@pytest.mark.parametrize('pq', [32, 64, 96, 224, 320])
def test_products_over_few_query_points(dev, pq):
    """Query points = the first pq points of a 512-point cloud (a rigid part's launch, a strided layer): fewer columns than one
    256-column block of the forward product / fewer than eight k-steps of the backward's, and FEWER QUERY POINTS THAN REFERENCED
    ROWS (the forward once cut its row range at the point count), against the float64 sums."""
    from vgtk import _hip
    s = _setup(dev, 2, 512, layer=2)
    head, _, rp = _geometry(s, dev)
    xyz = s['xyz']
    geo = _hip.DenseGeometry(xyz[:, :, :pq].contiguous(), xyz, head.memb[:, :pq].contiguous(), head.rows, rp, s['rk'], s['sigma'], NN, head.n_rows)
    B, o = 2, 256
    wd = _dense_weights64(s, head.rows, rp)[:, :pq]                           # [B,pq,rp,A,K]
    memb = _member(s, head.rows, rp)[:, :pq]
    gen = torch.Generator(device=dev).manual_seed(29)
    g = torch.randn(B, o, KS, rp, NA, device=dev, generator=gen)
    y = _hip.so3_dense_fwd(g.view(B, o, KS, rp * NA), geo, pq)
    ref = torch.einsum('bokra,bprak->bopa', g.double(), wd)
    mag = torch.einsum('bokra,bprak->bopa', g.double().abs(), wd)
    magm = torch.einsum('bokra,bpr->bopa', g.double().abs(), memb)
    assert float(((y.double() - ref).abs() / (1e-6 * mag + 5e-7 * magm).clamp(min=1e-30)).max()) < 1.0
    gy = torch.randn(B, o, pq, NA, device=dev, generator=gen)
    z = _hip.so3_dense_bwd(gy, geo).view(B, o, KS, NA, rp)
    ref = torch.einsum('bopa,bprak->bokar', gy.double(), wd)
    mag = torch.einsum('bopa,bprak->bokar', gy.double().abs(), wd)
    magm = torch.einsum('bopa,bpr->boar', gy.double().abs(), memb)[:, :, None]
    assert float(((z.double() - ref).abs() / (1e-6 * mag + 5e-7 * magm).clamp(min=1e-30)).max()) < 1.0


def _layer_run(dev, monkeypatch, mode, xyz, pose, feats0, W0, c, o, radius, sigma):
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    import vgtk.so3conv.functional as L
    monkeypatch.setattr(L, 'DENSE_MODE', mode)
    torch.manual_seed(2913)
    conv = sptk.InterSO3PoseConv(c, o, 1, 1, radius, sigma, NN, kanchor=NA, permute_modes=1).to(dev)
    with torch.no_grad():
        conv.basic_conv.W.copy_(W0)
    feats = feats0.clone().requires_grad_(True)
    L.BACKWARD_LOG = []
    y = conv(zptk.SphericalPointCloudPose(xyz, feats, None, pose))[3].feats
    gen = torch.Generator(device=dev).manual_seed(21)
    gy = torch.randn(y.shape, device=dev, generator=gen)
    gF, gW = torch.autograd.grad(y, [feats, conv.basic_conv.W], gy)
    log, L.BACKWARD_LOG = L.BACKWARD_LOG, None
    with torch.no_grad():
        y_ng = conv(zptk.SphericalPointCloudPose(xyz, feats0, None, pose))[3].feats
    return y.detach(), gF, gW, log, y_ng


@pytest.mark.parametrize('with_pose', [False, True])
def test_whole_layer_dense_against_lists(dev, monkeypatch, with_pose):
    """InterSO3PoseConv forward + backward at O = 256 with the dense product forced / switched off: y, dF, dW agree to the
    bars of the bench-shape tests; the no-grad forward takes the dense product too; identity POSES (what the model feeds) count
    as no rotation."""
    import synth_clouds
    B, P, c, o = 2, 512, 32, 256
    _, _, radius, sigma = synth_clouds.backbone_layers(4096)[2]
    xyz_np, _, _ = synth_clouds.laptop_batch(31, B, P)
    xyz = torch.from_numpy(xyz_np).to(dev)
    pose = torch.eye(4, device=dev).repeat(B, P, 1, 1) if with_pose else None
    gen = torch.Generator(device=dev).manual_seed(13)
    feats0 = torch.randn(B, c, P, NA, device=dev, generator=gen)
    W0 = torch.randn(o, c * KS, device=dev, generator=gen) * 0.05
    y0, gF0, gW0, log0, _ = _layer_run(dev, monkeypatch, 'off', xyz, pose, feats0, W0, c, o, radius, sigma)
    y1, gF1, gW1, log1, y1n = _layer_run(dev, monkeypatch, 'force', xyz, pose, feats0, W0, c, o, radius, sigma)
    assert [r['regime'] for r in log1] == ['dense rows'] and log0[0]['regime'] != 'dense rows'
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    assert rel(y1, y0) < 2e-5 and rel(gF1, gF0) < 2e-5 and rel(gW1, gW0) < 5e-5
    assert torch.equal(y1n, y1)


def test_clouds_the_dense_product_cannot_take_fall_back(dev, monkeypatch):
    """Padded (short) lists name a row twice; a pose rotation that is not the identity rotates the offsets: both must stay on
    the list kernels even with the dense product forced, with unchanged results."""
    import synth_clouds
    import vgtk.so3conv.functional as L
    B, P, c, o = 2, 512, 16, 256
    xyz_np, _, _ = synth_clouds.laptop_batch(41, B, P)
    xyz = torch.from_numpy(xyz_np).to(dev)
    gen = torch.Generator(device=dev).manual_seed(15)
    feats0 = torch.randn(B, c, P, NA, device=dev, generator=gen)
    W0 = torch.randn(o, c * KS, device=dev, generator=gen) * 0.05
    # (a) a small ball: most lists are short
    head = L._ListHead(__import__('vgtk.cuda.grouping', fromlist=['x']).ball_query(xyz, xyz, 0.05, NN), P, None, None, dense_probe=(None, None))
    assert not head.dense_possible()
    r0 = _layer_run(dev, monkeypatch, 'off', xyz, None, feats0, W0, c, o, 0.05, 0.002)
    r1 = _layer_run(dev, monkeypatch, 'force', xyz, None, feats0, W0, c, o, 0.05, 0.002)
    assert r1[3][0]['regime'] != 'dense rows' and all(torch.equal(a, b) for a, b in zip(r0[:3], r1[:3]))
    # (b) more distinct pose rotations in a cloud than the per-part launches take (DENSE_MAX_PARTS)
    _, _, radius, sigma = synth_clouds.backbone_layers(4096)[2]
    pose = torch.eye(4, device=dev).repeat(B, P, 1, 1)
    for j in range(L.DENSE_MAX_PARTS):
        a = 0.3 + 0.1 * j
        pose[1, 17 + 5 * j, :3, :3] = torch.tensor([[np.cos(a), -np.sin(a), 0.], [np.sin(a), np.cos(a), 0.], [0., 0., 1.]], device=dev)
    r0 = _layer_run(dev, monkeypatch, 'off', xyz, pose, feats0, W0, c, o, radius, sigma)
    r1 = _layer_run(dev, monkeypatch, 'force', xyz, pose, feats0, W0, c, o, radius, sigma)
    assert r1[3][0]['regime'] != 'dense rows' and all(torch.equal(a_, b_) for a_, b_ in zip(r0[:3], r1[:3]))


def _quat_rot(rng, n):
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                     2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(n, 3, 3).astype(np.float32)


@pytest.mark.parametrize('o,mode,narrow', [(256, 'force', True), (128, 'auto', False), (128, 'auto', True)])
def test_one_rotation_per_rigid_part_runs_the_dense_product_per_part(dev, monkeypatch, o, mode, narrow):
    """Articulated input: every point carries the rotation of its rigid part (2 parts in one cloud, 3 of very different sizes in
    another, 1 in the third; arbitrary rotations, so the relative rotation across a joint rotates the offsets AND selects a real
    anchor permutation: so3conv/functional.py:L1112-1160).  The dense product runs once per part slot (vgtk/so3conv/functional.py
    _PartsDense); y, dF, dW against the permuted list kernels (pinned by inter_pose_artmode.npz / inter_pose_perm.npz) to the bars
    of the identity-pose comparison.  O = 128 with DENSE_FWD_NARROW off (round 5's decision): list-kernel forward (bit-equal),
    per-part dense backward; on (the default since the empty k-steps are skipped): the forward runs the product too."""
    import synth_clouds
    import vgtk.so3conv.functional as L
    monkeypatch.setattr(L, 'DENSE_FWD_NARROW', narrow)
    B, P, c = 3, 512, 32
    _, _, radius, sigma = synth_clouds.backbone_layers(4096)[2]
    xyz_np, lab_np, _ = synth_clouds.laptop_batch(81, B, P)
    xyz = torch.from_numpy(xyz_np).to(dev)
    rng = np.random.default_rng(5)
    R = _quat_rot(rng, 6)
    part = np.zeros((B, P), np.int64)
    part[0] = lab_np[0] % 2                                        # the cloud's own two rigid parts
    part[1] = lab_np[1] % 2
    part[1, 100:117] = 2                                           # a third, tiny part
    pose_np = np.tile(np.eye(4, dtype=np.float32), (B, P, 1, 1))
    pose_np[0, :, :3, :3] = R[0:2][part[0]]
    pose_np[1, :, :3, :3] = R[2:5][part[1]]
    pose_np[2, :, :3, :3] = R[5]                                   # one rotation for the whole cloud
    pose = torch.from_numpy(pose_np).to(dev)
    parts = L._pose_parts(pose)
    assert parts is not None and parts.n == 3 and sorted(parts.sizes[1]) == sorted(np.bincount(part[1]).tolist()) and parts.sizes[2][1:] == [0, 0]
    gen = torch.Generator(device=dev).manual_seed(23)
    feats0 = torch.randn(B, c, P, NA, device=dev, generator=gen)
    W0 = torch.randn(o, c * KS, device=dev, generator=gen) * 0.05
    y0, gF0, gW0, log0, _ = _layer_run(dev, monkeypatch, 'off', xyz, pose, feats0, W0, c, o, radius, sigma)
    y1, gF1, gW1, log1, y1n = _layer_run(dev, monkeypatch, mode, xyz, pose, feats0, W0, c, o, radius, sigma)
    assert [r['regime'] for r in log1] == ['dense rows'] and log1[0].get('parts') == 3 and log0[0]['regime'] != 'dense rows'
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    if o == 128 and not narrow:
        assert torch.equal(y1, y0)
    else:
        assert rel(y1, y0) < 2e-5 and torch.equal(y1n, y1)
    assert rel(gF1, gF0) < 2e-5 and rel(gW1, gW0) < 5e-5
    # the same rotation everywhere in every cloud: relative rotations are the identity -- the plain product, no part launches
    pose_one = torch.from_numpy(np.tile(np.eye(4, dtype=np.float32), (B, P, 1, 1))).to(dev)
    pose_one[:, :, :3, :3] = torch.from_numpy(R[:3]).to(dev)[:, None]
    y2, gF2, gW2, log2, _ = _layer_run(dev, monkeypatch, mode, xyz, pose_one, feats0, W0, c, o, radius, sigma)
    y3, gF3, gW3, _, _ = _layer_run(dev, monkeypatch, 'off', xyz, pose_one, feats0, W0, c, o, radius, sigma)
    assert log2[0]['regime'] == 'dense rows' and 'parts' not in log2[0]
    assert rel(y2, y3) < 2e-5 and rel(gF2, gF3) < 2e-5 and rel(gW2, gW3) < 5e-5


@pytest.mark.parametrize('narrow', [False, True])
def test_narrow_layer_takes_the_dense_backward_only(dev, monkeypatch, narrow):
    """O = 128 in 'auto' mode.  DENSE_FWD_NARROW off (round 5's decision): the forward stays on grouping + contraction (bit-equal to
    DENSE_MODE 'off'), the backward takes the dense product on 128-row blocks; on (the default of round 6): both directions take it."""
    import synth_clouds
    import vgtk.so3conv.functional as L
    monkeypatch.setattr(L, 'DENSE_FWD_NARROW', narrow)
    B, P, c, o = 2, 512, 32, 128
    _, _, radius, sigma = synth_clouds.backbone_layers(4096)[2]
    xyz = torch.from_numpy(synth_clouds.laptop_batch(51, B, P)[0]).to(dev)
    gen = torch.Generator(device=dev).manual_seed(17)
    feats0 = torch.randn(B, c, P, NA, device=dev, generator=gen)
    W0 = torch.randn(o, c * KS, device=dev, generator=gen) * 0.05
    y0, gF0, gW0, log0, _ = _layer_run(dev, monkeypatch, 'off', xyz, None, feats0, W0, c, o, radius, sigma)
    y1, gF1, gW1, log1, y1n = _layer_run(dev, monkeypatch, 'auto', xyz, None, feats0, W0, c, o, radius, sigma)
    assert [r['regime'] for r in log1] == ['dense rows'] and log0[0]['regime'] == 'inverse lists'
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    if narrow:
        assert rel(y1, y0) < 2e-5 and torch.equal(y1n, y1)
    else:
        assert torch.equal(y1, y0) and torch.equal(y1n, y0)
    assert rel(gF1, gF0) < 2e-5 and rel(gW1, gW0) < 5e-5


def test_odd_batch_and_strided_centres(dev, monkeypatch):
    """(a) 3 clouds = 180 (cloud, anchor) products: not a multiple of the 8 XCDs -- the tail of the workgroup map; (b) a strided
    layer (so3conv/functional.py:L931-1013: half as many centres as support points, lazily sampled): the dense product with
    query points != support points.  Both against the list kernels."""
    import synth_clouds
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    import vgtk.so3conv.functional as L
    _, _, radius, sigma = synth_clouds.backbone_layers(4096)[2]
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    # (a)
    B, P, c, o = 3, 512, 16, 256
    xyz = torch.from_numpy(synth_clouds.laptop_batch(61, B, P)[0]).to(dev)
    gen = torch.Generator(device=dev).manual_seed(19)
    feats0 = torch.randn(B, c, P, NA, device=dev, generator=gen)
    W0 = torch.randn(o, c * KS, device=dev, generator=gen) * 0.05
    r0 = _layer_run(dev, monkeypatch, 'off', xyz, None, feats0, W0, c, o, radius, sigma)
    r1 = _layer_run(dev, monkeypatch, 'force', xyz, None, feats0, W0, c, o, radius, sigma)
    assert r1[3][0]['regime'] == 'dense rows'
    assert rel(r1[0], r0[0]) < 2e-5 and rel(r1[1], r0[1]) < 2e-5 and rel(r1[2], r0[2]) < 5e-5
    # (b)
    B, P = 2, 1024
    xyz = torch.from_numpy(synth_clouds.laptop_batch(71, B, P)[0]).to(dev)
    pose = torch.eye(4, device=dev).repeat(B, P, 1, 1)
    feats0 = torch.randn(B, c, P, NA, device=dev, generator=gen)
    outs = {}
    for mode in ('off', 'force'):
        monkeypatch.setattr(L, 'DENSE_MODE', mode)
        torch.manual_seed(2913)
        conv = sptk.InterSO3PoseConv(c, o, 1, 2, radius, sigma, NN, kanchor=NA, permute_modes=1, lazy_sample=True).to(dev)
        with torch.no_grad():
            conv.basic_conv.W.copy_(W0)
        feats = feats0.clone().requires_grad_(True)
        L.BACKWARD_LOG = []
        y = conv(zptk.SphericalPointCloudPose(xyz, feats, None, pose))[3].feats
        assert y.shape == (B, o, P // 2, NA)
        gy = torch.randn(y.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(23))
        gF, gW = torch.autograd.grad(y, [feats, conv.basic_conv.W], gy)
        log, L.BACKWARD_LOG = L.BACKWARD_LOG, None
        outs[mode] = (y.detach(), gF, gW, log[0]['regime'])
    assert outs['force'][3] == 'dense rows' and outs['off'][3] != 'dense rows'
    for i, bar in enumerate((2e-5, 2e-5, 5e-5)):
        assert rel(outs['force'][i], outs['off'][i]) < bar, i


def test_row_maxima_travel_from_the_batchnorm_backward_to_the_split(dev, monkeypatch):
    """conv -> BatchNormLeakyReLU -> loss: the BatchNorm backward writes the conv's output gradient AND its largest magnitude per
    (cloud, channel, anchor) (eap_bn_act_bwd_apply_rowmax_f32); the dense backward's split takes them instead of a pass of its
    own; the forward's re-ordering pass leaves the channel moments the BatchNorm starts with.  The maxima equal torch's."""
    import synth_clouds
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    import vgtk.so3conv.functional as L
    from vgtk import _hip
    from vgtk.so3conv.blocks import BatchNormLeakyReLU
    B, P, c, o = 2, 512, 16, 256
    _, _, radius, sigma = synth_clouds.backbone_layers(4096)[2]
    xyz = torch.from_numpy(synth_clouds.laptop_batch(95, B, P)[0]).to(dev)
    gen = torch.Generator(device=dev).manual_seed(29)
    feats0 = torch.randn(B, c, P, NA, device=dev, generator=gen)
    probe = torch.randn(B, o, P, NA, device=dev, generator=gen)
    monkeypatch.setattr(L, 'DENSE_MODE', 'force')
    res = []
    for use in (True, False):
        monkeypatch.setattr(_hip, 'USE_ROWMAX_HINT', use)
        torch.manual_seed(2913)
        conv = sptk.InterSO3PoseConv(c, o, 1, 1, radius, sigma, NN, kanchor=NA, permute_modes=1).to(dev)
        norm = BatchNormLeakyReLU(o).to(dev)
        feats = feats0.clone().requires_grad_(True)
        taken, staken = _hip.ROWMAX_HINTS_TAKEN, _hip.STATS_HINTS_TAKEN
        y = norm(conv(zptk.SphericalPointCloudPose(xyz, feats, None, None))[3].feats)
        (y * probe).sum().backward()
        assert _hip.ROWMAX_HINTS_TAKEN == taken + (1 if use else 0) and _hip.STATS_HINTS_TAKEN == staken + (1 if use else 0)
        res.append((feats.grad.clone(), conv.basic_conv.W.grad.clone(), y.detach().clone()))
    # (the moments that come with the re-ordering pass are summed in another order than the statistics pass's: equal to rounding)
    for i, bar in enumerate((2e-5, 2e-5, 2e-6)):
        assert float((res[0][i] - res[1][i]).abs().max() / res[1][i].abs().max()) < bar
    # the kernel's maxima against torch's, and its gradient against the plain apply entry
    x = torch.randn(B, 128, 64, NA, device=dev, generator=gen)
    gy = torch.randn(B, 128, 64, NA, device=dev, generator=gen)
    scale, shift, mean, invstd, k2, k3 = (torch.rand(128, device=dev, generator=gen) + 0.5 for _ in range(6))
    gx = _hip.bn_act_bwd_apply(gy, x, B, 128, 64 * NA, scale, shift, mean, invstd, k2, k3, 0.01)
    hint = _hip._ROWMAX_HINT[0]
    assert hint is not None and hint[0]() is gx
    assert torch.equal(hint[3].view(torch.float32), gx.abs().amax(dim=2))
    plain = torch.empty_like(x)
    _hip.call('eap_bn_act_bwd_apply_f32', x, B, 128, _hip._I64(64 * NA), _hip._F32(0.01), _hip._ptr(gy), _hip._ptr(x), _hip._ptr(scale), _hip._ptr(shift),
              _hip._ptr(mean), _hip._ptr(invstd), _hip._ptr(k2), _hip._ptr(k3), _hip._ptr(plain))
    assert torch.equal(plain, gx)


# ------------------------------------------------------------------------------------------------------------------------
# occupancy-sorted query points + skipped k-steps (round 6)
# ------------------------------------------------------------------------------------------------------------------------
def _group_membership(head, rp, P):
    """[B, P, rp / 16] bool on the host: point p's list names a row of 16-row group g (from the membership words)."""
    memb = head.memb.cpu().numpy().astype(np.uint32)                           # [B,P,16] words of 32 row slots
    bits = ((memb[:, :, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(memb.shape[0], P, 512)[:, :, :rp]
    return bits.reshape(bits.shape[0], P, rp // 16, 16).any(-1)


@pytest.mark.parametrize('P,layer', [(1024, 2), (2048, 1)])
def test_step_lists_name_exactly_the_nonempty_k_steps(dev, P, layer):
    """eap_so3_dense_point_keys / eap_so3_dense_steps against a host computation from the membership bits: the sort keys, the point
    order (stable sort by key), and for both directions the k-steps of every 256-column block in which some generated weight is not
    masked out -- and that a worthwhile share of the k-steps IS empty on these clouds (the point of the exercise)."""
    from vgtk import _hip
    s = _setup(dev, 2, P, layer=layer)
    head, geo, rp = _geometry(s, dev)
    assert geo.order is not None
    grp = _group_membership(head, rp, P)                                       # [B,P,G]
    B, G = grp.shape[0], grp.shape[2]
    keys = (grp.astype(np.int64) << np.arange(G, dtype=np.int64)).sum(-1)
    keys = np.where(keys >= 2 ** 31, keys - 2 ** 32, keys)                     # (int32 order)
    order = np.stack([np.argsort(keys[b], kind='stable') for b in range(B)])
    assert np.array_equal(geo.order.cpu().numpy(), order)
    assert int(geo.pivot_pos.item()) == int(np.nonzero(order[0] == 0)[0][0])
    n_rows = head.n_rows.cpu().numpy()
    sg = np.stack([grp[b][order[b]] for b in range(B)])                        # sorted points
    total = kept = 0
    # backward: k-steps of 32 points, column blocks of 256 dense indices (16 row slots x 24 kernel points per group)
    st = geo.steps(0).cpu().numpy()
    for b in range(B):
        used = ((min(n_rows[b], rp) + 15) // 16 * 16) * KS
        for bn in range(st.shape[1]):
            d0, d1 = 256 * bn, min(256 * bn + 256, KS * rp) - 1
            gs = np.arange(d0 // (16 * KS), d1 // (16 * KS) + 1)
            want = np.nonzero(sg[b].reshape(P // 32, 32, G)[:, :, gs].any((1, 2)))[0]
            want = want if len(want) else np.array([0])
            n = st[b, bn, 0]
            assert n == len(want) and np.array_equal(st[b, bn, 1:1 + n], want), (b, bn)
            if d0 < used:
                total += P // 32; kept += n
    frac_b = kept / total
    # forward: k-steps of 32 dense indices (two kernel points of one 16-row group), column blocks of 256 points
    st = geo.steps(1).cpu().numpy()
    total = kept = 0
    for b in range(B):
        for bn in range(st.shape[1]):
            occ = sg[b][256 * bn:256 * bn + 256].any(0)                         # [G]
            want = np.nonzero(np.repeat(occ, KS // 2))[0]
            want = want if len(want) else np.array([0])
            n = st[b, bn, 0]
            assert n == len(want) and np.array_equal(st[b, bn, 1:1 + n], want), (b, bn)
            total += ((min(n_rows[b], rp) + 15) // 16) * (KS // 2); kept += n
    frac_f = kept / total
    assert frac_b < 0.9 and frac_f < 0.95, (frac_b, frac_f)


@pytest.mark.parametrize('o', [256, 128])
def test_skipped_k_steps_contribute_exact_zeros(dev, o, monkeypatch):
    """Both products with the k-step lists and with every k-step (same point order): bit-equal -- a skipped k-step multiplies the stored
    operand by weights that are all exactly 0."""
    from vgtk import _hip
    s = _setup(dev, 2, 1024, layer=2 if o == 256 else 1)
    head, geo, rp = _geometry(s, dev)
    gen = torch.Generator(device=dev).manual_seed(31)
    gy = torch.randn(2, o, 1024, NA, device=dev, generator=gen)
    g = torch.randn(2, o, KS, rp * NA, device=dev, generator=gen)
    assert geo.steps(0) is not None and int(geo.steps(0)[:, :, 0].sum()) < geo.steps(0).shape[0] * geo.steps(0).shape[1] * (geo.steps(0).shape[2] - 1)
    z1, y1 = _hip.so3_dense_bwd(gy, geo), _hip.so3_dense_fwd(g, geo, 1024)
    monkeypatch.setattr(_hip, 'SKIP_DENSE_STEPS', False)
    assert geo.steps(0) is None
    z0, y0 = _hip.so3_dense_bwd(gy, geo), _hip.so3_dense_fwd(g, geo, 1024)
    assert torch.equal(z1, z0) and torch.equal(y1, y0)


def test_sorted_points_against_index_order(dev):
    """The same products with the query points in index order (round 5's form): the forward is bit-equal (a column's k-steps run in the
    same order either way), the backward -- whose contraction runs over the points -- to rounding; the channel moments the re-ordering
    pass leaves for the BatchNorm are those of the index-order pass to rounding, with the same pivot."""
    from vgtk import _hip
    s = _setup(dev, 2, 1024, layer=2)
    head, geo, rp = _geometry(s, dev)
    xyz = s['xyz']
    geo0 = _hip.DenseGeometry(xyz, xyz, head.memb, head.rows, rp, s['rk'], s['sigma'], NN, head.n_rows, sort=False)
    assert geo0.order is None and geo.order is not None
    gen = torch.Generator(device=dev).manual_seed(37)
    gy = torch.randn(2, 256, 1024, NA, device=dev, generator=gen)
    g = torch.randn(2, 256, KS, rp * NA, device=dev, generator=gen)
    y1 = _hip.so3_dense_fwd(g, geo, 1024)
    h1 = _hip.take_stats_hint(y1)
    y0 = _hip.so3_dense_fwd(g, geo0, 1024)
    h0 = _hip.take_stats_hint(y0)
    assert torch.equal(y1, y0)
    for a_, b_ in zip(h1, h0):
        assert torch.allclose(a_.sum(1, dtype=torch.float64), b_.sum(1, dtype=torch.float64), rtol=1e-5, atol=1e-3)
    z1, z0 = _hip.so3_dense_bwd(gy, geo), _hip.so3_dense_bwd(gy, geo0)
    assert float((z1 - z0).abs().max()) < 2e-6 * float(z0.abs().max())


# ------------------------------------------------------------------------------------------------------------------------
# conv + training-mode BatchNorm + leaky_relu as one node (round 6)
# ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('c,o,sort', [(32, 256, True), (32, 128, True), (32, 256, False), (64, 128, True), (128, 256, True)])
def test_conv_norm_node_against_separate_modules(dev, monkeypatch, c, o, sort):
    """vgtk.so3conv.conv_norm_act in training mode with the norm inside the conv's autograd node (the re-ordering pass applies it, the
    backward forms the gradient behind it inside the stored-operand split, the pre-activation is recovered from the OUTPUT) against the
    same conv followed by the BatchNormLeakyReLU module as a pass of its own (pinned against torch's BatchNorm2d + leaky_relu in
    tests/test_gpu_parity.py): y', dF, dW, d gamma, d beta and the running statistics."""
    import synth_clouds
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    import vgtk.so3conv.functional as L
    from vgtk import _hip
    monkeypatch.setattr(_hip, 'SORT_DENSE_POINTS', sort)
    B, P = 2, 512                                               # (c = 64 / 128: the operand kernel and the padded 64-channel gradient products)
    _, _, radius, sigma = synth_clouds.backbone_layers(4096)[2]
    xyz = torch.from_numpy(synth_clouds.laptop_batch(71, B, P)[0]).to(dev)
    pose = torch.eye(4, device=dev).repeat(B, P, 1, 1)
    gen = torch.Generator(device=dev).manual_seed(41)
    feats0 = torch.randn(B, c, P, NA, device=dev, generator=gen)
    W0 = torch.randn(o, c * KS, device=dev, generator=gen) * 0.05
    gamma0 = torch.rand(o, device=dev, generator=gen) + 0.5
    gamma0[3] = -0.7                                            # a negative scale: the sign of the pre-activation still comes from y'
    beta0 = torch.randn(o, device=dev, generator=gen) * 0.3
    gy = torch.randn(B, o, P, NA, device=dev, generator=gen)
    out = {}
    for fused in (False, True):
        monkeypatch.setattr(L, 'FUSE_CONV_NORM', fused)
        torch.manual_seed(2913)
        conv = sptk.InterSO3PoseConv(c, o, 1, 1, radius, sigma, NN, kanchor=NA, permute_modes=1).to(dev)
        norm = sptk.BatchNormLeakyReLU(o, negative_slope=0.01).to(dev)
        with torch.no_grad():
            conv.basic_conv.W.copy_(W0); norm.weight.copy_(gamma0); norm.bias.copy_(beta0)
        feats = feats0.clone().requires_grad_(True)
        L.BACKWARD_LOG = []
        y = sptk.conv_norm_act(conv, norm, zptk.SphericalPointCloudPose(xyz, feats, None, pose))[3].feats
        grads = torch.autograd.grad(y, [feats, conv.basic_conv.W, norm.weight, norm.bias], gy)
        log, L.BACKWARD_LOG = L.BACKWARD_LOG, None
        assert log[0]['regime'] == 'dense rows' and (log[0].get('norm') == 'in the node') == fused, log
        out[fused] = (y.detach(),) + grads + (norm.running_mean.clone(), norm.running_var.clone(), int(norm.num_batches_tracked))
        with torch.no_grad():                                  # (training mode, gradients off: the fused forward alone)
            y_ng = sptk.conv_norm_act(conv, norm, zptk.SphericalPointCloudPose(xyz, feats0, None, pose))[3].feats
        assert torch.equal(y_ng, y.detach())
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    a, b_ = out[True], out[False]
    assert rel(a[0], b_[0]) < 1e-5, rel(a[0], b_[0])                          # y'
    assert rel(a[1], b_[1]) < 2e-5 and rel(a[2], b_[2]) < 5e-5              # dF, dW
    assert rel(a[3], b_[3]) < 2e-5 and rel(a[4], b_[4]) < 2e-5              # d gamma, d beta
    assert rel(a[5], b_[5]) < 1e-5 and rel(a[6], b_[6]) < 1e-5 and a[7] == b_[7] == 1


@pytest.mark.parametrize('c,o,layer', [(128, 256, 2), (64, 128, 1)])
def test_forward_operand_made_by_one_kernel(dev, monkeypatch, c, o, layer):
    """vgtk._hip.so3_dense_gplanes (eap_so3_dense_gplanes_f32: G = W F over the referenced rows written straight as the product's planes,
    the row scales from a bound) against the GEMM + split it replaces and against float64, through the whole forward product; features and
    weights with magnitudes spread over several octaves per (cloud, anchor) / per row, so that the three power-of-two scales matter."""
    import vgtk.so3conv.functional as L
    from vgtk import _hip
    P = 512
    s = _setup(dev, 2, P, layer=layer)
    head, geo, rp = _geometry(s, dev)
    gen = torch.Generator(device=dev).manual_seed(43)
    feats = torch.randn(2, c, P, NA, device=dev, generator=gen) * torch.exp(2 * torch.randn(2, 1, 1, NA, device=dev, generator=gen))
    W = torch.randn(o, c * KS, device=dev, generator=gen) * 0.05 * torch.exp(1.5 * torch.randn(o, 1, device=dev, generator=gen))
    assert _hip.lib.eap_so3_dense_gplanes_supported(o, c, NA, KS, rp)
    calls = []
    orig = _hip.call
    monkeypatch.setattr(_hip, 'call', lambda name, *a, **k: (calls.append(name), orig(name, *a, **k))[1])
    y1 = L._dense_forward(feats, W, head.rows, geo, P)
    assert 'eap_so3_dense_gplanes_f32' in calls and 'eap_so3_dense_split_f32' not in calls
    monkeypatch.setattr(_hip, 'GPLANES', False)
    del calls[:]
    y0 = L._dense_forward(feats, W, head.rows, geo, P)
    assert 'eap_so3_dense_gplanes_f32' not in calls and 'eap_so3_dense_split_f32' in calls
    wd = _dense_weights64(s, head.rows, rp)                                   # [B,P,rp,A,K]
    fc = _hip.rows_gather(feats, head.rows, rp).double()                      # [B,c,rp,A]
    g64 = torch.einsum('ock,bcra->bokra', W.view(o, c, KS).double(), fc)
    ref = torch.einsum('bokra,bprak->bopa', g64, wd)
    mag = torch.einsum('bokra,bprak->bopa', g64.abs(), wd)
    gmag = torch.einsum('ock,bcra->bokra', W.view(o, c, KS).double().abs(), fc.abs())
    magm = torch.einsum('bokra,bpr->bopa', g64.abs(), _member(s, head.rows, rp))
    # the operand G itself now carries a split-product error (2^-21 of sum |W||F|), on top of the forward product's bound
    magg = torch.einsum('bokra,bprak->bopa', gmag, wd)
    for y in (y1, y0):
        assert float(((y.double() - ref).abs() / (1e-6 * mag + 5e-7 * magm + 1e-6 * magg).clamp(min=1e-30)).max()) < 1.0
    assert float((y1 - y0).abs().max()) < 2e-6 * float(y0.abs().max())
    assert torch.equal(y1, L._dense_forward(feats, W, head.rows, geo, P) if _hip.GPLANES else y1)


def test_many_rows_few_groups_take_the_dense_product(dev, monkeypatch):
    """Beyond 5 x nsample referenced rows the default decision looks at how many 16-row groups a point's list touches (the listed k-steps make
    the executed work proportional to that, not to the row count): 2048-point clouds with the 512-point plan's deepest radius reference
    ~410-420 rows per cloud but a list touches ~12 of their 26 groups -> dense rows in both directions, same results as the list kernels;
    with the bound lowered below that, back to the lists."""
    import synth_clouds
    import vgtk.so3conv.functional as L
    B, P, c, o = 2, 2048, 64, 256          # (c = 64: the forward's operand kernel passes the ~420 rows through LDS in two chunks)
    _, _, radius, sigma = synth_clouds.backbone_layers(512)[2]
    xyz = torch.from_numpy(synth_clouds.laptop_batch(91, B, P)[0]).to(dev)
    gen = torch.Generator(device=dev).manual_seed(47)
    feats0 = torch.randn(B, c, P, NA, device=dev, generator=gen)
    W0 = torch.randn(o, c * KS, device=dev, generator=gen) * 0.05
    y0, gF0, gW0, log0, _ = _layer_run(dev, monkeypatch, 'off', xyz, None, feats0, W0, c, o, radius, sigma)
    y1, gF1, gW1, log1, y1n = _layer_run(dev, monkeypatch, 'auto', xyz, None, feats0, W0, c, o, radius, sigma)
    assert log1[0]['regime'] == 'dense rows' and log1[0]['referenced_rows_max'] > 5 * NN, log1
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    assert rel(y1, y0) < 2e-5 and rel(gF1, gF0) < 2e-5 and rel(gW1, gW0) < 5e-5 and torch.equal(y1n, y1)
    monkeypatch.setattr(L, 'DENSE_MAX_GROUPS', 4.0)
    _, _, _, log2, _ = _layer_run(dev, monkeypatch, 'auto', xyz, None, feats0, W0, c, o, radius, sigma)
    assert log2[0]['regime'] != 'dense rows'


def test_frozen_conv_under_a_trainable_norm_keeps_the_separate_module(dev, monkeypatch):
    """Gradients wanted for the norm's parameters only (conv weights and input frozen): the node's backward is the conv's dense backward, which
    does not exist then -- conv_norm_act falls back to conv + norm module, and d gamma / d beta come out as from the modules."""
    import synth_clouds
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    B, P, c, o = 2, 512, 32, 256
    _, _, radius, sigma = synth_clouds.backbone_layers(4096)[2]
    xyz = torch.from_numpy(synth_clouds.laptop_batch(73, B, P)[0]).to(dev)
    pose = torch.eye(4, device=dev).repeat(B, P, 1, 1)
    gen = torch.Generator(device=dev).manual_seed(53)
    feats = torch.randn(B, c, P, NA, device=dev, generator=gen)
    gy = torch.randn(B, o, P, NA, device=dev, generator=gen)
    torch.manual_seed(2913)
    conv = sptk.InterSO3PoseConv(c, o, 1, 1, radius, sigma, NN, kanchor=NA, permute_modes=1).to(dev)
    conv.basic_conv.W.requires_grad_(False)
    norm = sptk.BatchNormLeakyReLU(o, negative_slope=0.01).to(dev)
    x = zptk.SphericalPointCloudPose(xyz, feats, None, pose)
    y = sptk.conv_norm_act(conv, norm, x)[3].feats
    gw, gb = torch.autograd.grad(y, [norm.weight, norm.bias], gy)
    with torch.no_grad():
        raw = conv(x)[3].feats
    norm2 = sptk.BatchNormLeakyReLU(o, negative_slope=0.01).to(dev)
    y2 = norm2(raw)
    gw2, gb2 = torch.autograd.grad(y2, [norm2.weight, norm2.bias], gy)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    assert rel(y.detach(), y2.detach()) < 1e-5 and rel(gw, gw2) < 2e-5 and rel(gb, gb2) < 2e-5
