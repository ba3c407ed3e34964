"""CPU: hand-built known-answer cases for oracle/native_ops.c -- the plain-C restatement of the reference's CUDA-only
native operators (no reference test and no importable implementation exists for them); every
expected value below is derived by hand from the cited kernel source, not from running the oracle.

  ball_query               vgtk/vgtk/cuda/grouping_cuda_kernel.cu:L68-113
  gather_points            vgtk/vgtk/cuda/gathering_cuda_kernel.cu:L43-98
  furthest_point_sampling  vgtk/vgtk/cuda/grouping_cuda_kernel.cu:L352-466
  inter / intra zpconv     vgtk/vgtk/cuda/zpconv_cuda_kernel.cu:L33-195
  chamfer                  extensions/chamfer_dist/chamfer.cu:L15-145, L173-201
  anchor queries           vgtk/vgtk/cuda/grouping_cuda_kernel.cu:L117-247

The same cases are asserted on the HIP kernels in tests/test_gpu_parity.py::test_native_known_answers."""
import numpy as np

from oracle import native


def line_cloud(n=8):
    """support points on the x axis at 0, 1, ..., n-1: xyz [1,3,n]."""
    xyz = np.zeros((1, 3, n), np.float32)
    xyz[0, 0] = np.arange(n)
    return xyz


BALL_CASES = [
    # (query x positions, radius, nsample, expected rows)
    ([0, 3, 100], 1.5, 4, [[0, 1, 0, 1],      # 2 hits < nsample-1 = 3: cyclic repeat of the hits (L100-105)
                           [2, 3, 4, 0],      # exactly nsample-1 hits: NO padding, the last slot keeps the host's zero init
                           [0, 0, 0, 0]]),    # empty ball: zeros (cnt = 0 < 3 copies zeros onto zeros)
    ([0, 3, 100], 1.0, 2, [[0, 0],            # strict d2 < r2: x=1 is at d == r and excluded; cnt = 1 == nsample-1
                           [3, 0],            # only x=3 itself
                           [0, 0]]),
    ([0, 3, 100], 10.0, 3, [[0, 1, 2],        # more hits than slots: the first nsample in INDEX order, not the nearest
                            [0, 1, 2],
                            [0, 0, 0]]),
    ([3], 1.5, 8, [[2, 3, 4, 2, 3, 4, 2, 3]]),  # 3 hits, 8 slots: idx[k + cnt] = idx[k] walks over what it just wrote
]


def known_ball_query(fn):
    for qx, radius, nsample, want in BALL_CASES:
        q = np.zeros((1, 3, len(qx)), np.float32)
        q[0, 0] = qx
        got = fn(q, line_cloud(), radius, nsample)
        np.testing.assert_array_equal(np.asarray(got)[0], np.array(want, np.int32), err_msg=str((qx, radius, nsample)))


def test_ball_query_known_answers():
    known_ball_query(native.ball_query)
    # float64 entry: radius2 is formed in float THEN widened (L78: `float radius2 = radius * radius`)
    q = np.zeros((1, 3, 1), np.float64)
    got = native.ball_query(q, line_cloud().astype(np.float64), 1.5, 4)
    np.testing.assert_array_equal(got[0, 0], [0, 1, 0, 1])


def test_ball_query_rounding_is_one_per_operation():
    """d2 = dx*dx + dy*dy + dz*dz with one rounding per operation (the oracle and the HIP kernel are both built with
    -ffp-contract=off): a point whose exactly-rounded d2 equals r2 is excluded, although the FMA-contracted value
    would fall just inside."""
    x = np.float32(0.1)
    r2 = np.float32(np.float32(np.float32(x * x) + np.float32(x * x)) + np.float32(x * x))
    radius = np.float32(np.sqrt(np.float64(r2)))
    if np.float32(radius * radius) != r2:          # the float radius whose square lands exactly on r2 may not exist
        return
    xyz = np.zeros((1, 3, 2), np.float32)
    xyz[0, :, 1] = x
    q = np.zeros((1, 3, 1), np.float32)
    got = native.ball_query(q, xyz, float(radius), 2)
    np.testing.assert_array_equal(got[0, 0], [0, 0])


def known_gather(fwd, bwd):
    pts = np.arange(2 * 3 * 5, dtype=np.float32).reshape(2, 3, 5)
    idx = np.array([[4, 0, 0, 2], [1, 1, 3, 3]], np.int32)
    out = np.asarray(fwd(pts, idx))
    want = np.stack([pts[b][:, idx[b]] for b in range(2)])
    np.testing.assert_array_equal(out, want)
    g = np.ones((2, 3, 4), np.float32)
    gp = np.asarray(bwd(g, idx, 5))
    want_g = np.zeros((2, 3, 5), np.float32)
    want_g[0, :, [4, 0, 2]] = np.array([1, 2, 1], np.float32)[:, None]       # index 0 is referenced twice: gradients add
    want_g[1, :, [1, 3]] = 2
    np.testing.assert_array_equal(gp, want_g)


def test_gather_points_known_answers():
    known_gather(native.gather_points_forward, native.gather_points_backward)


def known_fps(fn):
    # 1) first index is always 0; then the farthest from the chosen set, ties to the smaller index
    xyz = np.zeros((1, 3, 6), np.float32)
    xyz[0, 0] = [1, 2, 4, 8, 16, 8]
    got = np.asarray(fn(xyz, 4))[0]
    # from x=1: farthest is 16 (idx 4); then min-dist-to-{1,16}: x=8 -> min(49, 64) = 49 at idx 3 and idx 5 (tie) -> the
    # per-thread scan is strict '>' (first wins inside a thread) and the tree keeps the lower slot on ties: n = 6 ->
    # block = 4 threads; idx 3 lives in thread 3, idx 5 in thread 1; the reduction (s=2: t1 vs t3 -> `v2 > v1` is false, keeps
    # thread 1 = idx 5 ... ) -- so the tie resolves by THREAD slot, not by point index:
    #   s=2: slot0 <- max(slot0, slot2), slot1 <- max(slot1, slot3) with slot1 = (49, idx 5), slot3 = (49, idx 3): keeps idx 5
    #   s=1: slot0 vs slot1
    # slot0 after s=2: thread 0 holds idx 0 (x=1, d=0) and idx 4 (chosen, d=0) -> best 0; thread 2 holds idx 2 (x=4: min(9,144)=9)
    #   -> slot0 = 9 (idx 2); slot1 = 49 (idx 5) -> winner idx 5
    assert got[0] == 0 and got[1] == 4 and got[2] == 5, got
    # next: distances to {1,16,8(idx5)}: idx3 (x=8) -> 0, idx2 (x=4) -> min(9, 16) = 9, idx1 (x=2) -> 1 -> idx 2
    assert got[3] == 2, got
    # 2) points with |x|^2 <= 1e-3 are never selected (L385-387) -- except index 0, which is always the first pick
    xyz = np.zeros((1, 3, 5), np.float32)
    xyz[0, 0] = [0.0, 0.01, 5.0, 0.02, 3.0]          # idx 1 and 3 have |x|^2 = 1e-4, 4e-4 <= 1e-3
    got = np.asarray(fn(xyz, 3))[0]
    np.testing.assert_array_equal(got, [0, 2, 4])
    # 3) m larger than the number of eligible points: once every eligible point is taken all thread-bests stay at their
    #    initial (-1, 0) or reach 0, and the arg-max falls back to an already-chosen / index-0 entry -- never a skipped one
    got = np.asarray(fn(xyz, 5))[0]
    assert set(got[3:].tolist()) <= {0, 2, 4}, got


def test_furthest_point_sampling_known_answers():
    known_fps(native.furthest_point_sampling)


def known_zpconv(inter_fwd, inter_bwd, intra_fwd, intra_bwd):
    # inter: one cloud, 2 query points, 1 anchor, 1 kernel point, 2 neighbours, 1 channel, 3 support points
    idx = np.array([0, 2, 1, 1], np.int32).reshape(1, 2, 1, 1, 2)
    w = np.array([0.5, 2.0, 1.0, 3.0], np.float32).reshape(1, 2, 1, 1, 2)
    feats = np.array([10.0, 20.0, 30.0], np.float32).reshape(1, 1, 3, 1)
    out = np.asarray(inter_fwd(idx, w, feats))
    np.testing.assert_array_equal(out.reshape(-1), [0.5 * 10 + 2 * 30, 1 * 20 + 3 * 20])          # [b,c,k,p,a]
    g = np.array([1.0, 10.0], np.float32).reshape(1, 1, 1, 2, 1)
    gf = np.asarray(inter_bwd(idx, w, g, 3))
    np.testing.assert_array_equal(gf.reshape(-1), [0.5 * 1, 1 * 10 + 3 * 10, 2.0 * 1])            # a support point hit twice adds
    # the index really is per (a, k): two kernel points reading DIFFERENT neighbours
    idx2 = np.array([0, 1], np.int32).reshape(1, 1, 1, 2, 1)
    w2 = np.ones((1, 1, 1, 2, 1), np.float32)
    out2 = np.asarray(inter_fwd(idx2, w2, feats[:, :, :2]))
    np.testing.assert_array_equal(out2.reshape(-1), [10.0, 20.0])
    # intra: 2 output anchors reading 2 taps of 3 input anchors
    iidx = np.array([[0, 2], [1, 1]], np.int32)
    iw = np.array([[[1.0, 2.0]], [[3.0, 4.0]]], np.float32)                                       # [a_out, k=1, ann=2]
    f = np.array([1.0, 10.0, 100.0], np.float32).reshape(1, 1, 1, 3)
    out = np.asarray(intra_fwd(iidx, iw, f))
    np.testing.assert_array_equal(out.reshape(-1), [1 * 1 + 2 * 100, 3 * 10 + 4 * 10])
    gi = np.asarray(intra_bwd(iidx, iw, np.array([1.0, 1.0], np.float32).reshape(1, 1, 1, 1, 2), 3))
    np.testing.assert_array_equal(gi.reshape(-1), [1.0, 7.0, 2.0])


def test_zpconv_known_answers():
    known_zpconv(native.inter_zpconv_forward, native.inter_zpconv_backward, native.intra_zpconv_forward, native.intra_zpconv_backward)


def known_chamfer(fwd, bwd):
    # xyz1 = two points, xyz2 = four points with an exact tie: the FIRST minimum in index order wins (`dist < best`, L40-74)
    xyz1 = np.array([[[0, 0, 0], [10, 0, 0]]], np.float32)
    xyz2 = np.array([[[1, 0, 0], [-1, 0, 0], [0, 1, 0], [9, 0, 0]]], np.float32)
    d1, d2, i1, i2 = [np.asarray(t) for t in fwd(xyz1, xyz2)]
    np.testing.assert_array_equal(i1, [[0, 3]])                      # three points at distance 1 from the origin: index 0
    np.testing.assert_array_equal(d1, [[1.0, 1.0]])
    np.testing.assert_array_equal(i2, [[0, 0, 0, 1]])
    np.testing.assert_array_equal(d2, [[1.0, 1.0, 1.0, 1.0]])
    # a tie ACROSS the 512-point tile boundary of the CUDA kernel (k2 loop, L107-118: `dist[..] > best` is strict): first wins
    m = 600
    big = np.full((1, m, 3), 50.0, np.float32)
    big[0, 5] = [2, 0, 0]
    big[0, 520] = [-2, 0, 0]
    _, _, i1b, _ = [np.asarray(t) for t in fwd(xyz1[:, :1], big)]
    assert i1b[0, 0] == 5
    # gradient: 2 (p1 - p2) g scattered to both sides (L173-201)
    g1 = np.array([[1.0, 2.0]], np.float32)
    g2 = np.zeros((1, 4), np.float32)
    gx1, gx2 = [np.asarray(t) for t in bwd(xyz1, xyz2, i1, i2, g1, g2)]
    np.testing.assert_array_equal(gx1, [[[-2, 0, 0], [4, 0, 0]]])
    np.testing.assert_array_equal(gx2, [[[2, 0, 0], [0, 0, 0], [0, 0, 0], [-4, 0, 0]]])


def test_chamfer_known_answers():
    known_chamfer(native.chamfer_forward, native.chamfer_backward)


def test_anchor_queries_known_answers():
    # anchor_query (L181-247): one point at distance 2 along +z, anchor +z -> theta = acos(2 / (2 + 1e-6)); w = (kw-|x|)^2 + ((kh-theta)|x|)^2
    gxyz = np.zeros((1, 3, 1, 1), np.float64)
    gxyz[0, 2, 0, 0] = 2.0
    anchors = np.array([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0]], np.float64)
    kpts = np.array([[2.0, 0.0], [1.0, 0.5]], np.float64)
    w = native.anchor_query(np.zeros((1, 1), np.int32), np.zeros((1, 1, 1), np.int32), gxyz, anchors, kpts, 1)
    w = np.asarray(w[0] if isinstance(w, (list, tuple)) else w)
    norm = 2.0 + 1e-6
    th0, th1 = np.arccos(2.0 / norm), np.arccos(0.0)
    want = [[(2 - norm) ** 2 + ((0 - th0) * norm) ** 2, (1 - norm) ** 2 + ((0.5 - th0) * norm) ** 2],
            [(2 - norm) ** 2 + ((0 - th1) * norm) ** 2, (1 - norm) ** 2 + ((0.5 - th1) * norm) ** 2]]
    np.testing.assert_allclose(w.reshape(2, 2), want, rtol=1e-12)
    # initial_anchor_query (L117-167): centre at the origin, radius 1, one fragment point at (0.5,0,0), kernel point offsets 0 and (0.5,0,0)
    centers = np.zeros((1, 3, 1), np.float64)
    frag = np.array([[0.5, 0.0, 0.0], [3.0, 0.0, 0.0]], np.float64)                    # the second point is outside the radius
    k = np.array([[[0.0, 0.0, 0.0]], [[0.5, 0.0, 0.0]]], np.float64)                   # [ks=2, na=1, 3]
    wt, cnt = [np.asarray(t) for t in native.initial_anchor_query(centers, frag, k, 1.0, 0.5)]
    np.testing.assert_allclose(wt.reshape(-1), [1 - 0.25 / 0.5, 1.0], rtol=1e-12)
    np.testing.assert_array_equal(cnt.reshape(-1), [1.0, 1.0])
