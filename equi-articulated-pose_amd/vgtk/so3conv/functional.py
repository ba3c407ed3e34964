"""vgtk.so3conv.functional -- operator layer of the SE(3)-equivariant point convolution
(reference: vgtk/vgtk/so3conv/functional.py).  Same function names, argument order and return
structures as the reference; the compute runs in libeap_hip.so (HIP, gfx950):

    ball_query -> so3_prep (offsets, relative-rotation anchor) -> fused grouping
    (kernel weights + anchor permutation + gather + weighted sum) -> fp32-MFMA contraction

Differences a caller can observe (all documented in DESIGN.md):
  * `inter_w` is returned as a lazy `InterWeights` handle; call `.materialize()` (or set
    vgtk.so3conv.functional.MATERIALIZE_INTER_W = True) to get the [b,p,na,ks,nn] tensor the
    reference always builds (12 GB at B=8, P=4096).  Callers in SPConvNets only hand it back to
    the next conv, which ignores it for stride 1 (functional.py:L1025ff recomputes everything).
  * gradients flow to `feats` and `W` (what the reference trains); xyz / pose are treated as data.
  * float32 device tensors only -- there is no CPU path.
"""
import math
import os
import weakref

import numpy as np
import torch

import vgtk
import vgtk.functional as fr
import vgtk.pc as pctk
import vgtk.spconv as zpconv
import vgtk.cuda.grouping as cuda_nn

from .. import _hip

inter_so3conv_feat_grouping = zpconv.inter_zpconv_grouping_naive
batched_index_select = zpconv.batched_index_select
batched_index_select_other = zpconv.batched_index_select_other

MATERIALIZE_INTER_W = False

# ------------------------------------------------------------------------------------------------
# constants (functional.py:L111-121, L2630-2659)
# ------------------------------------------------------------------------------------------------
GAMMA_SIZE = 3
ROOT = vgtk.__path__[0]
Rs, R_idx, canonical_relative = fr.icosahedron_so3(GAMMA_SIZE)
_KP = np.load(os.path.join(ROOT, 'data', 'anchors', 'constants.npz'))


def select_anchor(anchors, k):
    if k == 1:
        return anchors[29][None]
    elif k == 20:
        return anchors[::3]
    elif k == 40:
        return anchors.reshape(20, 3, 3, 3)[:, :2].reshape(-1, 3, 3)
    return anchors


def get_anchors(k=60):
    return select_anchor(Rs, k)


def get_intra_idx():
    return R_idx


def get_canonical_relative():
    return canonical_relative


def get_kernel_points_np(radius, aperature, kernel_size, multiplier=1):
    """Kernel points [n,3] on a cone around +z (functional.py:L73-89): shell i of `kernel_size` shells sits at height
    z_i = i * radius / (kernel_size - 1) and carries i * multiplier + 1 polar angles alpha (interior points of
    [0, aperature / 2]); the j-th of them contributes a circle of 2 j + 1 points of radius z_i tan(alpha)."""
    assert isinstance(kernel_size, int)
    shells = []
    for i, z in enumerate(np.linspace(0, radius, kernel_size, dtype=np.float32)):
        for j, alpha in enumerate(zpconv.get_angular_kernel_points_np(aperature, i * multiplier + 1)):
            n = 2 * j + 1
            phi = np.linspace(0, 2 * np.pi, n, endpoint=False, dtype=np.float32)
            rho = z * np.tan(alpha)
            shells.append(np.stack([rho * np.cos(phi), rho * np.sin(phi), np.full(n, z)], axis=1))
    return np.concatenate(shells, axis=0)


def get_spherical_kernel_points_np(radius, kernel_size, multiplier=3):
    """Kernel points [n,3] on concentric spheres (functional.py:L91-109): sphere i of radius i * radius / (kernel_size - 1)
    carries an m x m longitude / colatitude grid, m = i * multiplier + 1 (longitudes without, colatitudes with end point)."""
    assert isinstance(kernel_size, int)
    spheres = []
    for i, r in enumerate(np.linspace(0, radius, kernel_size, dtype=np.float32)):
        m = i * multiplier + 1
        lon = np.linspace(0, 2 * np.pi, m, endpoint=False, dtype=np.float32)[:, None]
        col = np.linspace(0, np.pi, m, endpoint=True, dtype=np.float32)[None, :]
        xyz = np.stack([r * np.cos(lon) * np.sin(col), r * np.sin(lon) * np.sin(col), np.broadcast_to(r * np.cos(col), (m, m))], axis=-1)
        spheres.append(xyz.reshape(m * m, 3))
    return np.concatenate(spheres, axis=0)


def get_sphereical_kernel_points_from_ply(radius, kernel_size):
    """kernel_size 1/2/3 -> 24/30/66 kernel points rescaled so the max norm is `radius`."""
    assert 0 < kernel_size <= 3
    pts = _KP['kpsphere%d' % {1: 24, 2: 30, 3: 66}[kernel_size]].astype('float32')
    r = np.sqrt((pts ** 2).sum(1).max())
    return pts * radius / r


def get_2D_res_anchors():
    """4 rotations about the y axis by multiples of 90 degrees (functional.py:L29-46) -> [4,3,3]."""
    mats = []
    for i in range(4):
        th = i * (np.pi / 2.)
        c, s_ = np.cos(th), np.sin(th)
        mats.append(torch.from_numpy(np.array([[c, 0., s_], [0., 1., 0.], [-s_, 0., c]], dtype=np.float64)).float().unsqueeze(0))
    return torch.cat(mats, dim=0)


RES_ROT_2D = get_2D_res_anchors()


def initial_anchor_query(frag, centers, kernels, r, sigma):
    """frag [m,3], centers [b,3,nc], kernels [ks,na,3] -> (w, cnt) [b,ks,nc,na] (functional.py:L129-130)."""
    return cuda_nn.initial_anchor_query(centers, frag, kernels, r, sigma)


def inter_so3conv_blurring(xyz, feats, n_neighbor, radius, stride, inter_idx=None, lazy_sample=True,
                           radius_expansion=1.0):
    """Low-pass blur / pool before a strided conv (functional.py:L133-141)."""
    sample_idx = sample_xyz = None
    if inter_idx is None:
        _, inter_idx, sample_idx, sample_xyz = zpconv.inter_zpconv_grouping_ball(xyz, stride, radius * radius_expansion,
                                                                               n_neighbor, lazy_sample)
    if stride == 1:
        return zpconv.inter_blurring_naive(inter_idx, feats), xyz
    return zpconv.inter_pooling_naive(inter_idx, sample_idx, feats), sample_xyz


def canonicalize_points(xyz, pose):
    """R^T (x - t) per point (functional.py:L206-214): xyz [b,3,p], pose [b,p,4,4] -> [b,3,p]."""
    rotations = pose[:, :, :3, :3]
    translations = pose[:, :, :3, -1]
    cana = torch.matmul(torch.transpose(rotations, 2, 3),
                        (xyz.contiguous().transpose(1, 2).contiguous() - translations).unsqueeze(-1)).squeeze(-1)
    return cana.contiguous().transpose(1, 2).contiguous()


def get_occupancy_features(pc, n_anchor, use_center=False):
    """pc [nb,np,3] -> ones [nb,1,np,na] (functional.py:L50-69; normals are not supported --
    the reference branch for them is broken: `ns.anchors` at L61)."""
    nb, np_, nd = pc.shape
    if nd != 3:
        raise NotImplementedError('get_occupancy_features: xyz-only point clouds')
    features = torch.ones(nb, 1, np_, n_anchor, dtype=torch.float32, device=pc.device)
    if use_center:
        features[:, :, 0, :] = 0.0
    return features


# ------------------------------------------------------------------------------------------------
# per-(anchor set, kernel set, device) tables
# ------------------------------------------------------------------------------------------------
_TABLES = {}


def _group_tables(anchors):
    """mult table (uint8 [na,na]) + identity index when `anchors` is a group, else (None, None)."""
    # keyed on the tensor OBJECT and its version counter (a load_state_dict / copy_ into the buffer bumps
    # the version; a recycled allocation is a different object), never on the data pointer
    key = (id(anchors), anchors._version, str(anchors.device), anchors.shape[0])
    hit = _TABLES.get(key)
    if hit is not None and hit[0]() is anchors:
        return hit[1]
    A = anchors.detach().double().cpu().numpy()
    na = A.shape[0]
    prod = np.einsum('gij,ajk->gaik', A, A)
    score = np.einsum('gaij,cij->gac', prod, A)
    mult = score.argmax(-1)
    closed = np.allclose(score.max(-1), 3.0, atol=1e-4)
    ident = int(np.einsum('cii->c', A).argmax())
    has_identity = np.allclose(A[ident], np.eye(3), atol=1e-5)
    if closed and has_identity:
        out = (torch.from_numpy(mult.astype(np.uint8)).to(anchors.device).contiguous(), ident)
    else:
        out = (None, None)
    if len(_TABLES) > 256:
        _TABLES.clear()
    _TABLES[key] = (weakref.ref(anchors), out)
    return out


_MULTINV = {}


def _group_tables_inverse(mult):
    """multinv[r][a'] = a with mult[r][a] = a' (the inverse permutation of every row), cached per table."""
    key = (id(mult), mult._version)
    hit = _MULTINV.get(key)
    if hit is not None and hit[0]() is mult:
        return hit[1]
    na = mult.shape[0]
    multinv = torch.empty_like(mult)
    multinv.scatter_(1, mult.long(), torch.arange(na, device=mult.device, dtype=torch.uint8).repeat(na, 1))
    if len(_MULTINV) > 256:
        _MULTINV.clear()
    _MULTINV[key] = (weakref.ref(mult), multinv)
    return multinv


_COSETS = {}
STORE_ORDER_COLUMNS = True     # streamed forward: the transposed intermediate's columns in the grouping kernel's store order
COSET_OPERAND = True      # permuted-pose backward: coset-major LDS operand (False: per-anchor byte-table lookups, for A/B runs)


def _coset_tables(table, ident):
    """For a byte table of LEFT multiplications of an anchor group (table[r][a] = index of g_r . a for some bijection
    r -> g_r: `mult` or its row-wise inverse), a re-ordering of the anchors that turns every row into block moves:

        order [4*nb]   anchor index at coset-major position i: blocks of 4 = left cosets a.H of a Klein four-group
                       H = {e, h1, h2, h3} (h_i h_j = h_{i ^ j}), position j of a block = a.h_j;
        code [na, 16]  for row r and block b: sigma | x << 4 with table[r][order[4 b + j]] == order[4 sigma + (j ^ x)]

    (left multiplication maps left cosets onto left cosets; inside a block it multiplies the H-part: an index XOR).  A
    kernel that keeps the anchor axis of its LDS operand in `order` reads the four permuted anchors of a block as ONE
    16-byte word + a shuffle (csrc/so3_inter_inv.hip, COSET; csrc/so3_inter_lists2.hip, PERM).  -> (order uint8 [64] padded
    with the last anchor, code uint8 [na,16], pos uint8 [64] = position of every anchor in `order`) on the table's device, or
    None when the group has no such subgroup / the structure check fails."""
    key = (id(table), table._version, int(ident))
    hit = _COSETS.get(key)
    if hit is not None and hit[0]() is table:
        return hit[1]
    T = table.detach().cpu().numpy().astype(np.int64)
    na = T.shape[0]
    out = None
    if na % 4 == 0 and na // 4 <= 16:
        elem = T[:, ident]                                   # row r multiplies by the element with this anchor index
        row_of = np.empty(na, np.int64)
        row_of[elem] = np.arange(na)
        times = lambda a, b: int(T[row_of[a], b])            # a . b as anchor indices
        invol = [a for a in range(na) if a != ident and times(a, a) == ident]
        H = None
        for i, h1 in enumerate(invol):
            for h2 in invol[i + 1:]:
                if times(h1, h2) == times(h2, h1) and times(h1, h2) not in (ident, h1, h2):
                    H = [ident, h1, h2, times(h1, h2)]
                    break
            if H:
                break
        if H:
            # blocks that belong together stay together: the normaliser K of H (the tetrahedral subgroup of the icosahedral
            # group, K / H cyclic of order 3) -- a left multiplication maps the three H-cosets of a K-coset onto the three
            # H-cosets of ONE other K-coset, i.e. three of a 16-anchor group's four source blocks are neighbours in memory
            inv = {a: next(b for b in range(na) if times(a, b) == ident) for a in range(na)}
            K = [g for g in range(na) if sorted(times(times(g, h), inv[g]) for h in H) == sorted(H)]
            reps, covered = [], set()
            for k in K:
                if k not in covered:
                    reps.append(k)
                    covered.update(times(k, h) for h in H)
            order, seen = [], set()
            for a in range(na):
                if a not in seen:
                    for k in reps:
                        blk = [times(times(a, k), h) for h in H]
                        if blk[0] not in seen:
                            order += blk
                            seen.update(blk)
            pos = np.empty(na, np.int64)
            pos[order] = np.arange(na)
            code = np.zeros((na, 16), np.uint8)
            ok = len(order) == na
            for r in range(na):
                for b in range(na // 4):
                    src = pos[T[r, order[4 * b]]]
                    sigma, x = src >> 2, src & 3
                    code[r, b] = sigma | (x << 4)
                    ok = ok and all(T[r, order[4 * b + j]] == order[4 * sigma + (j ^ x)] for j in range(4))
            if ok:
                padded = np.asarray(order + [order[-1]] * (64 - na), np.uint8)
                where = np.zeros(64, np.uint8)
                where[:na] = pos
                out = (torch.from_numpy(padded).to(table.device), torch.from_numpy(code).to(table.device).contiguous(),
                       torch.from_numpy(where).to(table.device))
    if len(_COSETS) > 256:
        _COSETS.clear()
    _COSETS[key] = (weakref.ref(table), out)
    return out


def rotated_kernels(anchors, kernels):
    """rk [na,ks,3] = A_a kappa_k  (functional.py:L2519)."""
    return torch.matmul(anchors, kernels.transpose(0, 1)).permute(0, 2, 1).contiguous()


class InterWeights:
    """Lazy stand-in for the reference's inter_w [b,p,na,ks,nn] (functional.py:L2508-2549)."""

    def __init__(self, gx, rk, sigma):
        self.gx, self.rk, self.sigma = gx, rk, float(sigma)

    @property
    def shape(self):
        b, p, nn, _ = self.gx.shape
        return torch.Size((b, p, self.rk.shape[0], self.rk.shape[1], nn))

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def materialize(self):
        return _hip.so3_inter_weights(self.gx, self.rk, self.sigma)


def _as_gx(grouped_xyz):
    """[b,3,p,nn] -> float4 [b,p,nn,4] with r = 0."""
    b, _, p, nn = grouped_xyz.shape
    gx = torch.zeros(b, p, nn, 4, dtype=torch.float32, device=grouped_xyz.device)
    gx[..., :3] = grouped_xyz.permute(0, 2, 3, 1)
    return gx


def inter_so3conv_grouping_anchor(grouped_xyz, anchors, kernels, sigma, interpolate='linear'):
    """grouped_xyz [b,3,p,nn] -> materialised w [b,p,na,ks,nn] = relu(1 - |g - A_a k|^2/sigma)
    (functional.py:L2508-2549)."""
    if interpolate != 'linear':
        raise NotImplementedError('kernel function %s is not implemented!' % interpolate)
    _hip.check_input(grouped_xyz)
    return _hip.so3_inter_weights(_as_gx(grouped_xyz), rotated_kernels(anchors, kernels), float(sigma))


# ------------------------------------------------------------------------------------------------
# autograd ops
# ------------------------------------------------------------------------------------------------
class _InterGroup(torch.autograd.Function):
    """new_feats[b,c,k,p,a] = sum_n feats[b,c,idx_n,perm_n(a)] w(p,a,k,n)  (functional.py:L1221-1261)."""

    @staticmethod
    def forward(ctx, feats, idx, gx, rk, mult, sigma, ident=0, nonident=None):
        feats = feats.contiguous()
        ctx.ident = ident
        ctx.save_for_backward(idx, gx, rk, mult if mult is not None else torch.empty(0))
        ctx.has_mult = mult is not None
        ctx.sigma = sigma
        ctx.n = feats.shape[2]
        return _hip.so3_inter_group_fwd(feats, idx, gx, rk, mult, sigma, nonident)

    @staticmethod
    def backward(ctx, gout):
        idx, gx, rk, mult = ctx.saved_tensors
        g = _hip.so3_inter_group_bwd(gout.contiguous(), idx, gx, rk, mult if ctx.has_mult else None,
                                     ctx.sigma, ctx.n, ctx.ident)
        return g, None, None, None, None, None, None, None


class _IntraGroup(torch.autograd.Function):
    """out[b,c,t,p,a] = feats[b,c,p,intra_idx[a,t]]  (functional.py:L2553-2602)."""

    @staticmethod
    def forward(ctx, feats, intra_idx32):
        ctx.save_for_backward(intra_idx32)
        return _hip.so3_intra_group_fwd(feats.contiguous(), intra_idx32)

    @staticmethod
    def backward(ctx, gout):
        intra_idx32, = ctx.saved_tensors
        return _hip.so3_intra_group_bwd(gout.contiguous(), intra_idx32), None


class _Contract(torch.autograd.Function):
    """y[b,o,pa] = W[o,ck] x[b,ck,pa] on the matrix cores (BasicSO3Conv, modules.py:L48-55)."""

    @staticmethod
    def forward(ctx, W, x, epilogue=None):
        W = W.contiguous()
        x = x.contiguous()
        b, ck, pa = x.shape
        o = W.shape[0]
        y = torch.empty(b, o, pa, dtype=torch.float32, device=x.device)
        if epilogue is not None:
            if not epilogue.inference:
                raise RuntimeError('a folded epilogue is an inference-time fusion: build and use it under torch.no_grad()')
            res = None if epilogue.residual is None else epilogue.residual.contiguous().view(b, o, pa)
            epilogue.applied = _hip.gemm_epilogue(0, o, pa, ck, W, ck, x, pa, ck * pa, y, pa, o * pa, b, epilogue.scale, epilogue.shift,
                                                  epilogue.slope, res)
        if epilogue is None or not epilogue.applied:
            _hip.gemm(0, 0, o, pa, ck, W, ck, 0, x, pa, ck * pa, y, pa, o * pa, b)
        ctx.save_for_backward(W, x)
        return y

    @staticmethod
    def backward(ctx, gy):
        W, x = ctx.saved_tensors
        gy = gy.contiguous()
        b, ck, pa = x.shape
        o = W.shape[0]
        gW = gx = None
        if ctx.needs_input_grad[1]:
            gx = torch.empty_like(x)          # W^T gy : [ck,o] [o,pa]
            if _hip.SPLIT_BF16_CONTRACTION and ck >= 128 and ck % 128 == 0 and o % 16 == 0:
                Wt = W.t().contiguous()       # the split kernel reads its shared operand k-contiguous: [ck, o]
                _hip.gemm(0, 0, ck, pa, o, Wt, o, 0, gy, pa, o * pa, gx, pa, ck * pa, b)
            else:
                _hip.gemm(1, 0, ck, pa, o, W, ck, 0, gy, pa, o * pa, gx, pa, ck * pa, b)
        if ctx.needs_input_grad[0]:
            gW = torch.empty_like(W)          # sum_b gy_b x_b^T : [o,pa] [pa,ck]
            _hip.gemm_reduce(0, 1, o, ck, pa, gy, pa, o * pa, x, pa, ck * pa, gW, ck, b)
        return gW, gx, None


# Feature-gradient strategy of the fused inter convolution: "auto" picks the re-associated
# (inverse-list) path when at most 1/INV_ROW_FRACTION of the support points are referenced by
# any neighbour list, else dX = W^T dY followed by the transposed grouping.
BACKWARD_MODE = 'auto'      # 'auto' | 'inverse' | 'dx'
INV_ROW_FRACTION = 4


INV_LISTS_MAX_ROWS = 16384      # csrc/inv_lists.hip sorts a cloud's support rows in LDS


def _inv_lists_supported(idx, n_sup, na, ks):
    return na % 4 == 0 and ks <= 32 and n_sup <= INV_LISTS_MAX_ROWS and (idx.shape[1] * idx.shape[2]) % 4 == 0


LISTS_ON_SIDE_STREAM = True     # inverse neighbour lists built beside the forward's grouping / contraction kernels
_SIDE_STREAMS = {}


def _side_stream(dev):
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st


class _ListHead:
    """First half of the inverse neighbour lists of idx [b,p,nn] (csrc/inv_lists.hip): per cloud the
    referenced support rows (longest list first), their counts and offsets -- all on the device -- plus an
    ASYNCHRONOUS copy of the numbers the launch decisions need on the host (the largest number of referenced rows of a
    cloud; whether any cloud carries non-identity relative rotations; whether any cloud cannot take the dense product).
    Built in the forward, read in the backward: by then the copy has long landed, so nothing stalls."""

    def __init__(self, idx, n_sup, nonident, gx=None, prefill=False, dense_probe=None):
        """prefill (with gx): also the second half (csrc/inv_lists.hip fill: the entries of every referenced row) right away,
        for all rows -- the launch needs no host value that way.  Everything runs on a SIDE stream (LISTS_ON_SIDE_STREAM): these
        are small latency-bound kernels (0.3 + 0.45 ms per layer) that nothing in the forward waits for, and the grouping and
        contraction kernels that follow on the main stream leave them room; the backward waits for `event`.
        dense_probe = (rotation blocks or None, ...): also the membership bits of the dense product (csrc/so3_dense.hip) and
        whether every cloud can take it (no list names a row twice, at most 512 referenced rows, every given pose rotation
        EXACTLY the identity -- the dense operand has no rotation in it)."""
        dev = idx.device
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if LISTS_ON_SIDE_STREAM else main
        self.memb = None
        if side is not main:
            side.wait_stream(main)                        # idx / gx / nonident were produced on the main stream
            for t in (idx, gx, nonident) + tuple(x[1] if isinstance(x, tuple) else x for x in (dense_probe or ())):
                if t is not None:
                    t.record_stream(side)                 # (read on the side stream: the allocator must not recycle them under it)
        with torch.cuda.stream(side):
            self.rows, self.off, self.cnt, self.n_rows = _hip.inv_lists_rows(idx, n_sup)
            flag = nonident.max() if nonident is not None else torch.ones((), dtype=torch.int32, device=dev)
            dense_bad = torch.ones((), dtype=torch.int32, device=dev)
            if dense_probe is not None:
                self.memb, dflags = _hip.so3_dense_member(idx, self.rows, self.n_rows, n_sup)
                dense_bad = dflags.max()
                eye = torch.eye(3, dtype=torch.float32, device=dev)
                for rot in dense_probe:
                    if isinstance(rot, tuple):
                        # ('orthonormal', [b,3,3] one rotation per cloud): the plain product stands for R_rel = R R^T = I, which holds
                        # for an orthonormal block only -- a predicted, not-quite-orthonormal 3x3 must stay on the list kernels
                        # (they form R_p R_n^T as the reference does, so3conv/functional.py:L1112-1120)
                        r = rot[1].double()
                        dense_bad = dense_bad + ((torch.matmul(r, r.transpose(-1, -2)) - eye.double()).abs().max() > 1e-6).to(torch.int32)
                    elif rot is not None:
                        dense_bad = dense_bad + (rot[:, :, :3, :3] != eye).any().to(torch.int32)
            # how many 16-row groups of the referenced rows a point's list touches, averaged per cloud, the largest cloud average x 16:
            # what the dense product with listed k-steps executes is proportional to it (csrc/so3_dense.hip dense_keys_kernel)
            groups16 = torch.zeros((), dtype=torch.int32, device=dev)
            if self.memb is not None:
                touched = ((self.memb & 0xffff) != 0).sum((1, 2)) + ((self.memb & -65536) != 0).sum((1, 2))       # [b]
                groups16 = (touched.max().to(torch.float32) * (16.0 / max(idx.shape[1], 1))).to(torch.int32)
            stats = torch.stack([self.n_rows.max().to(torch.int32), flag.to(torch.int32), dense_bad.to(torch.int32), groups16])
            self.host = torch.empty(4, dtype=torch.int32, pin_memory=True)
            self.host.copy_(stats, non_blocking=True)
            self.entries = _hip.inv_lists_fill(idx, gx, self.rows, self.off, n_sup) if (prefill and gx is not None) else None
            self.event = torch.cuda.Event()
            self.event.record(side)
        if side is not main:                              # allocated under the side stream, used (and freed) under the main one
            for t in (self.rows, self.off, self.cnt, self.n_rows, self.memb) + (self.entries or ()):
                if t is not None:
                    t.record_stream(main)

    def fill(self, idx, gx, n_sup):
        """the second half of the lists after all (a forward that probed for the dense product and did not take it)"""
        dev = idx.device
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if LISTS_ON_SIDE_STREAM else main
        with torch.cuda.stream(side):
            self.entries = _hip.inv_lists_fill(idx, gx, self.rows, self.off, n_sup)
            self.event = torch.cuda.Event()
            self.event.record(side)
        if side is not main:
            for t in self.entries:
                t.record_stream(main)

    def wait(self):
        """the current stream waits for the lists (device side; no host stall)"""
        torch.cuda.current_stream(self.rows.device).wait_event(self.event)

    def decide(self):
        """-> (rcap, any_nonident) as Python values."""
        self.event.synchronize()
        rcap, flag, _, _ = self.host.tolist()
        return int(rcap), bool(flag)

    def groups_touched(self):
        """mean number of 16-row groups a point's list touches (the largest per-cloud mean); blocks on the host like decide()"""
        self.event.synchronize()
        return int(self.host[3]) / 16.0

    def dense_possible(self):
        """every cloud can take the dense product (see __init__); blocks on the host like decide()"""
        self.event.synchronize()
        return self.memb is not None and int(self.host[2]) == 0


def _inverse_lists(idx, gx, n_sup, ident, nonident=None):
    """Inverse neighbour lists of idx [b,p,nn] (which (point, slot) pairs reference each support row), the
    referenced rows compacted and ordered longest list first; entries of a row in (p, slot) order.
    Convenience form for tests and tools (blocks on the host for the row count); the fused conv uses
    _ListHead + _hip.inv_lists_fill without blocking.
    -> rows, off, cnt int32 [b,rcap]; ent_p int32 [b,p*nn]; ent_gx [b,p*nn,4]; rcap; all_ident."""
    if nonident is None:
        nonident = (gx[..., 3].contiguous().view(torch.int32) != ident).flatten(1).any(1).to(torch.int32)
    head = _ListHead(idx, n_sup, nonident)
    rcap, any_nonident = head.decide()
    head.wait()
    ent_p, ent_gx = _hip.inv_lists_fill(idx, gx, head.rows, head.off, rcap)
    return (head.rows[:, :rcap].contiguous(), head.off[:, :rcap].contiguous(), head.cnt[:, :rcap].contiguous(),
            ent_p, ent_gx, rcap, not any_nonident)


# Layout of the fused conv's intermediate X (where the kernels allow it, else the reference layout); the contraction
# is always a hand-written GEMM:
#   'transposed'  X as the plain [P*A, C*K] matrix, both GEMM operands k-contiguous: csrc/gemm_dma_f32.hip
#                 (141 TFLOP/s on the deepest layer; hipBLASLt on the same operands: 150 -- tools/gemm_only.py times both)
#   'blocked'     X blocked by anchor quads, contraction = csrc/gemm_f32.hip (eap_gemm_f32_xb)
#   'reference'   X [C*K, P*A] as the reference's einsum writes it
X_LAYOUT = os.environ.get('EAP_X_LAYOUT', 'transposed')
BLOCKED_X = True     # test knob: False forces the reference layout


# The intermediate X [B, C*K, P*A] of the fused conv (24 GB at C = 128, B = 8) is scratch, not state: the re-associated
# backward never reads it, so it is produced and consumed X_CHUNK_CLOUDS clouds at a time and never saved.  Only the
# textbook backward (many referenced rows) needs it for dW = dY X^T; which regime a layer is in is known on the host
# only when its backward runs, so the layer's previous decision is the hint: after a 'dx' backward the next forward of
# the same weights keeps X, and a wrong guess costs one re-run of the grouping kernel in the backward.
X_CHUNK_CLOUDS = int(os.environ.get('EAP_X_CHUNK_CLOUDS', '8'))   # measured: 2 / 4 / 8 clouds per slab = 146.8 / 146.9 / 145.6 ms per step
_KEEP_X_HINT = {}           # id(W) -> (weakref(W), bool)


def _keep_x_hint(W):
    hit = _KEEP_X_HINT.get(id(W))
    return bool(hit is not None and hit[0]() is W and hit[1])


def _set_keep_x_hint(W, keep):
    if len(_KEEP_X_HINT) > 1024:
        _KEEP_X_HINT.clear()
    _KEEP_X_HINT[id(W)] = (weakref.ref(W), bool(keep))


class FoldedEpilogue:
    """What an inference-mode BatchNorm2d + leaky_relu (+ skip sum) after a contraction amounts to: per output channel
    y = leaky_relu(scale * (W x) + shift, slope) (+ residual) (vgtk.so3conv.blocks.BatchNormLeakyReLU.folded).  A
    contraction that can apply it in its own epilogue (csrc/gemm_bf16x3.hip) sets `applied`; otherwise the caller runs
    the norm as a pass of its own."""

    def __init__(self, scale, shift, slope, residual=None):
        self.scale, self.shift, self.slope, self.residual = scale.contiguous(), shift.contiguous(), float(slope), residual
        self.applied = False
        self.inference = not torch.is_grad_enabled()      # made under torch.no_grad(): the only place it may be applied


class TrainEpilogue:
    """A TRAINING-mode BatchNorm2d + leaky_relu right behind an inter conv (vgtk.so3conv.blocks.conv_norm_act; `x = conv(x); feat =
    relu(norm(x.feats))`, SPConvNets/utils/base_so3poseconv.py:L205-222).  When the conv's forward runs the dense product it takes the
    normalisation into its own node and sets `applied`:
      forward   product -> Yt -> statistics pass over Yt -> the re-ordering pass writes y' = leaky(BatchNorm(y)) (no pass of the norm's own,
                the conv output y is never written);
      backward  one reduction pass over (dL/dy', y') -> the gradient behind the norm is formed WHILE it is split into the backward product's
                planes (never written either).  The pre-activation is recovered from y' (leaky_relu with a positive slope is invertible):
                the node keeps its output, not the conv output.
    Otherwise (list kernels, posed parts, ...) `applied` stays False and the caller runs the norm module as a pass of its own.
    A channel whose gamma is exactly 0 has a constant output: its input gradient is exactly 0 either way, its d gamma is reported as 0."""

    def __init__(self, norm):
        self.norm = norm
        self.applied = False
        self.inference = False
        self.residual = None

    def moments(self, s1, s2, pivot, count):
        """pivoted sums of the conv output per channel -> (scale, shift, slope) of the fused pass; updates the running statistics exactly as
        vgtk.so3conv.blocks._BNAct.forward does; leaves what the backward needs in self.saved"""
        from .blocks import batch_moments
        norm = self.norm
        mean, var, total = batch_moments(s1, s2, pivot, count, norm.sync)
        if norm.running_mean is not None:
            with torch.no_grad():
                m = norm.momentum
                norm.running_mean.mul_(1.0 - m).add_(mean.to(norm.running_mean.dtype), alpha=m)
                norm.running_var.mul_(1.0 - m).add_((var * (total / (total - 1))).to(norm.running_var.dtype), alpha=m)
        invstd = torch.rsqrt(var + norm.eps)
        gamma = norm.weight.detach().double()
        scale64 = gamma * invstd
        scale = scale64.float()
        shift = (norm.bias.detach().double() - mean * scale64).float()
        inv_gamma = torch.where(gamma == 0, torch.zeros_like(gamma), 1.0 / gamma).float()
        # (copies, not views of the parameters: the backward must see the values this forward normalised with)
        self.saved = (scale, norm.bias.detach().float().clone(), inv_gamma.contiguous(), total)
        return scale.contiguous(), shift.contiguous(), float(norm.negative_slope)


def _contract_into(W, x, y, layout, epilogue=None, b0=0, x_bound=None):
    """y[b,o,pa] = W . x for the intermediate in one of its three layouts (epilogue: see FoldedEpilogue; b0 = first
    cloud of this slab, for the residual; x_bound = (words [b, p], anchors per point, factor) bounding x per point, see
    _grouped_bound)."""
    b, c, ks, p, na = x.shape
    o = W.shape[0]
    if epilogue is not None and layout == 2:
        res = None if epilogue.residual is None else epilogue.residual[b0:b0 + b]
        if _hip.gemm_epilogue(1, o, p * na, c * ks, W, c * ks, x, c * ks, c * ks * p * na, y, p * na, o * p * na, b,
                              epilogue.scale, epilogue.shift, epilogue.slope, res, b_bound=x_bound):
            epilogue.applied = True
            return
        if b0 > 0 and epilogue.applied:
            raise RuntimeError('folded epilogue: the slabs of one contraction took different kernels')
    if layout == 2:                              # Y = W . (X^T)^T, both operands k-contiguous (csrc/gemm_dma_f32.hip)
        _hip.gemm(0, 1, o, p * na, c * ks, W, c * ks, 0, x, c * ks, c * ks * p * na, y, p * na, o * p * na, b, b_bound=x_bound)
    else:
        _hip.gemm(0, 0, o, p * na, c * ks, W, c * ks, 0, x, p * na, c * ks * p * na, y, p * na, o * p * na, b, b_blocked=layout == 1)


def _grouped_bound(feats, idx):
    """A bound on the grouped tensor per point without a pass over it: X[c,k,p,a] = sum over the neighbours of a feature times
    an interpolation weight in [0, 1] (so3conv/functional.py:L1112-1261), so |X[.,.,p,.]| <= sum_n max_{c,a} |feats[c,idx[p,n],a]|
    -- the two-plane contraction takes the scales of its operand's columns from it (vgtk/_hip.py SPLIT_PLANES).
    -> int32 [b, p] (float bit patterns) or None."""
    if not (_hip.SPLIT_BF16_CONTRACTION and _hip.SPLIT_PLANES == 2):
        return None
    return _hip.so3_grouped_bound(feats, idx)


BACKWARD_LOG = None      # a list while someone wants to know the backward regime of every inter conv (bench.py, tests)
FORWARD_LOG = None       # the same for the forward: {'channels', 'dense': the dense product ran, 'parts'}


# The dense product over the referenced rows (csrc/so3_dense.hip): 'auto' takes it when every cloud of the batch can (no pose
# rotation, no padded lists) and a cloud references few enough rows: it does rows / nsample times the flops of the list kernels on
# a pipe ~4.5 x as fast.  Both directions take it up to DENSE_ROW_FACTOR x nsample rows at any width it supports ('off' never; 'force'
# whenever the shapes are taken: tests).
DENSE_MODE = os.environ.get('EAP_DENSE', 'auto')
DENSE_ROW_FACTOR = 5.0
# ... and beyond that row count (up to the 512 row slots the membership words hold) while a point's list touches few enough 16-row groups: with
# the listed k-steps the product executes ~ groups_touched x 16 / nsample times the algorithmic flops (x 3 on the fp16 pipe at ~1.3 PFLOP/s)
# against the list kernels' 0.46 of the fp32 peak -- it wins below ~24 groups; 16 leaves margin for the per-block overheads
DENSE_MAX_GROUPS = float(os.environ.get('EAP_DENSE_MAX_GROUPS', '16'))
# the forward at widths that fill 128-row blocks only (the 64 -> 128 layer): with every k-step it tied with grouping + contraction
# (round 5: 9.0 against 9.1 ms); with the empty k-steps skipped (round 6) the product wins: 4.2 + 1.3 + 0.9 ms against 5.7 + 3.3, same run
# 129.2 against 128.0 clouds/s before the operand kernel, more after it
DENSE_FWD_NARROW = os.environ.get('EAP_DENSE_FWD_128', '1') != '0'
# a training-mode BatchNorm + leaky_relu behind a conv whose forward runs the dense product joins the conv's node (TrainEpilogue)
FUSE_CONV_NORM = os.environ.get('EAP_FUSE_CONV_NORM', '1') != '0'


def _dense_rows(rcap, n):
    """row slots of the dense product: whole groups of 16 (csrc/so3_dense.hip, dense index)"""
    return (rcap + 15) & ~15


def _dense_wanted(head, o, p, na, ks, nn, n):
    """-> (rp, forward too): rows per cloud of the dense product (0: not taken) and whether the forward takes it as well.
    Blocks on the host for the row count."""
    if DENSE_MODE == 'off' or head is None or head.memb is None:
        return 0, False
    rcap, _ = head.decide()
    rp = _dense_rows(rcap, n)
    if rp <= 0 or rp > n or not head.dense_possible() or not _hip.so3_dense_supported(p, na, ks, rp, o):
        return 0, False
    if DENSE_MODE == 'force':
        return rp, True
    if rp > DENSE_ROW_FACTOR * nn and not (rp <= _hip.DENSE_MAX_ROWS and head.groups_touched() <= DENSE_MAX_GROUPS):
        return 0, False
    return rp, (o % 256 == 0) or DENSE_FWD_NARROW


def _weight_grad_from_z(z, fc, b, c, o, ks, ra, ldz=None):
    """dW[o,(c,k)] = sum_{b,(r,a)} Z[b,o,k,(r,a)] Fc[b,c,(r,a)] for Z [b, o*ks, ra] (row pitch ldz >= ra) and the referenced feature
    rows Fc [b, c, ra] (any order of the (row, anchor) axis, the same in both)."""
    ldz = ra if ldz is None else ldz
    # (measured and dropped: 64 feature channels zero-padded to 128 rows to reach the split kernel -- 0.66 -> 0.53 ms per step, and the
    # kernel-against-kernel bar of tests/test_gpu_lists_and_modules.py::test_permuted_clouds_on_the_two_tile_kernel, 2e-6, went to 3.2e-6)
    if _hip.gemm_reduce_takes_split(c, o * ks, ra, fc, ra, c * ra, z, ldz, o * ks * ldz, o * ks):
        # the transposed product Fc_b Z_b^T [c, o*ks] has the tile shape the split-bf16 kernel takes (>= 128 rows,
        # >= 256 columns); Z_b Fc_b^T with its 64-128 columns would stay on the fp32 pipe
        dt = torch.empty(c, o * ks, dtype=torch.float32, device=z.device)
        _hip.gemm_reduce(0, 1, c, o * ks, ra, fc, ra, c * ra, z, ldz, o * ks * ldz, dt, o * ks, b)
        return dt.view(c, o, ks).permute(1, 0, 2).reshape(o, c * ks).contiguous()
    d = torch.empty(o * ks, c, dtype=torch.float32, device=z.device)    # sum_b Z_b Fc_b^T
    _hip.gemm_reduce(0, 1, o * ks, c, ra, z, ldz, o * ks * ldz, fc, ra, c * ra, d, c, b)
    return d.view(o, ks, c).permute(0, 2, 1).reshape(o, c * ks).contiguous()


# Posed clouds whose points carry ONE rotation per rigid part (articulated objects: every point takes the pose of its part): the
# reference rotates an entry's offset by R_rel = R_p R_r^T and permutes the neighbour's anchor axis by the group element nearest to
# R_rel (so3conv/functional.py:L1112-1160; csrc/so3_inter.hip so3_prep_kernel) -- both depend on (part of p, part of r) only.  The dense
# product then runs once per part over that part's query points: its k-side table takes R_rel^T per ROW (a row's part is fixed), the
# stored operand is built from the rows' features with their anchor axis permuted per row.  DENSE_MAX_PARTS bounds the launches.
DENSE_PARTS = os.environ.get('EAP_DENSE_PARTS', '1') != '0'
DENSE_MAX_PARTS = 6
_POSE_PARTS = {}            # (storage pointer, version, shape) of a pose tensor -> (weakref, _PoseParts or None)


class _PoseParts:
    """The points of every cloud grouped by their pose rotation (bit-equal 3x3 blocks): part slots 0 .. n-1 per cloud, largest part
    first; slot i of the batch is one launch over pts[i] int64 [b, p_i] (p_i = the largest part i of any cloud rounded to 32;
    entries past a cloud's own part size repeat a valid point and are marked in col_map[i] int32 [b, p_i] = -1)."""

    def __init__(self, labels, reps, sizes, p):
        dev = labels.device
        b = labels.shape[0]
        self.labels, self.reps, self.sizes, self.n = labels, reps, sizes, len(sizes[0])
        self.single = self.n == 1
        order = torch.argsort(labels, dim=1, stable=True)                   # points sorted by part
        self.pts, self.col_map, self.width = [], [], []
        start = [0] * b
        for i in range(self.n):
            width = (max(sizes[bi][i] for bi in range(b)) + 31) // 32 * 32
            st = torch.tensor(start, dtype=torch.int64, device=dev)[:, None]
            sz = torch.tensor([sizes[bi][i] for bi in range(b)], dtype=torch.int64, device=dev)[:, None]
            t = torch.arange(width, dtype=torch.int64, device=dev)[None, :]
            valid = t < sz
            pts = order.gather(1, (st + torch.where(valid, t, torch.zeros_like(t))).clamp(max=p - 1))
            self.pts.append(pts.contiguous())
            self.col_map.append(torch.where(valid, pts, torch.full_like(pts, -1)).to(torch.int32).contiguous())
            self.width.append(width)
            start = [start[bi] + sizes[bi][i] for bi in range(b)]


def _pose_parts(rot):
    """-> _PoseParts of pose [b,p,4,4], or None when a cloud has more than DENSE_MAX_PARTS distinct rotations.  One host read per pose
    tensor (remembered by tensor identity + version: the layers of a backbone share their poses)."""
    key = (rot.data_ptr(), rot._version, tuple(rot.shape))
    hit = _POSE_PARTS.get(key)
    if hit is not None and hit[0]() is rot:
        return hit[1]
    b, p = rot.shape[:2]
    dev = rot.device
    r9 = rot[:, :, :3, :3].reshape(b * p, 9).contiguous()
    bits = r9.view(torch.int32).to(torch.int64)
    mul = torch.tensor([0x9E3779B97F4A7C15 - (1 << 64), 0xC2B2AE3D27D4EB4F - (1 << 64), 0x165667B19E3779F9, 0x27D4EB2F165667C5, 0x85EBCA77C2B2AE63 - (1 << 64),
                        0x2545F4914F6CDD1D, 0x5851F42D4C957F2D, 0x14057B7EF767814F, 0x3C6EF372FE94F82B], dtype=torch.int64, device=dev)
    h = (bits * mul).sum(1)
    h = (h ^ (h >> 31)) & ((1 << 55) - 1)
    keys = h + (torch.arange(b, dtype=torch.int64, device=dev).repeat_interleave(p) << 55)
    uniq, inv, counts = torch.unique(keys, return_inverse=True, return_counts=True)     # (sorted: a cloud's parts are consecutive)
    result = None
    if uniq.numel() <= b * DENSE_MAX_PARTS:
        nu = uniq.numel()
        first = torch.full((nu,), b * p, dtype=torch.int64, device=dev).scatter_reduce(0, inv, torch.arange(b * p, dtype=torch.int64, device=dev), 'amin')
        rep = r9[first]                                                                     # [nu, 9]
        same = (rep[inv] == r9).all().to(torch.int64)                                       # (a hash collision would show here)
        host = torch.cat([same[None], uniq >> 55, counts]).tolist()
        ok, cloud, cnt = host[0], host[1:1 + nu], host[1 + nu:]
        per = [[] for _ in range(b)]
        for u in range(nu):
            per[cloud[u]].append((cnt[u], u))
        n = max(len(x) for x in per)
        if ok and n <= DENSE_MAX_PARTS and min(len(x) for x in per) >= 1:
            slot = [0] * nu
            sizes = [[0] * n for _ in range(b)]
            rep_idx = [[per[bi][0][1]] * n for bi in range(b)]
            for bi in range(b):
                for i, (c_, u) in enumerate(sorted(per[bi], key=lambda x: (-x[0], x[1]))):
                    slot[u], sizes[bi][i], rep_idx[bi][i] = i, c_, u
            labels = torch.tensor(slot, dtype=torch.int64, device=dev)[inv].view(b, p)
            reps = rep[torch.tensor(rep_idx, dtype=torch.int64, device=dev)].view(b, n, 3, 3)
            result = _PoseParts(labels, reps, sizes, p)
    if len(_POSE_PARTS) > 16:
        _POSE_PARTS.clear()
    _POSE_PARTS[key] = (weakref.ref(rot), result)
    return result


class _PartsDense:
    """Per part slot of a _PoseParts: the dense product's geometry over that part's query points (row rotations in its k-side table)
    and the per-row anchor permutation of the stored operand."""

    def __init__(self, parts, xyz, memb, rows, rp, rk, sigma, nn, n_rows, mult, anchors):
        b, p = parts.labels.shape
        dev = xyz.device
        self.parts, self.rp, self.ks, self.na = parts, int(rp), rk.shape[1], rk.shape[0]
        ld = rows.stride(0)
        rows_all = torch.as_strided(rows, (b, ld), (ld, 1))
        row_part = parts.labels.gather(1, rows_all.clamp(min=0, max=p - 1).long())                     # [b, ld] (empty slots: anything)
        R_row = parts.reps.double().gather(1, row_part[:, :, None, None].expand(b, ld, 3, 3))        # R_j of every row slot
        self.geo, self.perm = [], []
        A = anchors.double() if anchors is not None else None
        for i in range(parts.n):
            R_i = parts.reps[:, i].double()                                                           # [b,3,3]
            M = torch.matmul(R_row, R_i[:, None].transpose(-1, -2))                                   # R_rel^T = R_j R_i^T   [b,ld,3,3]
            pts = parts.pts[i]
            q_xyz = xyz.gather(2, pts[:, None, :].expand(b, 3, pts.shape[1])).contiguous()
            valid = (parts.col_map[i] >= 0)
            memb_i = (memb.gather(1, pts[:, :, None].expand(b, pts.shape[1], memb.shape[2])) * valid[:, :, None].to(memb.dtype)).contiguous()
            self.geo.append(_hip.DenseGeometry(q_xyz, xyz, memb_i, rows, rp, rk, sigma, nn, n_rows, row_rot=M.float().contiguous()))
            if mult is None:
                self.perm.append(None)
            else:
                # the group element nearest to R_rel = M^T: argmax_g tr(R_rel A_g) (so3_prep_kernel; first maximum)
                t = torch.einsum('brji,gji->brg', M[:, :rp], A)
                ridx = t.float().argmax(dim=2)                                                        # [b,rp]
                self.perm.append(mult.long()[ridx])                                                   # [b,rp,na]: F_i[.., r, a] = F[.., r, perm[r, a]]


def _dense_g(fc4, W, geo):
    """G = W F over the referenced rows as the dense forward's stored operand: fc4 [b,c,rp,na] -> (g [b,o,ks,ld], ld or None, None), or
    (None, None, (scale, planes)) when one kernel made the operand directly (vgtk._hip.so3_dense_gplanes).  The
    small GEMM runs on the split-operand kernels where its shapes allow: its columns (row, anchor) padded to whole 128-column
    tiles (the padding of F is zero, G's padded columns are skipped by the split that follows)."""
    b, c, rp, na = fc4.shape
    o, ks = W.shape[0], geo.ks
    ra = rp * na
    W3 = W.view(o, c, ks).permute(0, 2, 1).reshape(o * ks, c).contiguous()
    operand = _hip.so3_dense_gplanes(fc4, W3, geo)        # (round 6: product and planes in one kernel where the shape is taken)
    if operand is not None:
        return None, None, operand
    ld = _hip.dense_pitch(ra) if (c % 16 == 0 and c >= 16 and (o * ks) % 128 == 0) else ra
    if ld == ra:
        fc = fc4.reshape(b, c, ra)                                                  # [b,c,(r,a)]; empty slots: zeros
    else:
        fc = torch.zeros(b, c, ld, dtype=torch.float32, device=fc4.device)
        fc[:, :, :ra] = fc4.reshape(b, c, ra)
    g = torch.empty(b, o * ks, ld, dtype=torch.float32, device=fc4.device)
    _hip.gemm(0, 0, o * ks, ld, c, W3, c, 0, fc, ld, c * ld, g, ld, o * ks * ld, b)
    return g.view(b, o, ks, ld), (None if ld == ra else ld), None


def _dense_forward_parts(feats, W, rows, pd, p):
    """the dense forward of posed clouds, one launch per part slot (see _PartsDense) -> y [b,o,p,a]"""
    b, c, n, na = feats.shape
    o = W.shape[0]
    fc0 = _hip.rows_gather(feats, rows, pd.rp)                                      # [b,c,rp,na]
    y = torch.empty(b, o, p, na, dtype=torch.float32, device=feats.device)
    for i, geo in enumerate(pd.geo):
        perm = pd.perm[i]
        fc4 = fc0 if perm is None else fc0.gather(3, perm[:, None].expand(b, c, pd.rp, na))
        g, ldg, operand = _dense_g(fc4, W, geo)
        _hip.so3_dense_fwd(g, geo, pd.parts.width[i], c, ldg=ldg, out=y, col_map=pd.parts.col_map[i], operand=operand, o=o)
    return y


def _dense_forward(feats, W, rows, geo, p):
    """y [b,o,p,a] = sum_(k,r) G[o,(k,r),a] Wd[p,(k,r),a] with G = W F over the referenced rows (csrc/so3_dense.hip)."""
    g, ldg, operand = _dense_g(_hip.rows_gather(feats, rows, geo.rp), W, geo)
    return _hip.so3_dense_fwd(g, geo, p, feats.shape[1], ldg=ldg, operand=operand, o=W.shape[0])


class _InterConv(torch.autograd.Function):
    """Fused inter conv  y = W . group(feats)  (functional.py:L1221-1261 + modules.py:L48-55)
    with the re-associated feature gradient (csrc/so3_inter_inv.hip)."""

    @staticmethod
    def forward(ctx, feats, W_param, idx, gx, rk, mult, sigma, ident, nonident=None, anchors=None, epilogue=None, grad_mode=True, geometry=None,
                bn_weight=None, bn_bias=None):
        # (bn_weight, bn_bias: the parameters of a TrainEpilogue's norm -- inputs of the node so that it can hand back their gradients)
        feats = feats.contiguous()
        W = W_param.contiguous()
        ctx.anchors = anchors.detach().contiguous() if anchors is not None else None   # the rotations `mult` was built from
        # X is internal to this Function: where the kernels allow it, it is kept blocked by anchor
        # quads ([b,p,a/4,c,k,4]) -- coalesced row-end stores in the grouping kernel -- and the GEMMs
        # read it as a blocked B operand (include/eap_hip.h, "blocked intermediate")
        can = BLOCKED_X and X_LAYOUT != 'reference' and _hip.so3_inter_group_fwd_can_block(
            feats.shape[1], feats.shape[2], feats.shape[3], rk.shape[1], mult is not None, nonident is not None)
        layout = 0 if not can else (2 if X_LAYOUT == 'transposed' else 1)
        b, c, n, na = feats.shape
        p, ks, o = idx.shape[1], rk.shape[1], W.shape[0]
        # grad_mode = torch.is_grad_enabled() AT THE CALL (inside forward() autograd always has it off; and under torch.no_grad()
        # needs_input_grad still reports the parameters although nothing will be differentiated)
        needs_grad = (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) and grad_mode
        train_ep = epilogue if isinstance(epilogue, TrainEpilogue) else None
        if train_ep is not None:
            epilogue = None                                # (the list kernels know nothing of it: `applied` stays False there)
        ctx.bn = None
        if epilogue is not None and (needs_grad or not epilogue.inference):
            raise RuntimeError('a folded epilogue is an inference-time fusion: build and use it under torch.no_grad()')
        lists_ok = BACKWARD_MODE != 'dx' and _inv_lists_supported(idx, n, na, ks)
        keep = needs_grad and (not lists_ok or _keep_x_hint(W_param))
        # inverse neighbour lists (device only; the host-side numbers arrive asynchronously), started before the grouping so that
        # they run beside it; the entries too when the previous backward of this layer took the lists
        # geometry = (q_xyz, xyz, q_rot, rot): what the dense product over the referenced rows is built from (csrc/so3_dense.hip);
        # whether the batch can take it is known on the host once the lists' first half has run -- one host wait per layer,
        # only for layers whose width fills the dense kernel's blocks
        probe = parts = None
        # (a folded inference epilogue does not stop it: the dense forward's re-ordering pass applies it; posed parts leave
        # `epilogue.applied` False and the caller runs the norm as a pass of its own)
        # (without gradients only the forward can use the product, and 'auto' takes it at o % 256 == 0 only: no probe -- and no host wait --
        # for an inference call it could not change)
        if (DENSE_MODE != 'off' and geometry is not None and lists_ok and (epilogue is None or o % 256 == 0 or DENSE_FWD_NARROW)
                and (needs_grad or o % 256 == 0 or DENSE_FWD_NARROW or DENSE_MODE == 'force')
                and _hip.so3_dense_supported(p, na, ks, 16, o)):
            probe = (geometry[2], geometry[3])
            if (DENSE_PARTS and geometry[3] is not None and geometry[0] is geometry[1] and geometry[2] is geometry[3] and p == n):
                parts = _pose_parts(geometry[3])           # (one host read per pose tensor)
                if parts is not None:
                    probe = ()                             # the rotations are accounted for per part: no "exactly the identity" requirement
                    if parts.single:
                        # one rotation per cloud: every relative rotation R R^T is the identity -- the plain product -- PROVIDED the
                        # block is orthonormal (checked on the device with the other conditions)
                        probe = (('orthonormal', parts.reps[:, 0].contiguous()),)
                        parts = None
        head = None
        if (lists_ok and needs_grad) or probe is not None:
            head = _ListHead(idx, n, nonident, gx, prefill=(not keep) and probe is None, dense_probe=probe)
        rp, dense_fwd = _dense_wanted(head, o, p, na, ks, idx.shape[2], n) if probe is not None else (0, False)
        ctx.dense = None
        ctx.parts = parts if rp > 0 else None
        if rp > 0 and needs_grad:                         # (built by whoever needs it first: the forward below, or the backward)
            ctx.dense = [None, (geometry[0], geometry[1], head.memb, head.rows, rp, rk, sigma, idx.shape[2], head.n_rows)]
        if FORWARD_LOG is not None:
            FORWARD_LOG.append({'channels': (c, o), 'dense': bool(rp > 0 and dense_fwd), 'parts': None if parts is None else parts.n})
        if rp > 0 and dense_fwd:
            head.wait()
            if parts is None:
                geo = _hip.DenseGeometry(geometry[0], geometry[1], head.memb, head.rows, rp, rk, sigma, idx.shape[2], head.n_rows)
                # (a frozen conv under a trainable norm -- gradients wanted for the norm's parameters only -- keeps the separate module:
                # the node's backward is the conv's dense backward)
                bn_only = grad_mode and not needs_grad and len(ctx.needs_input_grad) > 14 and (ctx.needs_input_grad[13] or ctx.needs_input_grad[14])
                if train_ep is not None and FUSE_CONV_NORM and not bn_only:
                    g_, ldg, operand = _dense_g(_hip.rows_gather(feats, head.rows, geo.rp), W, geo)
                    y = _hip.so3_dense_fwd_bnact(g_, geo, p, c, ldg, train_ep.moments, operand=operand, o=o)
                    del g_, operand
                    train_ep.applied = True
                    ctx.bn = train_ep.saved + (float(train_ep.norm.negative_slope), bool(train_ep.norm.sync))
                elif epilogue is not None and epilogue.residual is None:
                    # an inference-mode norm folded into one per-channel map: applied by the re-ordering pass too
                    g_, ldg, operand = _dense_g(_hip.rows_gather(feats, head.rows, geo.rp), W, geo)
                    y = _hip.so3_dense_fwd_bnact(g_, geo, p, c, ldg, None, operand=operand, o=o, affine=(epilogue.scale, epilogue.shift, epilogue.slope))
                    del g_, operand
                    epilogue.applied = True
                else:
                    y = _dense_forward(feats, W, head.rows, geo, p)
            else:
                geo = _PartsDense(parts, geometry[1], head.memb, head.rows, rp, rk, sigma, idx.shape[2], head.n_rows, mult, ctx.anchors)
                y = _dense_forward_parts(feats, W, head.rows, geo, p)
            ctx.head = head if needs_grad else None
            if needs_grad:
                ctx.dense[0] = geo
            ctx.layout, ctx.kept_x = 0, False
            ctx.W_param = weakref.ref(W_param)
            ctx.save_for_backward(W, torch.empty(0), idx, gx, rk, mult if mult is not None else torch.empty(0),
                                  nonident if nonident is not None else torch.empty(0), feats, *((y,) if (ctx.bn is not None and needs_grad) else ()))
            ctx.has_mult = mult is not None
            ctx.has_flag = nonident is not None
            ctx.sigma, ctx.ident, ctx.n = sigma, ident, feats.shape[2]
            return y
        if head is not None and not (lists_ok and needs_grad):
            head = None
        elif head is not None and probe is not None and not keep and rp == 0:
            head.fill(idx, gx, n)                          # (the probe postponed it; a dense backward does not need the entries)
        ctx.head = head
        y = torch.empty(b, o, p, na, dtype=torch.float32, device=feats.device)
        coset = _coset_tables(mult, ident) if (mult is not None and nonident is not None and layout == 2 and COSET_OPERAND) else None
        x_bound = _grouped_bound(feats, idx) if layout == 2 else None          # [b, p] words
        if keep:
            x = _hip.so3_inter_group_fwd(feats, idx, gx, rk, mult, sigma, nonident, blocked=layout, coset=coset)   # [b,c,k,p,a] (nominal shape)
            _contract_into(W, x, y.view(b, o, p * na), layout, x_bound=None if x_bound is None else (x_bound, na, 1.0))
        else:
            x = None
            step = max(1, X_CHUNK_CLOUDS)
            # X is scratch between the grouping and the contraction here: its columns may be in the order the grouping kernel's
            # lanes hold them (1 KB store runs, csrc/so3_inter_lists2.hip LAYOUT 4) with W's columns permuted to match
            tp = (STORE_ORDER_COLUMNS and layout == 2 and _hip.so3_group_fwd_tp_takes(c, na, ks)
                  and (mult is None or (coset is not None and _hip.so3_group_perm_lists2_takes(c, na, ks, n))))
            Wc = W.index_select(1, _hip.so3_group_fwd_tp_columns(c, ks, W.device)) if tp else W
            for b0 in range(0, b, step):
                b1 = min(b, b0 + step)
                xs = _hip.so3_inter_group_fwd(feats[b0:b1], idx[b0:b1], gx[b0:b1], rk, mult, sigma,
                                              None if nonident is None else nonident[b0:b1], blocked=layout, coset=coset, store_order=tp)
                _contract_into(Wc, xs, y[b0:b1].view(b1 - b0, o, p * na), layout, epilogue, b0,
                               x_bound=None if x_bound is None else (x_bound[b0:b1], na, 1.0))
                del xs
        ctx.layout = layout
        ctx.kept_x = x is not None
        ctx.W_param = weakref.ref(W_param)
        # feats: needed by the re-associated weight gradient (saved, not copied: autograd's version check
        # then catches an in-place update of the previous block's output)
        ctx.save_for_backward(W, x if x is not None else torch.empty(0), idx, gx, rk, mult if mult is not None else torch.empty(0),
                              nonident if nonident is not None else torch.empty(0), feats)
        ctx.has_mult = mult is not None
        ctx.has_flag = nonident is not None
        ctx.sigma, ctx.ident, ctx.n = sigma, ident, feats.shape[2]
        return y

    @staticmethod
    def backward(ctx, gy):
        W, x, idx, gx, rk, mult, nonident, feats = ctx.saved_tensors[:8]
        yact = ctx.saved_tensors[8] if ctx.bn is not None else None
        mult = mult if ctx.has_mult else None
        nonident = nonident if ctx.has_flag else None
        gy = gy.contiguous()
        b, c, n, na = feats.shape
        p, ks = idx.shape[1], rk.shape[1]
        o, ck, pa = W.shape[0], c * ks, p * na
        gW = gF = None
        # Strategy: when few support rows are referenced (the reference's first-nsample-in-index-
        # order ball query with large radii), BOTH gradients follow from
        #     Z[o,k,q,a'] = sum_{(p,n)->q} dY[o,p,a] w(p,a,k,n)          (csrc/so3_inter_inv.hip)
        #     dF[c,q,a'] = sum_{o,k} W[o,(c,k)] Z[o,k,q,a']       dW[o,(c,k)] = sum_{q,a'} Z[o,k,q,a'] F[c,q,a']
        # two small GEMMs over the referenced rows only -- no dX = W^T dY, no scatter, and the
        # [O x P*A] x [P*A x C*K] weight-gradient GEMM shrinks by P / (referenced rows).
        head, rcap, any_nonident = ctx.head, 0, True
        if ctx.dense is not None:
            if ctx.dense[0] is None:                       # list-kernel forward, dense backward
                head.wait()
                if ctx.parts is None:
                    ctx.dense[0] = _hip.DenseGeometry(*ctx.dense[1])
                else:
                    a_ = ctx.dense[1]
                    ctx.dense[0] = _PartsDense(ctx.parts, a_[1], a_[2], a_[3], a_[4], a_[5], a_[6], a_[7], a_[8], mult, ctx.anchors)
            if ctx.parts is not None:
                return _InterConv._backward_parts(ctx, gy, W, feats, head, ctx.dense[0]) + (None,) * 13
            geo = ctx.dense[0]
            if BACKWARD_LOG is not None:
                BACKWARD_LOG.append({'channels': (c, o), 'support_rows': n, 'referenced_rows_max': int(geo.rp), 'regime': 'dense rows',
                                     **({'norm': 'in the node'} if ctx.bn is not None else {})})
            rp = geo.rp
            ra = na * rp
            # Z's rows padded to whole 128-column tiles where that puts the feature-gradient GEMM on the split-operand kernels
            # (the padding is never written: garbage columns of gFc nobody reads; the weight gradient contracts over ra columns)
            # (c = 64: W2 padded to 128 zero-extended rows for the same reason -- the 64-row product ran on the fp32 pipe: 0.92 ms)
            pad_c = 128 if (c == 64 and ctx.bn is not None) else c
            ldz = _hip.dense_pitch(ra) if (pad_c % 128 == 0 and (o * ks) % 16 == 0 and ctx.needs_input_grad[0]) else ra
            g_bn_w = g_bn_b = None
            z_bound = None
            if ctx.bn is not None:
                # the BatchNorm + leaky_relu backward of the node (csrc/bn_act.hip header): one reduction pass over (dL/dy', y'), then gx is
                # formed inside the split of the product's stored operand
                k1, beta, inv_gamma, count, slope, sync = ctx.bn
                from .blocks import all_reduce_sums
                sg, sgx, gmax, xmax = _hip.bn_act_bwd_reduce_fromy(gy, yact, beta, inv_gamma, slope)
                g_bn_w, g_bn_b = sgx.float(), sg.float()
                tg, tgx = all_reduce_sums(sg, sgx, sync=sync)                        # whole-batch means (SyncBatchNorm backward)
                k2 = (k1.double() * tg / count).float()
                k3 = (k1.double() * tgx / count).float()
                coef = torch.stack([k1, k2, k3, beta, inv_gamma]).contiguous()      # [5, o]
                bound = (k1.abs()[None, :, None] * gmax + k2.abs()[None, :, None] + k3.abs()[None, :, None] * xmax).contiguous()
                z = _hip.so3_dense_bwd_bn(gy, yact, geo, ldz, coef, bound, slope)
                if ldz % 4 == 0 and rp % 4 == 0:
                    # a bound on Z's columns for the feature-gradient GEMM below (it then runs with two fp16 planes per operand instead of
                    # three bf16 planes, without a pass over Z): |Z[o,k,(a,r)]| = |sum_p gx[o,p,a] w| <= max_o bound[o,a] x (points listing row r)
                    live = torch.arange(rp, device=gy.device)[None, :] < head.n_rows[:, None]                      # (slots past a cloud's rows: zero columns)
                    cntf = torch.where(live, head.cnt[:, :rp], torch.zeros_like(head.cnt[:, :rp])).clamp(min=0, max=p).to(torch.float32)
                    cols = bound.amax(1)[:, :, None] * cntf[:, None, :]                                           # [b,na,rp]
                    words = torch.zeros(b, ldz // 4, dtype=torch.float32, device=gy.device)
                    words[:, :ra // 4] = cols.view(b, na, rp // 4, 4).amax(3).view(b, ra // 4)
                    z_bound = (words.view(torch.int32), 4, 1.0)
            else:
                z = _hip.so3_dense_bwd(gy, geo, ldz)                                 # [b,o,ks,ldz] rows = [na,rp]: the lists' Z, anchor axis in front
            if ctx.needs_input_grad[0]:
                W2 = W.view(o, c, ks).permute(1, 0, 2).reshape(c, o * ks).contiguous()
                if pad_c != c:
                    W2 = torch.cat([W2, torch.zeros(pad_c - c, o * ks, dtype=torch.float32, device=gy.device)])
                gFc = torch.empty(b, pad_c, ldz, dtype=torch.float32, device=gy.device)
                _hip.gemm(0, 0, pad_c, ldz, o * ks, W2, o * ks, 0, z, ldz, o * ks * ldz, gFc, ldz, pad_c * ldz, b, b_bound=z_bound)
                gF = _hip.rows_scatter(gFc[:, :c, :ra].reshape(b, c, na, rp).transpose(2, 3).contiguous(), head.rows, n)
            if ctx.needs_input_grad[1]:
                fc = _hip.rows_gather(feats, head.rows, rp).transpose(2, 3).contiguous().view(b, c, ra)      # [b,c,(a,r)]
                gW = _weight_grad_from_z(z, fc, b, c, o, ks, ra, ldz)
            return gF, gW, None, None, None, None, None, None, None, None, None, None, None, g_bn_w, g_bn_b
        if head is not None:
            rcap, any_nonident = head.decide()
            if BACKWARD_MODE == 'auto' and rcap * INV_ROW_FRACTION > n:
                head.entries = None                        # (prefilled on last step's hint, not needed after all: 20 bytes per entry)
                head = None
        if BACKWARD_LOG is not None:      # diagnostics for bench.py / tests: which regime each layer's backward took
            BACKWARD_LOG.append({'channels': (c, o), 'support_rows': n, 'referenced_rows_max': int(rcap),
                                 'regime': 'inverse lists' if head is not None else 'textbook dX'})
        Wp = ctx.W_param()
        if Wp is not None:
            _set_keep_x_hint(Wp, head is None)            # the next forward of this layer keeps X iff this backward needed it
        if head is None and not ctx.kept_x:               # wrong guess (or the first step): one more run of the grouping kernel
            x = _hip.so3_inter_group_fwd(feats, idx, gx, rk, mult, ctx.sigma, nonident, blocked=ctx.layout,
                                         coset=_coset_tables(mult, ctx.ident) if (mult is not None and ctx.layout == 2 and COSET_OPERAND) else None)
        if head is not None:
            # slots past a cloud's last referenced row are empty (rows = -1): a multiple of 4 rows makes K = rcap * na of the
            # gradient GEMMs a multiple of 16.  (A multiple of 32 would put the dF GEMM on the split kernel -- measured: the 18 %
            # more rows at 136 referenced rows cost what the faster kernel gains.)
            rcap = min((rcap + 3) & ~3, n)
            head.wait()
            rows = head.rows[:, :rcap].contiguous()
            off, cnt = head.off[:, :rcap].contiguous(), head.cnt[:, :rcap].contiguous()
            ent_p, ent_gx = head.entries if head.entries is not None else _hip.inv_lists_fill(idx, gx, head.rows, head.off, rcap)
            multinv = _group_tables_inverse(mult) if (mult is not None and any_nonident) else None
            coset = _coset_tables(multinv, ctx.ident) if (multinv is not None and COSET_OPERAND) else None
            z_order = None
            if coset is not None and _hip.so3_group_perm_lists2_takes(o, na, ks, p):
                # permuted clouds on the two-tile kernel (csrc/so3_inter_lists2.hip, PERM): gy and Z with a coset-major anchor
                # axis, the per-entry words prepared once; the small tensors around the two GEMMs change their anchor order
                z_order, z_pos = coset[0], coset[2]
                ent_pc, ent_gx2 = _hip.so3_perm_entries(ent_p, ent_gx, coset[1], ctx.anchors, ctx.ident, na, p)
                z = _hip.so3_inter_group_inv_perm2(_hip.anchor_reorder(gy, z_order), rows, off, cnt, ent_pc, ent_gx2, rk, z_order,
                                                   ctx.sigma, idx.shape[2])
            else:
                z = _hip.so3_inter_group_inv(gy, rows, off, cnt, ent_p, ent_gx, rk, multinv, ctx.sigma, idx.shape[2],
                                             ctx.ident, ctx.anchors, coset)              # [b,o,ks,rcap,na]
            ra = rcap * na
            if ctx.needs_input_grad[0]:
                W2 = W.view(o, c, ks).permute(1, 0, 2).reshape(c, o * ks).contiguous()
                gFc = torch.empty(b, c, ra, dtype=torch.float32, device=gy.device)
                _hip.gemm(0, 0, c, ra, o * ks, W2, o * ks, 0, z, ra, o * ks * ra, gFc, ra, c * ra, b)
                if z_order is not None:
                    gFc = _hip.anchor_reorder(gFc.view(b, c, rcap, na), z_pos)
                gF = _hip.rows_scatter(gFc.view(b, c, rcap, na), rows, n)           # unreferenced rows: zero gradient
            if ctx.needs_input_grad[1]:
                fc = _hip.rows_gather(feats, rows, rcap)                             # [b,c,rcap,na]; unused slots: zeros
                if z_order is not None:
                    fc = _hip.anchor_reorder(fc, z_order)
                gW = _weight_grad_from_z(z, fc.view(b, c, ra), b, c, o, ks, ra)
        else:
            if ctx.needs_input_grad[1]:
                gW = torch.empty_like(W)          # sum_b gy_b x_b^T
                # (measured and dropped: every cloud's dY_b X^T_b on the split-operand 'nn' kernel instead of the batch-reducing fp32 kernel --
                # 512 x 3072 outputs over K = 30720 are 48 workgroups without a split of K: 15.5 -> 48 ms at 16 x 512 points)
                if ctx.layout == 2:     # X^T [pa, ck]: dW = dY X^T is a plain row-major product
                    _hip.gemm_reduce(0, 0, o, ck, pa, gy, pa, o * pa, x, ck, ck * pa, gW, ck, b)
                else:
                    _hip.gemm_reduce(0, 1, o, ck, pa, gy, pa, o * pa, x, pa, ck * pa, gW, ck, b, b_blocked=ctx.layout == 1)
            if ctx.needs_input_grad[0]:
                gx_ = torch.empty_like(x.view(b, ck, pa))      # W^T gy
                # (W^T written out -- a few MB: the product is then 'nn' with a shared A operand, which the split-operand kernels take;
                # with the transposition left to the GEMM it ran on the fp32 matrix pipe: 15.6 of the 61.6 ms step at 16 x 512 points)
                Wt = W.t().contiguous()
                _hip.gemm(0, 0, ck, pa, o, Wt, o, 0, gy.view(b, o, pa), pa, o * pa, gx_, pa, ck * pa, b)
                gF = _hip.so3_inter_group_bwd(gx_.view(b, c, ks, p, na), idx, gx, rk, mult, ctx.sigma, n, ctx.ident)
        return gF, gW, None, None, None, None, None, None, None, None, None, None, None, None, None


    @staticmethod
    def _backward_parts(ctx, gy, W, feats, head, pd):
        """the dense backward of posed clouds, one product per part slot (_PartsDense): Z_i over the part's query points, the two small
        GEMMs per part, the rows' anchor axis permuted back before the sum over the parts -> (gF, gW)"""
        b, c, n, na = feats.shape
        o, ks, rp = W.shape[0], pd.ks, pd.rp
        ra = na * rp
        if BACKWARD_LOG is not None:
            BACKWARD_LOG.append({'channels': (c, o), 'support_rows': n, 'referenced_rows_max': int(rp), 'regime': 'dense rows', 'parts': pd.parts.n})
        need_f, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        ldz = _hip.dense_pitch(ra) if (c % 128 == 0 and (o * ks) % 16 == 0 and need_f) else ra
        W2 = W.view(o, c, ks).permute(1, 0, 2).reshape(c, o * ks).contiguous() if need_f else None
        fc0 = _hip.rows_gather(feats, head.rows, rp) if need_w else None                        # [b,c,rp,na]
        gFr = gW = None
        # (the row maxima the BatchNorm backward left are those of ALL points: an upper bound for every part's columns)
        rowmax = _hip.take_rowmax_hint(gy)
        for i, geo in enumerate(pd.geo):
            perm = pd.perm[i]
            z = _hip.so3_dense_bwd(gy, geo, ldz, colmap=pd.parts.col_map[i], rowmax=rowmax)      # [b,o,ks,ldz] rows = [na,rp]
            pe = None if perm is None else perm[:, None].expand(b, c, rp, na)
            if need_f:
                gFc = torch.empty(b, c, ldz, dtype=torch.float32, device=gy.device)
                _hip.gemm(0, 0, c, ldz, o * ks, W2, o * ks, 0, z, ldz, o * ks * ldz, gFc, ldz, c * ldz, b)
                t = gFc[:, :, :ra].reshape(b, c, na, rp).transpose(2, 3)                          # [b,c,rp,na]: gradient of F_i[.., r, a] = F[.., r, perm[r, a]]
                if gFr is None:
                    gFr = torch.zeros(b, c, rp, na, dtype=torch.float32, device=gy.device)
                if pe is None:
                    gFr += t
                else:
                    gFr.scatter_add_(3, pe, t.contiguous())
            if need_w:
                fci = fc0 if pe is None else fc0.gather(3, pe)
                gw = _weight_grad_from_z(z, fci.transpose(2, 3).contiguous().view(b, c, ra), b, c, o, ks, ra, ldz)
                gW = gw if gW is None else gW + gw
            del z
        gF = _hip.rows_scatter(gFr, head.rows, n) if need_f else None
        return gF, gW


INTRA_DW_SLICE = 64      # channels whose 12-tap gather is materialised at a time for the intra weight gradient


class _IntraConv(torch.autograd.Function):
    """Intra SO(3) conv  y[b,o,p,a] = sum_{c,t} W[o, c*T + t] F[b,c,p,idx[a,t]]  (functional.py:L2553-2602 +
    modules.py:L48-55) WITHOUT the [B,C,T,P,A] gathered tensor (48 GB at C = 512, B = 8) in either direction:
      forward   implicit GEMM, the gather folded into the operand load (eap_so3_intra_conv_f32);
      dF        the same kernel: every column idx[:,t] is a permutation of the anchors, so
                dF[b,c,p,a'] = sum_{o,t} W[o, c*T + t] dY[b,o,p, inv_t(a')] -- weights regrouped to [C, O*T], gather
                table inv[a', t];
      dW        sum_{b,p,a} dY[b,o,p,a] F[b,c,p,idx[a,t]]: the gather is materialised for INTRA_DW_SLICE channels at
                a time (eap_so3_intra_group_fwd) and contracted by the k-split reduce GEMM (3 GB of scratch instead
                of 48)."""

    @staticmethod
    def forward(ctx, feats, W, idx32):
        feats, W = feats.contiguous(), W.contiguous()
        ctx.save_for_backward(feats, W, idx32)
        return _hip.so3_intra_conv(feats, W, idx32)

    @staticmethod
    def backward(ctx, gy):
        feats, W, idx32 = ctx.saved_tensors
        gy = gy.contiguous()
        b, c, p, na = feats.shape
        o, nt = W.shape[0], idx32.shape[1]
        gF = gW = None
        if ctx.needs_input_grad[0]:
            inv = torch.empty_like(idx32)                                         # inv[idx[a,t], t] = a
            inv.scatter_(0, idx32.long(), torch.arange(na, device=idx32.device, dtype=torch.int32)[:, None].expand(na, nt).contiguous())
            W2 = W.view(o, c, nt).permute(1, 0, 2).reshape(c, o * nt).contiguous()
            gF = _hip.so3_intra_conv(gy, W2, inv.contiguous())
        if ctx.needs_input_grad[1]:
            gW = torch.empty(o, c, nt, dtype=torch.float32, device=gy.device)
            pa = p * na
            for c0 in range(0, c, INTRA_DW_SLICE):
                c1 = min(c, c0 + INTRA_DW_SLICE)
                g = _hip.so3_intra_group_fwd(feats[:, c0:c1].contiguous(), idx32)       # [b, cs, nt, p, na]
                d = torch.empty(o, (c1 - c0) * nt, dtype=torch.float32, device=gy.device)
                _hip.gemm_reduce(0, 1, o, (c1 - c0) * nt, pa, gy, pa, o * pa, g, pa, (c1 - c0) * nt * pa, d, (c1 - c0) * nt, b)
                gW[:, c0:c1] = d.view(o, c1 - c0, nt)
            gW = gW.view(o, c * nt)
        return gF, gW, None


def intra_so3conv(feats, W, intra_idx):
    """feats [b,c,p,na], W [o, c*T], intra_idx [na,T] -> [b,o,p,na]; what IntraSO3Conv.forward runs."""
    if feats.dtype != torch.float32 or not feats.is_cuda:
        raise RuntimeError('intra_so3conv: float32 device tensors only')
    return _IntraConv.apply(feats, W, intra_idx.to(torch.int32).contiguous())


def _aligned16(t):
    """A contiguous tensor whose first element sits on a 16-byte boundary: `t` itself when it already does, else a copy (a
    contiguous VIEW into a larger storage -- a slice of autograd's gradient buffer, say -- can start at any 4-byte offset,
    and the streaming kernels move 16-byte words)."""
    t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone(memory_format=torch.contiguous_format)


class _NarrowContract(torch.autograd.Function):
    """_Contract for at most four output channels (the pose head's translation components, the attention logit): no matrix
    core has work for 1-4 rows, the op is a streaming pass over x in each direction (csrc/narrow_contract.hip)."""

    @staticmethod
    def forward(ctx, W, x):
        W, x = W.contiguous(), _aligned16(x)
        b, c, n = x.shape
        o = W.shape[0]
        y = torch.empty(b, o, n, dtype=torch.float32, device=x.device)
        _hip.call('eap_narrow_contract_fwd_f32', x, b, o, c, _hip._I64(n), _hip._ptr(W), _hip._ptr(x), _hip._ptr(y))
        ctx.save_for_backward(W, x)
        return y

    @staticmethod
    def backward(ctx, gy):
        W, x = ctx.saved_tensors
        gy = _aligned16(gy)
        b, c, n = x.shape
        o = W.shape[0]
        gW = gx = None
        if ctx.needs_input_grad[1]:
            gx = torch.empty_like(x)
            _hip.call('eap_narrow_contract_dx_f32', x, b, o, c, _hip._I64(n), _hip._ptr(W), _hip._ptr(gy), _hip._ptr(gx))
        if ctx.needs_input_grad[0]:
            slabs = int(_hip.lib.eap_narrow_contract_dw_slabs(_hip._I64(n)))
            part = torch.empty(slabs * b, o, c, dtype=torch.float32, device=x.device)
            _hip.call('eap_narrow_contract_dw_f32', x, b, o, c, _hip._I64(n), _hip._ptr(gy), _hip._ptr(x), _hip._ptr(part))
            gW = part.sum(0, dtype=torch.float64).float()
        return gW, gx


def so3_contract(W, x, epilogue=None):
    """W [O, C*K], x [b, C*K, P*A] -> [b, O, P*A] (epilogue: a FoldedEpilogue the product may apply, inference only)."""
    _hip.check_input(x)
    if x.dtype != torch.float32 or W.dtype != torch.float32:
        raise RuntimeError('so3_contract: float32 only')
    if not W.is_cuda:
        raise RuntimeError('so3_contract: W must be a device tensor')
    if epilogue is None and W.shape[0] <= 4 and _hip.lib.eap_narrow_contract_supported(x.shape[0], W.shape[0], x.shape[1], _hip._I64(x.shape[2])):
        return _NarrowContract.apply(W, x)
    return _Contract.apply(W, x, epilogue)


# ------------------------------------------------------------------------------------------------
# grouping entry points
# ------------------------------------------------------------------------------------------------
def _strided_centres(xyz, pose, stride, lazy_sample):
    """Centres of a strided conv (functional.py:L931-934 -> spconv/functional.py:L468-477): ceil(p / stride)
    points, furthest-point sampled (native FPS kernel) or, with lazy_sample, the first ones.
    -> sample_idx int32 [b,p2], sample_xyz [b,3,p2], sampled_pose [b,p2,4,4] or None."""
    n_sample = math.ceil(xyz.shape[2] / stride)
    sample_idx, sample_xyz = pctk.furthest_sample(xyz, n_sample, lazy_sample)
    sampled_pose = None if pose is None else batched_index_select(pose, 1, sample_idx.long()).contiguous()
    return sample_idx, sample_xyz.contiguous(), sampled_pose


def _inter_group(xyz, pose, feats, n_neighbor, anchors, kernels, radius, sigma, permute, q_xyz=None, q_pose=None):
    if feats.dtype != torch.float32 or xyz.dtype != torch.float32:
        raise RuntimeError('so3conv: float32 only')
    _hip.check_input(xyz)
    if not feats.is_cuda:
        raise RuntimeError('so3conv: feats must be a device tensor')
    q_xyz = xyz if q_xyz is None else q_xyz
    ball_idx = cuda_nn.ball_query(q_xyz, xyz, radius, n_neighbor)
    rk = rotated_kernels(anchors, kernels)
    mult = ident = None
    rot = q_rot = None
    if pose is not None:
        rot = pose.contiguous()
        q_rot = rot if q_pose is None else q_pose.contiguous()
        if rot.shape[-2:] != (4, 4) or rot.dtype != torch.float32:
            raise RuntimeError('so3conv: pose must be float32 [b,p,4,4]')
        if permute:
            mult, ident = _group_tables(anchors)
            if mult is None:
                raise NotImplementedError(
                    'anchor permutation with per-point poses needs a closed anchor set (kanchor 60 or 1)')
    gx, nonident = _hip.so3_prep(q_xyz, xyz, ball_idx, q_rot, rot, anchors.contiguous(), 0 if ident is None else ident)
    new_feats = _InterGroup.apply(feats, ball_idx, gx, rk, mult, float(sigma), 0 if ident is None else ident, nonident)
    inter_w = InterWeights(gx, rk, sigma)
    return ball_idx, (inter_w.materialize() if MATERIALIZE_INTER_W else inter_w), new_feats


def inter_so3conv_fused(xyz, pose, feats, W, n_neighbor, anchors, kernels, radius, sigma, permute, q_xyz=None, q_pose=None,
                        epilogue=None):
    """ball query + prep + fused (grouping . contraction) -> (ball_idx, InterWeights, y [b,o,p,a]).
    What InterSO3PoseConv / InterSO3Conv.forward run; q_xyz / q_pose = the sampled centres of a strided conv
    (default: every point is a centre).  epilogue: a FoldedEpilogue the contraction may apply (inference only)."""
    if feats.dtype != torch.float32 or xyz.dtype != torch.float32 or W.dtype != torch.float32:
        raise RuntimeError('so3conv: float32 only')
    _hip.check_input(xyz)
    if not feats.is_cuda or not W.is_cuda:
        raise RuntimeError('so3conv: feats and W must be device tensors')
    q_xyz = xyz if q_xyz is None else q_xyz
    ball_idx = cuda_nn.ball_query(q_xyz, xyz, radius, n_neighbor)
    rk = rotated_kernels(anchors, kernels)
    mult = ident = rot = q_rot = None
    if pose is not None:
        rot = pose.contiguous()
        q_rot = rot if q_pose is None else q_pose.contiguous()
        if rot.shape[-2:] != (4, 4) or rot.dtype != torch.float32:
            raise RuntimeError('so3conv: pose must be float32 [b,p,4,4]')
        if permute:
            mult, ident = _group_tables(anchors)
            if mult is None:
                raise NotImplementedError(
                    'anchor permutation with per-point poses needs a closed anchor set (kanchor 60 or 1)')
    gx, nonident = _hip.so3_prep(q_xyz, xyz, ball_idx, q_rot, rot, anchors.contiguous(), 0 if ident is None else ident)
    bn_w = bn_b = None
    if isinstance(epilogue, TrainEpilogue):
        bn_w, bn_b = epilogue.norm.weight, epilogue.norm.bias
    y = _InterConv.apply(feats, W, ball_idx, gx, rk, mult, float(sigma), 0 if ident is None else ident, nonident,
                         anchors if mult is not None else None, epilogue, torch.is_grad_enabled(),
                         (q_xyz.contiguous(), xyz.contiguous(), q_rot, rot), bn_w, bn_b)
    inter_w = InterWeights(gx, rk, sigma)
    return ball_idx, (inter_w.materialize() if MATERIALIZE_INTER_W else inter_w), y


def inter_so3conv_fused_art_mode(xyz, pose, feats, W, seg_labels, n_neighbor, anchors, kernels, radius, sigma, permute):
    """The stride-1 branch of inter_so3poseconv_grouping_strided_arti_mode (functional.py:L1420-1520) + the contraction:
    xyz [b, ns, 3, p] holds the cloud in ns articulation states; every state gets its own ball query, and point p takes the
    neighbour list and the (UNROTATED) offset vectors of state seg_labels[b, p]; the relative rotations of the poses only
    select the anchor permutation.  -> (InterWeights, y [b,o,p,a]).  Runs on the fused kernels: per-state ball query +
    offsets (eap_so3_prep_f32 without poses), the selection by label, the relative-rotation anchors from a second prep call
    on the selected lists, then the same autograd node as the regular conv."""
    if xyz.dim() != 4 or xyz.shape[2] != 3:
        raise RuntimeError('use_art_mode: xyz must be [b, n_states, 3, p]')
    if seg_labels is None:
        raise RuntimeError('use_art_mode: the per-point state labels `seg` are required')
    if feats.dtype != torch.float32 or xyz.dtype != torch.float32 or W.dtype != torch.float32:
        raise RuntimeError('so3conv: float32 only')
    _hip.check_input(xyz)
    b, ns, _, p = xyz.shape
    flat = xyz.reshape(b * ns, 3, p).contiguous()
    idx_all = cuda_nn.ball_query(flat, flat, radius, n_neighbor)                                     # [b*ns, p, nn]
    gx_all, _ = _hip.so3_prep(flat, flat, idx_all, None, None, anchors.contiguous(), 0)              # offsets x_n - x_p per state
    nn = idx_all.shape[2]
    pick = seg_labels.long().view(b, 1, p, 1)
    ball_idx = idx_all.view(b, ns, p, nn).gather(1, pick.expand(b, 1, p, nn)).squeeze(1).contiguous()
    gx = gx_all.view(b, ns, p, nn, 4).gather(1, pick.unsqueeze(-1).expand(b, 1, p, nn, 4)).squeeze(1).contiguous()
    rk = rotated_kernels(anchors, kernels)
    mult = nonident = None
    ident = 0
    if pose is not None and permute:
        rot = pose.contiguous()
        mult, ident = _group_tables(anchors)
        if mult is None:
            raise NotImplementedError('anchor permutation with per-point poses needs a closed anchor set (kanchor 60 or 1)')
        # the nearest anchor of every pair's relative rotation (4th word of gx) and the per-cloud "not all identity" flag; the
        # rotated offsets this call also produces are not used in this mode
        first = xyz[:, 0].contiguous()
        g_rot, nonident = _hip.so3_prep(first, first, ball_idx, rot, rot, anchors.contiguous(), ident)
        gx = torch.cat([gx[..., :3], g_rot[..., 3:]], dim=-1).contiguous()
    y = _InterConv.apply(feats, W, ball_idx, gx, rk, mult, float(sigma), ident, nonident, anchors if mult is not None else None, None, torch.is_grad_enabled())
    inter_w = InterWeights(gx, rk, sigma)
    return (inter_w.materialize() if MATERIALIZE_INTER_W else inter_w), y


# ------------------------------------------------------------------------------------------------
# the `use_2d` variant (functional.py:L1718-2130, modules.py:L249-255; scripts/train/eyeglasses.sh): an anchor axis of
# (na, 4) -- every anchor times four residual rotations about the y axis
# ------------------------------------------------------------------------------------------------
_RES_2D = {}
SLOW_2D_CHUNK = 64      # query points per slab of the general (really permuted) 2-D path
FORCE_GENERAL_2D = False    # test knob: the general path even when no residual index is permuted


def _res_rot_2d(device):
    r = _RES_2D.get(str(device))
    if r is None:
        r = _RES_2D[str(device)] = RES_ROT_2D.to(device)
    return r


def residual_rotation_index(pose, ball_idx, chunk=512):
    """rotated_anchor_idx int64 [b,p,nn,4] of the 2-D variant (functional.py:L1934-1938): per entry and residual rotation z the
    index j maximising tr(R_rel^T RES_z RES_j^T), R_rel = R_p R_n^T -- the reference's expression, evaluated on the device a slab
    of points at a time."""
    res = _res_rot_2d(pose.device)
    rot = pose[:, :, :3, :3].contiguous()
    b, p = rot.shape[:2]
    out = []
    for s0 in range(0, p, chunk):
        idx = ball_idx[:, s0:s0 + chunk].long()
        grouped = batched_index_select_other(rot, idx, dim=1)                                       # [b,pc,nn,3,3]
        rel = torch.matmul(rot[:, s0:s0 + chunk].unsqueeze(2), grouped.transpose(3, 4).contiguous())
        ra = torch.matmul(rel.transpose(-1, -2).contiguous().unsqueeze(3), res)                      # [b,pc,nn,4,3,3]
        d = torch.matmul(ra.unsqueeze(4), res.unsqueeze(0).transpose(2, 3).contiguous())            # [b,pc,nn,4,4,3,3]
        out.append(torch.argmax(d[..., 0, 0] + d[..., 1, 1] + d[..., 2, 2], dim=-1))
    return torch.cat(out, dim=1)


def _conv_2d(xyz, pose, feats, W, n_neighbor, anchors, kernels, radius, sigma, permute, contract=True):
    """Stride-1 branch of inter_so3poseconv_grouping_strided_2D (+ the contraction when `contract`).  feats [b,c,p,na*4].

    With the residual index unpermuted -- permute_modes == 0, or every entry's relative rotation closest to the identity among
    the four residual rotations (identity poses: what the shipped model feeds) -- output anchor (a, z) reads feature anchor (a, z)
    only, with the weights of rotation A_a RES_z: FOUR ordinary convolutions on the fused kernels, one per residual rotation,
    over the anchor sets {A_a RES_z}_a (the poses still rotate the offsets).  Otherwise the reference's expression in device
    tensor ops, a slab of SLOW_2D_CHUNK query points at a time (correct, not accelerated: the permuted 2-D case has no kernel)."""
    if feats.dtype != torch.float32 or xyz.dtype != torch.float32:
        raise RuntimeError('so3conv: float32 only')
    _hip.check_input(xyz)
    if not feats.is_cuda:
        raise RuntimeError('so3conv: feats must be a device tensor')
    b, c, p, na4 = feats.shape
    na = anchors.shape[0]
    if na4 != 4 * na:
        raise RuntimeError(f'use_2d: feats must carry {na} x 4 anchors, got {na4}')
    res = _res_rot_2d(feats.device)
    tot = torch.matmul(anchors.unsqueeze(1), res.unsqueeze(0)).contiguous()                          # [na,4,3,3] = A_a RES_z (L1921)
    ball_idx = cuda_nn.ball_query(xyz, xyz, radius, n_neighbor)
    shift = None
    if permute and pose is not None:
        shift = residual_rotation_index(pose, ball_idx)
        if not FORCE_GENERAL_2D and bool((shift == torch.arange(4, device=shift.device)).all()):
            shift = None
    f5 = feats.contiguous().view(b, c, p, na, 4)
    if shift is None:
        outs = []
        for z in range(4):
            fz = f5[..., z].contiguous()
            az = tot[:, z].contiguous()
            if contract:
                outs.append(inter_so3conv_fused(xyz, pose, fz, W, n_neighbor, az, kernels, radius, sigma, False)[2])
            else:
                outs.append(_inter_group(xyz, pose, fz, n_neighbor, az, kernels, radius, sigma, False)[2])
        y = torch.stack(outs, dim=-1)
        y = y.view(*y.shape[:-2], na4)
        rk = rotated_kernels(tot.view(na4, 3, 3), kernels)
        gx, _ = _hip.so3_prep(xyz, xyz, ball_idx, None if pose is None else pose.contiguous(), None if pose is None else pose.contiguous(),
                              anchors.contiguous(), 0)
        return ball_idx, InterWeights(gx, rk, sigma), y
    # the general case: really permuted residual indices
    rk = rotated_kernels(tot.view(na4, 3, 3), kernels)                                               # [na*4,ks,3]
    rot = pose.contiguous()
    gx, _ = _hip.so3_prep(xyz, xyz, ball_idx, rot, rot, anchors.contiguous(), 0)                      # rotated offsets (L1859-1862)
    ks = kernels.shape[0]
    fs = torch.cat([f5, torch.zeros(b, c, 1, na, 4, dtype=feats.dtype, device=feats.device)], dim=2)  # shadow row (L1944)
    trans = fs.permute(0, 2, 3, 4, 1)                                                                # [b,p+1,na,4,c]
    outs = []
    for s0 in range(0, p, SLOW_2D_CHUNK):
        s1 = min(p, s0 + SLOW_2D_CHUNK)
        w = _hip.so3_inter_weights(gx[:, s0:s1].contiguous(), rk, float(sigma)).view(b, s1 - s0, na, 4, ks, -1)     # [b,pc,na,4,ks,nn]
        idx = ball_idx[:, s0:s1].long()
        g = batched_index_select_other(trans, idx, dim=1)                                            # [b,pc,nn,na,4,c]
        sel = shift[:, s0:s1, :, None, :, None].expand(-1, -1, -1, na, -1, c)
        g = torch.gather(g, 4, sel)                                                                  # feature residual index per (entry, z) (L1954-1957)
        x = torch.einsum('bpnazc,bpazkn->bckpaz', g, w).reshape(b, c, ks, s1 - s0, na4)              # (L2124)
        outs.append(so3_contract(W, x.reshape(b, c * ks, (s1 - s0) * na4).contiguous()).view(b, W.shape[0], s1 - s0, na4) if contract else x)
    return ball_idx, InterWeights(gx, rk, sigma), torch.cat(outs, dim=2 if contract else 3)


def inter_so3conv_fused_2d(xyz, pose, feats, W, n_neighbor, anchors, kernels, radius, sigma, permute):
    """What InterSO3PoseConv(use_2d=True).forward runs for stride 1 -> (InterWeights, y [b,o,p,na*4])."""
    if W.dtype != torch.float32 or not W.is_cuda:
        raise RuntimeError('so3conv: W must be a float32 device tensor')
    _, w, y = _conv_2d(xyz, pose, feats, W, n_neighbor, anchors, kernels, radius, sigma, permute, contract=True)
    return (w.materialize() if MATERIALIZE_INTER_W else w), y


def inter_so3poseconv_grouping_strided_2D(xyz, pose, feats, stride, n_neighbor, anchors, kernels, radius, sigma, inter_idx=None,
                                          inter_w=None, lazy_sample=True, radius_expansion=1.0, pooling=None, permute_modes=0):
    """Grouping of the `use_2d` variant (functional.py:L1718-2130), stride-1 branch (L1812-2128: the neighbourhood is recomputed,
    inter_idx handed back unchanged).  -> inter_idx, inter_w, new_xyz, new_feats [b,c,ks,p,na*4], sample_idx, sampled_pose.
    The strided branch of the reference (L1754-1810) applies the 60-anchor expressions to the 240-anchor features and cannot run
    as written; the shipped configuration forces stride 1 (...pn_38_multi_stage.py:L2191)."""
    if pooling is not None and stride > 1 and feats.shape[1] > 1:
        raise ValueError('xyz_pooling is not None?!!')                 # functional.py:L1737
    if inter_idx is None and stride > 1:
        raise NotImplementedError('use_2d with stride > 1: the reference\'s strided branch (functional.py:L1754-1810) does not match its 240-anchor features')
    _, w, new_feats = _conv_2d(xyz, pose, feats, None, n_neighbor, anchors, kernels, radius, sigma, permute_modes != 0, contract=False)
    return inter_idx, (w.materialize() if MATERIALIZE_INTER_W else w), xyz, new_feats, None, pose


def inter_so3conv_grouping(xyz, feats, stride, n_neighbor, anchors, kernels, radius, sigma,
                           inter_idx=None, inter_w=None, lazy_sample=True, radius_expansion=1.0,
                           pooling=None):
    """Pose-free grouping (functional.py:L144-203), any stride.
    -> inter_idx [b,p2,nn], inter_w, new_xyz, new_feats [b,c,ks,p2,na], sample_idx."""
    if pooling is not None and stride > 1 and feats.shape[1] > 1:
        # low-pass blur before a strided conv (functional.py:L158-173)
        if pooling == 'stride':
            pool_stride, stride_nn, stride = stride, int(n_neighbor * stride ** 0.5), 1
        elif pooling == 'no-stride':
            pool_stride, stride_nn = 1, n_neighbor
        else:
            raise NotImplementedError(f"Pooling mode {pooling} is not implemented!")
        feats, xyz = inter_so3conv_blurring(xyz, feats, stride_nn, radius, pool_stride, inter_idx, lazy_sample)
        inter_idx = None
    if inter_idx is None:
        if stride > 1:
            sample_idx, new_xyz, _ = _strided_centres(xyz, None, stride, lazy_sample)
        else:
            new_xyz = xyz
            sample_idx = torch.arange(xyz.shape[2], dtype=torch.long, device=xyz.device).unsqueeze(0).repeat(xyz.shape[0], 1)
        inter_idx, inter_w, new_feats = _inter_group(xyz, None, feats, n_neighbor, anchors, kernels,
                                                     radius * radius_expansion, sigma, False, q_xyz=new_xyz)
        return inter_idx, inter_w, new_xyz, new_feats, sample_idx
    # cached neighbourhood from an earlier layer (functional.py:L195-201)
    if isinstance(inter_w, InterWeights):
        new_feats = _InterGroup.apply(feats, inter_idx, inter_w.gx, inter_w.rk, None, inter_w.sigma, 0, None)
    else:
        new_feats = inter_so3conv_feat_grouping(inter_idx, inter_w, zpconv.add_shadow_feature(feats))
    return inter_idx, inter_w, xyz, new_feats, None


def inter_so3poseconv_grouping_strided(xyz, pose, feats, stride, n_neighbor, anchors, kernels, radius,
                                       sigma, inter_idx=None, inter_w=None, lazy_sample=True,
                                       radius_expansion=1.0, pooling=None, permute_modes=0):
    """Pose-aware grouping (functional.py:L896-1286).  stride > 1 with no cached index: centres are
    furthest-point sampled, poses gathered, the ball query runs centres -> all points with the expanded
    radius (L931-1013) and the returned inter_idx is None (L1013); otherwise the stride-1 branch (L1025-1261:
    the neighbourhood is recomputed on every call, passed-in inter_idx / inter_w are ignored and inter_idx is
    handed back unchanged).
    -> inter_idx, inter_w, new_xyz, new_feats [b,c,ks,p2,na], sample_idx, sampled_pose."""
    if pooling is not None and stride > 1 and feats.shape[1] > 1:
        raise ValueError('xyz_pooling is not None?!!')                 # functional.py:L913
    if inter_idx is None and stride > 1:
        sample_idx, new_xyz, sampled_pose = _strided_centres(xyz, pose, stride, lazy_sample)
        _, w, new_feats = _inter_group(xyz, pose, feats, n_neighbor, anchors, kernels, radius * radius_expansion, sigma,
                                       permute_modes != 0, q_xyz=new_xyz, q_pose=sampled_pose)
        return None, w, new_xyz, new_feats, sample_idx.long(), sampled_pose          # spconv/functional.py:L476 `idx.long()`
    _, w, new_feats = _inter_group(xyz, pose, feats, n_neighbor, anchors, kernels, radius, sigma,
                                   permute_modes != 0)
    return inter_idx, w, xyz, new_feats, None, pose


def intra_so3conv_grouping(intra_idx, feature):
    """intra_idx [na,pnn], feature [nb,c,np,na] -> [nb,c,pnn,np,na] (functional.py:L2553-2602)."""
    if feature.dtype != torch.float32 or not feature.is_cuda:
        raise RuntimeError('intra_so3conv_grouping: float32 device tensors only')
    return _IntraGroup.apply(feature, intra_idx.to(torch.int32).contiguous())


def intra_so3conv_grouping_2D(intra_idx, feature):
    """The 2-D variant (functional.py:L2606-2628): the anchor axis is (na, 4); the 12-tap gather acts on na.
    feature [nb,c,np,na*4] -> [nb,c,pnn,np,na*4].  Index plumbing only (a view + one gather)."""
    nb, c_in, nq, tot_na = feature.shape
    f = feature.contiguous().view(nb, c_in, nq, -1, 4)
    na = f.size(-2)
    _, pnn = intra_idx.shape
    f1 = f.index_select(3, intra_idx.view(-1)).view(nb, c_in, nq, na, pnn, 4)
    return f1.permute([0, 1, 4, 2, 3, 5]).contiguous().view(nb, c_in, pnn, nq, tot_na).contiguous()


def anchor_permutation_index(xyz, pose, n_neighbor, anchors, radius):
    """The reference's rotated_anchor_idx int64 [b,p,nn,na] (functional.py:L1199-1204); test hook."""
    ball_idx = cuda_nn.ball_query(xyz, xyz, radius, n_neighbor)
    mult, ident = _group_tables(anchors)
    gx, _ = _hip.so3_prep(xyz, xyz, ball_idx, pose.contiguous(), pose.contiguous(), anchors.contiguous(), ident)
    return _hip.so3_anchor_perm(gx, mult)
