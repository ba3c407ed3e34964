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
import os

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


def get_sphereical_kernel_points_from_ply(radius, kernel_size):
    """kernel_size 1/2/3 -> 24/30/66 kernel points rescaled so the max norm is `radius`."""
    assert 0 < kernel_size <= 3
    pts = _KP['kpsphere%d' % {1: 24, 2: 30, 3: 66}[kernel_size]].astype('float32')
    r = np.sqrt((pts ** 2).sum(1).max())
    return pts * radius / r


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
    key = (anchors.data_ptr(), anchors.device, anchors.shape[0])
    hit = _TABLES.get(key)
    if hit is not None:
        return hit
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
    _TABLES[key] = out
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
    def forward(ctx, W, x):
        W = W.contiguous()
        x = x.contiguous()
        b, ck, pa = x.shape
        o = W.shape[0]
        y = torch.empty(b, o, pa, dtype=torch.float32, device=x.device)
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
            _hip.gemm(1, 0, ck, pa, o, W, ck, 0, gy, pa, o * pa, gx, pa, ck * pa, b)
        if ctx.needs_input_grad[0]:
            gW = torch.empty_like(W)          # sum_b gy_b x_b^T : [o,pa] [pa,ck]
            _hip.gemm_reduce(0, 1, o, ck, pa, gy, pa, o * pa, x, pa, ck * pa, gW, ck, b)
        return gW, gx


# Feature-gradient strategy of the fused inter convolution: "auto" picks the re-associated
# (inverse-list) path when at most 1/INV_ROW_FRACTION of the support points are referenced by
# any neighbour list, else dX = W^T dY followed by the transposed grouping.
BACKWARD_MODE = 'auto'      # 'auto' | 'inverse' | 'dx'
INV_ROW_FRACTION = 4


def _inverse_lists(idx, gx, n_sup, ident, nonident=None):
    """Inverse neighbour lists of idx [b,p,nn] (which (point, slot) pairs reference each support
    row), with the referenced rows compacted.  Small torch plumbing on the device; ONE host sync
    (the number of referenced rows sizes the workspace)."""
    b, p, nn = idx.shape
    keys = idx.reshape(b, p * nn).long()
    skeys, order = torch.sort(keys, dim=1, stable=True)          # entries of a row stay in (p,n) order
    counts = torch.zeros(b, n_sup + 1, dtype=torch.int64, device=idx.device)
    counts.scatter_add_(1, keys.clamp(max=n_sup), torch.ones_like(keys))
    counts = counts[:, :n_sup]
    offs = torch.cumsum(counts, 1) - counts
    nonempty = counts > 0
    n_rows = nonempty.sum(1)
    # one host sync for both facts the launch needs: workspace rows, and whether any relative
    # rotation differs from the identity (if none does, the permutation table is skipped)
    if nonident is not None:
        all_ident = (nonident == 0).all()
    else:
        all_ident = (gx[..., 3].contiguous().view(torch.int32) == ident).all()
    rcap, all_ident = torch.stack([n_rows.max(), all_ident.to(n_rows.dtype)]).tolist()
    rcap, all_ident = int(rcap), bool(all_ident)
    # referenced rows first, longest entry list first: neighbouring blocks then walk the query
    # points at the same pace (shared L2 window, csrc/so3_inter_inv.hip) and the long lists start early
    rows = torch.argsort(counts, dim=1, descending=True, stable=True)[:, :rcap]
    valid = torch.arange(rcap, device=idx.device)[None, :] < n_rows[:, None]
    off_c = torch.gather(offs, 1, rows)
    cnt_c = torch.gather(counts, 1, rows) * valid
    rows_c = torch.where(valid, rows, torch.full_like(rows, -1))
    ent_p = torch.div(order, nn, rounding_mode='floor').to(torch.int32).contiguous()
    ent_gx = torch.gather(gx.reshape(b, p * nn, 4), 1, order[..., None].expand(-1, -1, 4)).contiguous()
    return (rows_c.to(torch.int32).contiguous(), off_c.to(torch.int32).contiguous(),
            cnt_c.to(torch.int32).contiguous(), ent_p, ent_gx, rcap, all_ident)


# Layout of the fused conv's intermediate X and who contracts it (where the kernels allow it, else the
# reference layout + own GEMM):
#   'transposed'  X as the plain [P*A, C*K] matrix, contraction = library GEMM (hipBLASLt via torch.matmul:
#                 149 TFLOP/s on the deepest layer against 127 for csrc/gemm_f32.hip)
#   'blocked'     X blocked by anchor quads, contraction = csrc/gemm_f32.hip (eap_gemm_f32_xb)
#   'reference'   X [C*K, P*A] as the reference's einsum writes it, contraction = csrc/gemm_f32.hip
X_LAYOUT = 'transposed'
LIBRARY_SMALL_GEMMS = True    # the two Z-based gradient GEMMs (plain row-major operands) through the library as well; False: csrc/gemm_f32.hip
BLOCKED_X = True     # test knob: False forces the reference layout


class _InterConv(torch.autograd.Function):
    """Fused inter conv  y = W . group(feats)  (functional.py:L1221-1261 + modules.py:L48-55)
    with the re-associated feature gradient (csrc/so3_inter_inv.hip)."""

    @staticmethod
    def forward(ctx, feats, W, idx, gx, rk, mult, sigma, ident, nonident=None):
        feats = feats.contiguous()
        W = W.contiguous()
        # X is internal to this Function: where the kernels allow it, it is kept blocked by anchor
        # quads ([b,p,a/4,c,k,4]) -- coalesced row-end stores in the grouping kernel -- and the GEMMs
        # read it as a blocked B operand (include/eap_hip.h, "blocked intermediate")
        can = BLOCKED_X and X_LAYOUT != 'reference' and _hip.so3_inter_group_fwd_can_block(
            feats.shape[1], feats.shape[2], feats.shape[3], rk.shape[1], mult is not None, nonident is not None)
        layout = 0 if not can else (2 if X_LAYOUT == 'transposed' else 1)
        x = _hip.so3_inter_group_fwd(feats, idx, gx, rk, mult, sigma, nonident, blocked=layout)   # [b,c,k,p,a] (nominal shape)
        b, c, ks, p, na = x.shape
        o = W.shape[0]
        y = torch.empty(b, o, p, na, dtype=torch.float32, device=x.device)
        if layout == 2:
            _hip.library_contract(W, x.view(b, p * na, c * ks), y.view(b, o, p * na))
        else:
            _hip.gemm(0, 0, o, p * na, c * ks, W, c * ks, 0, x, p * na, c * ks * p * na, y, p * na, o * p * na, b, b_blocked=layout == 1)
        ctx.layout = layout
        ctx.save_for_backward(W, x, idx, gx, rk, mult if mult is not None else torch.empty(0),
                              nonident if nonident is not None else torch.empty(0))
        ctx.has_mult = mult is not None
        ctx.has_flag = nonident is not None
        ctx.feats_ref = feats          # needed by the re-associated weight gradient (not a new copy)
        ctx.sigma, ctx.ident, ctx.n = sigma, ident, feats.shape[2]
        return y

    @staticmethod
    def backward(ctx, gy):
        W, x, idx, gx, rk, mult, nonident = ctx.saved_tensors
        mult = mult if ctx.has_mult else None
        nonident = nonident if ctx.has_flag else None
        gy = gy.contiguous()
        b, c, ks, p, na = x.shape
        o, ck, pa = W.shape[0], c * ks, p * na
        gW = gF = None
        n = ctx.n
        # Strategy: when few support rows are referenced (the reference's first-nsample-in-index-
        # order ball query with large radii), BOTH gradients follow from
        #     Z[o,k,q,a'] = sum_{(p,n)->q} dY[o,p,a] w(p,a,k,n)          (csrc/so3_inter_inv.hip)
        #     dF[c,q,a'] = sum_{o,k} W[o,(c,k)] Z[o,k,q,a']       dW[o,(c,k)] = sum_{q,a'} Z[o,k,q,a'] F[c,q,a']
        # two small GEMMs over the referenced rows only -- no dX = W^T dY, no scatter, and the
        # [O x P*A] x [P*A x C*K] weight-gradient GEMM shrinks by P / (referenced rows).
        inv = None
        if BACKWARD_MODE != 'dx' and na % 4 == 0 and ks <= 32 and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            inv = _inverse_lists(idx, gx, n, ctx.ident, nonident)
            if BACKWARD_MODE == 'auto' and inv[5] * INV_ROW_FRACTION > n:
                inv = None
        if inv is not None:
            rows, off, cnt, ent_p, ent_gx, rcap, all_ident = inv
            multinv = None
            if mult is not None and not all_ident:   # multinv[r][a'] = a  with  mult[r][a] = a'
                multinv = torch.empty_like(mult)
                multinv.scatter_(1, mult.long(), torch.arange(na, device=mult.device, dtype=torch.uint8).repeat(na, 1))
            z = _hip.so3_inter_group_inv(gy, rows, off, cnt, ent_p, ent_gx, rk, multinv, ctx.sigma, idx.shape[2],
                                         ctx.ident)                                  # [b,o,ks,rcap,na]
            ra = rcap * na
            dest = rows.clamp(min=0).long()                                           # unused slots (rows < 0) carry zeros in Z
            dest4 = dest[:, None, :, None].expand(b, c, rcap, na)
            if ctx.needs_input_grad[0]:
                W2 = W.view(o, c, ks).permute(1, 0, 2).reshape(c, o * ks).contiguous()
                gFc = torch.empty(b, c, ra, dtype=torch.float32, device=gy.device)
                if LIBRARY_SMALL_GEMMS:
                    _hip.library_matmul(W2, z.view(b, o * ks, ra), gFc)
                else:
                    _hip.gemm(0, 0, c, ra, o * ks, W2, o * ks, 0, z, ra, o * ks * ra, gFc, ra, c * ra, b)
                # rows of unused slots are exactly zero (Z is), so adding them to row 0 is harmless
                gF = torch.zeros(b, c, n, na, dtype=torch.float32, device=gy.device)
                gF.scatter_add_(2, dest4, gFc.view(b, c, rcap, na))
            if ctx.needs_input_grad[1]:
                feats = ctx.feats_ref
                fc = torch.gather(feats, 2, dest4)                                   # [b,c,rcap,na]; unused slots meet zero rows of Z
                fc = fc.reshape(b, c, ra)
                if LIBRARY_SMALL_GEMMS:
                    d = _hip.library_matmul(z.view(b, o * ks, ra), fc.transpose(1, 2), None).sum(0)
                else:
                    d = torch.empty(o * ks, c, dtype=torch.float32, device=gy.device)    # sum_b Z_b Fc_b^T
                    _hip.gemm_reduce(0, 1, o * ks, c, ra, z, ra, o * ks * ra, fc, ra, c * ra, d, c, b)
                gW = d.view(o, ks, c).permute(0, 2, 1).reshape(o, c * ks).contiguous()
        else:
            if ctx.needs_input_grad[1]:
                gW = torch.empty_like(W)          # sum_b gy_b x_b^T
                if ctx.layout == 2:     # X^T [pa, ck]: dW = dY X^T is a plain row-major product
                    _hip.gemm_reduce(0, 0, o, ck, pa, gy, pa, o * pa, x, ck, ck * pa, gW, ck, b)
                else:
                    _hip.gemm_reduce(0, 1, o, ck, pa, gy, pa, o * pa, x, pa, ck * pa, gW, ck, b, b_blocked=ctx.layout == 1)
            if ctx.needs_input_grad[0]:
                gx_ = torch.empty_like(x.view(b, ck, pa))      # W^T gy
                _hip.gemm(1, 0, ck, pa, o, W, ck, 0, gy.view(b, o, pa), pa, o * pa, gx_, pa, ck * pa, b)
                gF = _hip.so3_inter_group_bwd(gx_.view(b, c, ks, p, na), idx, gx, rk, mult, ctx.sigma, n, ctx.ident)
        return gF, gW, None, None, None, None, None, None, None


def so3_contract(W, x):
    """W [O, C*K], x [b, C*K, P*A] -> [b, O, P*A]."""
    _hip.check_input(x)
    if x.dtype != torch.float32 or W.dtype != torch.float32:
        raise RuntimeError('so3_contract: float32 only')
    if not W.is_cuda:
        raise RuntimeError('so3_contract: W must be a device tensor')
    return _Contract.apply(W, x)


# ------------------------------------------------------------------------------------------------
# grouping entry points
# ------------------------------------------------------------------------------------------------
def _check_stride(stride, pooling, feats):
    if stride != 1:
        raise NotImplementedError(
            'stride > 1 (furthest-point-sampled centres) is outside the accelerated path: the shipped '
            'models force stride 1 (SPConvNets/models/unsup_seg_so3_pose_conv_pn_38_multi_stage.py:L2191)')


def _inter_group(xyz, pose, feats, n_neighbor, anchors, kernels, radius, sigma, permute):
    if feats.dtype != torch.float32 or xyz.dtype != torch.float32:
        raise RuntimeError('so3conv: float32 only')
    _hip.check_input(xyz)
    if not feats.is_cuda:
        raise RuntimeError('so3conv: feats must be a device tensor')
    ball_idx = cuda_nn.ball_query(xyz, xyz, radius, n_neighbor)
    rk = rotated_kernels(anchors, kernels)
    mult = ident = None
    rot = None
    if pose is not None:
        rot = pose.contiguous()
        if rot.shape[-2:] != (4, 4) or rot.dtype != torch.float32:
            raise RuntimeError('so3conv: pose must be float32 [b,p,4,4]')
        if permute:
            mult, ident = _group_tables(anchors)
            if mult is None:
                raise NotImplementedError(
                    'anchor permutation with per-point poses needs a closed anchor set (kanchor 60 or 1)')
    gx, nonident = _hip.so3_prep(xyz, xyz, ball_idx, rot, rot, anchors.contiguous(), 0 if ident is None else ident)
    new_feats = _InterGroup.apply(feats, ball_idx, gx, rk, mult, float(sigma), 0 if ident is None else ident, nonident)
    inter_w = InterWeights(gx, rk, sigma)
    return ball_idx, (inter_w.materialize() if MATERIALIZE_INTER_W else inter_w), new_feats


def inter_so3conv_fused(xyz, pose, feats, W, n_neighbor, anchors, kernels, radius, sigma, permute):
    """ball query + prep + fused (grouping . contraction) -> (ball_idx, InterWeights, y [b,o,p,a]).
    What InterSO3PoseConv / InterSO3Conv.forward run for stride 1."""
    if feats.dtype != torch.float32 or xyz.dtype != torch.float32 or W.dtype != torch.float32:
        raise RuntimeError('so3conv: float32 only')
    _hip.check_input(xyz)
    if not feats.is_cuda or not W.is_cuda:
        raise RuntimeError('so3conv: feats and W must be device tensors')
    ball_idx = cuda_nn.ball_query(xyz, xyz, radius, n_neighbor)
    rk = rotated_kernels(anchors, kernels)
    mult = ident = rot = None
    if pose is not None:
        rot = pose.contiguous()
        if rot.shape[-2:] != (4, 4) or rot.dtype != torch.float32:
            raise RuntimeError('so3conv: pose must be float32 [b,p,4,4]')
        if permute:
            mult, ident = _group_tables(anchors)
            if mult is None:
                raise NotImplementedError(
                    'anchor permutation with per-point poses needs a closed anchor set (kanchor 60 or 1)')
    gx, nonident = _hip.so3_prep(xyz, xyz, ball_idx, rot, rot, anchors.contiguous(), 0 if ident is None else ident)
    y = _InterConv.apply(feats, W, ball_idx, gx, rk, mult, float(sigma), 0 if ident is None else ident, nonident)
    inter_w = InterWeights(gx, rk, sigma)
    return ball_idx, (inter_w.materialize() if MATERIALIZE_INTER_W else inter_w), y


def inter_so3conv_grouping(xyz, feats, stride, n_neighbor, anchors, kernels, radius, sigma,
                           inter_idx=None, inter_w=None, lazy_sample=True, radius_expansion=1.0,
                           pooling=None):
    """Pose-free grouping (functional.py:L144-203), stride 1.
    -> inter_idx [b,p,nn], inter_w, new_xyz, new_feats [b,c,ks,p,na], sample_idx."""
    _check_stride(stride, pooling, feats)
    if inter_idx is None:
        inter_idx, inter_w, new_feats = _inter_group(xyz, None, feats, n_neighbor, anchors, kernels,
                                                     radius * radius_expansion, sigma, False)
        sample_idx = torch.arange(xyz.shape[2], dtype=torch.long, device=xyz.device).unsqueeze(0).repeat(xyz.shape[0], 1)
        return inter_idx, inter_w, xyz, new_feats, sample_idx
    # cached neighbourhood from an earlier layer (functional.py:L195-201)
    if isinstance(inter_w, InterWeights):
        new_feats = _InterGroup.apply(feats, inter_idx, inter_w.gx, inter_w.rk, None, inter_w.sigma, 0, None)
    else:
        new_feats = inter_so3conv_feat_grouping(inter_idx, inter_w, zpconv.add_shadow_feature(feats))
    return inter_idx, inter_w, xyz, new_feats, None


def inter_so3poseconv_grouping_strided(xyz, pose, feats, stride, n_neighbor, anchors, kernels, radius,
                                       sigma, inter_idx=None, inter_w=None, lazy_sample=True,
                                       radius_expansion=1.0, pooling=None, permute_modes=0):
    """Pose-aware grouping, stride-1 branch of functional.py:L896-1286 (the neighbourhood is
    recomputed on every call, exactly like the reference: passed-in inter_idx / inter_w are
    ignored and handed back unchanged).
    -> inter_idx (as passed in), inter_w, new_xyz, new_feats [b,c,ks,p,na], sample_idx (None),
       sampled_pose."""
    _check_stride(stride, pooling, feats)
    _, w, new_feats = _inter_group(xyz, pose, feats, n_neighbor, anchors, kernels, radius, sigma,
                                   permute_modes != 0)
    return inter_idx, w, xyz, new_feats, None, pose


def intra_so3conv_grouping(intra_idx, feature):
    """intra_idx [na,pnn], feature [nb,c,np,na] -> [nb,c,pnn,np,na] (functional.py:L2553-2602)."""
    if feature.dtype != torch.float32 or not feature.is_cuda:
        raise RuntimeError('intra_so3conv_grouping: float32 device tensors only')
    return _IntraGroup.apply(feature, intra_idx.to(torch.int32).contiguous())


def anchor_permutation_index(xyz, pose, n_neighbor, anchors, radius):
    """The reference's rotated_anchor_idx int64 [b,p,nn,na] (functional.py:L1199-1204); test hook."""
    ball_idx = cuda_nn.ball_query(xyz, xyz, radius, n_neighbor)
    mult, ident = _group_tables(anchors)
    gx, _ = _hip.so3_prep(xyz, xyz, ball_idx, pose.contiguous(), pose.contiguous(), anchors.contiguous(), ident)
    return _hip.so3_anchor_perm(gx, mult)
