"""vgtk.spconv.functional -- operator API of the S^2 ("ZP") convolution and the helpers shared
with the SO(3) convolution (reference: vgtk/vgtk/spconv/functional.py).

Every compute op runs in libeap_hip.so (HIP, gfx950); torch only carries tensors, autograd
plumbing and trivially cheap view/cat ops.  Nothing here falls back to CPU.
"""
import math

import numpy as np
import torch

import vgtk.cuda.gathering as gather
import vgtk.cuda.zpconv as cuda_zpconv
import vgtk.pc as pctk

from .. import _hip


# ------------------------------------------------------------------------------------------------
# shadow point / feature  (functional.py:L83-96)
# ------------------------------------------------------------------------------------------------
def add_shadow_point(x):
    """[b,c,n] -> [b,c,n+1] with a 1e4 column appended."""
    b, c, _ = x.shape
    shadow = torch.full((b, c, 1), 1e4, dtype=torch.float32, device=x.device)
    return torch.cat((x, shadow), dim=2).contiguous()


def add_shadow_feature(x):
    """[b,c,n,a] -> [b,c,n+1,a] with a zero row appended."""
    b, c, _, a = x.shape
    shadow = torch.zeros(b, c, 1, a, dtype=torch.float32, device=x.device)
    return torch.cat((x, shadow), dim=2).contiguous()


# ------------------------------------------------------------------------------------------------
# autograd wrappers around the native ops  (functional.py:L102-129, L211-238, L314-335)
# ------------------------------------------------------------------------------------------------
class Gathering(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx):
        """points [nb,c,np], idx int32 [nb,m] -> [nb,c,m]"""
        ctx.save_for_backward(idx)
        ctx.npoint = points.size(2)
        return gather.gather_points_forward(points.contiguous(), idx.contiguous())

    @staticmethod
    def backward(ctx, grad):
        idx, = ctx.saved_tensors
        return gather.gather_points_backward(grad.contiguous(), idx, ctx.npoint), None


class IntraZPConvGrouping(torch.autograd.Function):
    @staticmethod
    def forward(ctx, intra_idx, intra_w, feats):
        """intra_idx [na_out,ann], intra_w [na_out,ks,ann], feats [nb,c,np,na_in] -> [nb,c,ks,np,na_out]"""
        ctx.save_for_backward(intra_idx, intra_w)
        ctx.anchor_in = feats.shape[3]
        return cuda_zpconv.intra_zpconv_forward(intra_idx, intra_w, feats.contiguous())

    @staticmethod
    def backward(ctx, grad):
        intra_idx, intra_w = ctx.saved_tensors
        return None, None, cuda_zpconv.intra_zpconv_backward(intra_idx, intra_w, grad.contiguous(), ctx.anchor_in)


class InterZPConvGrouping(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inter_idx, inter_w, feats):
        """inter_idx,inter_w [nb,np,na,ks,ann], feats [nb,c,nq(+1),na] -> [nb,c,ks,np,na]"""
        ctx.save_for_backward(inter_idx, inter_w)
        ctx.nq = feats.size(2)
        return cuda_zpconv.inter_zpconv_forward(inter_idx, inter_w, feats.contiguous())

    @staticmethod
    def backward(ctx, grad):
        inter_idx, inter_w = ctx.saved_tensors
        return None, None, cuda_zpconv.inter_zpconv_backward(inter_idx, inter_w, grad.contiguous(), ctx.nq)


def intra_zpconv_grouping(intra_idx, intra_w, feats):
    return IntraZPConvGrouping.apply(intra_idx, intra_w, feats)


def inter_zpconv_grouping_native(inter_idx, inter_w, feats):
    """5-D index / weights [b,p,a,k,ann] + feats [b,c,q,a] -> [b,c,k,p,a] through the native op with autograd
    (the commented-out `InterZPConvGrouping.apply` call of functional.py:L602)."""
    return InterZPConvGrouping.apply(inter_idx, inter_w, feats)


# ------------------------------------------------------------------------------------------------
# index helpers (same names, argument order and results as functional.py:L364-372, L452-466)
# ------------------------------------------------------------------------------------------------
def batched_index_select(input, dim, index):
    """input [B, ...], index int64 [B, M] -> input with axis `dim` replaced by the M selected positions, the selection
    being per batch element and shared by every other axis: out[b, ..., m, ...] = input[b, ..., index[b, m], ...]."""
    view = [1] * input.dim()
    view[0], view[dim] = index.shape[0], index.shape[1]
    target = list(input.shape)
    target[dim] = index.shape[1]
    return input.gather(dim, index.reshape(view).expand(target))


def batched_index_select_other(values, indices, dim=1):
    """values [*lead, N, *rest], indices int64 [*lead, *extra] with values in [0, N) -> [*lead, *extra, *rest]:
    out[l, e, r] = values[l, indices[l, e], r] (the neighbour gather of the grouping: lead = batch, N = support points,
    extra = (query point, neighbour slot), rest = per-point payload)."""
    lead, rest = values.shape[:dim], values.shape[dim + 1:]
    extra = indices.shape[dim:]
    n_lead = int(np.prod(lead)) if len(lead) else 1
    n_rest = int(np.prod(rest)) if len(rest) else 1
    flat = values.reshape(n_lead, values.shape[dim], n_rest)
    pick = indices.reshape(n_lead, -1, 1).expand(-1, -1, n_rest)
    return flat.gather(1, pick).reshape(*lead, *extra, *rest)


# ------------------------------------------------------------------------------------------------
# ball query + grouping  (functional.py:L341-350, L428-449)
# ------------------------------------------------------------------------------------------------
def ball_query(query_points, support_points, radius, n_sample, support_feats=None):
    """[b,3,m] x [b,3,n] -> idx int32 [b,m,k], grouped xyz [b,3,m,k] (, grouped feats)."""
    idx = pctk.ball_query_index(query_points, support_points, radius, n_sample)
    support_points = add_shadow_point(support_points)
    if support_feats is None:
        return idx, pctk.group_nd(support_points, idx)
    return idx, pctk.group_nd(support_points, idx), pctk.group_nd(support_feats, idx)


def inter_zpconv_grouping_ball(xyz, stride, radius, n_neighbor, lazy_sample=True):
    n_sample = math.ceil(xyz.shape[2] / stride)
    if stride > 1:
        idx, sample_xyz = pctk.furthest_sample(xyz, n_sample, lazy_sample)
    else:
        sample_xyz = xyz
        idx = torch.arange(xyz.shape[2], dtype=torch.long, device=xyz.device).unsqueeze(0).repeat(xyz.shape[0], 1)
    ball_idx, grouped_xyz = ball_query(sample_xyz, xyz, radius, n_neighbor)
    grouped_xyz = grouped_xyz - sample_xyz.unsqueeze(3)
    return grouped_xyz, ball_idx, idx, sample_xyz


# ------------------------------------------------------------------------------------------------
# "naive" grouping with a neighbour index shared by all (anchor, kernel point) pairs
# (functional.py:L252-272, L375-406).  The reference runs gather + einsum in torch; here the
# shared index is expanded to the native op's 5-D signature as a stride-0 view made contiguous
# only for the index (int32), and the HIP zpconv kernels do the contraction.
# ------------------------------------------------------------------------------------------------
def inter_zpconv_grouping_naive(inter_idx, inter_w, feats):
    """inter_idx [b,p,nn], inter_w [b,p,a,k,nn], feats [b,c,q,a] -> [b,c,k,p,a]."""
    b, p, a, k, nn = inter_w.shape
    idx5 = inter_idx.int()[:, :, None, None, :].expand(b, p, a, k, nn).contiguous()
    return InterZPConvGrouping.apply(idx5, inter_w.contiguous(), feats)


def intra_zpconv_grouping_naive(intra_idx, intra_w, feats):
    """intra_idx [a,nn], intra_w [a,k,nn], feats [b,c,p,a_in] -> [b,c,k,p,a]."""
    return IntraZPConvGrouping.apply(intra_idx.int().contiguous(), intra_w.contiguous(), feats)


def _neighbour_mean(inter_idx, feats):
    """mean over the nn neighbours of every query point of feats [b,c,q,a] (shadow row = zeros, as the reference's
    add_shadow_feature): -> [b,c,p,a]."""
    b, p, nn = inter_idx.shape
    padded = add_shadow_feature(feats)                                              # [b,c,q+1,a]
    rows = batched_index_select(padded, 2, inter_idx.long().reshape(b, p * nn))     # [b,c,p*nn,a]
    return rows.reshape(b, feats.shape[1], p, nn, feats.shape[3]).mean(3)


def inter_pooling_naive(inter_idx, sample_idx, feats, alpha=0.5):
    """Low-pass pooling onto the sampled centres (functional.py:L274-292): alpha * the centre's own feature +
    (1 - alpha) * the mean over its neighbourhood.  inter_idx [b,p,nn], sample_idx [b,p], feats [b,c,q,a] -> [b,c,p,a]."""
    return alpha * batched_index_select(feats, 2, sample_idx.long()) + (1.0 - alpha) * _neighbour_mean(inter_idx, feats)


def inter_blurring_naive(inter_idx, feats, alpha=0.5):
    """The same blend with every point its own centre (functional.py:L295-311): p == q."""
    if inter_idx.shape[1] != feats.shape[2]:
        raise ValueError('inter_blurring_naive: one neighbour list per point expected')
    return alpha * feats + (1.0 - alpha) * _neighbour_mean(inter_idx, feats)


# ------------------------------------------------------------------------------------------------
# polar-angle kernel positions shared with vgtk.so3conv (functional.py:L132-135)
# ------------------------------------------------------------------------------------------------
def get_angular_kernel_points_np(aperature, kernel_size):
    """kernel_size interior points of [0, aperature / 2] (end points excluded)."""
    return np.linspace(0, 0.5 * aperature, kernel_size + 2, dtype=np.float32)[1:-1]


# ------------------------------------------------------------------------------------------------
# pose-aware ball grouping (functional.py:L468-500)
# ------------------------------------------------------------------------------------------------
def inter_zpposeconv_grouping_ball(xyz, pose, stride, radius, n_neighbor, lazy_sample=True):
    """-> grouped_xyz [b,3,p2,nn], ball_idx (long) [b,p2,nn], idx [b,p2], sample_xyz, grouped_pose
    [b,p2,nn,...], sampled_pose [b,p2,...]   (functional.py:L468-500)."""
    n_sample = math.ceil(xyz.shape[2] / stride)
    idx, sample_xyz = pctk.furthest_sample(xyz, n_sample, lazy_sample)
    idx = idx.long()
    sampled_pose = batched_index_select(pose, dim=1, index=idx)
    ball_idx, grouped_xyz = ball_query(sample_xyz, xyz, radius, n_neighbor)
    ball_idx = ball_idx.long()
    grouped_pose = batched_index_select_other(pose, ball_idx, dim=1)
    grouped_xyz = grouped_xyz - sample_xyz.unsqueeze(3)
    return grouped_xyz, ball_idx, idx, sample_xyz, grouped_pose, sampled_pose


# ------------------------------------------------------------------------------------------------
# S^2 ("ZP") convolution: anchor directions, kernel tables, anchor-to-anchor tables and the grouping entry of
# vgtk.spconv.modules (functional.py:L20-61, L134-208, L503-607, L610-655).  Host-side tables are built once per
# module; the contractions run in the HIP zpconv kernels through the *_naive entries above.
# ------------------------------------------------------------------------------------------------
_ANCHOR_TABLES = {}


def _unit_rows(points):
    """Rows of `points` [n,3] longer than 0.5, scaled to unit length (the icosphere files carry a centre vertex)."""
    length = np.linalg.norm(points, axis=1)
    long_enough = length > 0.5
    return (points[long_enough] / length[long_enough, None]).astype(np.float32)


def get_anchors(anchor):
    """Directions on S^2 as a CPU tensor [na,3]: an int names an icosphere of the data set (12 / 42 / 92 / 162 vertices,
    vgtk/data/anchors/constants.npz), a str is the path of a PLY file, a tensor is passed through (detached, on the host)."""
    if torch.is_tensor(anchor):
        return anchor.detach().cpu()
    if isinstance(anchor, str):
        return torch.from_numpy(_unit_rows(pctk.load_ply(anchor).astype(np.float32)))
    if not isinstance(anchor, int):
        raise ValueError(f'Not recognized anchor type {type(anchor)}')
    if anchor not in _ANCHOR_TABLES:
        import os
        import vgtk
        table = np.load(os.path.join(vgtk.__path__[0], 'data', 'anchors', 'constants.npz'))
        name = f'sphere{anchor:d}_vertices'
        if name not in table.files:
            raise ValueError(f'no icosphere with {anchor} vertices in the anchor data (12, 42, 92, 162)')
        _ANCHOR_TABLES[anchor] = _unit_rows(table[name].astype(np.float32))
    return torch.from_numpy(_ANCHOR_TABLES[anchor].copy())


def get_kernel_rings_np(radius, aperature, kernel_size, multiplier=1):
    """Kernel positions of the inter ZP conv as (distance from the centre, polar angle) rows, float32 [ks,2].
    int kernel_size: ring i of `kernel_size` rings sits at the i-th interior point of [0, radius] and carries
    multiplier * i + 1 polar angles; a pair (n_r, n_w): the full grid of n_r distances radius/n_r .. radius and n_w angles."""
    if isinstance(kernel_size, int):
        ring_radius = np.linspace(0, radius, kernel_size + 2, dtype=np.float32)[1:-1]
        rows = [(ring_radius[i], w) for i in range(kernel_size)
                for w in get_angular_kernel_points_np(aperature, multiplier * i + 1)]
        return np.asarray(rows, dtype=np.float32).reshape(-1, 2)
    n_r, n_w = kernel_size
    ring_radius = np.linspace(radius / n_r, radius, n_r, dtype=np.float32)
    angles = get_angular_kernel_points_np(aperature, n_w)
    grid = np.stack(np.meshgrid(ring_radius, angles, indexing='ij'), axis=-1)
    return grid.reshape(-1, 2).astype(np.float32)


def get_intra_kernels(aperature, kernel_size):
    """kernel_size angular bins covering [0, aperature / 2], end points included."""
    return torch.from_numpy(np.linspace(0, 0.5 * aperature, kernel_size, dtype=np.float32))


def acos_safe(x, eps=1e-4):
    """arccos continued linearly outside [-(1 - eps), 1 - eps] with the slope arccos(1 - eps) / eps, so that its gradient
    stays finite at +-1."""
    edge = 1.0 - eps
    slope = np.arccos(edge) / eps
    s = torch.sign(x)
    outside = torch.acos(s * edge) - slope * s * (x.abs() - 1 + eps)     # (|x| - 1) + eps in fp32, the reference's rounding order
    return torch.where(x.abs() <= edge, torch.acos(x), outside)


def anchor_knn(a_src, a_tgt, k=3, metric="spherical"):
    """The k anchors of a_src [n,3] nearest to every anchor of a_tgt [m,3] -> (values [m,k], indices [m,k]).
    'spherical': largest <s,t> - 1; 'angular': smallest acos_safe(<s,t>); anything else: smallest squared distance."""
    if metric in ('spherical', 'angular'):
        cosine = (a_tgt[:, None, :] * a_src[None, :, :]).sum(2)
        if metric == 'spherical':
            return (cosine - 1.0).topk(k=k, dim=1, largest=True)
        return acos_safe(cosine).topk(k=k, dim=1, largest=False)
    return (a_src[None, :, :] - a_tgt[:, None, :]).pow(2).sum(2).topk(k=k, dim=1, largest=False)


def get_intra_kernel_weights(anchor_in, anchor_out, kernels, ann, aperature, sigma=1e-1, use_suppression=False):
    """Tables of the intra ZP conv: for every output anchor its `ann` angularly nearest input anchors, idx int32 [a_out,ann],
    and the influence of each on every angular bin, [a_out,ks,ann] = relu(1 - |angle - bin| / pi / (3 sqrt(sigma / 2)))
    (optionally zero beyond half the aperture)."""
    if anchor_out is None:
        anchor_out = anchor_in
    angle, idx = anchor_knn(anchor_in, anchor_out, k=ann, metric='angular')                 # [a_out,ann]
    width = 3.0 * (sigma / 2.0) ** 0.5
    gap = (angle[:, None, :] - kernels[None, :, None]).abs() / np.pi                         # [a_out,ks,ann]
    influence = torch.relu(1.0 - gap / width)
    if use_suppression:
        influence = influence * angle.le(0.5 * aperature)[:, None, :].float()
    return idx.int().contiguous(), influence.contiguous()


def compute_anchor_weights(anchor_in, anchor_out, k=3, sigma=1e-1, interpolation="inv"):
    """k-nearest-anchor interpolation table from anchor_in [a1,3] to anchor_out [a2,3] -> idx [a2,k], w [a2,k] (rows sum to 1).
    'spherical': softmax((<i,o> - 1) / sigma); 'euclidean': softmax(-d^2 / sigma); 'inv': 1 / (sigma d^2 + 1e-6), normalised."""
    if interpolation == 'spherical':
        val, idx = anchor_knn(anchor_in, anchor_out, k=k, metric='spherical')
        return idx, torch.softmax(val / sigma, dim=1)
    val, idx = anchor_knn(anchor_in, anchor_out, k=k, metric='euclidean')
    if interpolation == 'euclidean':
        return idx, torch.softmax(-val / sigma, dim=1)
    if interpolation != 'inv':
        raise ValueError(f'unknown interpolation {interpolation!r}')
    closeness = 1.0 / (sigma * val + 1e-6)
    return idx, closeness / closeness.sum(1, keepdim=True)


def anchor_prop(x, idx, w):
    """Features [b,c,p,a1] carried to another anchor set: out[..., j] = sum_k w[j,k] x[..., idx[j,k]] -> [b,c,p,a2]."""
    a2, k = idx.shape
    picked = x.index_select(3, idx.reshape(-1).long()).reshape(*x.shape[:3], a2, k)
    return (picked * w).sum(-1)


def inter_zpconv_grouping_anchor(grouped_xyz, ball_idx, sample_idx, anchors, kernels, anchor_nn, n_support,
                                 radius, aperture, sigma):
    """Kernel weights of the inter ZP conv (the reference's live "linear kernel" branch): for the neighbour offsets
    grouped_xyz [b,3,p,nn], anchor directions [a,3] and kernel rows (rho_k, theta_k):
        w[b,p,a,k,n] = relu(1 - (|r - rho_k| + |r (theta - theta_k)| / 3) / sqrt(sigma)),
    r = |offset| + 1e-6, theta = acos_safe(<offset, anchor> / r).  -> (ball_idx unchanged [b,p,nn], w [b,p,a,ks,nn])."""
    r = grouped_xyz.pow(2).sum(1).sqrt() + 1e-6                                               # [b,p,nn]
    along = torch.einsum('bdpn,ad->bpan', grouped_xyz, anchors.to(grouped_xyz))              # [b,p,a,nn]
    theta = acos_safe(along / r[:, :, None, :])[:, :, :, None, :]                             # [b,p,a,1,nn]
    r5 = r[:, :, None, None, :]
    rho = kernels[:, 0].reshape(1, 1, 1, -1, 1)
    theta_k = kernels[:, 1].reshape(1, 1, 1, -1, 1)
    spread = (r5 - rho).abs() + (r5 * (theta - theta_k)).abs() / 3.0
    return ball_idx, torch.relu(1.0 - spread / sigma ** 0.5)


def inter_zpconv_grouping(xyz, feats, stride, n_neighbor, anchors, kernels, anchor_nn, radius, aperture, sigma,
                          inter_idx=None, inter_w=None, lazy_sample=True, radius_expansion=1.0):
    """Ball grouping + kernel weights (unless handed in) + the grouped contraction -> inter_idx [b,p,nn], inter_w,
    new_xyz, new_feats [b,c,k,p,a].  As in the reference the freshly built weights are stored with their anchor and
    kernel axes swapped ([b,p,ks,a,nn]) and the contraction reads axis 2 as the anchor axis, a singleton there being
    broadcast over the feature's anchors."""
    if inter_idx is None:
        grouped_xyz, ball_idx, centre_idx, new_xyz = inter_zpconv_grouping_ball(xyz, stride, radius * radius_expansion,
                                                                                n_neighbor, lazy_sample)
        inter_idx, inter_w = inter_zpconv_grouping_anchor(grouped_xyz, ball_idx, centre_idx, anchors, kernels, anchor_nn,
                                                          xyz.shape[2], radius, aperture, sigma)
        inter_w = inter_w.transpose(2, 3).contiguous()
    else:
        new_xyz = xyz
    padded = add_shadow_feature(feats)
    w = inter_w if inter_w.shape[2] == padded.shape[3] else inter_w.expand(-1, -1, padded.shape[3], -1, -1)
    return inter_idx, inter_w, new_xyz, inter_zpconv_grouping_naive(inter_idx, w, padded)
