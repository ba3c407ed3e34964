"""vgtk.spconv.functional -- operator API of the S^2 ("ZP") convolution and the helpers shared
with the SO(3) convolution (reference: vgtk/vgtk/spconv/functional.py).

Every compute op runs in libeap_hip.so (HIP, gfx950); torch only carries tensors, autograd
plumbing and trivially cheap view/cat ops.  Nothing here falls back to CPU.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

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
# index helpers (functional.py:L364-372, L452-466)
# ------------------------------------------------------------------------------------------------
def batched_index_select(input, dim, index):
    for ii in range(1, len(input.shape)):
        if ii != dim:
            index = index.unsqueeze(ii)
    expanse = list(input.shape)
    expanse[0] = -1
    expanse[dim] = -1
    return torch.gather(input, dim, index.expand(expanse))


def batched_index_select_other(values, indices, dim=1):
    value_dims = values.shape[(dim + 1):]
    indices_shape = list(indices.shape)
    indices = indices[(..., *((None,) * len(value_dims)))]
    indices = indices.expand(*((-1,) * len(indices_shape)), *value_dims)
    value_expand_len = len(indices_shape) - (dim + 1)
    values = values[(*((slice(None),) * dim), *((None,) * value_expand_len), ...)]
    value_expand_shape = [-1] * len(values.shape)
    expand_slice = slice(dim, (dim + value_expand_len))
    value_expand_shape[expand_slice] = indices.shape[expand_slice]
    values = values.expand(*value_expand_shape)
    dim += value_expand_len
    return values.gather(dim, indices)


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


def inter_pooling_naive(inter_idx, sample_idx, feats, alpha=0.5):
    b, p, pnn = inter_idx.shape
    _, c, q, a = feats.shape
    new_feats = batched_index_select(feats, 2, sample_idx.long())
    grouped = batched_index_select(add_shadow_feature(feats), 2, inter_idx.long().view(b, -1)).view(b, -1, p, pnn, a)
    return alpha * new_feats + (1 - alpha) * grouped.mean(3)


def inter_blurring_naive(inter_idx, feats, alpha=0.5):
    b, p, pnn = inter_idx.shape
    _, c, q, a = feats.shape
    assert p == q
    grouped = batched_index_select(add_shadow_feature(feats), 2, inter_idx.long().view(b, -1)).view(b, -1, p, pnn, a)
    return alpha * feats + (1 - alpha) * grouped.mean(3)


# ------------------------------------------------------------------------------------------------
# intra kernels of the S^2 convolution (functional.py:L132-208) -- tiny host-side tables
# ------------------------------------------------------------------------------------------------
def get_angular_kernel_points_np(aperature, kernel_size):
    return np.linspace(0, 0.5 * aperature, kernel_size + 2, dtype=np.float32)[1:-1]


def get_intra_kernels(aperature, kernel_size):
    return torch.from_numpy(np.linspace(0, 0.5 * aperature, kernel_size, dtype=np.float32))


def acos_safe(x, eps=1e-4):
    sign = torch.sign(x)
    slope = np.arccos(1 - eps) / eps
    return torch.where(abs(x) <= 1 - eps, torch.acos(x),
                       torch.acos(sign * (1 - eps)) - slope * sign * (abs(x) - 1 + eps))


def anchor_knn(a_src, a_tgt, k=3, metric='spherical'):
    a_src = a_src.unsqueeze(0)
    a_tgt = a_tgt.unsqueeze(1)
    if metric == 'spherical':
        dists = torch.sum(a_src * a_tgt, dim=2) - 1.0
        return dists.topk(k=k, dim=1, largest=True)
    if metric == 'angular':
        dists = acos_safe(torch.sum(a_src * a_tgt, dim=2))
        return dists.topk(k=k, dim=1, largest=False)
    dists = torch.sum((a_src - a_tgt) ** 2, dim=2)
    return dists.topk(k=k, dim=1, largest=False)


def get_intra_kernel_weights(anchor_in, anchor_out, kernels, ann, aperature, sigma=1e-1, use_suppression=False):
    anchor_out = anchor_in if anchor_out is None else anchor_out
    angles, idx = anchor_knn(anchor_in, anchor_out, k=ann, metric='angular')
    if use_suppression:
        suppression = angles.le(0.5 * aperature).unsqueeze(1).expand(-1, kernels.size(0), -1).float()
    angles = angles.unsqueeze(1)
    kernels = kernels.unsqueeze(0).unsqueeze(-1)
    influence = (angles - kernels).abs() / np.pi
    influence = F.relu(1.0 - influence / (3 * (sigma / 2.0) ** 0.5), inplace=True)
    if use_suppression:
        influence = influence * suppression
    return idx.int().contiguous(), influence.contiguous()


def compute_anchor_weights(anchor_in, anchor_out, k=3, sigma=1e-1, interpolation='inv'):
    if interpolation == 'spherical':
        dists = (anchor_in.unsqueeze(0) * anchor_out.unsqueeze(1)).sum(2) - 1.0
        val, idx = dists.topk(k=k, dim=1, largest=True)
        return idx, F.softmax(val / sigma, dim=1)
    dists = (anchor_in.unsqueeze(0) - anchor_out.unsqueeze(1)).pow(2).sum(2)
    val, idx = dists.topk(k=k, dim=1, largest=False)
    if interpolation == 'euclidean':
        return idx, F.softmax(-val / sigma, dim=1)
    inv_val = 1. / (sigma * val + 1e-6)
    return idx, inv_val / inv_val.sum(1, keepdim=True)


def anchor_prop(x, idx, w):
    """[b,c,p,a1] -> [b,c,p,a2] 3-NN interpolation over anchors."""
    return (x[:, :, :, idx] * w).sum(-1)


# ------------------------------------------------------------------------------------------------
# S^2 anchors and kernels of the ZP convolution (functional.py:L20-66) -- host-side tables
# ------------------------------------------------------------------------------------------------
_SPHERES = None


def get_anchors(anchor):
    """int (12 / 42 / 92 / 162: vertices of the icosphere data file with norm > 0.5, normalised), a PLY path,
    or a tensor -> [na, 3] CPU tensor (functional.py:L20-39)."""
    global _SPHERES
    if isinstance(anchor, torch.Tensor):
        return anchor.detach().cpu()
    if isinstance(anchor, int):
        if _SPHERES is None:
            import os
            import vgtk
            _SPHERES = np.load(os.path.join(vgtk.__path__[0], 'data', 'anchors', 'constants.npz'))
        key = 'sphere%d_vertices' % anchor
        if key not in _SPHERES.files:
            raise ValueError('no S^2 anchor set with %d directions (12, 42, 92, 162)' % anchor)
        pts = _SPHERES[key].astype('float32')
    elif isinstance(anchor, str):
        pts = pctk.load_ply(anchor).astype('float32')
    else:
        raise ValueError(f'Not recognized anchor type {type(anchor)}')
    norms = np.sqrt(np.sum(pts ** 2, axis=1))
    keep = np.where(norms > 0.5)
    return torch.from_numpy(pts[keep] / np.expand_dims(norms[keep], 1))


def get_kernel_rings_np(radius, aperature, kernel_size, multiplier=1):
    """(radius, polar angle) kernel points of the inter ZP conv (functional.py:L42-61) -> [ks, 2]."""
    if isinstance(kernel_size, int):
        rrange = np.linspace(0, radius, kernel_size + 2, dtype=np.float32)[1:-1]
        kps = []
        for ri in range(kernel_size):
            for wi in get_angular_kernel_points_np(aperature, multiplier * ri + 1):
                kps.append([rrange[ri], wi])
    else:
        rrange = np.linspace(radius / kernel_size[0], radius, kernel_size[0], dtype=np.float32)
        wrange = get_angular_kernel_points_np(aperature, kernel_size[1])
        rr = np.tile(rrange[:, None, None], [1, wrange.shape[0], 1])
        ww = np.tile(wrange[None, :, None], [rrange.shape[0], 1, 1])
        kps = np.concatenate((rr, ww), axis=2).reshape(-1, 2)
    return np.array(kps).astype('float32')


# ------------------------------------------------------------------------------------------------
# inter ZP conv: ball + anchor weights + grouping (functional.py:L468-607)
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


def inter_zpconv_grouping_anchor(grouped_xyz, ball_idx, sample_idx, anchors, kernels, anchor_nn, n_support,
                                 radius, aperture, sigma):
    """S^2 kernel weights (functional.py:L503-573, the live "linear kernel" branch):
    grouped_xyz [b,3,p,nn], anchors [a,3], kernels [ks,2] = (radius, polar angle)
    -> inter_idx = ball_idx [b,p,nn], inter_w [b,p,a,ks,nn]."""
    norm = grouped_xyz.pow(2).sum(1).sqrt() + 1e-6                                   # [b,p,nn]
    cos_theta = (grouped_xyz.unsqueeze(3) * anchors.t()[:, None, :, None]).sum(1) / norm.unsqueeze(2)   # [b,p,a,nn]
    theta = acos_safe(cos_theta).unsqueeze(3)                                        # [b,p,a,1,nn]
    norm2 = norm[:, :, None, None, :]
    knorm2 = kernels[:, :1]
    theta2 = kernels[:, 1:]
    ratio = 3.0
    dist1 = (norm2 - knorm2).abs() + (norm2 * (theta - theta2)).abs() / ratio
    inter_w = F.relu(1.0 - dist1 / sigma ** 0.5, inplace=True)
    return ball_idx, inter_w


def inter_zpconv_grouping(xyz, feats, stride, n_neighbor, anchors, kernels, anchor_nn, radius, aperture, sigma,
                          inter_idx=None, inter_w=None, lazy_sample=True, radius_expansion=1.0):
    """functional.py:L576-607 -> inter_idx, inter_w, new_xyz, new_feats [b,c,ks,p,a]; the contraction runs in
    the HIP zpconv kernel (inter_zpconv_grouping_naive)."""
    if inter_idx is None:
        grouped_xyz, ball_idx, idx, new_xyz = inter_zpconv_grouping_ball(xyz, stride, radius * radius_expansion,
                                                                         n_neighbor, lazy_sample)
        inter_idx, inter_w = inter_zpconv_grouping_anchor(grouped_xyz, ball_idx, idx, anchors, kernels, anchor_nn,
                                                          xyz.shape[2], radius, aperture, sigma)
        inter_w = inter_w.contiguous().permute(0, 1, 3, 2, 4).contiguous()
    else:
        new_xyz = xyz
    feats = add_shadow_feature(feats)
    w = inter_w
    if w.shape[2] == 1 and feats.shape[3] != 1:      # the reference's einsum broadcasts a singleton anchor axis
        w = w.expand(-1, -1, feats.shape[3], -1, -1)
    new_feats = inter_zpconv_grouping_naive(inter_idx, w, feats)
    return inter_idx, inter_w, new_xyz, new_feats
