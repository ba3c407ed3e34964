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
