"""Index-producing point-cloud operators (reference: vgtk/vgtk/pc/sample.py:L46-77)."""
import torch

from ..cuda import grouping as _native
from ..utils import batch_gather

__all__ = ['group_nd', 'ball_query_index', 'furthest_sample_index', 'furthest_sample']


def group_nd(pc, idx):
    """[b,c,n] x [b,m1(,m2,...)] -> [b,c,m1(,m2,...)]  (sample.py:L46-50)."""
    flat = idx.reshape(idx.shape[0], -1).contiguous()
    return batch_gather(pc, flat, dim=2).view(idx.shape[0], -1, *idx.shape[1:])


def ball_query_index(query_points, support_points, radius, n_sample):
    """[b,3,m] x [b,3,n] x r x k -> int32 [b,m,k]  (sample.py:L54-59)."""
    return _native.ball_query(query_points.contiguous(), support_points.contiguous(), radius, n_sample)


def furthest_sample_index(pc, n_sample, lazy_sample):
    """[b,3,n] -> int32 [b,m]  (sample.py:L63-72): the first n_sample points when sampling is lazy or would keep every
    point, else the native furthest-point sampler."""
    if lazy_sample or n_sample == pc.shape[2]:
        first = torch.arange(n_sample, device=pc.device, dtype=torch.int32)
        return first.unsqueeze(0).repeat(pc.shape[0], 1)
    return _native.furthest_point_sampling(pc.contiguous(), n_sample)


def furthest_sample(pc, n_sample, lazy_sample=True):
    """-> (indices [b,m], the sampled points [b,3,m])."""
    chosen = furthest_sample_index(pc, n_sample, lazy_sample)
    return chosen, group_nd(pc, chosen)
