"""Index-producing point-cloud operators (reference: vgtk/vgtk/pc/sample.py:L46-77)."""
import torch

import vgtk.cuda.grouping as cuda_nn
import vgtk.utils as utils

__all__ = ['group_nd', 'ball_query_index', 'furthest_sample_index', 'furthest_sample']


def group_nd(pc, idx):
    """[b,c,n] x [b,m1(,m2,...)] -> [b,c,m1(,m2,...)]  (sample.py:L46-50)."""
    b = idx.shape[0]
    out = utils.batch_gather(pc, idx.view(b, -1).contiguous(), dim=2)
    return out.view(b, -1, *idx.shape[1:])


def ball_query_index(query_points, support_points, radius, n_sample):
    """[b,3,m] x [b,3,n] x r x k -> int32 [b,m,k]  (sample.py:L54-59)."""
    return cuda_nn.ball_query(query_points.contiguous(), support_points.contiguous(), radius, n_sample)


def furthest_sample_index(pc, n_sample, lazy_sample):
    """[b,3,n] -> int32 [b,m]  (sample.py:L63-72)."""
    if pc.shape[2] == n_sample or lazy_sample:
        nb = pc.shape[0]
        return torch.arange(n_sample, device=pc.device).view(1, -1).expand(nb, -1).int().contiguous()
    return cuda_nn.furthest_point_sampling(pc.contiguous(), n_sample)


def furthest_sample(pc, n_sample, lazy_sample=True):
    idx = furthest_sample_index(pc, n_sample, lazy_sample)
    return idx, group_nd(pc, idx)
