"""vgtk.cuda.grouping -- replaces the pybind module of vgtk/vgtk/cuda/grouping_cuda.cpp."""
import ctypes

import torch

from .. import _hip

_F32 = ctypes.c_float


def ball_query(new_xyz, xyz, radius, nsample):
    """(new_xyz T [B,3,M], xyz T [B,3,N], float radius, int nsample) -> int32 [B,M,nsample].
    grouping_cuda.cpp:L71-86."""
    _hip.check_input(new_xyz, xyz)
    if new_xyz.dtype != xyz.dtype:
        raise RuntimeError('ball_query: new_xyz and xyz must have the same dtype')
    b, _, m = new_xyz.shape
    n = xyz.shape[2]
    idx = torch.empty(b, m, nsample, dtype=torch.int32, device=xyz.device)
    _hip.call('eap_ball_query_' + _hip.suffix(xyz), xyz, b, n, m, _F32(radius), int(nsample),
              _hip._ptr(new_xyz), _hip._ptr(xyz), _hip._ptr(idx))
    return idx


def furthest_point_sampling(xyz, m):
    """(xyz T [B,3,N], int m) -> int32 [B,m]; grouping_cuda.cpp:L160-174."""
    _hip.check_input(xyz)
    if xyz.dtype != torch.float32:
        raise RuntimeError('furthest_point_sampling: float32 only')
    b, _, n = xyz.shape
    idx = torch.empty(b, m, dtype=torch.int32, device=xyz.device)
    temp = torch.empty(b, n, dtype=torch.float32, device=xyz.device)
    _hip.call('eap_furthest_point_sampling_f32', xyz, b, n, int(m), _hip._ptr(xyz), _hip._ptr(temp), _hip._ptr(idx))
    return idx


def anchor_query(sample_idx, grouped_idx, grouped_xyz, anchors, kernel_pts, nq):
    """-> [w T [B,P,A,K,NN]]; grouping_cuda.cpp:L88-108."""
    _hip.check_input(sample_idx, grouped_idx, grouped_xyz, anchors, kernel_pts)
    if grouped_xyz.dtype != torch.float32:
        raise RuntimeError('anchor_query: float32 only')
    b, _, np_, nn = grouped_xyz.shape
    na, ks = anchors.shape[0], kernel_pts.shape[0]
    w = torch.empty(b, np_, na, ks, nn, dtype=torch.float32, device=grouped_xyz.device)
    _hip.call('eap_anchor_query_f32', w, b, np_, nn, na, ks, _hip._ptr(grouped_xyz), _hip._ptr(anchors),
              _hip._ptr(kernel_pts), _hip._ptr(w))
    return [w]


def initial_anchor_query(centers, xyz, kernel_pts, radius, sigma):
    """-> [w, cnt] T [B,K,NC,A]; grouping_cuda.cpp:L138-158."""
    _hip.check_input(centers, xyz, kernel_pts)
    if centers.dtype != torch.float32:
        raise RuntimeError('initial_anchor_query: float32 only')
    b, _, nc = centers.shape
    m = xyz.shape[0]
    ks, na, _ = kernel_pts.shape
    w = torch.empty(b, ks, nc, na, dtype=torch.float32, device=centers.device)
    cnt = torch.empty_like(w)
    _hip.call('eap_initial_anchor_query_f32', w, b, nc, m, na, ks, _F32(radius), _F32(sigma),
              _hip._ptr(centers), _hip._ptr(xyz), _hip._ptr(kernel_pts), _hip._ptr(w), _hip._ptr(cnt))
    return [w, cnt]
