"""vgtk.cuda.gathering -- replaces the pybind module of vgtk/vgtk/cuda/gathering_cuda.cpp."""
import torch

from .. import _hip


def gather_points_forward(pts, idx):
    """(pts T [B,C,N], idx int32 [B,M]) -> float32 [B,C,M]; gathering_cuda.cpp:L29-43
    (the reference allocates a Float output whatever the input dtype)."""
    _hip.check_input(pts, idx)
    if pts.dtype != torch.float32:
        raise RuntimeError('gather_points_forward: float32 points only')
    if idx.dtype != torch.int32:
        raise RuntimeError('gather_points_forward: idx must be int32')
    b, c, n = pts.shape
    m = idx.shape[1]
    out = torch.empty(b, c, m, dtype=torch.float32, device=pts.device)
    _hip.call('eap_gather_points_fwd_f32', out, b, c, n, m, _hip._ptr(pts), _hip._ptr(idx), _hip._ptr(out))
    return out


def gather_points_backward(grad_out, idx, npoint):
    """(grad T [B,C,M], idx int32 [B,M], int npoint) -> T [B,C,npoint]; gathering_cuda.cpp:L45-60."""
    _hip.check_input(grad_out, idx)
    if idx.dtype != torch.int32:
        raise RuntimeError('gather_points_backward: idx must be int32')
    b, c, m = grad_out.shape
    out = torch.empty(b, c, npoint, dtype=grad_out.dtype, device=grad_out.device)
    _hip.call('eap_gather_points_bwd_' + _hip.suffix(grad_out), out, b, c, int(npoint), m,
              _hip._ptr(grad_out), _hip._ptr(idx), _hip._ptr(out))
    return out
