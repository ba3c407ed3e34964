"""Top-level `chamfer` module -- replaces the reference's compiled pybind module of the same name
(extensions/chamfer_dist/chamfer_cuda.cpp:L36-39: `forward`, `backward`) with calls into
libeap_hip.so."""
import torch

from vgtk import _hip


def forward(xyz1, xyz2):
    """(xyz1 f32 [B,n,3], xyz2 f32 [B,m,3]) -> [dist1 [B,n], dist2 [B,m], idx1 i32 [B,n], idx2 i32 [B,m]]"""
    _hip.check_input(xyz1, xyz2)
    if xyz1.dtype != torch.float32 or xyz2.dtype != torch.float32:
        raise RuntimeError('chamfer.forward: float32 only (as the reference)')
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dev = xyz1.device
    d1 = torch.empty(b, n, dtype=torch.float32, device=dev)
    d2 = torch.empty(b, m, dtype=torch.float32, device=dev)
    i1 = torch.empty(b, n, dtype=torch.int32, device=dev)
    i2 = torch.empty(b, m, dtype=torch.int32, device=dev)
    _hip.call('eap_chamfer_fwd_f32', xyz1, b, n, m, _hip._ptr(xyz1), _hip._ptr(xyz2), _hip._ptr(d1), _hip._ptr(d2),
              _hip._ptr(i1), _hip._ptr(i2))
    return [d1, d2, i1, i2]


def backward(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2):
    """-> [grad_xyz1 [B,n,3], grad_xyz2 [B,m,3]]"""
    grad_dist1 = grad_dist1.contiguous()
    grad_dist2 = grad_dist2.contiguous()
    _hip.check_input(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = torch.empty_like(xyz1)
    g2 = torch.empty_like(xyz2)
    _hip.call('eap_chamfer_bwd_f32', xyz1, b, n, m, _hip._ptr(xyz1), _hip._ptr(xyz2), _hip._ptr(idx1), _hip._ptr(idx2),
              _hip._ptr(grad_dist1), _hip._ptr(grad_dist2), _hip._ptr(g1), _hip._ptr(g2))
    return [g1, g2]
