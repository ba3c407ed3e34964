"""oracle/native.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy front-end (ctypes) for oracle/native_ops.c, the plain-C CPU restatement of
the reference's CUDA-only native operators.  Function names and argument order
mirror the reference's pybind entries (vgtk/vgtk/cuda/*.cpp) so tests read like
calls into the reference extension.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile liboracle_native.so with gcc (idempotent)."""
    so = os.path.join(_HERE, 'liboracle_native.so')
    src = os.path.join(_HERE, 'native_ops.c')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return 'f32'
    if dtype == np.float64:
        return 'f64'
    raise TypeError(f'unsupported dtype {dtype}')


def _c(a, dtype=None):
    return np.ascontiguousarray(a, dtype=dtype)


# grouping_cuda.cpp:L71-86
def ball_query(new_xyz, xyz, radius, nsample):
    new_xyz = _c(new_xyz); xyz = _c(xyz, new_xyz.dtype)
    b, _, m = new_xyz.shape
    n = xyz.shape[2]
    idx = np.zeros((b, m, nsample), np.int32)
    getattr(lib(), 'oracle_ball_query_' + _sfx(new_xyz.dtype))(
        b, n, m, ctypes.c_float(radius), nsample, _p(new_xyz), _p(xyz), _p(idx))
    return idx


# gathering_cuda.cpp:L29-43 (output dtype is always float32)
def gather_points_forward(pts, idx):
    pts = _c(pts, np.float32); idx = _c(idx, np.int32)
    b, c, n = pts.shape
    m = idx.shape[1]
    out = np.zeros((b, c, m), np.float32)
    lib().oracle_gather_points_fwd_f32(b, c, n, m, _p(pts), _p(idx), _p(out))
    return out


# gathering_cuda.cpp:L45-60
def gather_points_backward(grad_out, idx, npoint):
    grad_out = _c(grad_out); idx = _c(idx, np.int32)
    b, c, m = grad_out.shape
    out = np.zeros((b, c, npoint), grad_out.dtype)
    getattr(lib(), 'oracle_gather_points_bwd_' + _sfx(grad_out.dtype))(
        b, c, npoint, m, _p(grad_out), _p(idx), _p(out))
    return out


# zpconv_cuda.cpp:L41-56
def inter_zpconv_forward(idx, w, feats):
    idx = _c(idx, np.int32); w = _c(w); feats = _c(feats, w.dtype)
    b, np_, na, ks, ann = idx.shape
    _, c, nq, _ = feats.shape
    out = np.zeros((b, c, ks, np_, na), w.dtype)
    getattr(lib(), 'oracle_inter_zpconv_fwd_' + _sfx(w.dtype))(
        b, np_, nq, na, ks, ann, c, _p(idx), _p(w), _p(feats), _p(out))
    return out


# zpconv_cuda.cpp:L58-75
def inter_zpconv_backward(idx, w, grad, npoint):
    idx = _c(idx, np.int32); w = _c(w); grad = _c(grad, w.dtype)
    b, np_, na, ks, ann = idx.shape
    c = grad.shape[1]
    out = np.zeros((b, c, npoint, na), w.dtype)
    getattr(lib(), 'oracle_inter_zpconv_bwd_' + _sfx(w.dtype))(
        b, np_, npoint, na, ks, ann, c, _p(idx), _p(w), _p(grad), _p(out))
    return out


# zpconv_cuda.cpp:L77-92
def intra_zpconv_forward(idx, w, feats):
    idx = _c(idx, np.int32); w = _c(w); feats = _c(feats, w.dtype)
    na_out, ann = idx.shape
    ks = w.shape[1]
    b, c, np_, na_in = feats.shape
    out = np.zeros((b, c, ks, np_, na_out), w.dtype)
    getattr(lib(), 'oracle_intra_zpconv_fwd_' + _sfx(w.dtype))(
        b, np_, na_in, na_out, ks, ann, c, _p(idx), _p(w), _p(feats), _p(out))
    return out


# zpconv_cuda.cpp:L94-110
def intra_zpconv_backward(idx, w, grad, anchor_in):
    idx = _c(idx, np.int32); w = _c(w); grad = _c(grad, w.dtype)
    na_out, ann = idx.shape
    ks = w.shape[1]
    b, c, _, np_, _ = grad.shape
    out = np.zeros((b, c, np_, anchor_in), w.dtype)
    getattr(lib(), 'oracle_intra_zpconv_bwd_' + _sfx(w.dtype))(
        b, np_, anchor_in, na_out, ks, ann, c, _p(idx), _p(w), _p(grad), _p(out))
    return out


# grouping_cuda.cpp:L160-174
def furthest_point_sampling(xyz, m):
    xyz = _c(xyz)
    b, _, n = xyz.shape
    idx = np.zeros((b, m), np.int32)
    getattr(lib(), 'oracle_fps_' + _sfx(xyz.dtype))(b, n, m, _p(xyz), _p(idx))
    return idx


# chamfer_cuda.cpp:L22-25 / chamfer.cu:L147-171
def chamfer_forward(xyz1, xyz2):
    xyz1 = _c(xyz1, np.float32); xyz2 = _c(xyz2, np.float32)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1 = np.zeros((b, n), np.float32); i1 = np.zeros((b, n), np.int32)
    d2 = np.zeros((b, m), np.float32); i2 = np.zeros((b, m), np.int32)
    lib().oracle_chamfer_nn_f32(b, n, _p(xyz1), m, _p(xyz2), _p(d1), _p(i1))
    lib().oracle_chamfer_nn_f32(b, m, _p(xyz2), n, _p(xyz1), _p(d2), _p(i2))
    return d1, d2, i1, i2


# chamfer_cuda.cpp:L27-34 / chamfer.cu:L203-231
def chamfer_backward(xyz1, xyz2, idx1, idx2, g1, g2):
    xyz1 = _c(xyz1, np.float32); xyz2 = _c(xyz2, np.float32)
    idx1 = _c(idx1, np.int32); idx2 = _c(idx2, np.int32)
    g1 = _c(g1, np.float32); g2 = _c(g2, np.float32)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    gx1 = np.zeros_like(xyz1); gx2 = np.zeros_like(xyz2)
    lib().oracle_chamfer_grad_f32(b, n, _p(xyz1), m, _p(xyz2), _p(g1), _p(idx1), _p(gx1), _p(gx2))
    lib().oracle_chamfer_grad_f32(b, m, _p(xyz2), n, _p(xyz1), _p(g2), _p(idx2), _p(gx2), _p(gx1))
    return gx1, gx2


# grouping_cuda.cpp:L88-108 (output initialised to 1e6, then fully overwritten)
def anchor_query(sample_idx, grouped_idx, grouped_xyz, anchors, kernel_pts, nq):
    grouped_xyz = _c(grouped_xyz)
    anchors = _c(anchors, grouped_xyz.dtype); kernel_pts = _c(kernel_pts, grouped_xyz.dtype)
    b, _, np_, nn = grouped_xyz.shape
    na, ks = anchors.shape[0], kernel_pts.shape[0]
    w = np.full((b, np_, na, ks, nn), 1e6, grouped_xyz.dtype)
    getattr(lib(), 'oracle_anchor_query_' + _sfx(grouped_xyz.dtype))(
        b, np_, nn, na, ks, _p(grouped_xyz), _p(anchors), _p(kernel_pts), _p(w))
    return [w]


# grouping_cuda.cpp:L138-158
def initial_anchor_query(centers, xyz, kernel_pts, radius, sigma):
    centers = _c(centers); xyz = _c(xyz, centers.dtype); kernel_pts = _c(kernel_pts, centers.dtype)
    b, _, nc = centers.shape
    m = xyz.shape[0]
    ks, na, _ = kernel_pts.shape
    w = np.zeros((b, ks, nc, na), centers.dtype)
    cnt = np.zeros_like(w)
    getattr(lib(), 'oracle_initial_anchor_query_' + _sfx(centers.dtype))(
        b, nc, m, na, ks, ctypes.c_float(radius), ctypes.c_float(sigma),
        _p(centers), _p(xyz), _p(kernel_pts), _p(w), _p(cnt))
    return [w, cnt]
