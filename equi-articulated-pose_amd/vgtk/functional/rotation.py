"""Icosahedral SO(3) anchors, the 60x12 intra-conv index and torch rotation helpers.

Mirrors the public surface of the reference's vgtk/vgtk/functional/rotation.py
(`icosahedron_so3_trimesh` L236-343 and the torch helpers L379-518) without the
trimesh dependency: the unit icosahedron (12 vertices / 20 outward-wound faces,
the numbers of vgtk/vgtk/data/anchors/sphere12.ply) ships as data in
vgtk/data/anchors/constants.npz and everything else is derived here.

Conventions reproduced from the reference (checked against fixtures generated
by importing the reference, tests/golden/constants.npz):
  * anchor 3f+g  = Rx(gamma_g + off_f) . Ry(beta_f) . Rz(alpha_f), with
    sin(beta) = n_z, (cos, sin)(alpha) = (n_x, n_y)/cos(beta), gamma_g = -2 pi g/3
    and a 60 degree in-plane offset for the face rings at n_z ~ -0.19 and ~ +0.79
    (rotation.py:L141-219);
  * every anchor is right-multiplied by anchor_29^T, so anchor 29 is the
    identity (rotation.py:L257);
  * intra_idx[n, k] = index of A_n . A_0 . A_{nbr_k}^T where nbr_0..11 are the 9
    anchors of the 3 faces adjacent to face 0 (face-major within each in-plane
    step) followed by the 3 anchors of face 0 itself (rotation.py:L117-139,
    L258-306).  The ORDER of the 3 adjacent faces comes from trimesh's
    `face_adjacency` row order, which is not pinned by anything in the
    reference repo; this file orders adjacent faces by the shared edge's sorted
    vertex pair ("intra column order: parity unpinned").
"""
import os

import numpy as np
import torch

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'data', 'anchors', 'constants.npz')


def _icosahedron():
    d = np.load(_DATA)
    return d['sphere12_vertices'].astype(np.float64), d['sphere12_faces'].astype(np.int64)


def _face_normals(v, f):
    n = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    return n / np.linalg.norm(n, axis=1, keepdims=True)


def _face_neighbours(f):
    """For every face, its 3 edge-adjacent faces ordered by the shared edge (v_lo, v_hi)."""
    by_edge = {}
    for fi, tri in enumerate(f):
        for a, b in ((tri[0], tri[1]), (tri[1], tri[2]), (tri[2], tri[0])):
            by_edge.setdefault((min(a, b), max(a, b)), []).append(fi)
    nbrs = [[] for _ in range(len(f))]
    for edge in sorted(by_edge):
        fa, fb = sorted(by_edge[edge])
        nbrs[fa].append(fb)
        nbrs[fb].append(fa)
    return np.asarray(nbrs, dtype=np.int64)


def _euler_frames(normals, gsize):
    """[nf*gsize, 3, 3] float32 frames Rx(gamma) Ry(beta) Rz(alpha) per (face, in-plane step)."""
    nf = normals.shape[0]
    sb = normals[:, 2]
    cb = np.sqrt(1.0 - sb ** 2)
    ca = normals[:, 0] / cb
    sa = normals[:, 1] / cb
    gamma = -np.linspace(0, 2 * np.pi, gsize, endpoint=False, dtype=np.float32)
    # rings at n_z ~ -0.19 and ~ +0.79 carry a 60 degree in-plane offset
    shifted = (np.abs(sb + 0.19) < 0.01) | (np.abs(sb - 0.79) < 0.01)
    plain = (np.abs(sb + 0.79) < 0.01) | (np.abs(sb - 0.19) < 0.01)
    out = np.zeros((nf, gsize, 3, 3), dtype=np.float32)
    for fi in range(nf):
        Rz = np.array([[ca[fi], sa[fi], 0], [-sa[fi], ca[fi], 0], [0, 0, 1]], dtype=np.float32)
        Ry = np.array([[cb[fi], 0, sb[fi]], [0, 1, 0], [-sb[fi], 0, cb[fi]]], dtype=np.float32)
        for g in range(gsize):
            gm = gamma[g] + np.float32(60 / 180 * np.pi) if shifted[fi] else gamma[g]
            c, s = np.cos(np.float32(gm)), np.sin(np.float32(gm))
            Rx = np.array([[1, 0, 0], [0, c, s], [0, -s, c]], dtype=np.float32)
            if shifted[fi] or plain[fi]:
                out[fi, g] = np.einsum('ij,jh->ih', np.einsum('ij,jh->ih', Rx, Ry), Rz)
    return out.reshape(nf * gsize, 3, 3)


def _nearest_anchor(R, anchors):
    """argmax_j tr(R A_j^T) for R [...,3,3]."""
    return np.argmax(np.einsum('...ij,cij->...c', R, anchors), axis=-1)


_CACHE = {}


def icosahedron_so3(gsize=3):
    """-> (anchors float32 [20*gsize,3,3], intra_idx int64 [20*gsize, 9+gsize], None)."""
    if gsize in _CACHE:
        return _CACHE[gsize]
    v, f = _icosahedron()
    normals = _face_normals(v, f)
    frames = _euler_frames(normals, gsize)
    anchors = np.einsum('bij,kj', frames, frames[29]).astype(np.float32)

    nbr_faces = _face_neighbours(f)[0]                       # faces adjacent to face 0
    nbr = [int(nf) * gsize + g for g in range(gsize) for nf in nbr_faces]
    nbr += list(range(gsize))                                # face 0's own in-plane steps
    A = anchors.astype(np.float64)
    target = np.einsum('nij,jk,mlk->nmil', A, A[0], A[nbr])  # A_n A_0 A_nbr^T
    intra_idx = _nearest_anchor(target, A).astype(np.int64)
    _CACHE[gsize] = (anchors, intra_idx, None)
    return _CACHE[gsize]


def icosahedron_so3_trimesh(mesh_path=None, gsize=3, use_quats=False):
    """Signature-compatible alias of the reference entry (rotation.py:L236); the mesh
    path is ignored -- the icosahedron ships as data."""
    if use_quats:
        raise NotImplementedError('quaternion anchors are not used by the conv path')
    return icosahedron_so3(gsize)


def anchor_group_tables(anchors):
    """Multiplication / inverse tables of the anchor group.

    mult[g, a] = index of A_g . A_a ;  inv[g] = index of A_g^T.
    Used by the HIP conv to turn the reference's per-(point, neighbour) 60x60
    trace arg-max (so3conv/functional.py:L1199-1204) into one nearest-anchor
    search plus a table lookup.
    """
    A = np.asarray(anchors, dtype=np.float64)
    prod = np.einsum('gij,ajk->gaik', A, A)
    mult = _nearest_anchor(prod, A).astype(np.uint8)
    inv = _nearest_anchor(np.transpose(A, (0, 2, 1)), A).astype(np.uint8)
    return mult, inv


# ---------------------------------------------------------------------------
# torch helpers used by SPConvNets (vgtk.functional.{compute_rotation_matrix_from_*, so3_mean})
# ---------------------------------------------------------------------------
def _normalize(v):
    mag = torch.sqrt(v.pow(2).sum(1)).clamp_min(1e-8)
    return v / mag[:, None]


def compute_rotation_matrix_from_quaternion(quaternion):
    """[B,4] (w,x,y,z) -> [B,3,3]; rotation.py:L379-416."""
    q = _normalize(quaternion)
    qw, qx, qy, qz = q[:, 0:1], q[:, 1:2], q[:, 2:3], q[:, 3:4]
    xx, yy, zz = qx * qx, qy * qy, qz * qz
    xy, xz, yz = qx * qy, qx * qz, qy * qz
    xw, yw, zw = qx * qw, qy * qw, qz * qw
    row0 = torch.cat((1 - 2 * yy - 2 * zz, 2 * xy - 2 * zw, 2 * xz + 2 * yw), 1)
    row1 = torch.cat((2 * xy + 2 * zw, 1 - 2 * xx - 2 * zz, 2 * yz - 2 * xw), 1)
    row2 = torch.cat((2 * xz - 2 * yw, 2 * yz + 2 * xw, 1 - 2 * xx - 2 * yy), 1)
    return torch.stack((row0, row1, row2), 1)


def compute_rotation_matrix_from_ortho6d(ortho6d):
    """[B,6] -> [B,3,3] with columns (x, y, z); rotation.py:L443-478."""
    x = _normalize(ortho6d[:, 0:3])
    z = _normalize(torch.cross(x, ortho6d[:, 3:6], dim=1))
    y = torch.cross(z, x, dim=1)
    return torch.stack((x, y, z), 2)


def so3_mean(Rs, weights=None):
    """Chordal L2 mean of rotations [B,N,3,3] -> [B,3,3]; rotation.py:L481-518."""
    w = 1.0 if weights is None else weights[:, :, None, None]
    Ce = torch.sum(w * Rs, dim=1)
    cu, _, cv = torch.svd(Ce)
    cvT = cv.transpose(1, 2).contiguous()
    dets = torch.det(torch.matmul(cu, cvT))
    D = torch.diag_embed(torch.stack((torch.ones_like(dets), torch.ones_like(dets), dets), 1))
    return torch.einsum('bij,bjk,bkl->bil', cu, D, cvT)
