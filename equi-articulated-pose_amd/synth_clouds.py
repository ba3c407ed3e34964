"""Synthetic 'laptop' point clouds (SURVEY.md section 8d / BASELINE.md section 3).

Two thin boxes (base 0.5 x 0.02 x 0.35 and an identical lid hinged along the back
edge, opening angle ~ U(30, 120) degrees), points sampled uniformly on the box
surfaces, centred to zero mean (cf. the reference's
SPConvNets/datasets/MotionDataset.py:L583-589), then a Haar-random global
rotation.  Labels are 0 (base) / 1 (lid); per-point pose is the identity, which
is what the shipped model feeds the convolution
(...pn_38_multi_stage.py:L2023-2026).  Cloud i uses seed 2913 + i (2913 is the
reference's default seed, SPConvNets/options.py:L17).
"""
import math

import numpy as np

BASE_SEED = 2913
BOX = (0.5, 0.02, 0.35)


def _box_surface(rng, n, size):
    sx, sy, sz = size
    areas = np.array([sy * sz, sy * sz, sx * sz, sx * sz, sx * sy, sx * sy])
    face = rng.choice(6, size=n, p=areas / areas.sum())
    u = rng.random((n, 3)) * np.array(size)
    axis = face // 2
    side = face % 2
    u[np.arange(n), axis] = side * np.array(size)[axis]
    return u


def _haar(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _visible_subset(rng, pts, label, n_points, cam=(0.0, 0.0, -1.8), yfov=math.radians(60.0), ph=480, cell=2, tol=0.012):
    """Depth-buffer visibility from a pinhole camera at `cam` looking down +z -- the synthetic counterpart of the reference's
    rendered partial views (SPConvNets/datasets/MotionHOIDatasetPartial.py:L134 create_partial_pts: camera mean pose (0, 0, -1.8),
    yfov 60 degrees, 640 x 480 depth image back-projected to points): the raw surface samples are splatted into cells of
    `cell` x `cell` pixels, a sample survives if its depth is within `tol` of its cell's nearest sample (so only the surfaces
    facing the camera remain), and `n_points` of the survivors are drawn (without replacement when there are enough)."""
    f = 0.5 * ph / math.tan(0.5 * yfov)
    rel = pts - np.asarray(cam)
    z = rel[:, 2]
    u = np.floor(rel[:, 0] / z * f / cell).astype(np.int64)
    v = np.floor(rel[:, 1] / z * f / cell).astype(np.int64)
    key = (u - u.min()) * (v.max() - v.min() + 1) + (v - v.min())
    order = np.argsort(key, kind='stable')
    ks, zs = key[order], z[order]
    first = np.r_[True, ks[1:] != ks[:-1]]
    zmin = np.minimum.reduceat(zs, np.flatnonzero(first))[np.cumsum(first) - 1]
    vis = order[zs <= zmin + tol]
    vis.sort()
    sel = rng.choice(vis, size=n_points, replace=len(vis) < n_points)
    return pts[sel], label[sel]


def laptop_cloud(index, n_points, partial=False):
    """-> (xyz float32 [3,N], label int64 [N], pose float32 [N,4,4])."""
    rng = np.random.default_rng(BASE_SEED + index)
    n_raw = n_points * 8 if partial else n_points
    n_base = n_raw // 2
    base = _box_surface(rng, n_base, BOX)
    lid = _box_surface(rng, n_raw - n_base, BOX)
    ang = math.radians(rng.uniform(30.0, 120.0))
    # hinge = back edge (z = 0) on top of the base; open the lid about the x axis
    c, s = math.cos(ang), math.sin(ang)
    ly, lz = lid[:, 1].copy(), lid[:, 2].copy()
    lid[:, 1] = BOX[1] + c * ly + s * lz
    lid[:, 2] = -s * ly + c * lz
    pts = np.concatenate([base, lid], 0)
    label = np.concatenate([np.zeros(n_base, np.int64), np.ones(n_raw - n_base, np.int64)])
    pts = pts - pts.mean(0, keepdims=True)
    pts = pts @ _haar(rng).T
    if partial:
        pts, label = _visible_subset(rng, pts, label, n_points)
        pts = pts - pts.mean(0, keepdims=True)
    perm = rng.permutation(n_points)
    pts, label = pts[perm], label[perm]
    pose = np.tile(np.eye(4, dtype=np.float32), (n_points, 1, 1))
    return np.ascontiguousarray(pts.T.astype(np.float32)), label, pose


def laptop_batch(start, batch, n_points, partial=False):
    """-> xyz [B,3,N] float32, label [B,N] int64, pose [B,N,4,4] float32."""
    xs, ls, ps = zip(*[laptop_cloud(start + i, n_points, partial) for i in range(batch)])
    return np.stack(xs), np.stack(ls), np.stack(ps)


# Layer hyper-parameters produced by the reference's build_model
# (...pn_38_multi_stage.py:L2089-2092, L2115-2126, L2146-2163, L2174-2191)
def backbone_layers(input_num):
    """[(C_in, C_out, radius, sigma)] for the 3-block inter backbone; NN=64, K=24, A=60."""
    init_r, input_radius, sigma_ratio = 0.2, 0.4, 0.5
    strides = [2, 2, 2, 2]
    if input_num > 1024:
        strides[0] = int(2 * (input_num / 1024))
    mult = [1]
    for i in range(3):
        mult.append(mult[-1] * strides[i])
    radii = [init_r * m ** 0.5 * input_radius for m in mult]
    sig = [sigma_ratio * radii[0] ** 2]
    for s in strides:
        sig.append(sig[-1] * s)
    chans = [1, 64, 128, 512]
    out = []
    for i in range(3):
        nidx = i if i == 0 else i + 1
        out.append((chans[i], chans[i + 1], radii[nidx], sig[nidx]))
    return out
