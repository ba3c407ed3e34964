"""Host logic of the per-part dense product (vgtk/so3conv/functional.py _pose_parts / _PoseParts): grouping the points of posed clouds
by their rotation, the per-slot point lists and column maps.  Torch ops only: runs on the CPU."""
import numpy as np
import torch


def _rot(rng, n):
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                     2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(n, 3, 3).astype(np.float32)


def _pose(part, R):
    b, p = part.shape
    pose = np.tile(np.eye(4, dtype=np.float32), (b, p, 1, 1))
    for bi in range(b):
        pose[bi, :, :3, :3] = R[bi][part[bi]]
    return torch.from_numpy(pose)


def test_parts_of_posed_clouds():
    import vgtk.so3conv.functional as L
    rng = np.random.default_rng(3)
    b, p = 3, 200
    part = np.zeros((b, p), np.int64)
    part[0] = rng.integers(0, 2, p)
    part[1] = rng.integers(0, 3, p)
    part[1, :5] = 3                                    # a tiny fourth part
    R = _rot(rng, b * 4).reshape(b, 4, 3, 3)
    pose = _pose(part, R)
    parts = L._pose_parts(pose)
    assert parts is not None and parts.n == 4 and not parts.single
    for bi in range(b):
        want = sorted(np.bincount(part[bi], minlength=4).tolist(), reverse=True)
        assert parts.sizes[bi] == want                                                 # largest part first, empty slots last
        lab = parts.labels[bi].numpy()
        for i in range(parts.n):                                                       # a slot = one rotation
            pts = np.nonzero(lab == i)[0]
            assert len(pts) == parts.sizes[bi][i]
            if len(pts):
                assert len(set(part[bi][pts].tolist())) == 1
                assert np.array_equal(parts.reps[bi, i].numpy(), R[bi][part[bi][pts[0]]])
    seen = torch.zeros(b, p, dtype=torch.int64)
    for i in range(parts.n):
        width = parts.width[i]
        assert width % 32 == 0 and width >= max(parts.sizes[bi][i] for bi in range(b)) and parts.pts[i].shape == (b, width)
        cm = parts.col_map[i]
        assert cm.dtype == torch.int32 and cm.shape == (b, width)
        for bi in range(b):
            n = parts.sizes[bi][i]
            assert bool((cm[bi, :n] >= 0).all()) and bool((cm[bi, n:] < 0).all())
            assert torch.equal(cm[bi, :n].long(), parts.pts[i][bi, :n])
            assert bool((parts.labels[bi][parts.pts[i][bi, :n]] == i).all())
            assert bool(((parts.pts[i][bi] >= 0) & (parts.pts[i][bi] < p)).all())      # padding entries name real points too
            seen[bi, parts.pts[i][bi, :n]] += 1
    assert bool((seen == 1).all())                                                     # every point in exactly one launch
    assert L._pose_parts(pose) is parts                                                # remembered per tensor ...
    pose[0, 0, 0, 3] += 1.0                                                            # ... and version (a translation edit: same parts, new object)
    again = L._pose_parts(pose)
    assert again is not parts and again.sizes == parts.sizes


def test_one_rotation_per_cloud_and_too_many():
    import vgtk.so3conv.functional as L
    rng = np.random.default_rng(4)
    b, p = 2, 64
    R = _rot(rng, b * 8).reshape(b, 8, 3, 3)
    single = L._pose_parts(_pose(np.zeros((b, p), np.int64), R))
    assert single is not None and single.single and single.n == 1 and single.sizes == [[p], [p]]
    many = np.zeros((b, p), np.int64)
    many[1, :L.DENSE_MAX_PARTS] = 1 + np.arange(L.DENSE_MAX_PARTS)                     # DENSE_MAX_PARTS + 1 rotations in one cloud
    assert L._pose_parts(_pose(many, R)) is None
    every = _pose(np.zeros((b, p), np.int64), R)
    every[:, :, :3, :3] = torch.from_numpy(_rot(rng, b * p).reshape(b, p, 3, 3))       # a rotation per point
    assert L._pose_parts(every) is None
