"""Minimal stand-in for `trimesh` (pinned 3.2.0 by the reference, absent here).

Test infrastructure only (see plyfile shim).  Exposes what
vgtk/vgtk/functional/rotation.py:L117-125, L240-243 touches: ``load(path)`` →
object with ``faces``, ``face_normals``, ``face_adjacency``, ``fix_normals()``.

Pinned vs. unpinned: ``faces`` (file order) and ``face_normals`` are fully
determined by the PLY (all faces of sphere12.ply are already outward-wound, so
``fix_normals`` is a no-op).  The ROW ORDER of ``face_adjacency`` is a trimesh
internal; here rows are ordered by the shared edge's sorted vertex pair
(v_lo, v_hi), each pair ascending.  That choice fixes only the column order of
the 60x12 intra index ("intra column order: parity unpinned").
"""
import numpy as np
from plyfile import PlyData


class _Mesh:
    def __init__(self, vertices, faces):
        self.vertices = np.asarray(vertices, dtype=np.float64)
        self.faces = np.asarray(faces, dtype=np.int64)

    def fix_normals(self):
        v = self.vertices
        for i, f in enumerate(self.faces):
            n = np.cross(v[f[1]] - v[f[0]], v[f[2]] - v[f[0]])
            if np.dot(n, v[f].mean(0)) < 0:
                self.faces[i] = f[::-1]

    @property
    def face_normals(self):
        v = self.vertices
        f = self.faces
        n = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
        return n / np.linalg.norm(n, axis=1, keepdims=True)

    @property
    def face_adjacency(self):
        edges = {}
        for fi, f in enumerate(self.faces):
            for a, b in ((f[0], f[1]), (f[1], f[2]), (f[2], f[0])):
                key = (min(a, b), max(a, b))
                edges.setdefault(key, []).append(fi)
        rows = []
        for key in sorted(edges):
            fs = edges[key]
            if len(fs) == 2:
                rows.append(sorted(fs))
        return np.asarray(rows, dtype=np.int64)


def load(path, **kwargs):
    ply = PlyData.read(path)
    v = ply['vertex']
    vertices = np.vstack([v['x'], v['y'], v['z']]).T
    faces = np.vstack(ply['face']['vertex_indices'])
    return _Mesh(vertices, faces)
