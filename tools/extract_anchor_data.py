"""tools/extract_anchor_data.py -- run once in the build container.

Reads the reference's geometric DATA files (vgtk/vgtk/data/anchors/*.ply: the unit
icosahedron and the kernel-point sets) and stores the raw numbers in
equi-articulated-pose_amd/vgtk/data/anchors/constants.npz.  Data only -- the 60
anchors, the 60x12 intra index and the scaled kernel points are derived from
these numbers by the package's own code (vgtk/functional/rotation.py,
vgtk/so3conv/functional.py).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', 'tests', 'golden', '_ref_shims'))
from plyfile import PlyData  # noqa: E402

REF = '/root/reference/vgtk/vgtk/data/anchors'
OUT = os.path.join(HERE, '..', 'equi-articulated-pose_amd', 'vgtk', 'data', 'anchors', 'constants.npz')


def verts(name):
    v = PlyData.read(os.path.join(REF, name))['vertex']
    return np.vstack([v['x'], v['y'], v['z']]).T.astype(np.float32)


ico = PlyData.read(os.path.join(REF, 'sphere12.ply'))
np.savez(OUT,
         sphere12_vertices=verts('sphere12.ply'),
         sphere12_faces=np.vstack(ico['face']['vertex_indices']).astype(np.int32),
         kpsphere24=verts('kpsphere24.ply'),
         kpsphere30=verts('kpsphere30.ply'),
         kpsphere66=verts('kpsphere66.ply'),
         # S^2 anchor sets of the ZP convolution (vgtk.spconv.functional.get_anchors(int): the vertices with
         # norm > 0.5, normalised); sphere12_vertices doubles as the 12-anchor set
         sphere42_vertices=verts('sphere42.ply'),
         sphere92_vertices=verts('sphere92.ply'),
         sphere162_vertices=verts('sphere162.ply'))
d = np.load(OUT)
for k in d.files:
    print(k, d[k].shape, d[k].dtype)
