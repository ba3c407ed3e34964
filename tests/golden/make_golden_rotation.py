"""tests/golden/make_golden_rotation.py -- known answers of the reference's angle -> rotation helpers
(SPConvNets/models/model_utils.py:L954-1043: from_rotation_mtx_to_axis, compute_rotation_matrix_from_angle), produced
by RUNNING them on CPU in the build container.  Inputs: the 60 icosahedral anchors (which contain the identity and the
15 half-turns, i.e. every branch of the axis extraction), seeded angles, and a seeded non-unit `defined_axis`.
Data only.  Re-run:  python tests/golden/make_golden_rotation.py"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

vgtk, sptk, L, zptk = ref_import.import_reference()
spec = importlib.util.spec_from_file_location('ref_model_utils', '/root/reference/SPConvNets/models/model_utils.py')
MU = importlib.util.module_from_spec(spec)
spec.loader.exec_module(MU)

g = torch.Generator().manual_seed(31)
anchors = torch.from_numpy(np.ascontiguousarray(L.get_anchors(60))).float()
angles = (torch.rand(5, 60, 1, generator=g) * 2.0 - 1.0) * np.pi
axis_shared = torch.randn(1, 3, generator=g)
axis_shared = axis_shared / axis_shared.norm() * 1.0003          # unit only to rounding, as a predicted axis is
axis_full = torch.nn.functional.normalize(torch.randn(5, 60, 3, generator=g), dim=-1)
out = {'anchors': anchors.numpy(), 'angles': angles.numpy(), 'axis_shared': axis_shared.numpy(), 'axis_full': axis_full.numpy(),
       'anchor_axes': MU.from_rotation_mtx_to_axis(anchors).numpy(),
       'R_anchor_axes': MU.compute_rotation_matrix_from_angle(anchors, angles).numpy(),
       'R_shared': MU.compute_rotation_matrix_from_angle(anchors, angles, defined_axis=axis_shared).numpy(),
       'R_full': MU.compute_rotation_matrix_from_angle(anchors, angles, defined_axis=axis_full).numpy()}
tr = (np.trace(out['anchors'], axis1=1, axis2=2) - 1) / 2
print('identity anchors:', int((np.abs(tr - 1) < 1e-8).sum()), ' half-turn anchors (|tr+1|<1e-8):', int((np.abs(tr + 1) < 1e-8).sum()),
      ' near half-turns (<1e-5):', int((np.abs(tr + 1) < 1e-5).sum()))
path = os.path.join(HERE, 'rotation.npz')
np.savez_compressed(path, **out)
print(f'rotation.npz: {os.path.getsize(path) / 1024:.0f} KiB')
