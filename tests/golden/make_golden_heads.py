"""tests/golden/make_golden_heads.py -- golden vectors of the pose head by RUNNING THE REFERENCE class
SO3OutBlockRTWithMaskSep (SPConvNets/models/model_utils.py:L363-677) on CPU in the build container
(through tests/golden/ref_import.py, like make_golden.py).  Stored: the module's state_dict after seeding, the
inputs, and the output dictionary in training mode (batch statistics, running-stat update) and in eval mode.
Data only.  Re-run with:  python tests/golden/make_golden_heads.py"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, '..', '..', 'equi-articulated-pose_amd'))
import ref_import  # noqa: E402

vgtk, sptk, L, zptk = ref_import.import_reference()
spec = importlib.util.spec_from_file_location('ref_model_utils', '/root/reference/SPConvNets/models/model_utils.py')
MU = importlib.util.module_from_spec(spec)
spec.loader.exec_module(MU)

B, C, N, A = 2, 16, 24, 60
CFG = dict(params={'dim_in': C, 'mlp': [32, 24], 'kanchor': A, 'temperature': 3.0}, norm=1, pooling_method='max', pred_axis=True,
           pred_pv_points=True, pred_central_points=True, num_heads=1, representation='quat')
torch.manual_seed(1234)
head = MU.SO3OutBlockRTWithMaskSep(**CFG)
with torch.no_grad():
    for m in head.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.5, 0.5)
            m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 2.0)
state = {k: v.clone() for k, v in head.state_dict().items()}
g = torch.Generator().manual_seed(77)
xyz = torch.randn(B, 3, N, generator=g) * 0.3
feats = torch.randn(B, C, N, A, generator=g)
trans_feats = torch.randn(B, C, N, A, generator=g)
mask = (torch.rand(B, N, generator=g) > 0.4).float()
anchors = torch.from_numpy(np.ascontiguousarray(L.get_anchors(A))).float().unsqueeze(0).repeat(B, 1, 1, 1)

out = {}
for mode in ('train', 'eval'):
    head.load_state_dict(state)
    head.train(mode == 'train')
    x = zptk.SphericalPointCloud(xyz, feats.clone(), None)
    with torch.no_grad():
        res = head(x, mask, trans_feats.clone(), trans_xyz=xyz, anchors=anchors)
    for k, v in res.items():
        out[f'{mode}_{k}'] = v.numpy().copy()
    if mode == 'train':
        for k, v in head.state_dict().items():
            if 'running' in k:
                out[f'after_{k}'] = v.numpy().copy()
arrs = {'xyz': xyz.numpy(), 'feats': feats.numpy(), 'trans_feats': trans_feats.numpy(), 'mask': mask.numpy(), 'anchors': anchors.numpy()}
arrs.update({f'state_{k}': v.numpy() for k, v in state.items()})
arrs.update(out)
path = os.path.join(HERE, 'pose_head.npz')
np.savez_compressed(path, **arrs)
print(f'pose_head.npz: {os.path.getsize(path) / 1024:.0f} KiB;', ', '.join(f'{k}{list(v.shape)}' for k, v in out.items()))
