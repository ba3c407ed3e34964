"""tests/golden/ref_import.py -- import the reference's Python operator path on CPU.

Build-container only (needs /root/reference).  Follows SURVEY.md Appendix B:
  1. sys.path = [_ref_shims, /root/reference/vgtk]; np.float alias; no-op
     torch.cuda.synchronize.
  2. a synthetic `vgtk.cuda` package whose `grouping.ball_query` and
     `gathering.gather_points_forward` are the CPU oracle (oracle/native.py) --
     the only two native calls on the live path; `zpconv` also points at the
     oracle so the autograd wrappers can be exercised.
  3. plyfile / trimesh shims from tests/golden/_ref_shims.
Used ONLY by make_golden.py (fixture generation) and by the optional
`-m refimport` tests; never by product code, never on the GPU box.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..'))
REF_VGTK = '/root/reference/vgtk'


def available():
    return os.path.isdir(REF_VGTK)


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def import_reference():
    """Returns the reference modules (vgtk, vgtk.so3conv, vgtk.so3conv.functional, vgtk.spconv)."""
    if 'vgtk' in sys.modules and not getattr(sys.modules['vgtk'], '_is_reference', False):
        raise RuntimeError('a non-reference `vgtk` is already imported in this process')
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    from oracle import native

    sys.path.insert(0, REF_VGTK)
    sys.path.insert(0, os.path.join(HERE, '_ref_shims'))
    if not hasattr(np, 'float'):
        np.float = float
    torch.cuda.synchronize = lambda *a, **k: None

    cuda_pkg = types.ModuleType('vgtk.cuda')
    cuda_pkg.__path__ = []
    grouping = types.ModuleType('vgtk.cuda.grouping')
    gathering = types.ModuleType('vgtk.cuda.gathering')
    zpconv = types.ModuleType('vgtk.cuda.zpconv')

    grouping.ball_query = lambda q, s, r, n: _t(native.ball_query(q.detach().numpy(), s.detach().numpy(), r, n))
    grouping.furthest_point_sampling = lambda x, m: _t(native.furthest_point_sampling(x.detach().numpy(), m))
    gathering.gather_points_forward = lambda p, i: _t(native.gather_points_forward(p.detach().numpy(), i.numpy()))
    gathering.gather_points_backward = lambda g, i, n: _t(native.gather_points_backward(g.detach().numpy(), i.numpy(), n))
    zpconv.inter_zpconv_forward = lambda i, w, f: _t(native.inter_zpconv_forward(i.numpy(), w.detach().numpy(), f.detach().numpy()))
    zpconv.inter_zpconv_backward = lambda i, w, g, n: _t(native.inter_zpconv_backward(i.numpy(), w.detach().numpy(), g.detach().numpy(), n))
    zpconv.intra_zpconv_forward = lambda i, w, f: _t(native.intra_zpconv_forward(i.numpy(), w.detach().numpy(), f.detach().numpy()))
    zpconv.intra_zpconv_backward = lambda i, w, g, n: _t(native.intra_zpconv_backward(i.numpy(), w.detach().numpy(), g.detach().numpy(), n))

    cuda_pkg.grouping, cuda_pkg.gathering, cuda_pkg.zpconv = grouping, gathering, zpconv
    sys.modules['vgtk.cuda'] = cuda_pkg
    sys.modules['vgtk.cuda.grouping'] = grouping
    sys.modules['vgtk.cuda.gathering'] = gathering
    sys.modules['vgtk.cuda.zpconv'] = zpconv

    # the reference's stride-1 grouping calls .cuda() on a torch.arange (spconv/functional.py:L437)
    torch.Tensor.cuda = lambda self, *a, **k: self

    # optional third-party imports of modules that are off the hot path
    for name in ('parse', 'colour', 'tensorboardX', 'ipdb'):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                stub = types.ModuleType(name)
                stub.Color = object
                stub.SummaryWriter = object
                sys.modules[name] = stub

    import vgtk
    vgtk.cuda = cuda_pkg
    vgtk._is_reference = True
    import vgtk.spconv as zptk
    import vgtk.so3conv as sptk
    import vgtk.so3conv.functional as L
    return vgtk, sptk, L, zptk
