"""Boundary B1, the package root: with the reference's OWN `vgtk/app/` and `vgtk/loss.py` dropped in beside the product's
files (Option A -- nothing of the reference is stored in this repo or shipped to the GPU box; the overlay is
made in a scratch directory), `import vgtk` re-exports the runtime names exactly as vgtk/vgtk/__init__.py:L8-9 does and the
module-level `class Trainer(vgtk.Trainer)` of SPConvNets/trainer_unsup_arti_align.py:L49 can be defined.

Build-container only (needs /root/reference).  Runs in a child interpreter: the overlay must not leak into this process's
`vgtk`.  Third-party modules the reference imports at module level and this image lacks (tensorboardX, colour, parse, ...)
and the dataset / rendering modules of SPConvNets are stubbed -- they are not part of the boundary."""
import os
import shutil
import subprocess
import sys
import textwrap

import pytest

REF = '/root/reference'
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'vgtk', 'vgtk', 'app')), reason='needs /root/reference (build container only)')


def test_root_exports_with_the_reference_runtime_overlaid(tmp_path):
    pkg = tmp_path / 'overlay'
    shutil.copytree(os.path.join(ROOT, 'equi-articulated-pose_amd'), pkg, ignore=shutil.ignore_patterns('csrc', '__pycache__', '*.o'))
    shutil.copytree(os.path.join(REF, 'vgtk', 'vgtk', 'app'), pkg / 'vgtk' / 'app', ignore=shutil.ignore_patterns('__pycache__'))
    shutil.copy(os.path.join(REF, 'vgtk', 'vgtk', 'loss.py'), pkg / 'vgtk' / 'loss.py')
    script = textwrap.dedent(f'''
        import importlib, importlib.abc, importlib.machinery, sys, types
        import numpy as np
        np.float = float

        class _Stub(types.ModuleType):
            """anything: attribute access makes classes, calls return None"""
            __path__ = []
            def __getattr__(self, name):
                if name.startswith('__'):
                    raise AttributeError(name)
                t = type(name, (), {{'__init__': lambda self, *a, **k: None}})
                setattr(self, name, t)
                return t

        class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
            """stubs every module that cannot be found, below the listed roots"""
            ROOTS = ('tensorboardX', 'colour', 'parse', 'trimesh', 'plyfile', 'imageio', 'skimage', 'ipdb', 'pyrender', 'transforms3d',
                     'torch_cluster', 'torch_scatter', 'open3d', 'cv2', 'matplotlib', 'SPConvNets.datasets', 'SPConvNets.pose_utils',
                     'SPConvNets.ransac', 'SPConvNets.models.common_utils', 'SPConvNets.utils.loss_util', 'model_util')
            def find_spec(self, name, path=None, target=None):
                if any(name == r or name.startswith(r + '.') for r in self.ROOTS):
                    return importlib.machinery.ModuleSpec(name, self)
                return None
            def create_module(self, spec):
                return _Stub(spec.name)
            def exec_module(self, module):
                pass
        sys.meta_path.append(_Finder())

        sys.path[:0] = [{str(pkg)!r}, {REF!r}]
        import vgtk
        assert {str(pkg)!r} in vgtk.__file__, vgtk.__file__
        for name in ('Trainer', 'Logger', 'Summary', 'Timer', 'HierarchyArgmentParser', 'dump_args', 'LearningRateScheduler', 'batch_gather',
                     'CrossEntropyLossPerP', 'AttentionCrossEntropyLoss'):
            assert hasattr(vgtk, name), name
        assert vgtk.Trainer.__module__ == 'vgtk.app.trainer' and vgtk.CrossEntropyLossPerP.__module__ == 'vgtk.loss'
        # the operator layer is still the product's
        import vgtk.so3conv as sptk
        assert 'overlay' in sptk.__file__ and hasattr(sptk.functional, '_InterConv')
        # the reference's trainer module: its module-level class statement needs vgtk.Trainer
        import SPConvNets.trainer_unsup_arti_align as T
        assert issubclass(T.Trainer, vgtk.Trainer)
        print('OVERLAY-OK')
    ''')
    env = dict(os.environ, PYTHONPATH='')
    out = subprocess.run([sys.executable, '-c', script], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert out.returncode == 0 and 'OVERLAY-OK' in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_root_without_the_overlay_has_no_runtime_names():
    import vgtk
    assert not hasattr(vgtk, 'Trainer') and hasattr(vgtk, 'LearningRateScheduler') and hasattr(vgtk, 'batch_gather')
