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


def _overlay(tmp_path):
    pkg = tmp_path / 'overlay'
    shutil.copytree(os.path.join(ROOT, 'equi-articulated-pose_amd'), pkg, ignore=shutil.ignore_patterns('csrc', '__pycache__', '*.o'))
    shutil.copytree(os.path.join(REF, 'vgtk', 'vgtk', 'app'), pkg / 'vgtk' / 'app', ignore=shutil.ignore_patterns('__pycache__'))
    shutil.copy(os.path.join(REF, 'vgtk', 'vgtk', 'loss.py'), pkg / 'vgtk' / 'loss.py')
    return pkg


def _prelude(pkg, extra_roots=()):
    """Child-interpreter prelude: stubs for the third-party / dataset modules, the overlay first on sys.path, the reference behind it."""
    return textwrap.dedent(f'''
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
                     'SPConvNets.ransac') + {tuple(extra_roots)!r}
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
    ''')


def _run(script, tmp_path, token):
    env = dict(os.environ, PYTHONPATH='', HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='', LOCAL_RANK='0')    # (the model reads LOCAL_RANK: ...pn_38_multi_stage.py:L103)
    out = subprocess.run([sys.executable, '-c', script], capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert out.returncode == 0 and token in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    return out.stdout


def test_root_exports_with_the_reference_runtime_overlaid(tmp_path):
    pkg = _overlay(tmp_path)
    script = _prelude(pkg, ('SPConvNets.models.common_utils', 'SPConvNets.utils.loss_util', 'model_util')) + textwrap.dedent('''
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
    _run(script, tmp_path, 'OVERLAY-OK')


def test_reference_model_builder_yields_the_product_operators_with_the_reference_state_dict(tmp_path):
    """SPConvNets/models/unsup_seg_so3_pose_conv_pn_38_multi_stage.py:L2067-2325 `build_model_from(opt)` -- the call the trainer
    makes (trainer_unsup_arti_align.py:L333-334) -- on product + overlay, with the reference's own option parser
    (SPConvNets/options.py) fed the shipped script's model arguments (scripts/train/laptop_syn.sh).  The model it returns must be
    made of the PRODUCT's conv modules and carry the reference's parameter names and shapes (its checkpoints load)."""
    pkg = _overlay(tmp_path)
    script = _prelude(pkg) + textwrap.dedent(f'''
        import os
        import torch
        sys.path.insert(2, os.path.join({REF!r}, 'SPConvNets', 'models'))         # `from DGCNN import PrimitiveNet`
        sys.argv = ['run.py', 'experiment', '-d', {str(tmp_path)!r}, '--experiment-id', 'overlay',
                    'model', '--model', 'unsup_seg_so3_pose_conv_pn_38_multi_stage', '--input-num', '512', '--kanchor', '60']
        from SPConvNets.options import opt
        opt.device = torch.device('cpu')
        # (no GPU in the build container; the model's constructor calls .cuda() on its constants: ...pn_38_multi_stage.py:L117)
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
        if not hasattr(opt, 'nmasks'):
            opt.nmasks = 2
        import SPConvNets.models.unsup_seg_so3_pose_conv_pn_38_multi_stage as MS
        model = MS.build_model_from(opt)
        assert type(model).__name__ == 'ClsSO3ConvModel'
        import vgtk.so3conv as sptk
        assert 'overlay' in sptk.__file__
        convs = [m for m in model.modules() if type(m).__name__ in ('InterSO3PoseConv', 'InterSO3Conv', 'IntraSO3Conv', 'BasicSO3Conv')]
        assert convs and all(type(m).__module__.startswith('vgtk.so3conv') and 'overlay' in sys.modules[type(m).__module__].__file__ for m in convs), \
            sorted({{type(m).__module__ for m in convs}})
        sd = model.state_dict()
        # the reference's names and shapes (BasicSO3PoseConvBlock -> InterSO3PoseConvBlock -> InterSO3PoseConv -> BasicSO3Conv.W:
        # SPConvNets/utils/base_so3poseconv.py:L205-222, vgtk/vgtk/so3conv/modules.py:L33-55)
        want = {{'glb_backbone.0.blocks.0.inter_conv.conv.basic_conv.W': (64, 1, 24), 'backbone.1.blocks.0.conv.basic_conv.W': (128, 64, 24),
                'backbone_sec.2.blocks.0.conv.basic_conv.W': (512, 128, 24),
                'glb_backbone.2.blocks.0.intra_conv.conv.basic_conv.W': (512, 512, 12)}}
        for k, shp in want.items():
            assert k in sd, (k, [n for n in sd if n.startswith(k.split('.')[0])][:8])
            assert tuple(sd[k].shape) == shp or sd[k].numel() == shp[0] * shp[1] * shp[2], (k, tuple(sd[k].shape))
        n_params = sum(p.numel() for p in model.parameters())
        print('BUILD-OK', n_params, len(sd), len(convs))
    ''')
    out = _run(script, tmp_path, 'BUILD-OK')
    n_params = int(out.split('BUILD-OK')[1].split()[0])
    assert n_params > 10_000_000, n_params          # (21.2 M with the shipped hyper-parameters)


def test_root_without_the_overlay_has_no_runtime_names():
    import vgtk
    assert not hasattr(vgtk, 'Trainer') and hasattr(vgtk, 'LearningRateScheduler') and hasattr(vgtk, 'batch_gather')
