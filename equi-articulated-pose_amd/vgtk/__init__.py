"""vgtk -- drop-in operator package for the SE(3)-equivariant point-convolution hot path of
Meowuu7/equi-articulated-pose, implemented for MI355X (gfx950): hand-written HIP kernels behind
a C ABI (libeap_hip.so, include/eap_hip.h), Python operator layer with the reference's module
paths and signatures (boundary B1).

Scope: vgtk.spconv, vgtk.so3conv, vgtk.pc, vgtk.functional, vgtk.cuda.{zpconv,grouping,gathering}.
The reference's training runtime (vgtk/app: Trainer, Logger, Summary, Timer, HierarchyArgmentParser, dump_args) and its losses
(vgtk/loss.py) are pure Python beside the hot path and are NOT part of this package; a maintainer who keeps the reference's own
`vgtk/app/` and `vgtk/loss.py` next to these files (Option A) gets them re-exported from the package root exactly
as the reference's vgtk/vgtk/__init__.py:L8-9 does, so `class Trainer(vgtk.Trainer)` (SPConvNets/trainer_unsup_arti_align.py:L49)
resolves.  tests/test_reference_trainer_overlay.py does that overlay in a scratch directory.
"""
from . import _hip  # noqa: F401  (raises if libeap_hip.so is missing -- no CPU fallback)
from . import functional  # noqa: F401
from . import point3d  # noqa: F401
from . import pc  # noqa: F401
from . import spconv  # noqa: F401
from . import so3conv  # noqa: F401
from .utils import batch_gather, batch_zip, LearningRateScheduler  # noqa: F401

# the reference's runtime, when its files were dropped in beside this one (vgtk/vgtk/__init__.py:L8-9); absent otherwise
import importlib.util as _ilu
if _ilu.find_spec(__name__ + '.app') is not None:
    from .app import *  # noqa: F401,F403
if _ilu.find_spec(__name__ + '.loss') is not None:
    from .loss import *  # noqa: F401,F403
del _ilu
