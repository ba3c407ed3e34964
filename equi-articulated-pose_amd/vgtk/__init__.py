"""vgtk -- drop-in operator package for the SE(3)-equivariant point-convolution hot path of
Meowuu7/equi-articulated-pose, implemented for MI355X (gfx950): hand-written HIP kernels behind
a C ABI (libeap_hip.so, include/eap_hip.h), Python operator layer with the reference's module
paths and signatures (boundary B1).

Scope: vgtk.spconv, vgtk.so3conv, vgtk.pc, vgtk.functional, vgtk.cuda.{zpconv,grouping,gathering}.
The reference's training runtime (vgtk.app: Trainer, Logger, ...) and losses are out of scope.
"""
from . import _hip  # noqa: F401  (raises if libeap_hip.so is missing -- no CPU fallback)
from . import functional  # noqa: F401
from . import point3d  # noqa: F401
from . import pc  # noqa: F401
from . import spconv  # noqa: F401
from . import so3conv  # noqa: F401
from .utils import batch_gather, batch_zip, LearningRateScheduler  # noqa: F401
