"""vgtk.cuda -- the reference's native extension namespace (vgtk/vgtk/cuda), served by
libeap_hip.so through ctypes.  Same module names and function signatures as the reference's
pybind11 modules `zpconv`, `grouping`, `gathering` (boundary B2)."""
from . import gathering, grouping, zpconv  # noqa: F401
