from .base import SphericalPointCloud, SphericalPointCloudPose  # noqa: F401
from .functional import *  # noqa: F401,F403
from . import functional  # noqa: F401
from .modules import BasicZPConv, IntraZPConv, InterZPConv, AnchorProp  # noqa: F401
