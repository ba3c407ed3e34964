from vgtk.spconv import SphericalPointCloud, SphericalPointCloudPose  # noqa: F401
from .functional import *  # noqa: F401,F403
from .modules import *  # noqa: F401,F403
from .blocks import BatchNormLeakyReLU, InstanceNormLeakyReLU, conv_norm_act, pointwise_norm_act  # noqa: F401
from .heads import (masked_max, pointwise_conv, pose_head_over_slot_groups, slot_point_groups, InvPPOutBlockOurs, SO3OutBlockRTWithMaskSep, anchor_attention_pool, orbit_selection, slot_masked_mean,  # noqa: F401
                    rotation_from_angle_axis, pose_head_over_subsets, rotation_axes, compute_rotation_matrix_from_angle, orbit_slot_distances)
