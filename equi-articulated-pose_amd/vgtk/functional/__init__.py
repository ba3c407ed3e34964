from .rotation import *  # noqa: F401,F403
from .rotation import (anchor_group_tables, compute_rotation_matrix_from_ortho6d,  # noqa: F401
                       compute_rotation_matrix_from_quaternion, icosahedron_so3,
                       icosahedron_so3_trimesh, so3_mean)
