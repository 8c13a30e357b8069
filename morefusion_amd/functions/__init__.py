# flake8: noqa
# mirrors morefusion/functions/__init__.py:3-15

from .geometry import average_voxelization_3d
from .geometry import compose_transform
from .geometry import interpolate_voxel_grid
from .geometry import max_voxelization_3d
from .geometry import occupancy_grid_3d
from .geometry import pseudo_occupancy_voxelization
from .geometry import quaternion_matrix
from .geometry import transform_points
from .geometry import transformation_matrix
from .geometry import translation_matrix
from .geometry import truncated_distance_function

from .loss import average_distance
from .loss import average_distance_batch  # the same op over B objects (one fused launch)
