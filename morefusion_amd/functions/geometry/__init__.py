# flake8: noqa
# mirrors morefusion/functions/geometry/__init__.py:3-28

from .average_voxelization_3d import average_voxelization_3d

from .compose_transform import compose_transform

from .max_voxelization_3d import max_voxelization_3d

from .occupancy_grid_1d import occupancy_grid_1d
from .occupancy_grid_2d import occupancy_grid_2d
from .occupancy_grid_3d import occupancy_grid_3d

from .interpolate_voxel_grid import interpolate_voxel_grid

from .quaternion_matrix import quaternion_matrix

from .transform_points import transform_points

from .transformation_matrix import transformation_matrix

from .translation_matrix import translation_matrix

from .truncated_distance_function import truncated_distance_function
from .truncated_distance_function import pseudo_occupancy_voxelization
