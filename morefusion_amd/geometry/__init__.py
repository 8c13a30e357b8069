# flake8: noqa
# the subset of morefusion/geometry/__init__.py:3-25 that is on (or feeds) the hot path
from .compose_transform import compose_transform
from .masks_to_bboxes import masks_to_bboxes
from .nn import nn
from .pointcloud_from_depth import pointcloud_from_depth
from .quaternion_from_matrix import quaternion_from_matrix, translation_from_matrix
from .instance_crops import grid_origin, instance_crops
