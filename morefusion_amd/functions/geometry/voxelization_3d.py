"""Argument validation shared by the voxelization ops.

Mirrors ``Voxelization3D.__init__`` / ``check_type_forward``
(morefusion/functions/geometry/voxelization_3d.py:5-32): same conditions, raised
as ValueError / TypeError instead of chainer's InvalidType.
"""
import torch

from ... import _lib


def check_dimensions(dimensions):
    if not (
        isinstance(dimensions, tuple)
        and len(dimensions) == 3
        and all(isinstance(d, int) for d in dimensions)
    ):
        # message kept verbatim from the reference (voxelization_3d.py:16)
        raise ValueError("dimensions must be a tuple of 4 integers")
    return dimensions


def check_inputs(values, points, batch_indices):
    _lib.require_gpu(values, points, batch_indices)
    if values.dtype != torch.float32 or values.ndim != 2:
        raise TypeError("values must be float32 [P, C]")
    if points.dtype != torch.float32 or points.ndim != 2 or points.shape[1] != 3:
        raise TypeError("points must be float32 [P, 3]")
    if points.shape[0] != values.shape[0]:
        raise TypeError("points and values must have the same length")
    if batch_indices.dtype != torch.int32 or batch_indices.ndim != 1:
        raise TypeError("batch_indices must be int32 [P]")
    if batch_indices.shape[0] != values.shape[0]:
        raise TypeError("batch_indices and values must have the same length")
