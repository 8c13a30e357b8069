"""occupancy_grid_1d -- soft occupancy of a 1-D grid, the toy sibling of occupancy_grid_3d.

morefusion/functions/geometry/occupancy_grid_1d.py:9-60: ``m[i] = max_p relu(1 - |i - (x_p -
origin) / pitch|)``.  Unused by the pose pipeline (exported for completeness of
``morefusion.functions.geometry``); a handful of elementwise ops, so plain differentiable torch.
"""
import numbers

import torch


def occupancy_grid_1d(points, *, pitch, origin, dimension):
    if not (isinstance(pitch, numbers.Real) and isinstance(origin, numbers.Real) and isinstance(dimension, int)):
        raise AssertionError("pitch, origin must be real numbers and dimension an int")
    if points.dtype != torch.float32 or points.ndim != 1:
        raise TypeError("points must be float32 [P]")
    cells = torch.arange(dimension, dtype=points.dtype, device=points.device)
    d = cells[None, :] - ((points[:, None] - origin) / pitch)  # [P, dimension]
    return torch.relu(1 - d.abs()).max(dim=0).values
