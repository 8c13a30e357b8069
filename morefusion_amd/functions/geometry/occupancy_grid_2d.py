"""occupancy_grid_2d -- soft occupancy of a 2-D grid, the toy sibling of occupancy_grid_3d.

morefusion/functions/geometry/occupancy_grid_2d.py:10-75: distance of every cell to every point
in cell units, ``min(relu(threshold - d), 1)``, max over points.  The reference builds its grid
with ``meshgrid(arange(d0), arange(d1), arange(P))`` in 'xy' order, so its result is laid out
``[dimension[1], dimension[0]]`` -- reproduced here.  Unused by the pose pipeline; plain
differentiable torch.
"""
import numbers
from collections.abc import Sequence

import torch


def occupancy_grid_2d(points, *, pitch, origin, dimension, threshold=1):
    if not isinstance(pitch, numbers.Real):
        raise AssertionError("pitch must be a real number")
    if not (isinstance(origin, Sequence) and len(origin) == 2 and isinstance(dimension, Sequence)
            and len(dimension) == 2):
        raise AssertionError("origin and dimension must be sequences of length 2")
    if points.dtype != torch.float32 or points.ndim != 2 or points.shape[1] != 2:
        raise TypeError("points must be float32 [P, 2]")
    kw = dict(dtype=points.dtype, device=points.device)
    ci = torch.arange(dimension[0], **kw)[None, :, None]  # varies along axis 1 ('xy' meshgrid)
    cj = torch.arange(dimension[1], **kw)[:, None, None]
    d_ik = ci - ((points[:, 0] - origin[0]) / pitch)[None, None, :]
    d_jk = cj - ((points[:, 1] - origin[1]) / pitch)[None, None, :]
    d = torch.sqrt(d_ik ** 2 + d_jk ** 2)  # [dimension[1], dimension[0], P]
    m = torch.relu(threshold - d.abs()).clamp(max=1)
    return m.max(dim=2).values
