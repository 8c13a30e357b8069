"""average_distance -- ADD / ADD-S pose loss.

morefusion/functions/loss/average_distance.py:40-85.  The ADD-S branch uses
``geometry.nn`` (fused HIP 1-NN, no R x Q distance matrix).
"""
import torch

from ... import geometry as geometry_module
from ..geometry import transform_points


def average_distance(points, transform_true, transforms_pred, symmetric=False):
    n_points = points.shape[0]
    n_pred = transforms_pred.shape[0]
    assert points.shape == (n_points, 3)
    assert transform_true.shape == (4, 4)
    assert transforms_pred.shape == (n_pred, 4, 4)

    points_true = transform_points(points, transform_true)
    points_pred = transform_points(points, transforms_pred)
    assert points_true.shape == (n_points, 3)
    assert points_pred.shape == (n_pred, n_points, 3)

    if symmetric:
        ref = points_true.detach()
        query = points_pred.detach().reshape(n_pred * n_points, 3)
        indices = geometry_module.nn(ref, query)
        points_true = points_true[indices]
        points_true = points_true.reshape(n_pred, n_points, 3)
    else:
        points_true = points_true[None].expand(n_pred, n_points, 3)

    return torch.sqrt(((points_true - points_pred) ** 2).sum(dim=2)).mean(dim=1)
