"""ADD and ADD-S between two poses of a model point cloud (host side, float64).

Behaviour of morefusion/metrics/average_distance.py:6-35: ``average_distance(points,
transform1, transform2)`` takes three equally long lists and returns ``(adds, add_ss)``:
ADD = mean distance between corresponding transformed points; ADD-S = mean distance from
each point under transform1 to its nearest neighbour under transform2.
"""
import numpy as np
from scipy.spatial import cKDTree


def _apply(points, T, translate):
    moved = points @ T[:3, :3].T
    return moved + T[:3, 3] if translate else moved


def _pair(points, T1, T2, translate=True):
    points = np.asarray(points, dtype=np.float64)
    T1, T2 = np.asarray(T1, dtype=np.float64), np.asarray(T2, dtype=np.float64)
    if points.ndim != 2 or points.shape[1] != 3 or T1.shape != (4, 4) or T2.shape != (4, 4):
        raise ValueError("points must be [n,3] and the transforms 4x4")
    a, b = _apply(points, T1, translate), _apply(points, T2, translate)
    add = np.sqrt(((a - b) ** 2).sum(axis=1)).mean()
    nearest, _ = cKDTree(b).query(a, k=1)
    return add, nearest.mean()


def average_distance(points, transform1, transform2, translate=True):
    if not isinstance(points, list):
        raise TypeError("points must be a list of [n,3] arrays (one per instance)")
    if not (len(points) == len(transform1) == len(transform2)):
        raise ValueError("points, transform1 and transform2 must have the same length")
    pairs = [_pair(p, a, b, translate) for p, a, b in zip(points, transform1, transform2)]
    adds = np.array([p[0] for p in pairs], dtype=float).reshape(len(points))
    add_ss = np.array([p[1] for p in pairs], dtype=float).reshape(len(points))
    return adds, add_ss
