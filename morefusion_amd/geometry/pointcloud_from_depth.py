"""pointcloud_from_depth -- back-project a depth image.

morefusion/geometry/pointcloud_from_depth.py:4-26 (host-side pre-processing that feeds
the path; NumPy, same as the reference).
"""
import numpy as np


def pointcloud_from_depth(depth, fx, fy, cx, cy, depth_type="z"):
    assert depth_type in ["z", "euclidean"], "Unexpected depth_type"
    assert depth.dtype.kind == "f", "depth must be float and have meter values"
    rows, cols = depth.shape
    c, r = np.meshgrid(np.arange(cols), np.arange(rows), sparse=True)
    valid = ~np.isnan(depth)
    z = np.where(valid, depth, np.nan)
    x = np.where(valid, z * (c - cx) / fx, np.nan)
    y = np.where(valid, z * (r - cy) / fy, np.nan)
    pc = np.dstack((x, y, z))
    if depth_type == "euclidean":
        norm = np.linalg.norm(pc, axis=2)
        pc = pc * (z / norm)[:, :, None]
    return pc
