"""pointcloud_from_depth -- pinhole back-projection of a metric depth image.

Behaviour of morefusion/geometry/pointcloud_from_depth.py:4-26 (host-side pre-processing
feeding the path): ``depth`` float [H,W] in metres with NaN = invalid; returns [H,W,3] camera
-frame points (NaN where invalid).  ``depth_type="euclidean"`` treats the value as range
along the viewing ray instead of z.
"""
import numpy as np


def pointcloud_from_depth(depth, fx, fy, cx, cy, depth_type="z"):
    if depth_type not in ("z", "euclidean"):
        raise AssertionError("Unexpected depth_type")
    if depth.dtype.kind != "f":
        raise AssertionError("depth must be float and have meter values")
    height, width = depth.shape
    u = np.arange(width)[None, :]   # column index, broadcast over rows
    v = np.arange(height)[:, None]  # row index, broadcast over columns
    ok = ~np.isnan(depth)
    z = np.where(ok, depth, np.nan)
    cloud = np.dstack((np.where(ok, z * (u - cx) / fx, np.nan),
                       np.where(ok, z * (v - cy) / fy, np.nan), z))
    if depth_type == "euclidean":
        cloud = cloud * (z / np.linalg.norm(cloud, axis=2))[:, :, None]
    return cloud
