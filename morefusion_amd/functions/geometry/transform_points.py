"""transform_points -- [M,4,4] x [N,3] -> [M,N,3].

morefusion/functions/geometry/transform_points.py:6-30.  Evaluated as
((R0 x + R1 y) + R2 z) + t, un-fused, element-wise (no BLAS): the same order the
HIP kernels and the oracle use, so downstream voxel indices agree bit for bit.
"""


def transform_points(points, transform):
    N = points.shape[0]
    assert points.shape == (N, 3)
    squeeze_axis0 = False
    if transform.ndim == 2:
        transform = transform[None]
        squeeze_axis0 = True
    M = transform.shape[0]
    assert transform.shape == (M, 4, 4)
    R = transform[:, :3, :3]
    t = transform[:, :3, 3]
    x, y, z = points[None, :, 0, None], points[None, :, 1, None], points[None, :, 2, None]
    out = ((R[:, None, :, 0] * x + R[:, None, :, 1] * y) + R[:, None, :, 2] * z) + t[:, None, :]
    if squeeze_axis0:
        out = out[0, :, :]
    return out
