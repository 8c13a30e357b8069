"""transformation_matrix(quaternion, translation) -> homogeneous 4x4 transform(s).

API of morefusion/functions/geometry/transformation_matrix.py:5-18: a batch ``[N,4]`` /
``[N,3]`` gives ``[N,4,4]``, a single ``[4]`` / ``[3]`` pair gives ``[4,4]``.
"""
from .compose_transform import compose_transform
from .quaternion_matrix import quaternion_matrix


def transformation_matrix(quaternion, translation):
    single = quaternion.ndim == 1
    q = quaternion[None] if single else quaternion
    t = translation[None] if single else translation
    if q.ndim != 2 or q.shape[1] != 4 or t.shape != (q.shape[0], 3):
        raise ValueError(
            f"expected quaternion [N,4] and translation [N,3] (or [4] and [3]), got "
            f"{tuple(quaternion.shape)} and {tuple(translation.shape)}")
    rotation = quaternion_matrix(q)[:, :3, :3]
    T = compose_transform(rotation, t)
    return T[0] if single else T
