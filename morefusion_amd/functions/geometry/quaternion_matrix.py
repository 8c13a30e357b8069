"""quaternion_matrix -- wxyz quaternion (any norm) -> 4x4 homogeneous rotation.

morefusion/functions/geometry/quaternion_matrix.py:65-78 (+ Function :6-51).
Plain torch ops (device-agnostic, autograd by torch): a handful of 4x4 scalars --
launch overhead only; the refinement hot loop fuses this into k_icc_step instead.
"""
import torch


def quaternion_matrix(quaternion):
    squeeze_axis0 = False
    if quaternion.ndim == 1:
        squeeze_axis0 = True
        quaternion = quaternion[None]
    norm = (quaternion ** 2).sum(dim=1, keepdim=True)
    q = quaternion * torch.sqrt(2.0 / norm)
    Q = q[:, :, None] * q[:, None, :]
    one = torch.ones_like(Q[:, 0, 0])
    zero = torch.zeros_like(one)
    rows = [
        torch.stack([one - Q[:, 2, 2] - Q[:, 3, 3], Q[:, 1, 2] - Q[:, 3, 0], Q[:, 1, 3] + Q[:, 2, 0], zero], 1),
        torch.stack([Q[:, 1, 2] + Q[:, 3, 0], one - Q[:, 1, 1] - Q[:, 3, 3], Q[:, 2, 3] - Q[:, 1, 0], zero], 1),
        torch.stack([Q[:, 1, 3] - Q[:, 2, 0], Q[:, 2, 3] + Q[:, 1, 0], one - Q[:, 1, 1] - Q[:, 2, 2], zero], 1),
        torch.stack([zero, zero, zero, one], 1),
    ]
    matrix = torch.stack(rows, 1)
    if squeeze_axis0:
        matrix = matrix[0, :, :]
    return matrix
