"""quaternion_from_matrix -- rotation matrix -> unit quaternion (w, x, y, z).

Stands in for ``trimesh.transformations.quaternion_from_matrix`` (third party, not in the
reference tree) as called by the refinement links
(morefusion/contrib/iterative_collision_check_link.py:22,
contrib/iterative_closest_point_link.py:13): the eigenvector of the symmetric 4x4
"K" matrix with the largest eigenvalue, sign fixed so that w >= 0.  Host side, float64.
"""
import numpy as np


def quaternion_from_matrix(matrix):
    M = np.asarray(matrix, dtype=np.float64)[:4, :4]
    (m00, m01, m02), (m10, m11, m12), (m20, m21, m22) = M[0, :3], M[1, :3], M[2, :3]
    K = np.array([
        [m00 - m11 - m22, 0.0, 0.0, 0.0],
        [m01 + m10, m11 - m00 - m22, 0.0, 0.0],
        [m02 + m20, m12 + m21, m22 - m00 - m11, 0.0],
        [m21 - m12, m02 - m20, m10 - m01, m00 + m11 + m22],
    ]) / 3.0
    w, V = np.linalg.eigh(K)  # uses the lower triangle
    q = V[[3, 0, 1, 2], np.argmax(w)]
    if q[0] < 0.0:
        q = -q
    return q


def translation_from_matrix(matrix):
    return np.asarray(matrix)[:3, 3].copy()
