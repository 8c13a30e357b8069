"""compose_transform(R=None, t=None) -> 4x4 (no gradient).

morefusion/geometry/compose_transform.py:7-24.  NumPy in, NumPy out; torch in, torch out.
"""
import numpy as np
import torch


def compose_transform(R=None, t=None):
    if isinstance(R, torch.Tensor) or isinstance(t, torch.Tensor):
        ref = R if isinstance(R, torch.Tensor) else t
        T = torch.eye(4, dtype=ref.dtype, device=ref.device)
        if R is not None:
            T[:3, :3] = R.detach()
        if t is not None:
            T[:3, 3] = t.detach()
        return T
    T = np.eye(4)
    if R is not None:
        T = T.astype(np.asarray(R).dtype)
        T[:3, :3] = R
    if t is not None:
        if R is None:
            T = T.astype(np.asarray(t).dtype)
        T[:3, 3] = t
    return T
