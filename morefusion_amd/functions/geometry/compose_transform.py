"""compose_transform -- [R | t] -> 4x4.  morefusion/functions/geometry/compose_transform.py:5-48."""
import torch


def compose_transform(R, t):
    squeeze_axis0 = False
    if R.ndim == 2 and t.ndim == 1:
        R = R[None]
        t = t[None]
        squeeze_axis0 = True
    if R.ndim != 3 or R.shape[1:] != (3, 3) or t.ndim != 2 or t.shape[1] != 3 or R.shape[0] != t.shape[0]:
        raise TypeError("R must be [N,3,3] and t [N,3]")
    N = R.shape[0]
    top = torch.cat([R, t[:, :, None]], dim=2)
    # (built on the device: a host list would be a pageable host-to-device copy, which a hipGraph capture refuses)
    bottom = torch.zeros((1, 1, 4), dtype=R.dtype, device=R.device)
    bottom[..., 3] = 1
    bottom = bottom.expand(N, 1, 4)
    matrix = torch.cat([top, bottom], dim=1)
    if squeeze_axis0:
        matrix = matrix[0, :, :]
    return matrix
