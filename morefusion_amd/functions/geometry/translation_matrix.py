"""translation_matrix -- t -> 4x4.  morefusion/functions/geometry/translation_matrix.py:5-39."""
import torch


def translation_matrix(translation):
    squeeze_axis0 = False
    if translation.ndim == 1:
        translation = translation[None]
        squeeze_axis0 = True
    if translation.ndim != 2 or translation.shape[1] != 3:
        raise TypeError("translation must be [N, 3]")
    N = translation.shape[0]
    eye = torch.eye(3, dtype=translation.dtype, device=translation.device).expand(N, 3, 3)
    top = torch.cat([eye, translation[:, :, None]], dim=2)
    bottom = torch.zeros((1, 1, 4), dtype=translation.dtype, device=translation.device)  # (device-built: capture-safe)
    bottom[..., 3] = 1
    bottom = bottom.expand(N, 1, 4)
    matrix = torch.cat([top, bottom], dim=1)
    if squeeze_axis0:
        matrix = matrix[0, :, :]
    return matrix
