"""median with the reference's even-n rule (mean of the two middle values).

morefusion/extra/_cupy.py:47-62 -- ``torch.median`` returns the LOWER middle value for
even n, which would shift every grid origin (model.py:202-205).
"""
import torch


def median(x, axis=None):
    if axis is None:
        x = x.flatten()
        axis = 0
    n = x.shape[axis]
    s, _ = torch.sort(x, dim=axis)
    m_odd = s.select(axis, n // 2)
    if n % 2 == 1:
        return m_odd
    m_even = s.select(axis, n // 2 - 1)
    return (m_odd + m_even) / 2
