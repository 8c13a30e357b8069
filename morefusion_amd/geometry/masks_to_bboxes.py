"""masks_to_bboxes -- (y1, x1, y2, x2) of boolean masks.

morefusion/geometry/masks_to_bboxes.py:4-38.
"""
import numpy as np


def masks_to_bboxes(masks):
    masks = np.asarray(masks)
    assert masks.dtype == bool
    ndim = masks.ndim
    assert ndim in [2, 3], "masks must be 2 or 3 dimensional"
    if ndim == 2:
        masks = masks[None]
    bboxes = np.zeros((len(masks), 4), dtype=np.float64)
    for i, mask in enumerate(masks):
        where = np.argwhere(mask)
        if where.size == 0:
            continue
        (y1, x1), (y2, x2) = where.min(0), where.max(0) + 1
        bboxes[i] = y1, x1, y2, x2
    if ndim == 2:
        return bboxes[0]
    return bboxes
