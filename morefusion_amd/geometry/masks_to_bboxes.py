"""masks_to_bboxes -- tight (y1, x1, y2, x2) boxes of boolean masks, end-exclusive.

Behaviour of morefusion/geometry/masks_to_bboxes.py:4-38: input ``[N,H,W]`` or ``[H,W]``
bool, output float64 ``[N,4]`` or ``[4]``; an empty mask gives a zero box.
"""
import numpy as np


def _extent(flags):
    """First and one-past-last True position of a 1-D bool array, or (0, 0)."""
    hit = np.flatnonzero(flags)
    return (hit[0], hit[-1] + 1) if hit.size else (0, 0)


def masks_to_bboxes(masks):
    masks = np.asarray(masks)
    if masks.dtype != bool:
        raise AssertionError("masks must be boolean")
    if masks.ndim not in (2, 3):
        raise AssertionError("masks must be 2 or 3 dimensional")
    stack = masks[None] if masks.ndim == 2 else masks
    boxes = np.zeros((stack.shape[0], 4), dtype=np.float64)
    for n, m in enumerate(stack):
        (y1, y2), (x1, x2) = _extent(m.any(axis=1)), _extent(m.any(axis=0))
        boxes[n] = (y1, x1, y2, x2)
    return boxes[0] if masks.ndim == 2 else boxes
