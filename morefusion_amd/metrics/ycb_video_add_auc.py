"""YCB-Video ADD(-S) AUC: area under the accuracy-vs-threshold curve up to ``max_value``.

Behaviour of morefusion/metrics/ycb_video_add_auc.py:5-51 (itself following the YCB_Video
toolbox's plot_accuracy_keyframe.m): errors above ``max_value`` count as misses, the curve
is made monotone (VOC-style) and integrated over the distinct error values.
"""
import numpy as np


def VOCap(rec, prec, max_value=0.1):
    """VOC average precision of (rec, prec) extended with (0, 0) and (max_value, prec[-1])."""
    x = np.concatenate(([0.0], np.asarray(rec, dtype=float), [max_value]))
    y = np.concatenate(([0.0], np.asarray(prec, dtype=float), [prec[-1]]))
    y = np.maximum.accumulate(y)
    step = np.flatnonzero(x[1:] != x[:-1]) + 1
    return float(((x[step] - x[step - 1]) * y[step]).sum() / max_value)


def ycb_video_add_auc(adds, *, max_value=0.1, return_xy=False):
    adds = np.asarray(adds, dtype=float)
    if adds.ndim != 1:
        raise AssertionError("adds must be 1-D")
    if adds.min() < 0:
        raise AssertionError(f"min of adds must be >=0: {adds.min()}")
    n = adds.size
    hits = np.sort(adds[adds <= max_value])  # misses sort to +inf and are dropped
    if hits.size:
        accuracy = np.arange(1, hits.size + 1) / n
        auc = VOCap(hits, accuracy, max_value=max_value)
        x = np.concatenate(([0.0], hits, [max_value]))
        y = np.concatenate(([0.0], accuracy, [accuracy[-1]]))
    else:
        auc = 0
        x, y = np.array([0.0, max_value]), np.zeros(2)
    return (auc, x, y) if return_xy else auc
