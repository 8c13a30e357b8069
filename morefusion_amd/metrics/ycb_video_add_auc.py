"""YCB-Video ADD(-S) AUC (VOC-style area up to 0.1 m).

morefusion/metrics/ycb_video_add_auc.py:5-51 (itself a port of the YCB_Video_toolbox
plot_accuracy_keyframe.m).
"""
import numpy as np


def ycb_video_add_auc(adds, *, max_value=0.1, return_xy=False):
    adds = np.asarray(adds)
    assert adds.ndim == 1
    assert adds.min() >= 0, f"min of adds must be >=0: {adds.min()}"
    D = adds.astype(float).copy()
    D[D > max_value] = np.inf
    d = np.sort(D)
    n = len(d)
    accuracy = np.cumsum(np.ones((1, n))) / n
    keep = np.isfinite(d)
    if keep.any():
        d = d[keep]
        accuracy = accuracy[keep]
        auc = VOCap(d, accuracy, max_value=max_value)
        x = np.r_[0, d, max_value]
        y = np.r_[0, accuracy, accuracy[-1]]
    else:
        auc = 0
        x = np.array([0, max_value], dtype=float)
        y = np.array([0, 0], dtype=float)
    if return_xy:
        return auc, x, y
    return auc


def VOCap(rec, prec, max_value=0.1):
    mrec = np.r_[0, rec, max_value]
    mpre = np.r_[0, prec, prec[-1]]
    for i in range(1, len(mpre)):
        mpre[i] = max(mpre[i], mpre[i - 1])
    i = np.argwhere(mrec[1:] != mrec[:-1]) + 1
    return np.sum((mrec[i] - mrec[i - 1]) * mpre[i]) / max_value
