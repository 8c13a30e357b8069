"""auc_for_errors -- normalised area under the accuracy-vs-threshold curve.

Behaviour of morefusion/metrics/auc_for_errors.py:5-25: accuracy(th) = fraction of errors
<= th, sampled at ``nstep`` thresholds evenly spaced on [0, max_threshold], integrated with
the trapezoid rule and divided by ``max_threshold`` (so the result lies in [0, 1]).
"""
import numpy as np


def auc_for_errors(errors, max_threshold, *, nstep=1000, return_xy=False):
    e = np.sort(np.asarray(errors, dtype=float).ravel())
    if np.asarray(errors).ndim != 1:
        raise AssertionError("errors must be 1-D")
    if e.size and e[0] < 0:
        raise AssertionError(f"min of errors must be >=0: {e[0]}")
    thresholds = np.linspace(0.0, max_threshold, nstep)
    # number of errors <= each threshold, in one pass over the sorted errors
    accuracy = np.searchsorted(e, thresholds, side="right") / float(e.size)
    widths = np.diff(thresholds)
    area = float((widths * (accuracy[1:] + accuracy[:-1]) * 0.5).sum())
    auc = area / float(max_threshold)
    return (auc, thresholds, accuracy) if return_xy else auc
