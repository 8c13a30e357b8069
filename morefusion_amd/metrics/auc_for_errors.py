"""auc_for_errors.  morefusion/metrics/auc_for_errors.py:5-25 (trapezoid of the
accuracy-vs-threshold curve, scaled to [0, 1])."""
import numpy as np


def auc_for_errors(errors, max_threshold, *, nstep=1000, return_xy=False):
    errors = np.asarray(errors)
    assert errors.ndim == 1
    assert errors.min() >= 0, f"min of errors must be >=0: {errors.min()}"
    x = np.linspace(0, max_threshold, nstep)
    y = np.array([1.0 * (errors <= th).sum() / errors.size for th in x], dtype=float)
    auc = np.trapezoid(y, x) if hasattr(np, "trapezoid") else np.trapz(y, x)
    auc = auc / (1.0 * max_threshold)
    if return_xy:
        return auc, x, y
    return auc
