"""Evaluation aggregation of the pose network (second half).

``PoseEstimationEvaluator`` follows morefusion/training/extensions/pose_estimation_evaluator.py:
18-160 without the Chainer trainer around it: run ``eval_func`` over an iterable of batches,
collect one observation dict per batch, gather the dicts on rank 0 when a process group is
active (the reference gathers a pandas frame through ChainerMN), then

* regroup ``validation/main/{add,add_s,add_or_add_s}/{class_id}/{instance}`` per class,
* average every key over the batches that reported it,
* per (type, class): YCB-Video AUC up to 0.1 m and the fraction of errors below 2 cm,
* fill each parent key (``.../add``, ``.../auc/add_s`` ...) with the mean over its classes.
"""
import collections
import math
import os.path as osp
import re

import numpy as np
import torch
import torch.distributed as dist

from . import metrics

ADD_TYPES = ("add", "add_s", "add_or_add_s")
_PARENTS = (
    ["loss", "loss_quaternion", "loss_translation"] + list(ADD_TYPES)
    + [f"auc/{t}" for t in ADD_TYPES] + [f"<2cm/{t}" for t in ADD_TYPES]
)


class _MeanOfKeys:
    """chainer.DictSummary.compute_mean: per-key mean over the dicts that carry the key."""

    def __init__(self):
        self._sum = collections.defaultdict(float)
        self._n = collections.Counter()

    def add(self, d):
        for k, v in d.items():
            self._sum[k] += float(v)
            self._n[k] += 1

    def compute_mean(self):
        return {k: self._sum[k] / self._n[k] for k in self._sum}


def summarize_observations(observations, prefix="validation/main/"):
    """List of per-batch observation dicts -> the evaluator's result dict."""
    pattern = re.compile(re.escape(prefix) + "(" + "|".join(ADD_TYPES) + ")/([0-9]+)/.+")
    summary = _MeanOfKeys()
    adds = collections.defaultdict(list)
    for row in observations:
        processed = {}
        for key, value in row.items():
            if value is None or (isinstance(value, float) and math.isnan(value)):
                continue  # DataFrame.dropna()
            match = pattern.match(key)
            if match:
                add_type, class_id = match.groups()
                key = f"{prefix}{add_type}/{class_id}"
                adds[f"{add_type}/{class_id}"].append(value)
            processed[key] = value  # several instances of a class in one batch: last one wins (as upstream)
        summary.add(processed)
    result = summary.compute_mean()
    for name, values in adds.items():
        result[f"{prefix}auc/{name}"] = metrics.ycb_video_add_auc(values, max_value=0.1)
        result[f"{prefix}<2cm/{name}"] = float((np.asarray(values) < 0.02).sum() / len(values))
    parents = _MeanOfKeys()
    for parent in (prefix + p for p in _PARENTS):
        if parent in result:
            continue
        for key, value in result.items():
            if osp.dirname(key) == parent:
                parents.add({parent: value})
    result.update(parents.compute_mean())
    return result


def _to_python(v):
    if isinstance(v, torch.Tensor):
        v = v.detach().cpu()
    return v.item() if hasattr(v, "item") else v


class PoseEstimationEvaluator:
    """``PoseEstimationEvaluator(batches, eval_func)()`` -> result dict on rank 0, ``{}`` elsewhere.

    ``eval_func(**batch)`` (or ``(*batch)`` / ``(batch)``) returns a dict of scalars; keys are
    prefixed with ``validation/main/`` like ``chainer.report`` under the trainer does."""

    def __init__(self, iterator, eval_func, converter=None, prefix="validation/main/", group=None):
        self._iterator = iterator
        self._eval_func = eval_func
        self._converter = converter
        self._prefix = prefix
        self._group = group

    def evaluate(self):
        it = self._iterator
        if hasattr(it, "reset"):
            it.reset()
        local = []
        for batch in it:
            if self._converter is not None:
                batch = self._converter(batch)
            with torch.no_grad():
                if isinstance(batch, tuple):
                    out = self._eval_func(*batch)
                elif isinstance(batch, dict):
                    out = self._eval_func(**batch)
                else:
                    out = self._eval_func(batch)
            local.append({self._prefix + k: _to_python(v) for k, v in (out or {}).items()})
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self._group) > 1:
            rank = dist.get_rank(self._group)
            gathered = [None] * dist.get_world_size(self._group) if rank == 0 else None
            dist.gather_object(local, gathered, dst=dist.get_global_rank(self._group, 0)
                               if self._group is not None else 0, group=self._group)
            if rank != 0:
                return {}
            local = [row for part in gathered for row in part]
        return summarize_observations(local, self._prefix)

    __call__ = evaluate
