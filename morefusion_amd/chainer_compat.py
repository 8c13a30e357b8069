"""The handful of Chainer / CuPy names the reference's driver scripts use, over torch.

SURVEY.md 8b: the drivers on the path (examples/ycb_video/pose_refinement/
check_iterative_collision_check_link.py:3-4,29-79, check_iterative_closest_point_link.py:3-4,
singleview_3d/demo.py:51-112, evaluate.py:257-291) take ``chainer.optimizers.Adam``,
``chainer.backends.cuda.to_gpu / to_cpu``, ``Variable.array``, ``chainer.no_backprop_mode``,
``chainer.using_config``, ``chainer.dataset.concat_examples`` and
``chainer.serializers.load_npz`` from Chainer itself.  With

    import morefusion_amd as morefusion
    from morefusion_amd import chainer_compat as chainer
    from morefusion_amd.chainer_compat import cuda

their loop bodies run textually unchanged: arrays are ``torch.Tensor`` on the MI355X instead of
``cupy.ndarray``.  Nothing here computes anything; the HIP library stays the only implementation
of the ops (moving a tensor "to the GPU" without one raises).
"""
import contextlib
import types

import numpy as np
import torch

from . import optimizers as _optimizers
from . import serializers as _serializers

optimizers = types.SimpleNamespace(Adam=_optimizers.Adam)
serializers = types.SimpleNamespace(load_npz=_serializers.load_npz, save_npz=_serializers.save_npz)
config = types.SimpleNamespace(train=True, enable_backprop=True)


def _array_alias():
    """``Variable.array`` (and ``ndarray.get()``-style hand-over): a detached view."""
    if not hasattr(torch.Tensor, "array"):
        torch.Tensor.array = property(lambda self: self.detach())


_array_alias()


class _Cuda:
    """chainer.backends.cuda"""

    @staticmethod
    def to_gpu(array, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("cuda.to_gpu: no MI355X visible (morefusion_amd has no CPU fallback)")
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        if isinstance(array, torch.Tensor):
            return array.to(dev)
        return torch.as_tensor(np.ascontiguousarray(array)).to(dev)

    @staticmethod
    def to_cpu(array):
        if isinstance(array, torch.Tensor):
            return array.detach().cpu().numpy()
        return np.asarray(array)

    @staticmethod
    def get_array_module(*arrays):
        return torch if any(isinstance(a, torch.Tensor) for a in arrays) else np


cuda = _Cuda()
backends = types.SimpleNamespace(cuda=cuda)


class ArrayModule:
    """``link.xp`` of a Chainer link on the GPU (cupy): the handful of array constructors the
    reference's drivers call through it (``link.xp.asarray``, ``model.xp.arange / argmax``,
    ``xp.concatenate / hstack / vstack / isnan / zeros / array``), producing torch tensors on the
    link's device."""

    def __init__(self, device_of):
        self._device_of = device_of

    def _dev(self):
        return self._device_of()

    def asarray(self, a, dtype=None):
        t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(a))
        t = t.to(self._dev())
        return t if dtype is None else t.to(_torch_dtype(dtype))

    array = asarray

    def arange(self, *a, dtype=None):
        return torch.arange(*a, device=self._dev(), dtype=None if dtype is None else _torch_dtype(dtype))

    def zeros(self, shape, dtype=np.float32):
        return torch.zeros(shape, device=self._dev(), dtype=_torch_dtype(dtype))

    def argmax(self, a, axis=None):
        return torch.argmax(a) if axis is None else torch.argmax(a, dim=axis)

    def isnan(self, a):
        return torch.isnan(a)

    def concatenate(self, xs, axis=0):
        return torch.cat([self.asarray(x) for x in xs], dim=axis)

    def hstack(self, xs):
        return torch.hstack([self.asarray(x) for x in xs])

    def vstack(self, xs):
        return torch.vstack([self.asarray(x) for x in xs])


def _torch_dtype(dtype):
    if isinstance(dtype, torch.dtype):
        return dtype
    return torch.from_numpy(np.zeros(0, dtype=np.dtype(dtype))).dtype


def link_xp(module):
    """``xp`` for a torch module standing in for a Chainer link: arrays land where its parameters are."""
    return ArrayModule(lambda: next(module.parameters()).device)


def no_backprop_mode():
    return torch.no_grad()


@contextlib.contextmanager
def using_config(name, value):
    """``chainer.using_config('train', False)`` / ``('enable_backprop', False)``."""
    if not hasattr(config, name):
        raise AttributeError(f"unknown config entry {name!r}")
    old = getattr(config, name)
    setattr(config, name, value)
    try:
        if name == "enable_backprop" and not value:
            with torch.no_grad():
                yield
        else:
            yield
    finally:
        setattr(config, name, old)


def _stack(items, device, padding):
    arrays = [x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x) for x in items]
    if padding is None:
        out = np.stack(arrays)
    else:  # pad every example up to the largest shape (chainer.dataset.concat_examples)
        shape = np.max([a.shape for a in arrays], axis=0)
        out = np.full((len(arrays),) + tuple(shape), padding, dtype=arrays[0].dtype)
        for i, a in enumerate(arrays):
            out[(i,) + tuple(slice(0, s) for s in a.shape)] = a
    if device is None or (isinstance(device, int) and device < 0):
        return out
    return cuda.to_gpu(out, device)


def concat_examples(batch, device=None, padding=None):
    """List of examples (dicts, tuples or arrays) -> one batched container of the same kind;
    ``device=None`` / negative keeps NumPy arrays, ``device >= 0`` gives tensors on that GPU."""
    if len(batch) == 0:
        raise ValueError("batch is empty")
    first = batch[0]
    if isinstance(first, dict):
        pad = padding if isinstance(padding, dict) else {k: padding for k in first}
        return {k: _stack([ex[k] for ex in batch], device, pad[k]) for k in first}
    if isinstance(first, tuple):
        pad = padding if isinstance(padding, tuple) else (padding,) * len(first)
        return tuple(_stack([ex[i] for ex in batch], device, pad[i]) for i in range(len(first)))
    return _stack(batch, device, padding)


dataset = types.SimpleNamespace(concat_examples=concat_examples)
