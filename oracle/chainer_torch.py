"""Forward-only stand-in for the slice of Chainer the reference's pose network uses, evaluated with
torch on the CPU -- TEST INFRASTRUCTURE ONLY (``oracle/gen_golden_predict.py``, build container).

Purpose: run the reference's OWN network code -- ``models/dense_fusion/resnet.py``, ``pspnet.py``
and ``contrib/singleview_3d/models/model.py`` (``Model.__init__``, ``predict``, ``_extract``,
``_voxelize``) -- end to end, with weights injected from a ``morefusion_amd`` model through the
pinned parameter paths, to obtain golden outputs for ``Model.predict``.  What this module supplies
is only the library layer underneath that code:

* ``chainer.Chain / Link`` (``init_scope``, child registration, ``namedparams``),
  ``chainercv.links.PickableSequentialChain`` (children applied in registration order);
* ``L.Convolution{1,2,3}D`` (``W [out,in,*k]``, ``b``; stride / pad / dilate -> ``torch.conv{1,2,3}d``),
  ``L.PReLU`` (one shared slope);
* ``F.relu, concat, stack, dropout (inference: identity), resize_images (bilinear, align_corners),
  max_pooling_2d(cover_all), average_pooling_2d, log_softmax, sigmoid, normalize (x / (|x| + eps),
  eps = 1e-5: chainer/functions/normalization/l2_normalization.py), mean, sum, argmax``.
The reference's own voxel ops inside the network (average_voxelization_3d, interpolate_voxel_grid)
run as CUDA text through ``oracle/cuda_text.py``.
"""
import contextlib

import numpy as np
import torch
import torch.nn.functional as TF

from . import chainer_tape as T

Variable = T.Variable


def _a(x):
    return np.asarray(T.unwrap(x))


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(_a(x), dtype=np.float32))


def _v(t):
    return Variable(t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t))


class Link:
    xp = np

    def __init__(self):
        object.__setattr__(self, "_params", {})
        object.__setattr__(self, "_children", {})
        object.__setattr__(self, "_order", [])
        object.__setattr__(self, "_in_scope", False)

    @contextlib.contextmanager
    def init_scope(self):
        object.__setattr__(self, "_in_scope", True)
        try:
            yield
        finally:
            object.__setattr__(self, "_in_scope", False)

    def __setattr__(self, name, value):
        if getattr(self, "_in_scope", False):
            if isinstance(value, Link):
                self._children[name] = value
            if callable(value) or isinstance(value, Link):
                if name not in self._order:
                    self._order.append(name)
        object.__setattr__(self, name, value)

    def namedlinks(self, prefix=""):
        yield prefix, self
        for name, child in self._children.items():
            yield from child.namedlinks(prefix + "/" + name)

    def __call__(self, *a, **k):
        return self.forward(*a, **k)


class Chain(Link):
    pass


class PickableSequentialChain(Chain):
    """chainercv.links.PickableSequentialChain with the default ``pick`` (the last layer)."""

    def __call__(self, x):
        for name in self._order:
            x = getattr(self, name)(x)
        return x


def _conv(nd):
    class Conv(Link):
        def __init__(self, in_channels, out_channels, ksize=None, stride=1, pad=0, nobias=False, dilate=1, **kw):
            super().__init__()
            self.stride, self.pad, self.dilate, self.nobias = stride, pad, dilate, nobias
            self.W, self.b = None, None  # injected (numpy)

        def __call__(self, x):
            fn = (TF.conv1d, TF.conv2d, TF.conv3d)[nd - 1]
            b = None if self.nobias else torch.from_numpy(self.b)
            with torch.no_grad():
                return _v(fn(_t(x), torch.from_numpy(self.W), b, stride=self.stride, padding=self.pad,
                             dilation=self.dilate))
    return Conv


class PReLU(Link):
    def __init__(self, shape=(), init=0.25):
        super().__init__()
        self.W = None

    def __call__(self, x):
        a = _a(x)
        return Variable(np.where(a > 0, a, np.float32(self.W) * a).astype(np.float32))


# ---- chainer.functions (inference) ------------------------------------------------------------------
def relu(x):
    return Variable(np.maximum(_a(x), 0))


def concat(xs, axis=1):
    return Variable(np.concatenate([_a(x) for x in xs], axis=axis))


def stack(xs, axis=0):
    return Variable(np.stack([_a(x) for x in xs], axis=axis))


def dropout(x, ratio=0.5):
    return x  # chainer.config.train is False


def resize_images(x, output_shape):
    with torch.no_grad():
        return _v(TF.interpolate(_t(x), size=tuple(int(v) for v in output_shape), mode="bilinear", align_corners=True))


def max_pooling_2d(x, ksize, stride=None, pad=0, cover_all=True):
    with torch.no_grad():
        return _v(TF.max_pool2d(_t(x), ksize, stride or ksize, pad, ceil_mode=bool(cover_all)))


def average_pooling_2d(x, ksize, stride=None, pad=0):
    with torch.no_grad():
        return _v(TF.avg_pool2d(_t(x), tuple(int(k) for k in ksize) if not np.isscalar(ksize) else int(ksize),
                                tuple(int(k) for k in stride) if not np.isscalar(stride) else int(stride), pad))


def log_softmax(x, axis=1):
    with torch.no_grad():
        return _v(TF.log_softmax(_t(x), dim=axis))


def sigmoid(x):
    with torch.no_grad():
        return _v(torch.sigmoid(_t(x)))


def normalize(x, eps=1e-5, axis=1):
    a = _a(x)
    norm = np.sqrt(np.sum(a * a, axis=axis, keepdims=True)) + np.float32(eps)
    return Variable(a / norm)


def mean(x, axis=None):
    return Variable(np.mean(_a(x), axis=axis))


def fsum(x, axis=None):
    return Variable(np.sum(_a(x), axis=axis))


def argmax(x, axis=None):
    return Variable(np.argmax(_a(x), axis=axis))
