"""A reverse-mode tape over NumPy with chainer's differentiation rules -- TEST INFRASTRUCTURE ONLY.

Purpose: obtain the GRADIENTS of the reference's ICC / ICP links by running the reference's own
code (``oracle/gen_golden_cuda.py``).  The links' ``forward`` is a graph of
* the reference's own ``chainer.Function`` subclasses, whose ``backward`` / ``backward_gpu`` methods
  are reference code and are EXECUTED as they stand (QuaternionMatrix, ComposeTransform,
  TruncatedDistanceFunction -- the latter's backward is CUDA text, run through oracle/cuda_text.py);
* chainer built-ins (``+ - * / **``, ``F.sum / sqrt / repeat / concat / stack / matmul / maximum /
  argmin``, ``getitem``, ``transpose``), whose elementary rules this module restates (chainer is not
  installed).  The only rule that is not textbook calculus is ``F.maximum``: the gradient goes to
  ``x1`` where ``x1 >= x2`` and to ``x2`` elsewhere (chainer/functions/math/maximum.py).
Everything is evaluated in the dtype of the operands (float32 for the links), accumulation of
several gradient contributions into one variable in the order the tape replays them (reverse
creation order), like chainer's backward pass.
"""
import numpy as np


def _unb(g, shape):
    """Sum a broadcast gradient back to ``shape``."""
    g = np.asarray(g)
    while g.ndim > len(shape):
        g = g.sum(axis=0)
    for ax, n in enumerate(shape):
        if n == 1 and g.shape[ax] != 1:
            g = g.sum(axis=ax, keepdims=True)
    return g


class Variable:
    __array_ufunc__ = None
    _tape = []  # creation-ordered list of (outputs tuple, inputs, backward_fn)

    def __init__(self, array, requires_grad=False):
        self.array = np.asarray(array)
        self.grad = None
        self.requires_grad = requires_grad

    data = property(lambda self: self.array)
    shape = property(lambda self: self.array.shape)
    ndim = property(lambda self: self.array.ndim)
    dtype = property(lambda self: self.array.dtype)
    size = property(lambda self: self.array.size)

    @property
    def T(self):
        return _op(self.array.T, [self], lambda g: (g.T,))

    def __len__(self):
        return len(self.array)

    def __getitem__(self, k):
        k = unwrap(k)
        shape = self.array.shape
        dt = self.array.dtype

        if isinstance(k, np.ndarray) and k.dtype == bool:  # boolean mask -> integer indices
            k = np.nonzero(k)
        kk = k if isinstance(k, tuple) else (k,)
        advanced = any(isinstance(i, (np.ndarray, list)) for i in kk)

        def bwd(g):
            gx = np.zeros(shape, dt)
            if advanced:
                np.add.at(gx, k, g)  # repeated indices accumulate
            else:
                gx[k] += g           # basic indexing (ints, slices, None): a view, no repeats
            return (gx,)

        return _op(self.array[k], [self], bwd)

    def transpose(self, *axes):
        axes = axes[0] if len(axes) == 1 and isinstance(axes[0], (tuple, list)) else axes
        inv = np.argsort(axes)
        return _op(self.array.transpose(*axes), [self], lambda g: (g.transpose(*inv),))

    def reshape(self, *shape):
        old = self.array.shape
        return _op(self.array.reshape(*shape), [self], lambda g: (g.reshape(old),))

    def __neg__(self):
        return _op(-self.array, [self], lambda g: (-g,))

    def __add__(self, o):
        return _binary(self, o, np.add, lambda g, a, b: (g, g))

    __radd__ = __add__

    def __sub__(self, o):
        return _binary(self, o, np.subtract, lambda g, a, b: (g, -g))

    def __rsub__(self, o):
        return _binary(o, self, np.subtract, lambda g, a, b: (g, -g))

    def __mul__(self, o):
        return _binary(self, o, np.multiply, lambda g, a, b: (g * b, g * a))

    __rmul__ = __mul__

    def __truediv__(self, o):
        return _binary(self, o, np.divide, lambda g, a, b: (g / b, -g * a / (b * b)))

    def __rtruediv__(self, o):
        return _binary(o, self, np.divide, lambda g, a, b: (g / b, -g * a / (b * b)))

    def __pow__(self, p):
        assert np.isscalar(p)
        a = self.array
        return _op(a ** p, [self], lambda g: (g * (a.dtype.type(p) * a ** (p - 1)),))

    def backward(self):
        """Reverse sweep from this (scalar) variable; fills ``.grad`` of every variable on the tape."""
        self.grad = np.ones_like(self.array)
        for outs, inputs, bwd in reversed(Variable._tape):
            if all(o.grad is None for o in outs):
                continue
            gys = tuple(o.grad if o.grad is not None else np.zeros_like(o.array) for o in outs)
            grads = bwd(gys[0]) if len(outs) == 1 else bwd(gys)
            for v, g in zip(inputs, grads):
                if g is None or not isinstance(v, Variable):
                    continue
                g = np.asarray(g, dtype=v.array.dtype)
                v.grad = g.copy() if v.grad is None else v.grad + g


def unwrap(x):
    if isinstance(x, Variable):
        return x.array
    if isinstance(x, tuple):
        return tuple(unwrap(v) for v in x)
    return x


def _op(value, inputs, bwd):
    out = Variable(value)
    Variable._tape.append(((out,), inputs, bwd))
    return out


def _binary(a, b, fn, rule):
    av, bv = unwrap(a), unwrap(b)
    av_, bv_ = np.asarray(av), np.asarray(bv)
    if not isinstance(a, Variable) and av_.dtype != bv_.dtype and av_.ndim == 0:
        av_ = av_.astype(bv_.dtype)  # Python scalars take the array's dtype (chainer / NumPy rule)
    if not isinstance(b, Variable) and av_.dtype != bv_.dtype and bv_.ndim == 0:
        bv_ = bv_.astype(av_.dtype)

    def bwd(g):
        ga, gb = rule(g, av_, bv_)
        return (_unb(ga, av_.shape), _unb(gb, bv_.shape))

    return _op(fn(av_, bv_), [a, b], bwd)


def reset():
    Variable._tape = []


# ---- chainer.functions ---------------------------------------------------------------------------
def F_sum(x, axis=None, keepdims=False):
    a = unwrap(x)
    shape = a.shape

    def bwd(g):
        g = np.asarray(g)
        if axis is not None and not keepdims:
            g = np.expand_dims(g, axis)
        return (np.broadcast_to(g, shape).copy(),)

    return _op(np.sum(a, axis=axis, keepdims=keepdims), [x], bwd)


def F_mean(x, axis=None):
    a = unwrap(x)
    n = a.size if axis is None else a.shape[axis]
    return F_sum(x, axis=axis) / a.dtype.type(n)


def F_log(x):
    a = np.asarray(unwrap(x))
    return _op(np.log(a), [x], lambda g: (g / a,))


def F_sqrt(x):
    y = np.sqrt(unwrap(x))
    return _op(y, [x], lambda g: (g / (2 * y),))


def F_repeat(x, n, axis):
    a = unwrap(x)
    shape = a.shape

    def bwd(g):
        return (g.reshape(shape[:axis] + (shape[axis], n) + shape[axis + 1:]).sum(axis=axis + 1),)

    return _op(np.repeat(a, n, axis=axis), [x], bwd)


def F_concat(xs, axis=1):
    arrs = [np.asarray(unwrap(x)) for x in xs]
    sizes = np.cumsum([a.shape[axis] for a in arrs])[:-1]
    return _op(np.concatenate(arrs, axis=axis), list(xs), lambda g: tuple(np.split(g, sizes, axis=axis)))


def F_stack(xs, axis=0):
    arrs = [np.asarray(unwrap(x)) for x in xs]
    return _op(np.stack(arrs, axis=axis), list(xs),
               lambda g: tuple(np.take(g, i, axis=axis) for i in range(len(arrs))))


def F_matmul(a, b):
    av, bv = np.asarray(unwrap(a)), np.asarray(unwrap(b))

    def bwd(g):
        ga = np.matmul(g, np.swapaxes(bv, -1, -2))
        gb = np.matmul(np.swapaxes(av, -1, -2), g)
        return (_unb(ga, av.shape), _unb(gb, bv.shape))

    return _op(np.matmul(av, bv), [a, b], bwd)


def F_maximum(a, b):
    av, bv = np.asarray(unwrap(a)), np.asarray(unwrap(b))
    cond = av >= bv  # chainer/functions/math/maximum.py: ties go to x1
    return _op(np.maximum(av, bv), [a, b], lambda g: (np.where(cond, g, 0), np.where(cond, 0, g)))


def F_min(x, axis=None):
    a = np.asarray(unwrap(x))
    y = np.min(a, axis=axis)
    cond = a == (y if axis is None else np.expand_dims(y, axis))  # chainer's SelectorBase: every tied position

    def bwd(g):
        g = np.asarray(g)
        return ((g if axis is None else np.expand_dims(g, axis)) * cond,)

    return _op(y, [x], bwd)


def F_relu(x):
    a = np.asarray(unwrap(x))
    return _op(np.maximum(a, 0), [x], lambda g: (g * (a > 0),))


def F_minimum(a, b):
    av, bv = np.asarray(unwrap(a)), np.asarray(unwrap(b))
    cond = av <= bv  # chainer/functions/math/minimum.py: ties go to x1
    return _op(np.minimum(av, bv), [a, b], lambda g: (np.where(cond, g, 0), np.where(cond, 0, g)))


def F_argmin(x, axis=None):
    return Variable(np.argmin(unwrap(x), axis=axis))


class Function:
    """chainer.Function (old style): ``forward(inputs) -> tuple``, ``backward(inputs, grad_outputs)``;
    the GPU pair is preferred when the subclass defines it (the arrays are NumPy, the code under
    test is the reference's GPU branch)."""

    def _pick(self, base):
        owned = {k for c in type(self).__mro__[:-2] for k in vars(c)}
        for name in (base + "_gpu", base + "_cpu", base):
            if name in owned:
                return getattr(self, name)
        raise NotImplementedError(base)

    def __call__(self, *inputs):
        arrays = tuple(np.asarray(unwrap(x)) for x in inputs)
        outs = self._pick("forward")(arrays)
        fn = self
        vs = tuple(Variable(o) for o in outs)

        def bwd(gys):  # ONE call with the gradients of all outputs, like chainer
            gys = gys if isinstance(gys, tuple) else (gys,)
            return tuple(fn._pick("backward")(arrays, gys))

        Variable._tape.append((vs, list(inputs), bwd))
        return vs[0] if len(vs) == 1 else vs

    def retain_inputs(self, idx):
        pass
