"""jax.numpy subset used by the reference's update path, on torch tensors.  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import math

import numpy as np
import torch

from ._core import Array, asarray as _as, canon_dtype, float_dtype, raw

ndarray = Array
float32 = torch.float32
float64 = torch.float64
int32 = torch.int32
int64 = torch.int64
uint8 = torch.uint8
uint32 = torch.int64
bool_ = torch.bool
dtype = torch.dtype
newaxis = None
pi = math.pi
inf = math.inf


def _w(t):
    return t.as_subclass(Array)


def _ax(axis):
    return tuple(axis) if isinstance(axis, list) else axis


def array(x, dtype=None):
    return _as(x, dtype)


def zeros(shape, dtype=None):
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    return _w(torch.zeros(shape, dtype=canon_dtype(dtype) or float_dtype()))


def ones(shape, dtype=None):
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    return _w(torch.ones(shape, dtype=canon_dtype(dtype) or float_dtype()))


def full(shape, fill_value, dtype=None):
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    fv = raw(fill_value)
    out = torch.zeros(shape, dtype=canon_dtype(dtype) or (fv.dtype if fv.is_floating_point() else float_dtype())) + fv
    return _w(out)


def zeros_like(x, dtype=None):
    return _w(torch.zeros_like(raw(x), dtype=canon_dtype(dtype)))


def ones_like(x, dtype=None):
    return _w(torch.ones_like(raw(x), dtype=canon_dtype(dtype)))


def arange(*a, dtype=None):
    return _w(torch.arange(*a, dtype=canon_dtype(dtype)))


def linspace(a, b, n):
    return _w(torch.linspace(a, b, n, dtype=float_dtype()))


def meshgrid(*xs, indexing="xy"):
    return [_w(t) for t in torch.meshgrid(*[raw(x) for x in xs], indexing=indexing)]


def reshape(x, shape):
    return _as(x).reshape(tuple(int(s) for s in shape) if not isinstance(shape, int) else (shape,))


def transpose(x, axes=None):
    return _as(x).transpose(*(axes or ()))


def concatenate(xs, axis=0):
    ts = [raw(x) for x in xs]
    dt = ts[0].dtype
    for t in ts[1:]:
        dt = torch.promote_types(dt, t.dtype)
    return _w(torch.cat([t.to(dt) for t in ts], dim=axis))


def stack(xs, axis=0):
    return _w(torch.stack([raw(x) for x in xs], dim=axis))


def expand_dims(x, axis):
    return _w(raw(x).unsqueeze(axis))


def squeeze(x, axis=None):
    return _as(x).squeeze(axis)


def ndim(x):
    return raw(x).dim() if not isinstance(x, (int, float)) else 0


def shape(x):
    return tuple(raw(x).shape)


def broadcast_to(x, shape):
    return _w(raw(x).broadcast_to(tuple(shape)))


def _red(name):
    def f(x, axis=None, keepdims=False, **kw):
        return getattr(_as(x), name)(axis=_ax(axis), keepdims=keepdims)
    return f


mean, sum, min, max = _red("mean"), _red("sum"), _red("min"), _red("max")  # noqa: A001


def _un(fn):
    def f(x):
        t = raw(x)
        if not t.is_floating_point():
            t = t.to(float_dtype())
        return _w(fn(t))
    return f


exp, log, sqrt, tanh, abs, square, log1p, sign = (_un(torch.exp), _un(torch.log), _un(torch.sqrt), _un(torch.tanh),  # noqa: A001
                                                 _un(torch.abs), _un(torch.square), _un(torch.log1p), _un(torch.sign))
logaddexp = lambda a, b: _w(torch.logaddexp(raw(a), raw(b)))  # noqa: E731


def clip(x, a_min=None, a_max=None, **kw):
    a_min = kw.get("min", a_min)
    a_max = kw.get("max", a_max)
    return _w(torch.clamp(raw(x), a_min, a_max))


def minimum(a, b):
    return _w(torch.minimum(raw(a), raw(b)))


def maximum(a, b):
    return _w(torch.maximum(raw(a), raw(b)))


def where(c, a, b):
    ta, tb = raw(a), raw(b)
    dt = torch.promote_types(ta.dtype, tb.dtype)
    return _w(torch.where(raw(c).bool(), ta.to(dt), tb.to(dt)))


def einsum(eq, *xs):
    return _w(torch.einsum(eq, *[raw(x) for x in xs]))


def matmul(a, b):
    return _w(raw(a) @ raw(b))


dot = matmul


def pad(x, pad_width, mode="constant", constant_values=0):
    t = raw(x)
    pw = [tuple(int(v) for v in p) for p in pad_width]
    assert len(pw) == t.dim()
    if mode == "edge":   # replicate the border element (numpy 'edge'); index arithmetic keeps integer dtypes exact
        for d, (lo, hi) in enumerate(pw):
            if lo or hi:
                n = t.shape[d]
                idx = torch.clamp(torch.arange(-lo, n + hi), 0, n - 1)
                t = t.index_select(d, idx)
        return _w(t)
    assert mode == "constant"
    flat = []
    for lo, hi in reversed(pw):
        flat += [lo, hi]
    return _w(torch.nn.functional.pad(t, flat, value=constant_values))


def isfinite(x):
    return _w(torch.isfinite(raw(x)))


def all(x):  # noqa: A001
    return _w(raw(x).all())


def argmax(x, axis=None):
    return _w(raw(x).argmax(dim=axis))


def cumsum(x, axis=0):
    return _w(raw(x).cumsum(dim=axis))


def asarray(x, dtype=None):
    return _as(x, dtype)
