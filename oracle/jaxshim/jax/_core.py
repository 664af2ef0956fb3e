"""Array type of the jax stand-in: a torch.Tensor subclass with numpy method semantics.  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import numpy as np
import torch

_FLOAT = [torch.float64]


def set_float_dtype(dt):
    """torch.float64 (truth) or torch.float32 (expected rounding); jnp.float32 maps to this."""
    _FLOAT[0] = dt


def float_dtype():
    return _FLOAT[0]


def canon_dtype(dt):
    if dt is None:
        return None
    if dt in (float, np.float32, np.float64, torch.float32, torch.float64, "float32", "float64"):
        return _FLOAT[0]
    if dt in (int, np.int32, torch.int32, "int32"):
        return torch.int32
    if dt in (np.int64, torch.int64):
        return torch.int64
    if dt in (np.uint8, torch.uint8):
        return torch.uint8
    if dt in (bool, np.bool_, torch.bool):
        return torch.bool
    if dt in (np.uint32,):
        return torch.int64
    if isinstance(dt, torch.dtype):
        return dt
    raise TypeError(f"unsupported dtype {dt!r}")


def _axis(axis):
    if isinstance(axis, list):
        return tuple(axis)
    return axis


class Array(torch.Tensor):
    """numpy-flavoured tensor: .astype, .min/.max(axis) -> values, .repeat = np.repeat, .transpose = permutation."""

    def astype(self, dt):
        return self.to(canon_dtype(dt))

    def copy(self):
        return self.clone()

    def __array__(self, dtype=None, copy=None):
        a = self.detach().as_subclass(torch.Tensor).cpu().numpy()
        return a.astype(dtype) if dtype is not None else a

    def mean(self, axis=None, keepdims=False, dtype=None):
        t = self.as_subclass(torch.Tensor)
        if not t.is_floating_point():
            t = t.to(_FLOAT[0])
        r = t.mean() if axis is None else t.mean(dim=_axis(axis), keepdim=keepdims)
        return r.as_subclass(Array)

    def sum(self, axis=None, keepdims=False, dtype=None):
        t = self.as_subclass(torch.Tensor)
        r = t.sum() if axis is None else t.sum(dim=_axis(axis), keepdim=keepdims)
        return r.as_subclass(Array)

    def min(self, axis=None, keepdims=False):
        t = self.as_subclass(torch.Tensor)
        r = t.amin() if axis is None else t.amin(dim=_axis(axis), keepdim=keepdims)
        return r.as_subclass(Array)

    def max(self, axis=None, keepdims=False):
        t = self.as_subclass(torch.Tensor)
        r = t.amax() if axis is None else t.amax(dim=_axis(axis), keepdim=keepdims)
        return r.as_subclass(Array)

    def repeat(self, repeats, axis=None):
        t = self.as_subclass(torch.Tensor)
        if axis is None:
            t, axis = t.reshape(-1), 0
        return torch.repeat_interleave(t, int(repeats), dim=axis).as_subclass(Array)

    def transpose(self, *axes):
        if len(axes) == 1 and isinstance(axes[0], (tuple, list)):
            axes = tuple(axes[0])
        t = self.as_subclass(torch.Tensor)
        if not axes:
            axes = tuple(reversed(range(t.dim())))
        return t.permute(*axes).as_subclass(Array)

    @property
    def T(self):
        return self.transpose()

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        return self.as_subclass(torch.Tensor).reshape(*[int(s) for s in shape]).as_subclass(Array)

    def squeeze(self, axis=None):
        t = self.as_subclass(torch.Tensor)
        return (t.squeeze() if axis is None else t.squeeze(axis)).as_subclass(Array)

    def item(self):
        return self.detach().as_subclass(torch.Tensor).item()

    def __hash__(self):
        return id(self)

    def __deepcopy__(self, memo):
        return self.detach().clone().as_subclass(Array)


def asarray(x, dtype=None):
    """anything array-like -> Array (floats in the configured float dtype unless a dtype is given)."""
    dt = canon_dtype(dtype)
    if isinstance(x, torch.Tensor):
        t = x if dt is None or x.dtype == dt else x.to(dt)
        return t if isinstance(t, Array) else t.as_subclass(Array)
    if isinstance(x, (list, tuple)) and len(x) and any(isinstance(e, torch.Tensor) for e in x):
        t = torch.stack([asarray(e).as_subclass(torch.Tensor) for e in x])
        return (t if dt is None else t.to(dt)).as_subclass(Array)
    a = np.asarray(x)
    if dt is None:
        if a.dtype.kind == "f":
            dt = _FLOAT[0]
        elif a.dtype == np.uint32:
            dt = torch.int64
        elif a.dtype.kind in "iu" and a.dtype.itemsize == 8:
            dt = torch.int64 if a.dtype.kind == "i" else torch.int64
    if a.dtype == np.uint32 or a.dtype == np.uint64:
        a = a.astype(np.int64)
    t = torch.from_numpy(np.ascontiguousarray(a)) if a.ndim else torch.tensor(a.item() if a.dtype.kind != "b" else bool(a))
    if a.ndim == 0 and a.dtype.kind == "f" and dt is None:
        dt = _FLOAT[0]
    if dt is not None and t.dtype != dt:
        t = t.to(dt)
    return t.as_subclass(Array)


def raw(x):
    return asarray(x).as_subclass(torch.Tensor)
