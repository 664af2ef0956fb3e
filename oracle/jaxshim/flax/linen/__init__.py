"""flax.linen stand-in: Module system (compact naming, adoption by attachment name, sharing by reference), the layers
and transforms the reference's networks use.  TEST INFRASTRUCTURE ONLY -- see oracle/jaxshim/README.md.

Restated flax behaviour (flax >= 0.8 defaults):
  * modules are dataclasses; `parent` / `name` are keyword-only fields;
  * a module constructed inside a compact method is a child of the running module, auto-named `<Class>_<i>` unless a
    name is given;
  * a module passed from outside as an attribute (possibly inside a dict / list / tuple) is ADOPTED by the first bound
    module that owns it and NAMED BY ATTACHMENT (`<attr>` or `<attr>_<key>`), whatever name it was constructed with
    (flax_preserve_adopted_names = False); an instance shared between several owners is adopted once (sharing by
    reference), so its parameters live under the first owner in structural order (dataclass-field order, dict keys
    sorted) -- which is why the reference's load_resnet10_params patches `modules_actor` only;
  * `init` returns plain nested dicts; `apply` reads them; rng streams are folded with the module path.
"""
from __future__ import annotations

import dataclasses
import functools
import math
from typing import Any, Callable, Optional

import numpy as np
import torch
import torch.nn.functional as F

import jax
from jax import numpy as jnp
from jax import random as jrandom
from jax._core import Array, asarray, raw
from jax.nn import elu, gelu, initializers, leaky_relu, log_softmax, relu, sigmoid, silu, softmax, softplus, swish, tanh  # noqa: F401

from . import module  # noqa: F401  (the reference annotates a field with `nn.module`)

_STACK = []        # modules whose method is currently running


class _Scope:
    """Root state of one init/apply."""

    def __init__(self, params, rngs, initializing):
        self.params, self.rngs, self.initializing = params, rngs or {}, initializing
        self.rng_counters = {}


def _walk_modules(value, suffix=""):
    """(suffix, module) for every Module in an attribute value; dict keys sorted (jax pytree order)."""
    if isinstance(value, Module):
        yield suffix, value
    elif isinstance(value, dict):
        for k in sorted(value.keys()):
            yield from _walk_modules(value[k], f"{suffix}_{k}")
    elif isinstance(value, (list, tuple)):
        for i, v in enumerate(value):
            yield from _walk_modules(v, f"{suffix}_{i}")


def _clone_value(value, memo):
    if isinstance(value, Module):
        return value._deep_clone(memo)
    if isinstance(value, dict):
        return type(value)({k: _clone_value(v, memo) for k, v in value.items()}) if type(value) is not dict else \
            {k: _clone_value(v, memo) for k, v in value.items()}
    if isinstance(value, list):
        return [_clone_value(v, memo) for v in value]
    if isinstance(value, tuple) and not hasattr(value, "_fields"):
        return tuple(_clone_value(v, memo) for v in value)
    return value


def compact(fn):
    fn._compact = True
    return fn


def _wrap_method(fn):
    if getattr(fn, "_wrapped", False):
        return fn

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        if self._scope is None:
            raise RuntimeError(f"module {type(self).__name__} is not bound: call it through init()/apply()")
        top = not _STACK or _STACK[-1] is not self
        if top:
            self._autonames = {}
        _STACK.append(self)
        try:
            return fn(self, *a, **k)
        finally:
            _STACK.pop()
    wrapped._wrapped = True
    return wrapped


class Module:
    name: Optional[str] = dataclasses.field(default=None, kw_only=True)
    parent: Any = dataclasses.field(default=None, kw_only=True, repr=False)

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        if "__call__" in cls.__dict__:
            cls.__call__ = _wrap_method(cls.__dict__["__call__"])
        ann = cls.__dict__.get("__annotations__", {})
        # (re)apply dataclass: own annotations become fields, inherited ones are kept
        cls.__annotations__ = dict(ann)
        dataclasses.dataclass(eq=False, repr=False)(cls)

    # ---- life cycle --------------------------------------------------------------------------
    def __post_init__(self):
        object.__setattr__(self, "_scope", None)
        object.__setattr__(self, "_path", None)
        object.__setattr__(self, "_autonames", {})
        object.__setattr__(self, "_children", {})
        if self.parent is None and _STACK:          # constructed inside a running module: its child
            par = _STACK[-1]
            nm = self.name
            if nm is None:
                base = type(self).__name__
                i = par._autonames.get(base, 0)
                par._autonames[base] = i + 1
                nm = f"{base}_{i}"
            self._attach(par, nm)

    def _attach(self, parent, name):
        object.__setattr__(self, "parent", parent)
        object.__setattr__(self, "name", name)
        object.__setattr__(self, "_scope", parent._scope)
        object.__setattr__(self, "_path", parent._path + (name,))
        self._adopt_fields()

    def _adopt_fields(self):
        """Eager, recursive adoption of attribute modules at bind time, in field order."""
        for f in dataclasses.fields(self):
            if f.name in ("parent", "name"):
                continue
            for suffix, m in _walk_modules(getattr(self, f.name)):
                if m.parent is None:
                    m._attach(self, f"{f.name}{suffix}")   # named by attachment; a constructor name is dropped

    def _deep_clone(self, memo):
        if id(self) in memo:
            return memo[id(self)]
        kw = {f.name: _clone_value(getattr(self, f.name), memo) for f in dataclasses.fields(self)
              if f.init and f.name not in ("parent", "name")}
        saved = list(_STACK)
        _STACK.clear()                                # the clone must come out unbound
        try:
            new = type(self)(**kw, name=None)
        finally:
            _STACK.extend(saved)
        memo[id(self)] = new
        return new

    def clone(self, **upd):
        new = self._deep_clone({})
        for k, v in upd.items():
            object.__setattr__(new, k, v)
        return new

    def _bind_root(self, scope):
        root = self._deep_clone({})
        object.__setattr__(root, "_scope", scope)
        object.__setattr__(root, "_path", ())
        root._adopt_fields()
        return root

    # ---- public API --------------------------------------------------------------------------
    def init(self, rngs, *args, method=None, **kwargs):
        if not isinstance(rngs, dict):
            rngs = {"params": rngs}
        scope = _Scope({}, rngs, True)
        root = self._bind_root(scope)
        (getattr(root, method) if isinstance(method, str) else root)(*args, **kwargs)
        return {"params": scope.params}

    def apply(self, variables, *args, rngs=None, method=None, mutable=False, **kwargs):
        scope = _Scope(variables["params"], rngs, False)
        root = self._bind_root(scope)
        return (getattr(root, method) if isinstance(method, str) else root)(*args, **kwargs)

    def param(self, name, init_fn, *init_args):
        d = self._scope.params
        for p in self._path:
            if p not in d:
                if not self._scope.initializing:
                    raise KeyError(f"no parameters at {'/'.join(self._path)} (looking for '{name}')")
                d[p] = {}
            d = d[p]
        if name not in d:
            if not self._scope.initializing:
                raise KeyError(f"parameter '{name}' missing at {'/'.join(self._path)}")
            d[name] = asarray(init_fn(self.make_rng("params"), *init_args))
        return asarray(d[name])

    def has_rng(self, name):
        return name in self._scope.rngs

    def make_rng(self, name):
        if name not in self._scope.rngs:
            raise KeyError(f"{type(self).__name__} needs PRNG stream '{name}' (rngs={{...}})")
        key = (name, self._path)
        c = self._scope.rng_counters.get(key, 0)
        self._scope.rng_counters[key] = c + 1
        if jrandom._threefry_on():   # flax >= 0.8's own derivation (threefry.flax_fold_in_static): path names + call counter from 1
            from jax import threefry as _tf
            k = _tf.flax_fold_in_static(jrandom._tf_key(self._scope.rngs[name]), tuple(self._path) + (c + 1,))
            return asarray(np.asarray(k, np.int64))
        h = int.from_bytes(("/".join(self._path) + f"#{c}").encode(), "little") % (1 << 62)
        return jrandom.fold_in(self._scope.rngs[name], h)

    @property
    def path(self):
        return self._path

    def __repr__(self):
        return f"{type(self).__name__}(name={self.name!r})"


dataclasses.dataclass(eq=False, repr=False)(Module)


# ------------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------------
class Dense(Module):
    features: int
    use_bias: bool = True
    dtype: Any = None
    param_dtype: Any = None
    kernel_init: Callable = initializers.lecun_normal()
    bias_init: Callable = initializers.zeros

    @compact
    def __call__(self, x):
        x = asarray(x)
        kernel = self.param("kernel", self.kernel_init, (x.shape[-1], self.features))
        y = raw(x) @ raw(kernel)
        if self.use_bias:
            y = y + raw(self.param("bias", self.bias_init, (self.features,)))
        return y.as_subclass(Array)


def _pair(v, n=2):
    if v is None:
        return (1,) * n
    if isinstance(v, int):
        return (v,) * n
    return tuple(int(e) for e in v)


def _same_pads(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2     # XLA SAME: the extra element goes to the high side


class Conv(Module):
    features: int
    kernel_size: Any = (3, 3)
    strides: Any = 1
    padding: Any = "SAME"
    use_bias: bool = True
    dtype: Any = None
    param_dtype: Any = None
    kernel_init: Callable = initializers.lecun_normal()
    bias_init: Callable = initializers.zeros

    @compact
    def __call__(self, x):
        x = raw(x)
        ks, st = _pair(self.kernel_size), _pair(self.strides)
        kernel = raw(self.param("kernel", self.kernel_init, ks + (x.shape[-1], self.features)))   # HWIO
        unbatched = x.dim() == 3
        if unbatched:
            x = x[None]
        H, W = x.shape[1], x.shape[2]
        pad = self.padding
        if isinstance(pad, str):
            pads = [_same_pads(H, ks[0], st[0]), _same_pads(W, ks[1], st[1])] if pad.upper() == "SAME" else [(0, 0), (0, 0)]
        elif isinstance(pad, int):
            pads = [(pad, pad), (pad, pad)]
        else:
            pads = [(p, p) if isinstance(p, int) else tuple(p) for p in pad]
        xn = F.pad(x.permute(0, 3, 1, 2), (pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
        y = F.conv2d(xn, kernel.permute(3, 2, 0, 1), stride=st).permute(0, 2, 3, 1)
        if self.use_bias:
            y = y + raw(self.param("bias", self.bias_init, (self.features,)))
        if unbatched:
            y = y[0]
        return y.as_subclass(Array)


def _fast_stats(x, axes):
    """flax _compute_stats with use_fast_variance=True: var = max(0, E[x^2] - E[x]^2)."""
    mean = x.mean(dim=axes, keepdim=True)
    mean2 = (x * x).mean(dim=axes, keepdim=True)
    return mean, torch.clamp(mean2 - mean * mean, min=0.0)


class LayerNorm(Module):
    epsilon: float = 1e-6
    dtype: Any = None
    param_dtype: Any = None
    use_bias: bool = True
    use_scale: bool = True
    bias_init: Callable = initializers.zeros
    scale_init: Callable = initializers.ones

    @compact
    def __call__(self, x):
        x = raw(x)
        mean, var = _fast_stats(x, (-1,))
        y = (x - mean) * torch.rsqrt(var + self.epsilon)
        if self.use_scale:
            y = y * raw(self.param("scale", self.scale_init, (x.shape[-1],)))
        if self.use_bias:
            y = y + raw(self.param("bias", self.bias_init, (x.shape[-1],)))
        return y.as_subclass(Array)


class GroupNorm(Module):
    num_groups: Optional[int] = 32
    group_size: Optional[int] = None
    epsilon: float = 1e-6
    dtype: Any = None
    param_dtype: Any = None
    use_bias: bool = True
    use_scale: bool = True
    bias_init: Callable = initializers.zeros
    scale_init: Callable = initializers.ones

    @compact
    def __call__(self, x):
        x = raw(x)
        Cc = x.shape[-1]
        groups = self.num_groups if self.num_groups is not None else Cc // self.group_size
        # statistics per sample over every non-batch axis and the channels of a group
        xg = x.reshape(x.shape[:-1] + (groups, Cc // groups))
        axes = tuple(range(1, xg.dim() - 2)) + (xg.dim() - 1,)
        mean, var = _fast_stats(xg, axes)
        y = ((xg - mean) * torch.rsqrt(var + self.epsilon)).reshape(x.shape)
        if self.use_scale:
            y = y * raw(self.param("scale", self.scale_init, (Cc,)))
        if self.use_bias:
            y = y + raw(self.param("bias", self.bias_init, (Cc,)))
        return y.as_subclass(Array)


class Dropout(Module):
    rate: float
    deterministic: Optional[bool] = None
    rng_collection: str = "dropout"

    @compact
    def __call__(self, inputs, deterministic=None):
        det = self.deterministic if deterministic is None else deterministic
        if det is None:
            raise ValueError("Dropout needs `deterministic`")
        if self.rate == 0.0 or det:
            return inputs
        keep = 1.0 - self.rate
        x = raw(inputs)
        with jrandom.context("dropout:" + "/".join(self._path)):
            mask = raw(jrandom.bernoulli(self.make_rng(self.rng_collection), keep, tuple(x.shape)))
        return torch.where(mask, x / keep, torch.zeros_like(x)).as_subclass(Array)


def _pool(x, window, strides, padding, init, fn):
    x = raw(x)
    unbatched = x.dim() == 3
    if unbatched:
        x = x[None]
    st = _pair(strides, len(window))
    if isinstance(padding, str):
        pads = [_same_pads(x.shape[1 + i], window[i], st[i]) if padding.upper() == "SAME" else (0, 0) for i in range(len(window))]
    else:
        pads = [tuple(p) for p in padding]
    xn = F.pad(x.permute(0, 3, 1, 2), (pads[1][0], pads[1][1], pads[0][0], pads[0][1]), value=init)
    y = fn(xn, tuple(window), st).permute(0, 2, 3, 1)
    return (y[0] if unbatched else y).as_subclass(Array)


def max_pool(x, window_shape, strides=None, padding="VALID"):
    return _pool(x, window_shape, strides, padding, float("-inf"), F.max_pool2d)


def avg_pool(x, window_shape, strides=None, padding="VALID"):
    return _pool(x, window_shape, strides, padding, 0.0, F.avg_pool2d)


# ------------------------------------------------------------------------------------------------
# nn.vmap over parameters (actor_critic_nets.py:156-164 `ensemblize`)
# ------------------------------------------------------------------------------------------------
class _Vmapped(Module):
    """`nn.vmap(target, variable_axes={'params': 0}, split_rngs={'params': True}, in_axes=None, out_axes=k,
    axis_size=N)`: the transformed module IS the target (same parameter names) with a leading axis of size N on every
    parameter; inputs are broadcast, outputs stacked along `out_axes`."""
    target: Any = None
    target_args: tuple = ()
    target_kwargs: Any = None
    axis_size: int = 1
    out_axes: int = 0

    @compact
    def __call__(self, *args, **kwargs):
        sc = self._scope
        store = sc.params
        for p in self._path[:-1]:
            store = store.setdefault(p, {}) if sc.initializing else store[p]
        mine = store.get(self._path[-1]) if self._path else store
        member_params, outs = [], []
        for i in range(self.axis_size):
            if sc.initializing and mine is None:
                sub_params = {}
                rngs = dict(sc.rngs)
                rngs["params"] = jrandom.fold_in(self.make_rng("params"), i)      # split_rngs={'params': True}
            else:
                sub_params = jax.tree_map(lambda a: asarray(a)[i], mine)
                rngs = sc.rngs
            sub = _Scope(sub_params, rngs, sc.initializing and mine is None)
            saved = list(_STACK)
            _STACK.clear()
            try:
                inner = self.target(*self.target_args, **(self.target_kwargs or {}))
            finally:
                _STACK.extend(saved)
            inner = inner._bind_root(sub)
            outs.append(inner(*args, **kwargs))
            member_params.append(sub.params)
        if sc.initializing and mine is None:
            stacked = jax.tree_map(lambda *v: torch.stack([raw(x) for x in v]).as_subclass(Array), *member_params)
            if self._path:
                store[self._path[-1]] = stacked
            else:
                store.update(stacked)
        return jax.tree_map(lambda *v: torch.stack([raw(x) for x in v], dim=self.out_axes).as_subclass(Array), *outs)


def vmap(target, variable_axes=None, split_rngs=None, in_axes=0, out_axes=0, axis_size=None, **kw):
    assert variable_axes == {"params": 0} and split_rngs == {"params": True} and in_axes is None, \
        "the stand-in nn.vmap supports the ensemblize() configuration only"

    def make(*args, name=None, parent=None, **kwargs):
        return _Vmapped(target=target, target_args=tuple(args), target_kwargs=dict(kwargs), axis_size=axis_size,
                        out_axes=out_axes, name=name)
    return make
