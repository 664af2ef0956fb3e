"""Stand-in for the parts of jax the reference's SAC/DrQ update touches, on PyTorch-CPU tensors.
TEST INFRASTRUCTURE ONLY -- see oracle/jaxshim/README.md.  (The reference README pins jax 0.4.35.)"""
from __future__ import annotations

import functools

import torch

from . import _core, lax, nn, numpy, random, tree_util  # noqa: F401
from ._core import Array
from .tree_util import tree_flatten, tree_leaves, tree_map, tree_structure, tree_unflatten  # noqa: F401

__version__ = "0.4.35-standin"


def jit(fun=None, *, static_argnames=None, static_argnums=None, **kw):
    """No tracing: the function simply runs eagerly.  Returns a plain function so that it still binds as a method."""
    if fun is None:
        return functools.partial(jit, static_argnames=static_argnames, static_argnums=static_argnums, **kw)

    @functools.wraps(fun)
    def wrapped(*a, **k):
        return fun(*a, **k)
    return wrapped


def device_put(x, device=None):
    return x


def devices(*a):
    return ["cpu:0"]


def local_devices():
    return ["cpu:0"]


def _value_and_grad(fun, argnums=0, has_aux=False):
    assert argnums == 0

    def run(x, *rest, **kw):
        leaves, treedef = tree_flatten(x)
        req = []
        for l in leaves:
            t = _core.asarray(l).detach().clone()
            if t.is_floating_point():
                t.requires_grad_(True)
            req.append(t)
        out = fun(tree_unflatten(treedef, req), *rest, **kw)
        val, aux = (out if has_aux else (out, None))
        flt = [t for t in req if t.requires_grad]
        if isinstance(val, torch.Tensor) and val.requires_grad:
            gs = torch.autograd.grad(val, flt, allow_unused=True)
        else:   # constant loss (the reference's `lambda params, rng: (0.0, {})`): exact zero gradients
            gs = [None] * len(flt)
        it = iter(gs)
        grads = []
        for t in req:
            g = next(it) if t.requires_grad else None
            grads.append((g if g is not None else torch.zeros_like(t)).detach().as_subclass(Array))
        detach = lambda a: a.detach() if isinstance(a, torch.Tensor) else a  # noqa: E731
        gtree = tree_unflatten(treedef, grads)
        return (tree_map(detach, val), tree_map(detach, aux)), gtree
    return run


def value_and_grad(fun, argnums=0, has_aux=False):
    run = _value_and_grad(fun, argnums, has_aux)

    def f(*a, **k):
        (val, aux), g = run(*a, **k)
        return ((val, aux), g) if has_aux else (val, g)
    return f


def grad(fun, argnums=0, has_aux=False):
    run = _value_and_grad(fun, argnums, has_aux)

    def f(*a, **k):
        (val, aux), g = run(*a, **k)
        return (g, aux) if has_aux else g
    return f


def vmap(fun, in_axes=0, out_axes=0):
    """Python loop over the mapped axis (axis 0 only, or None = broadcast)."""
    def f(*args):
        axes = in_axes if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
        n = None
        for a, ax in zip(args, axes):
            if ax is not None:
                assert ax == 0, "stand-in vmap maps axis 0 only"
                n = int(_core.raw(tree_leaves(a)[0]).shape[0])
        outs = []
        for i in range(n):
            sl = [tree_map(lambda t: _core.asarray(t)[i], a) if ax is not None else a for a, ax in zip(args, axes)]
            outs.append(fun(*sl))
        return tree_map(lambda *v: torch.stack([_core.raw(x) for x in v], dim=out_axes).as_subclass(Array), *outs)
    return f


class _Config:
    def update(self, *a, **k):
        pass


config = _Config()
