"""jax.lax subset.  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import torch

from ._core import Array, asarray, raw
from . import tree_util


def stop_gradient(x):
    return tree_util.tree_map(lambda t: t.detach() if isinstance(t, torch.Tensor) else t, x)


def dynamic_slice(operand, start_indices, slice_sizes):
    t = raw(operand)
    starts = [int(s) for s in (raw(start_indices).tolist() if not isinstance(start_indices, (list, tuple)) else
                               [raw(s).item() for s in start_indices])]
    sl = []
    for d, (s, n) in enumerate(zip(starts, slice_sizes)):
        s = max(0, min(int(s), t.shape[d] - int(n)))   # XLA clamps the start so the slice stays in bounds
        sl.append(slice(s, s + int(n)))
    return t[tuple(sl)].as_subclass(Array)


def scan(f, init, xs, length=None):
    leaves = tree_util.tree_leaves(xs)
    n = int(length) if length is not None else int(raw(leaves[0]).shape[0])
    carry, ys = init, []
    for i in range(n):
        carry, y = f(carry, tree_util.tree_map(lambda a: asarray(a)[i], xs))
        ys.append(y)

    def stack(*vals):
        return torch.stack([raw(v) for v in vals]).as_subclass(Array)

    out = tree_util.tree_map(stack, *ys) if ys else None
    return carry, out


def pmean(x, axis_name):
    raise NotImplementedError("pmap collectives are not available in the stand-in (the reference never enables them)")
