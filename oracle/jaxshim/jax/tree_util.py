"""jax.tree_util subset: dict (sorted keys), list, tuple, namedtuple, None, registered classes.  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import collections

_REGISTRY = {}   # cls -> (flatten(obj) -> (children, aux), unflatten(aux, children))


def register_pytree_node(cls, flatten, unflatten):
    _REGISTRY[cls] = (flatten, unflatten)


def _is_namedtuple(x):
    return isinstance(x, tuple) and hasattr(x, "_fields")


def _node_kind(x):
    """-> None for leaves, else (kind, keys, children)."""
    t = type(x)
    if t in _REGISTRY:
        children, aux = _REGISTRY[t][0](x)
        return ("reg", (t, aux), list(children))
    if x is None:
        return ("none", None, [])
    if isinstance(x, dict):   # includes FrozenDict / OrderedDict / defaultdict: keys are sorted like jax does
        keys = sorted(x.keys())
        return ("dict", (t, tuple(keys)), [x[k] for k in keys])
    if _is_namedtuple(x):
        return ("namedtuple", t, list(x))
    if isinstance(x, (list, tuple)):
        return ("seq", (t, len(x)), list(x))
    return None


def _rebuild(kind, meta, children):
    if kind == "reg":
        t, aux = meta
        return _REGISTRY[t][1](aux, children)
    if kind == "none":
        return None
    if kind == "dict":
        t, keys = meta
        d = dict(zip(keys, children))
        if t is dict or t is collections.defaultdict or t is collections.OrderedDict:
            return d
        return t(d)
    if kind == "namedtuple":
        return meta(*children)
    t, _ = meta
    return t(children)


class PyTreeDef:
    def __init__(self, spec, num_leaves):
        self.spec, self.num_leaves = spec, num_leaves

    def __eq__(self, o):
        return isinstance(o, PyTreeDef) and self.spec == o.spec

    def __repr__(self):
        return f"PyTreeDef({self.spec})"


def _flatten(x, is_leaf, leaves):
    if is_leaf is not None and is_leaf(x):
        leaves.append(x)
        return "*"
    nk = _node_kind(x)
    if nk is None:
        leaves.append(x)
        return "*"
    kind, meta, children = nk
    return (kind, meta, tuple(_flatten(c, is_leaf, leaves) for c in children))


def tree_flatten(tree, is_leaf=None):
    leaves = []
    spec = _flatten(tree, is_leaf, leaves)
    return leaves, PyTreeDef(spec, len(leaves))


def tree_leaves(tree, is_leaf=None):
    return tree_flatten(tree, is_leaf)[0]


def tree_structure(tree, is_leaf=None):
    return tree_flatten(tree, is_leaf)[1]


def _unflatten(spec, it):
    if spec == "*":
        return next(it)
    kind, meta, children = spec
    return _rebuild(kind, meta, [_unflatten(c, it) for c in children])


def tree_unflatten(treedef, leaves):
    leaves = list(leaves)
    assert len(leaves) == treedef.num_leaves, (len(leaves), treedef.num_leaves)
    return _unflatten(treedef.spec, iter(leaves))


def tree_map(f, tree, *rest, is_leaf=None):
    """`rest` trees only need `tree`'s structure as a PREFIX (jax semantics): where `tree` has a leaf, the
    corresponding sub-trees of `rest` are passed whole."""
    if is_leaf is not None and is_leaf(tree):
        return f(tree, *rest)
    nk = _node_kind(tree)
    if nk is None:
        return f(tree, *rest)
    kind, meta, children = nk
    rest_children = []
    for r in rest:
        rk = _node_kind(r)
        if rk is None or rk[0] != kind or len(rk[2]) != len(children):
            raise ValueError(f"tree_map: structure mismatch at {type(tree).__name__} vs {type(r).__name__}")
        if kind == "dict" and rk[1][1] != meta[1]:
            raise ValueError(f"tree_map: dict keys differ: {meta[1]} vs {rk[1][1]}")
        rest_children.append(rk[2])
    out = [tree_map(f, c, *[rc[i] for rc in rest_children], is_leaf=is_leaf) for i, c in enumerate(children)]
    return _rebuild(kind, meta, out)


tree_multimap = tree_map
