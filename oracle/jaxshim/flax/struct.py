"""flax.struct stand-in: frozen dataclasses registered as pytrees.  TEST INFRASTRUCTURE ONLY."""
import dataclasses

from jax import tree_util


def field(pytree_node=True, **kw):
    md = dict(kw.pop("metadata", {}) or {})
    md["pytree_node"] = pytree_node
    return dataclasses.field(metadata=md, **kw)


def dataclass(cls):
    if getattr(cls, "_flax_struct", None) is cls:
        return cls
    dc = dataclasses.dataclass(frozen=True)(cls)
    data = [f.name for f in dataclasses.fields(dc) if f.metadata.get("pytree_node", True)]
    meta = [f.name for f in dataclasses.fields(dc) if not f.metadata.get("pytree_node", True)]

    def replace(self, **upd):
        return dataclasses.replace(self, **upd)

    dc.replace = replace
    dc._flax_struct = dc

    def flatten(x):
        return [getattr(x, n) for n in data], tuple(getattr(x, n) for n in meta)

    def unflatten(aux, children):
        return dc(**dict(zip(data, children)), **dict(zip(meta, aux)))

    tree_util.register_pytree_node(dc, flatten, unflatten)
    return dc


class PyTreeNode:
    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        dataclass(cls)

    def replace(self, **upd):   # overwritten by dataclass(); here for type checkers
        raise NotImplementedError
