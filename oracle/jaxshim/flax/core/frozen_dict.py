"""flax.core.frozen_dict stand-in: an immutable mapping that is a pytree node.  TEST INFRASTRUCTURE ONLY."""


class FrozenDict(dict):
    """Immutable dict (item assignment raises); `copy(add_or_replace=...)` like flax's."""

    def __class_getitem__(cls, item):
        return cls

    def _ro(self, *a, **k):
        raise TypeError("FrozenDict is immutable")

    __setitem__ = __delitem__ = _ro
    pop = popitem = clear = update = setdefault = _ro

    def copy(self, add_or_replace=None):
        d = dict(self)
        if add_or_replace:
            d.update(add_or_replace)
        return FrozenDict(d)

    def unfreeze(self):
        return unfreeze(self)

    def __hash__(self):
        return id(self)

    def __reduce__(self):
        return (FrozenDict, (dict(self),))


def freeze(d):
    return FrozenDict({k: (freeze(v) if isinstance(v, dict) else v) for k, v in d.items()})


def unfreeze(d):
    if isinstance(d, dict):
        return {k: unfreeze(v) for k, v in d.items()}
    return d
