"""chex stand-in: the shape assertions the reference calls (real checks).  TEST INFRASTRUCTURE ONLY."""
import jax
from jax._core import raw


def _shape(x):
    return tuple(raw(x).shape)


def assert_shape(x, expected):
    got = _shape(x)
    exp = tuple(expected)
    assert len(got) == len(exp) and all(e is None or e == g for g, e in zip(got, exp)), f"shape {got} != {exp}"


def assert_equal_shape(xs):
    shapes = [_shape(x) for x in xs]
    assert all(s == shapes[0] for s in shapes), f"shapes differ: {shapes}"


def assert_tree_shape_prefix(tree, prefix):
    prefix = tuple(prefix)
    for leaf in jax.tree_leaves(tree):
        s = _shape(leaf)
        assert s[:len(prefix)] == prefix, f"leaf shape {s} does not start with {prefix}"
