"""jax.random stand-in: keys are int64[2] Arrays; every draw is a deterministic function of the key and is recorded on a
tape so the golden-vector script can replay the same noise through the HIP path.  Default streams: keyed Philox (NOT the
numbers JAX would draw).  SERL_JAXSHIM_PRNG=threefry: JAX's own threefry2x32 key chain (threefry.py, pinned against the
Random123 known-answer vectors and the values printed in JAX's documentation) -- split / fold_in / randint / normal /
uniform / bernoulli then return what `jax.random` returns for the same key (normals to ~1 float32 ulp).
TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import hashlib
import os

import numpy as np
import torch

from ._core import Array, asarray, canon_dtype, float_dtype, raw
from . import threefry as _tf


def _threefry_on():
    return os.environ.get("SERL_JAXSHIM_PRNG", "") == "threefry"


def _tf_key(key):
    k0, k1 = _key_ints(key)
    return np.array([k0, k1], np.uint32)

TAPE = None            # list of records when recording is on
_CONTEXT = []          # labels pushed by callers (e.g. the flax Dropout module's path)


def start_tape():
    global TAPE
    TAPE = []
    return TAPE


def stop_tape():
    global TAPE
    t, TAPE = TAPE, None
    return t


class context:
    def __init__(self, label):
        self.label = label

    def __enter__(self):
        _CONTEXT.append(self.label)

    def __exit__(self, *a):
        _CONTEXT.pop()


def _rec(kind, value, **kw):
    if TAPE is not None:
        TAPE.append({"kind": kind, "value": np.array(value), "context": tuple(_CONTEXT), **kw})


def _key_ints(key):
    k = np.asarray(raw(key).detach().cpu().numpy()).reshape(-1)
    assert k.size == 2, f"expected one PRNG key, got shape {tuple(raw(key).shape)}"
    return int(k[0]) & 0xFFFFFFFF, int(k[1]) & 0xFFFFFFFF


def _gen(key, salt=b""):
    k0, k1 = _key_ints(key)
    h = hashlib.blake2b(salt + k0.to_bytes(4, "little") + k1.to_bytes(4, "little"), digest_size=16).digest()
    return np.random.Generator(np.random.Philox(key=int.from_bytes(h, "little") & ((1 << 128) - 1)))


def PRNGKey(seed):
    return asarray(np.array([0, int(seed) & 0xFFFFFFFF], np.int64))


key = PRNGKey


def split(key, num=2):
    if _threefry_on():
        return asarray(_tf.split(_tf_key(key), num).astype(np.int64))
    ks = _gen(key, b"split").integers(0, 1 << 32, size=(int(num), 2), dtype=np.int64)
    return asarray(ks)


def fold_in(key, data):
    if _threefry_on():
        return asarray(_tf.fold_in(_tf_key(key), int(data)).astype(np.int64))
    g = _gen(key, b"fold" + int(data).to_bytes(8, "little", signed=True))
    return asarray(g.integers(0, 1 << 32, size=(2,), dtype=np.int64))


def normal(key, shape=(), dtype=None):
    v = _tf.normal(_tf_key(key), tuple(shape)).astype(np.float64) if _threefry_on() else _gen(key, b"normal").standard_normal(tuple(shape))
    _rec("normal", v)
    return asarray(v, dtype or float_dtype())


def uniform(key, shape=(), dtype=None, minval=0.0, maxval=1.0):
    v = (_tf.uniform(_tf_key(key), tuple(shape), minval, maxval).astype(np.float64) if _threefry_on()
         else _gen(key, b"uniform").random(tuple(shape)) * (maxval - minval) + minval)
    _rec("uniform", v)
    return asarray(v, dtype or float_dtype())


def randint(key, shape, minval, maxval, dtype=None):
    v = (_tf.randint(_tf_key(key), tuple(shape), int(minval), int(maxval)).astype(np.int64) if _threefry_on()
         else _gen(key, b"randint").integers(int(minval), int(maxval), size=tuple(shape), dtype=np.int64))
    _rec("randint", v, minval=int(minval), maxval=int(maxval))
    return asarray(v, canon_dtype(dtype) if dtype is not None else torch.int32)


def bernoulli(key, p=0.5, shape=None):
    shape = tuple(shape) if shape is not None else tuple(np.shape(p))
    v = _tf.bernoulli(_tf_key(key), float(p), shape) if _threefry_on() else _gen(key, b"bernoulli").random(shape) < float(p)
    _rec("bernoulli", v, p=float(p))
    return asarray(v)


def truncated_normal(key, lower, upper, shape=(), dtype=None):
    g = _gen(key, b"truncnormal")
    v = g.standard_normal(tuple(shape))
    bad = (v < lower) | (v > upper)
    while bad.any():
        v[bad] = g.standard_normal(int(bad.sum()))
        bad = (v < lower) | (v > upper)
    return asarray(v, dtype or float_dtype())
