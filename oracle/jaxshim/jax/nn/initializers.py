"""jax.nn.initializers (variance_scaling family) restated; draws come from the stand-in jax.random.
TEST INFRASTRUCTURE ONLY."""
import math

import numpy as np

from .. import random as jrandom
from .._core import asarray, float_dtype


def _fans(shape, in_axis=-2, out_axis=-1):
    if len(shape) < 2:
        raise ValueError("variance_scaling needs at least 2 dims")
    receptive = int(np.prod(shape)) // (shape[in_axis] * shape[out_axis])
    return shape[in_axis] * receptive, shape[out_axis] * receptive


def variance_scaling(scale, mode, distribution, in_axis=-2, out_axis=-1):
    def init(key, shape, dtype=None):
        shape = tuple(int(s) for s in shape)
        fan_in, fan_out = _fans(shape, in_axis, out_axis)
        denom = {"fan_in": fan_in, "fan_out": fan_out, "fan_avg": (fan_in + fan_out) / 2}[mode]
        var = scale / denom
        if distribution == "truncated_normal":
            std = math.sqrt(var) / 0.87962566103423978   # stddev of a unit normal truncated to [-2, 2]
            return jrandom.truncated_normal(key, -2.0, 2.0, shape) * std
        if distribution == "normal":
            return jrandom.normal(key, shape) * math.sqrt(var)
        if distribution == "uniform":
            lim = math.sqrt(3.0 * var)
            return jrandom.uniform(key, shape, minval=-lim, maxval=lim)
        raise ValueError(distribution)
    return init


def lecun_normal(in_axis=-2, out_axis=-1):
    return variance_scaling(1.0, "fan_in", "truncated_normal", in_axis, out_axis)


def kaiming_normal(in_axis=-2, out_axis=-1):
    return variance_scaling(2.0, "fan_in", "normal", in_axis, out_axis)


he_normal = kaiming_normal


def xavier_uniform(in_axis=-2, out_axis=-1):
    return variance_scaling(1.0, "fan_avg", "uniform", in_axis, out_axis)


glorot_uniform = xavier_uniform


def xavier_normal(in_axis=-2, out_axis=-1):
    return variance_scaling(1.0, "fan_avg", "truncated_normal", in_axis, out_axis)


def zeros(key, shape, dtype=None):
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    return asarray(np.zeros(shape), float_dtype())


def ones(key, shape, dtype=None):
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    return asarray(np.ones(shape), float_dtype())


def constant(value):
    def init(key, shape, dtype=None):
        return asarray(np.full(tuple(shape), value), float_dtype())
    return init


def uniform(scale=1e-2):
    def init(key, shape, dtype=None):
        return jrandom.uniform(key, tuple(shape)) * scale
    return init


def normal(stddev=1e-2):
    def init(key, shape, dtype=None):
        return jrandom.normal(key, tuple(shape)) * stddev
    return init
