"""jax.nn subset.  TEST INFRASTRUCTURE ONLY."""
import torch
import torch.nn.functional as F

from .._core import Array, raw
from . import initializers  # noqa: F401


def _w(t):
    return t.as_subclass(Array)


def relu(x):
    return _w(torch.relu(raw(x)))


def tanh(x):
    return _w(torch.tanh(raw(x)))


def sigmoid(x):
    return _w(torch.sigmoid(raw(x)))


def softplus(x):
    t = raw(x)
    return _w(torch.logaddexp(t, torch.zeros_like(t)))   # jax.nn.softplus = logaddexp(x, 0)


def swish(x):
    t = raw(x)
    return _w(t * torch.sigmoid(t))


silu = swish


def gelu(x, approximate=True):
    return _w(F.gelu(raw(x), approximate="tanh" if approximate else "none"))


def softmax(x, axis=-1):
    return _w(torch.softmax(raw(x), dim=axis))


def log_softmax(x, axis=-1):
    return _w(torch.log_softmax(raw(x), dim=axis))


def leaky_relu(x, negative_slope=0.01):
    return _w(F.leaky_relu(raw(x), negative_slope))


def elu(x, alpha=1.0):
    return _w(F.elu(raw(x), alpha))
