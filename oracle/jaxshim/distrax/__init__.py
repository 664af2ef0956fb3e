"""distrax stand-in: the distribution / bijector classes the reference's policy uses, restated from distrax's published
definitions (serl_launcher/requirements.txt: distrax >= 0.1.2).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import math

import torch

import jax
from jax import random as jrandom
from jax._core import Array, asarray, raw


def _w(t):
    return t.as_subclass(Array)


class Distribution:
    def sample(self, *, seed, sample_shape=()):
        return self._sample_n(seed, tuple(sample_shape) if not isinstance(sample_shape, int) else (sample_shape,))

    def sample_and_log_prob(self, *, seed, sample_shape=()):
        x = self.sample(seed=seed, sample_shape=sample_shape)
        return x, self.log_prob(x)


class MultivariateNormalDiag(Distribution):
    def __init__(self, loc=None, scale_diag=None):
        self._loc = raw(loc)
        self._scale = raw(scale_diag) if scale_diag is not None else torch.ones_like(self._loc)
        self._scale = self._scale.broadcast_to(torch.broadcast_shapes(self._loc.shape, self._scale.shape))
        self._loc = self._loc.broadcast_to(self._scale.shape)

    @property
    def loc(self):
        return _w(self._loc)

    @property
    def scale_diag(self):
        return _w(self._scale)

    @property
    def event_shape(self):
        return tuple(self._loc.shape[-1:])

    @property
    def batch_shape(self):
        return tuple(self._loc.shape[:-1])

    def _sample_n(self, key, shape):
        eps = raw(jrandom.normal(key, shape + tuple(self._loc.shape)))
        return _w(self._loc + self._scale * eps)

    def log_prob(self, value):
        z = (raw(value) - self._loc) / self._scale
        return _w((-0.5 * z * z - torch.log(self._scale) - 0.5 * math.log(2.0 * math.pi)).sum(-1))

    def mode(self):
        return _w(self._loc)

    mean = mode

    def stddev(self):
        return _w(self._scale)

    def entropy(self):
        return _w((0.5 + 0.5 * math.log(2.0 * math.pi) + torch.log(self._scale)).sum(-1))


class Bijector:
    def forward(self, x):
        return self.forward_and_log_det(x)[0]

    def forward_log_det_jacobian(self, x):
        return self.forward_and_log_det(x)[1]

    def inverse(self, y):
        return self.inverse_and_log_det(y)[0]


class Tanh(Bijector):
    event_ndims_in = 0

    def forward_and_log_det(self, x):
        t = raw(x)
        # distrax.Tanh: log|d tanh/dx| = 2 (log 2 - x - softplus(-2x))
        return _w(torch.tanh(t)), _w(2.0 * (math.log(2.0) - t - torch.nn.functional.softplus(-2.0 * t)))

    def inverse_and_log_det(self, y):
        t = raw(y)
        x = torch.atanh(t)
        return _w(x), _w(-2.0 * (math.log(2.0) - x - torch.nn.functional.softplus(-2.0 * x)))


class Block(Bijector):
    def __init__(self, bijector, ndims):
        self.bijector, self.ndims = bijector, ndims

    def _sum(self, ld):
        t = raw(ld)
        return _w(t.sum(dim=tuple(range(-self.ndims, 0)))) if self.ndims else ld

    def forward_and_log_det(self, x):
        y, ld = self.bijector.forward_and_log_det(x)
        return y, self._sum(ld)

    def inverse_and_log_det(self, y):
        x, ld = self.bijector.inverse_and_log_det(y)
        return x, self._sum(ld)


class Lambda(Bijector):
    def __init__(self, forward, inverse=None, forward_log_det_jacobian=None, inverse_log_det_jacobian=None,
                 event_ndims_in=None, event_ndims_out=None, is_constant_jacobian=None):
        self._f, self._i, self._fl, self._il = forward, inverse, forward_log_det_jacobian, inverse_log_det_jacobian

    def forward_and_log_det(self, x):
        return self._f(x), self._fl(x)

    def inverse_and_log_det(self, y):
        return self._i(y), self._il(y)


class Chain(Bijector):
    """Composition: the LAST bijector of the list is applied first (distrax.Chain)."""

    def __init__(self, bijectors):
        self.bijectors = list(bijectors)

    def forward_and_log_det(self, x):
        total = None
        for b in reversed(self.bijectors):
            x, ld = b.forward_and_log_det(x)
            total = ld if total is None else _w(raw(total) + raw(ld))
        return x, total

    def inverse_and_log_det(self, y):
        total = None
        for b in self.bijectors:
            y, ld = b.inverse_and_log_det(y)
            total = ld if total is None else _w(raw(total) + raw(ld))
        return y, total


class Transformed(Distribution):
    def __init__(self, distribution, bijector):
        self._distribution, self._bijector = distribution, bijector

    @property
    def distribution(self):
        return self._distribution

    @property
    def bijector(self):
        return self._bijector

    def _sample_n(self, key, shape):
        return self._bijector.forward(self._distribution._sample_n(key, shape))

    def sample_and_log_prob(self, *, seed, sample_shape=()):
        # distrax.Transformed: y = f(x), log p(y) = log p_base(x) - log|det J_f(x)|
        x, lp = self._distribution.sample_and_log_prob(seed=seed, sample_shape=sample_shape)
        y, fldj = self._bijector.forward_and_log_det(x)
        return y, _w(raw(lp) - raw(fldj))

    def log_prob(self, value):
        x, ildj = self._bijector.inverse_and_log_det(value)
        return _w(raw(self._distribution.log_prob(x)) + raw(ildj))

    def mode(self):
        return self._bijector.forward(self._distribution.mode())
