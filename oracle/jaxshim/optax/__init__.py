"""optax stand-in: adam / adamw / clip_by_global_norm / chain / inject_hyperparams / schedules, restated from optax's
published algorithms (serl_launcher/requirements.txt: optax >= 0.1.5).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import functools
import inspect
from typing import Any, Callable, NamedTuple

import numpy as np
import torch

import jax
from jax import numpy as jnp
from jax._core import Array, asarray, raw


class GradientTransformation(NamedTuple):
    init: Callable
    update: Callable


class EmptyState(NamedTuple):
    pass


class ScaleByAdamState(NamedTuple):
    count: Any
    mu: Any
    nu: Any


class ScaleByScheduleState(NamedTuple):
    count: Any


class InjectHyperparamsState(NamedTuple):
    count: Any
    hyperparams: Any
    inner_state: Any


def _zeros_like_tree(params):
    return jax.tree_map(lambda p: jnp.zeros_like(p), params)


def _count():
    return asarray(np.zeros((), np.int32))


def scale_by_adam(b1=0.9, b2=0.999, eps=1e-8, eps_root=0.0):
    def init(params):
        return ScaleByAdamState(_count(), _zeros_like_tree(params), _zeros_like_tree(params))

    def update(updates, state, params=None):
        mu = jax.tree_map(lambda g, m: (1 - b1) * raw(g) + b1 * raw(m), updates, state.mu)
        nu = jax.tree_map(lambda g, v: (1 - b2) * raw(g) ** 2 + b2 * raw(v), updates, state.nu)
        count = raw(state.count) + 1
        t = float(count)
        bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
        out = jax.tree_map(lambda m, v: ((m / bc1) / (torch.sqrt(v / bc2 + eps_root) + eps)).as_subclass(Array), mu, nu)
        w = lambda tr: jax.tree_map(lambda a: a.as_subclass(Array), tr)  # noqa: E731
        return out, ScaleByAdamState(count.as_subclass(Array), w(mu), w(nu))
    return GradientTransformation(init, update)


def scale(step_size):
    def init(params):
        return EmptyState()

    def update(updates, state, params=None):
        return jax.tree_map(lambda g: (raw(g) * step_size).as_subclass(Array), updates), state
    return GradientTransformation(init, update)


def scale_by_schedule(step_size_fn):
    def init(params):
        return ScaleByScheduleState(_count())

    def update(updates, state, params=None):
        s = step_size_fn(int(raw(state.count)))
        return (jax.tree_map(lambda g: (raw(g) * raw(s)).as_subclass(Array), updates),
                ScaleByScheduleState((raw(state.count) + 1).as_subclass(Array)))
    return GradientTransformation(init, update)


def scale_by_learning_rate(learning_rate, flip_sign=True):
    m = -1 if flip_sign else 1
    if callable(learning_rate):
        return scale_by_schedule(lambda c: m * learning_rate(c))
    return scale(m * learning_rate)


def chain(*txs):
    def init(params):
        return tuple(t.init(params) for t in txs)

    def update(updates, state, params=None):
        new = []
        for t, s in zip(txs, state):
            updates, ns = t.update(updates, s, params)
            new.append(ns)
        return updates, tuple(new)
    return GradientTransformation(init, update)


def adam(learning_rate, b1=0.9, b2=0.999, eps=1e-8, eps_root=0.0):
    return chain(scale_by_adam(b1, b2, eps, eps_root), scale_by_learning_rate(learning_rate))


def add_decayed_weights(weight_decay=0.0, mask=None):
    def init(params):
        return EmptyState()

    def update(updates, state, params=None):
        return jax.tree_map(lambda g, p: (raw(g) + weight_decay * raw(p)).as_subclass(Array), updates, params), state
    return GradientTransformation(init, update)


def adamw(learning_rate, b1=0.9, b2=0.999, eps=1e-8, eps_root=0.0, weight_decay=1e-4, mask=None):
    return chain(scale_by_adam(b1, b2, eps, eps_root), add_decayed_weights(weight_decay, mask),
                 scale_by_learning_rate(learning_rate))


def global_norm(updates):
    return torch.sqrt(sum((raw(g) ** 2).sum() for g in jax.tree_leaves(updates))).as_subclass(Array)


def clip_by_global_norm(max_norm):
    def init(params):
        return EmptyState()

    def update(updates, state, params=None):
        g_norm = raw(global_norm(updates))
        trigger = g_norm < max_norm
        return jax.tree_map(lambda t: (raw(t) if trigger else raw(t) / g_norm * max_norm).as_subclass(Array), updates), state
    return GradientTransformation(init, update)


def apply_updates(params, updates):
    return jax.tree_map(lambda p, u: (raw(p) + raw(u)).as_subclass(Array), params, updates)


# ---- schedules ---------------------------------------------------------------------------------------
def constant_schedule(value):
    return lambda count: value


def polynomial_schedule(init_value, end_value, power, transition_steps, transition_begin=0):
    def schedule(count):
        if transition_steps <= 0:
            return init_value
        c = min(max(count - transition_begin, 0), transition_steps)
        frac = 1 - c / transition_steps
        return (init_value - end_value) * (frac ** power) + end_value
    return schedule


def linear_schedule(init_value, end_value, transition_steps, transition_begin=0):
    return polynomial_schedule(init_value, end_value, 1, transition_steps, transition_begin)


def cosine_decay_schedule(init_value, decay_steps, alpha=0.0):
    import math

    def schedule(count):
        c = min(count, decay_steps)
        cosine = 0.5 * (1 + math.cos(math.pi * c / decay_steps))
        return init_value * ((1 - alpha) * cosine + alpha)
    return schedule


def join_schedules(schedules, boundaries):
    def schedule(step):
        out = schedules[0](step)
        for b, s in zip(boundaries, schedules[1:]):
            if step >= b:
                out = s(step - b)
        return out
    return schedule


def warmup_cosine_decay_schedule(init_value, peak_value, warmup_steps, decay_steps, end_value=0.0):
    return join_schedules([linear_schedule(init_value, peak_value, warmup_steps),
                           cosine_decay_schedule(peak_value, decay_steps - warmup_steps,
                                                 alpha=end_value / peak_value if peak_value else 0.0)], [warmup_steps])


# ---- inject_hyperparams ------------------------------------------------------------------------------
def inject_hyperparams(inner_factory, static_args=(), hyperparam_dtype=None):
    """optax.inject_hyperparams: numeric hyper-parameters (and schedules, evaluated at the state's count BEFORE it is
    incremented) are kept in the state; non-numeric ones (e.g. weight_decay=None) are passed through untouched."""
    sig = inspect.signature(inner_factory)

    @functools.wraps(inner_factory)
    def wrapped(*args, **kwargs):
        bound = sig.bind(*args, **kwargs)
        bound.apply_defaults()
        sched = {k: v for k, v in bound.arguments.items() if callable(v) and k not in static_args}
        numeric = {k: v for k, v in bound.arguments.items()
                   if isinstance(v, (int, float, np.ndarray, torch.Tensor)) and not isinstance(v, bool) and k not in static_args}
        other = {k: v for k, v in bound.arguments.items() if k not in sched and k not in numeric}

        def hp_at(count):
            hp = {k: asarray(np.float32(v)) for k, v in numeric.items()}
            hp.update({k: asarray(np.float32(f(int(raw(count))))) for k, f in sched.items()})
            return hp

        def init(params):
            count = _count()
            hp = hp_at(count)
            return InjectHyperparamsState(count, hp, inner_factory(**other, **{k: float(raw(v)) for k, v in hp.items()}).init(params))

        def update(updates, state, params=None):
            hp = hp_at(state.count)
            inner = inner_factory(**other, **{k: float(raw(v)) for k, v in hp.items()})
            updates, inner_state = inner.update(updates, state.inner_state, params)
            return updates, InjectHyperparamsState((raw(state.count) + 1).as_subclass(Array), hp, inner_state)
        return GradientTransformation(init, update)
    return wrapped
