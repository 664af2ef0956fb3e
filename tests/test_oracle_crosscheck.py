"""CPU: the update oracle (oracle/drq_oracle.py) restates flax / optax / distrax arithmetic by hand because none of
those packages is installable here (the reference's own code pins it through tests/test_reference_update.py, the
library primitives underneath are restatements).  These tests cross-check
every restated library primitive against PyTorch's own, independently written implementation of the same
published algorithm -- not a substitute for reference outputs, but it rules out transcription slips in the
formulas the HIP kernels are then held to."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import drq_oracle as O

torch.manual_seed(0)
D = torch.float64


def test_group_norm_matches_torch():
    x = torch.randn(3, 8, 6, 64, dtype=D) * 3 + 1
    g, b = torch.randn(64, dtype=D), torch.randn(64, dtype=D)
    ref = F.group_norm(x.permute(0, 3, 1, 2), 4, g, b, eps=1e-5).permute(0, 2, 3, 1)
    assert torch.allclose(O.group_norm(x, g, b), ref, rtol=1e-10, atol=1e-10)


def test_layer_norm_matches_torch():
    x = torch.randn(5, 10, 256, dtype=D) * 2 - 0.5
    g, b = torch.randn(256, dtype=D), torch.randn(256, dtype=D)
    assert torch.allclose(O.layer_norm(x, g, b), F.layer_norm(x, (256,), g, b, eps=1e-6), rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("n,k,s,expect", [(128, 3, 2, (0, 1)), (128, 3, 1, (1, 1)), (128, 1, 2, (0, 0)), (63, 3, 2, (1, 1)),
                                           (64, 7, 2, (2, 3))])
def test_xla_same_padding(n, k, s, expect):
    """XLA SAME: total = max((ceil(n/s)-1)*s + k - n, 0), lo = total // 2."""
    assert O.same_pad(n, k, s) == expect


def test_conv_same_matches_explicit_padding():
    x = torch.randn(2, 9, 12, 5, dtype=D)
    w = torch.randn(3, 3, 5, 7, dtype=D)
    got = O.conv_same(x, w, 2)
    # independent: zero-pad by hand to XLA's (lo, hi) and correlate with unfold
    (pt, pb), (pl, pr) = O.same_pad(9, 3, 2), O.same_pad(12, 3, 2)
    xp = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    cols = F.unfold(xp, (3, 3), stride=2)                                   # [N, Cin*9, L]
    wm = w.permute(3, 2, 0, 1).reshape(7, -1)                              # [Cout, Cin*9]
    ref = (wm @ cols).reshape(2, 7, got.shape[1], got.shape[2]).permute(0, 2, 3, 1)
    assert got.shape == (2, 5, 6, 7) and torch.allclose(got, ref, rtol=1e-10, atol=1e-10)


def test_tanh_gaussian_log_prob_matches_torch_distributions():
    """distrax MultivariateNormalDiag + Block(Tanh) restated (actor_critic_nets.py:230-272)."""
    from torch.distributions import Independent, Normal, TransformedDistribution
    from torch.distributions.transforms import TanhTransform
    mean, std = torch.randn(6, 4, dtype=D), torch.rand(6, 4, dtype=D) + 0.2
    eps = torch.randn(6, 4, dtype=D) * 0.8
    a, logp = O.sample_and_log_prob(mean, std, eps)
    dist = TransformedDistribution(Independent(Normal(mean, std), 1), [TanhTransform(cache_size=1)])
    assert torch.allclose(a, torch.tanh(mean + std * eps))
    assert torch.allclose(logp, dist.log_prob(a), rtol=1e-8, atol=1e-8)


def test_adam_with_schedule_matches_torch_optim():
    """optax.adam(b1 .9, b2 .999, eps 1e-8) under inject_hyperparams + linear warm-up (optimizers.py:23-46), including
    the zero-gradient steps a tx takes while it is not in networks_to_update (sac.py:276-277)."""
    cfg = O.Config(image_keys=(), S=3, A=2, warmup=5, temp_warmup=0, lr=3e-4)
    _, theta = O.init_params(cfg, 1)
    st = O.TrainState(cfg, {}, theta)
    name = "actor/w2"
    p = torch.nn.Parameter(st.params[name].clone())
    opt = torch.optim.Adam([p], lr=cfg.lr, betas=(0.9, 0.999), eps=1e-8)
    # (the scheduled rate is a float32 value, as optax.inject_hyperparams stores it)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda t: float(np.float32(cfg.lr * min(1.0, t / cfg.warmup))) / cfg.lr)
    rng = np.random.default_rng(0)
    ref_other = {k: v.clone() for k, v in st.params.items()}
    for step in range(9):
        g = torch.tensor(rng.standard_normal(p.shape)) * (0.0 if step % 3 == 2 else 1.0)   # every third step: g = 0
        O.apply_gradients(st, {"actor": {name: g}})
        p.grad = g.clone()
        opt.step()
        sched.step()
        assert torch.allclose(st.params[name], p.detach(), rtol=1e-9, atol=1e-12), step
    # leaves without gradients did not move (their moments stay exactly zero)
    assert torch.equal(st.params["critic/w1"], ref_other["critic/w1"])
    assert st.step == 9 and st.opt["temperature"]["count"] == 9


def test_temperature_multiplier():
    """GeqLagrangeMultiplier(init 1e-2): lambda0 = softplus^-1(init), alpha = softplus(lambda) (lagrange.py:28-50)."""
    cfg = O.Config(image_keys=(), S=3, A=2)
    _, theta = O.init_params(cfg, 1)
    lam = torch.tensor(float(theta["temp/lagrange"]), dtype=D)
    assert abs(F.softplus(lam).item() - cfg.temperature_init) < 1e-8
    assert abs(lam.item() - math.log(math.expm1(cfg.temperature_init))) < 1e-6
