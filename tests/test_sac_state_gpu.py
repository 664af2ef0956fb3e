"""GPU parity of the state-only SAC agent (BASELINE.json configs[0] `async_sac_state_sim`; SACAgent.create_states,
sac.py:486-542: no encoder, flat observations, one Dense(1) Q head per ensemble member, actor/critic optimizers
with warm-up and a temperature optimizer without) against the fp64 CPU oracle.  Tolerance 1e-4 as in
test_agent_gpu.py."""
import numpy as np
import pytest
import torch

from oracle import drq_oracle as O
import agent_helpers as AH
from test_agent_gpu import TOL, _check_grads, _compare_state

pytestmark = pytest.mark.gpu


def _cfg(**kw):
    # launcher.py:50-76: discount 0.99 default, ensemble 10 / subsample 2; optimizer defaults sac.py:333-343
    base = dict(image_keys=(), S=10, A=4, discount=0.99, warmup=4, temp_warmup=0)
    base.update(kw)
    return O.Config(**base)


@pytest.mark.parametrize("B", [16, 40])
def test_state_update_critics_matches_oracle(gpu, B):
    cfg = _cfg()
    st, core = AH.make_pair(cfg, B)
    st.step = 0
    for it in range(3):      # the first step has lr = 0 (linear warm-up from 0): run a few
        b = AH.synth_batch(cfg, B, seed=3 + it)
        noise = O.make_noise(cfg, B, seed=7 + it)
        info, aux = O.update_critics(st, AH.batch_to_torch(b, torch.float64), O.noise_to_torch(noise, torch.float64))
        core.update_critics(AH.batch_to_device(cfg, b), AH.noise_to_device(cfg, noise))
        got = core.read_info()
        for k in ("critic_loss", "predicted_qs", "target_qs"):
            assert abs(got[k] - info[k]) < TOL * max(1.0, abs(info[k])), (it, k, got[k], info[k])
        q = core.debug("q", cfg.ensemble * B).reshape(cfg.ensemble, B)
        assert AH.rel_err(q, aux["q"].numpy()) < TOL
        assert AH.rel_err(core.debug("target_q", B), aux["target_q"].numpy()) < TOL
        _check_grads(cfg, core, aux["grads"], "g_critic", 0)
    _compare_state(cfg, st, core, steps=3)
    assert core.step == st.step == 3


@pytest.mark.parametrize("utd", [1, 2, 8])
def test_state_update_high_utd_matches_oracle(gpu, utd):
    cfg = _cfg()
    B = 16
    st, core = AH.make_pair(cfg, B)
    sl, _ = AH.leaf_slices(cfg)
    for it in range(2):
        b = AH.synth_batch(cfg, B, seed=4 + it)
        noise = O.make_noise(cfg, B, seed=8 + it, utd_ratio=utd)
        info, aux = O.update_high_utd(st, AH.batch_to_torch(b, torch.float64), O.noise_to_torch(noise, torch.float64), utd)
        core.update_high_utd(AH.batch_to_device(cfg, b), utd, AH.noise_to_device(cfg, noise))
        got = core.read_info()
        for k in ("critic_loss", "predicted_qs", "target_qs", "actor_loss", "temperature", "entropy", "temperature_loss"):
            assert abs(got[k] - info[k]) < TOL * max(1.0, abs(info[k])), (it, k, got[k], info[k])
        _check_grads(cfg, core, aux["g_actor"], "g_actor", sl["actor/w1"][0])
    _compare_state(cfg, st, core, tol=5e-4 if utd > 2 else TOL, steps=2 * (utd + 1))
    assert core.step == st.step == 2 * (utd + 1)
    # the temperature optimizer has no warm-up, the others do (sac.py:333-343)
    got = core.read_info()
    last = 2 * (utd + 1) - 1   # optimizer count of the last step
    assert got["temperature_lr"] == pytest.approx(cfg.lr)
    assert got["actor_lr"] == pytest.approx(cfg.lr * min(1.0, last / cfg.warmup))


def test_state_sample_actions(gpu):
    cfg = _cfg()
    st, core = AH.make_pair(cfg, 8)
    state = np.random.default_rng(0).standard_normal((5, cfg.S)).astype(np.float32)
    mean, std = O.policy_head(st.params, cfg, torch.tensor(state, dtype=torch.float64))
    mode = core.sample_actions(None, torch.tensor(state, device="cuda"), None).cpu().numpy()
    assert AH.rel_err(mode, torch.tanh(mean).numpy()) < TOL
    eps = np.random.default_rng(1).standard_normal((5, cfg.A)).astype(np.float32)
    a, _ = O.sample_and_log_prob(mean, std, torch.tensor(eps, dtype=torch.float64))
    got = core.sample_actions(None, torch.tensor(state, device="cuda"), torch.tensor(eps, device="cuda")).cpu().numpy()
    assert AH.rel_err(got, a.numpy()) < TOL
