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
    _compare_state(cfg, st, core, tol=TOL, steps=2 * (utd + 1))
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


# ---- reference-named Python surface: make_sac_agent / ReplayBufferDataStore (launcher.py:50-76,236-243) -----------
import itertools
import os


class _Box:
    def __init__(self, shape):
        self.shape = shape


class _Env:
    def __init__(self, S, A):
        self.observation_space, self.action_space = _Box((S,)), _Box((A,))


@pytest.mark.parametrize("name", ["plain_wrap", "plain_nowrap"])
def test_plain_store_matches_reference_golden(gpu, name):
    """ReplayBufferDataStore (HBM) vs fixtures generated from the reference's own ReplayBuffer: index stream bit-exact,
    gathered rows byte-identical (tests/golden/make_golden_replay.py)."""
    from serl_amd.utils.launcher import make_replay_buffer
    from serl_amd.utils.synthetic import flat_stream
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"replay_{name}.npz"))
    S, A, cap, n_ins, ep, sseed, rseed, B, ns = [int(x) for x in g["meta"]]
    rb = make_replay_buffer(_Env(S, A), capacity=cap, type="replay_buffer")
    rb.seed(rseed)
    for tr in itertools.islice(flat_stream(S, A, ep, sseed), n_ins):
        rb.insert(tr)
    assert len(rb) == int(g["size"]) and rb.latest_data_id() == int(g["insert_index"])
    for s in range(ns):
        b = rb.sample(B)
        for f in ("observations", "next_observations", "actions", "rewards", "masks", "dones"):
            assert np.array_equal(b[f].cpu().numpy(), g[f"{f}_{s}"]), (s, f)
    rb.seed(rseed)
    assert np.array_equal(rb.sample_indices(B), g["idx_0"])


def test_sac_learner_loop(gpu, tmp_path):
    """examples/async_sac_state_sim/async_sac_state_sim.py:280-300: iterator -> update_high_utd(utd_ratio=8)."""
    from serl_amd.utils.checkpoint import restore_checkpoint, save_checkpoint
    from serl_amd.utils.launcher import make_replay_buffer, make_sac_agent
    from serl_amd.utils.synthetic import flat_stream
    S, A, B, utd = 10, 4, 64, 8
    env = _Env(S, A)
    rb = make_replay_buffer(env, capacity=500, type="replay_buffer")
    rb.seed(0)
    for tr in itertools.islice(flat_stream(S, A, 20, 3), 300):
        rb.insert(tr)
    agent = make_sac_agent(7, np.zeros((S,), np.float32), np.zeros((A,), np.float32), batch_size=B)
    twin = make_sac_agent(7, np.zeros((S,), np.float32), np.zeros((A,), np.float32), batch_size=B)
    assert agent.config["discount"] == 0.99 and agent.config["critic_ensemble_size"] == 10
    p = agent.state.params
    assert p["modules_critic"]["network"]["Dense_0"]["kernel"].shape == (10, S + A, 256)
    assert p["modules_critic"]["Dense_0"]["kernel"].shape == (10, 256, 1)
    assert p["modules_actor"]["network"]["Dense_0"]["kernel"].shape == (S, 256)
    assert p["modules_actor"]["Dense_1"]["bias"].shape == (A,)
    it = rb.get_iterator(sample_args={"batch_size": B, "lazy": True})
    w0 = agent.core.get("params", "critic/w1").copy()
    noise_free = None
    for step in range(4):
        lazy = next(it)
        eager = rb.gather(lazy.parts[0][1])                         # the same rows as a reference-format dict
        agent, info = agent.update_high_utd(lazy, utd_ratio=utd)
        twin, _ = twin.update_high_utd(eager, utd_ratio=utd)
        d = info.resolve()
        assert set(d) >= {"critic", "actor", "temperature", "actor_lr", "critic_lr", "temperature_lr"}
        assert all(np.isfinite(v) for v in d["critic"].values()) and all(np.isfinite(v) for v in d["actor"].values())
    assert agent.state.step == 4 * (utd + 1)
    assert not np.array_equal(w0, agent.core.get("params", "critic/w1"))
    for leaf in ("critic/w1", "critic/head/kernel", "actor/w2", "temp/lagrange"):   # lazy and dict batches: same bytes in
        assert np.array_equal(agent.core.get("params", leaf), twin.core.get("params", leaf)), leaf
    # warm-up 2000 for actor/critic, none for the temperature (sac.py:333-343)
    assert d["actor_lr"] == pytest.approx(3e-4 * (agent.state.step - 1) / 2000) and d["temperature_lr"] == pytest.approx(3e-4)
    a = agent.sample_actions(np.zeros((S,), np.float32), argmax=True)
    assert a.shape == (A,) and np.all(np.abs(a) <= 1)
    a2 = agent.sample_actions(np.zeros((3, S), np.float32), seed=np.array([0, 5], np.uint32))
    assert a2.shape == (3, A)
    with pytest.raises(AssertionError, match="divisible by UTD"):
        agent.update_high_utd(next(it), utd_ratio=7)
    # checkpoint round trip
    save_checkpoint(str(tmp_path), agent, step=agent.state.step)
    fresh = make_sac_agent(1, np.zeros((S,), np.float32), np.zeros((A,), np.float32), batch_size=B)
    restore_checkpoint(str(tmp_path), fresh)
    assert fresh.state.step == agent.state.step
    for sec, leaf in (("params", "critic/head/bias"), ("target_params", "critic/w2"), ("opt/critic/mu", "critic/w1"),
                      ("opt/actor/nu", "actor/w1"), ("opt/temperature/mu", "temp/lagrange")):
        assert np.array_equal(fresh.core.get(sec, leaf), agent.core.get(sec, leaf)), (sec, leaf)
