"""GPU parity of the trainable SmallEncoder path (reference vision/small_encoders.py:9-55, selected by
DrQAgent.create_drq(encoder_type="small"), drq.py:137-153): conv stack forward, and backward of the critic loss into
every conv kernel and bias, against the fp64 oracle (which tests/test_reference_update.py pins to the reference's own
SmallEncoder).  Tolerance 1e-4 as everywhere."""
import numpy as np
import pytest
import torch

from oracle import drq_oracle as O
import agent_helpers as AH
from test_agent_gpu import TOL, _check_grads, _compare_state

pytestmark = pytest.mark.gpu


def _cfg(H=64, W=64, keys=("front", "wrist"), S=5, A=3):
    return O.Config(image_keys=keys, H=H, W=W, S=S, A=A, encoder_type="small")


@pytest.mark.parametrize("H,W,B,keys", [(64, 64, 16, ("front", "wrist")), (128, 128, 8, ("front", "wrist")), (96, 64, 5, ("image",))])
def test_small_encoder_update_critics(gpu, H, W, B, keys):
    cfg = _cfg(H, W, keys)
    st, core = AH.make_pair(cfg, B)
    b = AH.synth_batch(cfg, B, seed=3)
    noise = O.make_noise(cfg, B, seed=7)
    info, aux = O.update_critics(st, AH.batch_to_torch(b, torch.float64), O.noise_to_torch(noise, torch.float64))
    core.update_critics(AH.batch_to_device(cfg, b), AH.noise_to_device(cfg, noise))
    got = core.read_info()
    for k in ("critic_loss", "predicted_qs", "target_qs"):
        assert abs(got[k] - info[k]) < TOL * max(1.0, abs(info[k])), (k, got[k], info[k])
    q = core.debug("q", cfg.ensemble * B).reshape(cfg.ensemble, B)
    assert AH.rel_err(q, aux["q"].numpy()) < TOL
    assert AH.rel_err(core.debug("target_q", B), aux["target_q"].numpy()) < TOL
    x = core.debug("x", B * (cfg.enc_dim + cfg.A)).reshape(B, -1)
    assert AH.rel_err(x[:, :cfg.enc_dim], aux["enc_obs"].numpy()) < TOL       # conv stack -> pool -> Dense -> LN -> tanh
    assert any("conv0/kernel" in k for k in aux["grads"]) and any("conv3/bias" in k for k in aux["grads"])
    _check_grads(cfg, core, aux["grads"], "g_critic", 0)
    _compare_state(cfg, st, core)
    assert core.step == st.step == 1


def test_small_encoder_sequence(gpu):
    """critic steps, a UTD-2 scan, an all-networks update: the target encoder (EMA of trained convs) matters here"""
    cfg = _cfg()
    B = 8
    st, core = AH.make_pair(cfg, B)
    sl, _ = AH.leaf_slices(cfg)
    for it, kind in enumerate(("critics", "high_utd", "update", "critics", "high_utd")):
        b = AH.synth_batch(cfg, B, seed=40 + it)
        utd = 2 if kind == "high_utd" else 1
        noise = O.make_noise(cfg, B, seed=50 + it, utd_ratio=utd)
        tb, tn = AH.batch_to_torch(b, torch.float64), O.noise_to_torch(noise, torch.float64)
        db, dn = AH.batch_to_device(cfg, b), AH.noise_to_device(cfg, noise)
        if kind == "critics":
            info, _ = O.update_critics(st, tb, tn)
            core.update_critics(db, dn)
        elif kind == "high_utd":
            info, aux = O.update_high_utd(st, tb, tn, utd)
            core.update_high_utd(db, utd, dn)
            _check_grads(cfg, core, aux["g_actor"], "g_actor", sl["enc/proprio/dense/kernel"][0])
        else:
            info = O.update(st, tb, tn)
            core.update(db, ("actor", "critic", "temperature"), dn)
        got = core.read_info()
        for k, v in info.items():
            assert abs(got[k] - v) < 2 * TOL * max(1.0, abs(v)), (it, kind, k, got[k], v)
    _compare_state(cfg, st, core, steps=9)
    assert core.step == st.step == 9    # 1 + (2 + 1) + 1 + 1 + (2 + 1) optimizer steps


def test_small_encoder_python_surface(gpu):
    """make_drq_agent(encoder_type="small"): flax-layout export (Conv_0..3 under every camera), sample_actions."""
    from serl_amd.utils.launcher import make_drq_agent
    keys, H, W, S, A = ("front", "wrist"), 64, 64, 7, 4
    obs = {"front": np.zeros((1, H, W, 3), np.uint8), "wrist": np.zeros((1, H, W, 3), np.uint8), "state": np.zeros((1, S), np.float32)}
    agent = make_drq_agent(3, obs, np.zeros((A,), np.float32), image_keys=keys, encoder_type="small", batch_size=8)
    p = agent.state.params
    enc = p["modules_actor"]["encoder"]
    for k in keys:
        assert enc[f"encoder_{k}"]["Conv_0"]["kernel"].shape == (3, 3, 3, 32) and enc[f"encoder_{k}"]["Conv_3"]["bias"].shape == (256,)
        assert enc[f"encoder_{k}"]["Dense_0"]["kernel"].shape == (256, 256) and "pretrained_encoder" not in enc[f"encoder_{k}"]
    rng = np.random.default_rng(0)
    o = {"front": rng.integers(0, 256, (1, H, W, 3), dtype=np.uint8), "wrist": rng.integers(0, 256, (1, H, W, 3), dtype=np.uint8),
         "state": rng.standard_normal((1, S)).astype(np.float32)}
    a = agent.sample_actions(o, argmax=True)
    assert a.shape == (A,) and np.all(np.abs(a) <= 1) and np.isfinite(a).all()
    # one learner iteration through the reference-named API moves the conv kernels
    import itertools
    from helpers import make_spaces
    from serl_amd.utils.launcher import make_replay_buffer
    from serl_amd.utils.synthetic import transition_stream

    class _Env:
        observation_space, action_space = make_spaces(keys, H, W, 3, 1, S, A)
    rb = make_replay_buffer(_Env(), capacity=200, type="memory_efficient_replay_buffer", image_keys=keys)
    rb.seed(0)
    for tr in itertools.islice(transition_stream(keys, H, W, 3, 1, S, A, 20, 5), 100):
        rb.insert(tr)
    w0 = agent.core.get("params", "enc/0/conv1/kernel").copy()
    it = rb.get_iterator(sample_args={"batch_size": 8, "pack_obs_and_next_obs": True, "lazy": True})
    agent, _ = agent.update_critics(next(it))
    agent, info = agent.update_high_utd(next(it), utd_ratio=1)
    assert np.isfinite(info["critic"]["critic_loss"]) and not np.array_equal(w0, agent.core.get("params", "enc/0/conv1/kernel"))
