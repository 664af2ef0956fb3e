"""GPU: the reference-named Python surface (make_drq_agent, DrQAgent.update_*, sample_actions,
agent.state, get_iterator / concat_batches learner loop) on top of the C ABI."""
import itertools

import numpy as np
import pytest
import torch

from helpers import make_spaces

pytestmark = pytest.mark.gpu
KEYS, H, W, S, A = ("front", "wrist"), 64, 64, 7, 4


class _Env:
    def __init__(self):
        self.observation_space, self.action_space = make_spaces(KEYS, H, W, 3, 1, S, A)


def _setup(B=16, demo=False):
    from serl_amd.utils.launcher import make_drq_agent, make_replay_buffer
    from serl_amd.utils.synthetic import transition_stream
    env = _Env()
    rb = make_replay_buffer(env, capacity=300, type="memory_efficient_replay_buffer", image_keys=KEYS)
    rb.seed(0)
    for tr in itertools.islice(transition_stream(KEYS, H, W, 3, 1, S, A, 20, 5), 150):
        rb.insert(tr)
    obs = {"front": np.zeros((1, H, W, 3), np.uint8), "wrist": np.zeros((1, H, W, 3), np.uint8),
           "state": np.zeros((1, S), np.float32)}
    agent = make_drq_agent(3, obs, np.zeros((A,), np.float32), image_keys=KEYS,
                           encoder_type="resnet-pretrained", batch_size=B)
    return env, rb, agent


def test_learner_loop_shape(gpu):
    """examples/async_drq_sim/async_drq_sim.py:238-292 with critic_actor_ratio=2 and a demo buffer."""
    from serl_amd.utils.launcher import make_replay_buffer
    from serl_amd.utils.synthetic import transition_stream
    from serl_amd.utils.train_utils import concat_batches
    env, rb, agent = _setup(B=16)
    demo = make_replay_buffer(env, capacity=100, type="memory_efficient_replay_buffer", image_keys=KEYS)
    demo.seed(1)
    for tr in itertools.islice(transition_stream(KEYS, H, W, 3, 1, S, A, 10, 9), 40):
        demo.insert(tr)
    args = {"batch_size": 8, "pack_obs_and_next_obs": True, "lazy": True}
    it, dit = rb.get_iterator(sample_args=args), demo.get_iterator(sample_args=args)
    w0 = agent.core.get("params", "critic/w1").copy()
    for step in range(3):
        batch = concat_batches(next(it), next(dit), axis=0)
        agent, cinfo = agent.update_critics(batch)
        assert set(cinfo.keys()) == {"critic", "actor_lr", "critic_lr", "temperature_lr"}
        batch = concat_batches(next(it), next(dit), axis=0)
        agent, info = agent.update_high_utd(batch, utd_ratio=1)
        d = info.resolve()
        assert set(d["critic"]) == {"critic_loss", "predicted_qs", "target_qs"}
        assert set(d["actor"]) == {"actor_loss", "temperature", "entropy"}
        assert set(d["temperature"]) == {"temperature_loss"}
        assert all(np.isfinite(v) for v in d["critic"].values())
    assert agent.state.step == 9            # 3 x (1 critic + (1 critic + 1 actor/temp)) update() calls
    assert not np.array_equal(w0, agent.core.get("params", "critic/w1"))
    _, stale = agent.update_critics(concat_batches(next(it), next(dit), axis=0))
    agent.update_critics(concat_batches(next(it), next(dit), axis=0))
    with pytest.raises(RuntimeError):
        stale.resolve()                      # never read, and overwritten by the later update


def test_dict_batch_equals_lazy_batch(gpu):
    """eager reference-format batch (packed frames) and the fused lazy path feed identical bytes."""
    env, rb, agent = _setup(B=8)
    idx = rb.sample_indices(8)
    from serl_amd.data.data_store import LazyBatch
    crops = (np.random.default_rng(0).integers(0, 9, (8, 2)).astype(np.int32),
             np.random.default_rng(1).integers(0, 9, (8, 2)).astype(np.int32))
    a = agent.prepare(LazyBatch([(rb, idx)]), crops)
    fa, sa, aa = a.frames.clone(), a.state.clone(), a.action.clone()
    b = agent.prepare(rb.gather(idx), crops)
    torch.cuda.synchronize()
    assert torch.equal(fa, b.frames) and torch.equal(sa, b.state) and torch.equal(aa, b.action)


def test_state_export_and_sample_actions(gpu):
    env, rb, agent = _setup(B=8)
    p = agent.state.params
    enc = p["modules_actor"]["encoder"]
    assert enc["encoder_front"]["pretrained_encoder"]["conv_init"]["kernel"].shape == (7, 7, 3, 64)
    assert enc["encoder_wrist"]["SpatialLearnedEmbeddings_0"]["kernel"].shape == (2, 2, 512, 8)
    assert enc["Dense_0"]["kernel"].shape == (S, 64)
    assert p["modules_critic"]["network"]["Dense_0"]["kernel"].shape == (10, 2 * 256 + 64 + A, 256)
    assert p["modules_actor"]["Dense_1"]["bias"].shape == (A,)
    assert p["modules_temperature"]["lagrange"].shape == ()
    tp = agent.state.target_params
    assert np.array_equal(tp["modules_actor"]["network"]["Dense_0"]["kernel"], p["modules_actor"]["network"]["Dense_0"]["kernel"])
    os_ = agent.state.opt_states
    assert set(os_) == {"actor", "critic", "temperature"} and os_["critic"]["count"] == 0
    obs = {"front": np.random.default_rng(0).integers(0, 256, (1, H, W, 3), dtype=np.uint8),
           "wrist": np.random.default_rng(1).integers(0, 256, (1, H, W, 3), dtype=np.uint8),
           "state": np.zeros((1, S), np.float32)}
    a = agent.sample_actions(obs, argmax=True)
    assert a.shape == (A,) and np.all(np.abs(a) <= 1)
    a2 = agent.sample_actions(obs, seed=np.array([0, 5], np.uint32))
    assert a2.shape == (A,) and not np.allclose(a, a2)
    with pytest.raises(AssertionError):
        agent.sample_actions(obs, seed=np.array([0, 5], np.uint32), argmax=True)   # sac.py:316-317


def test_reference_error_behaviour(gpu):
    env, rb, agent = _setup(B=6)
    batch = rb.sample(6, pack_obs_and_next_obs=True, lazy=True)
    with pytest.raises(AssertionError, match="divisible by UTD"):
        agent.update_high_utd(batch, utd_ratio=4)
    from serl_amd.agents.drq import DrQAgent
    with pytest.raises(NotImplementedError, match="Unknown encoder type"):
        DrQAgent.create_drq(0, {"front": np.zeros((1, H, W, 3)), "state": np.zeros((1, S))}, np.zeros(A),
                            encoder_type="bogus", image_keys=("front",))


def test_checkpoint_roundtrip(gpu, tmp_path):
    """N1: save_checkpoint(agent.state) -> restore into a fresh agent -> identical next update."""
    from serl_amd.utils.checkpoint import latest_checkpoint, restore_checkpoint, save_checkpoint
    env, rb, agent = _setup(B=8)
    for _ in range(2):
        agent.update_high_utd(rb.sample(8, pack_obs_and_next_obs=True, lazy=True), utd_ratio=1)
    p1 = save_checkpoint(str(tmp_path), agent, step=agent.state.step, keep=2)
    agent.update_critics(rb.sample(8, pack_obs_and_next_obs=True, lazy=True))
    p2 = save_checkpoint(str(tmp_path), agent, step=agent.state.step, keep=2)
    assert latest_checkpoint(str(tmp_path)) == p2 and p1 != p2
    env2, rb2, fresh = _setup(B=8)
    restore_checkpoint(str(tmp_path), fresh)
    assert fresh.state.step == agent.state.step
    for leaf in ("critic/w1", "actor/w2", "enc/0/dense/kernel", "temp/lagrange", "trunk/block2/conv1"):
        for sec in ("params", "target_params"):
            assert np.array_equal(fresh.core.get(sec, leaf), agent.core.get(sec, leaf)), (sec, leaf)
    for sec, leaf in (("opt/critic/mu", "critic/w2"), ("opt/actor/nu", "actor/w1"), ("opt/temperature/mu", "temp/lagrange"),
                      ("opt/critic/nu", "enc/proprio/dense/kernel"), ("opt/actor/mu", "enc/proprio/dense/kernel")):
        assert np.array_equal(fresh.core.get(sec, leaf), agent.core.get(sec, leaf)), (sec, leaf)


def test_iterator_prefetch_is_transparent(gpu):
    """Lazy batches from get_iterator let the agent run gather + crop + trunk of the NEXT batch on a second stream
    under the current update; the results are bit-identical to the unpipelined agent (same crop / noise streams)."""
    from serl_amd.utils.launcher import make_replay_buffer
    from serl_amd.utils.synthetic import transition_stream
    from serl_amd.utils.train_utils import concat_batches

    def run(prefetch):
        env, rb, agent = _setup(B=16)
        demo = make_replay_buffer(env, capacity=100, type="memory_efficient_replay_buffer", image_keys=KEYS)
        demo.seed(1)
        for tr in itertools.islice(transition_stream(KEYS, H, W, 3, 1, S, A, 10, 9), 40):
            demo.insert(tr)
        agent.prefetch = prefetch
        args = {"batch_size": 8, "pack_obs_and_next_obs": True, "lazy": True}
        it, dit = rb.get_iterator(sample_args=args), demo.get_iterator(sample_args=args)
        infos = []
        for step in range(4):
            agent, _ = agent.update_critics(concat_batches(next(it), next(dit), axis=0))
            agent, info = agent.update_high_utd(concat_batches(next(it), next(dit), axis=0), utd_ratio=2)
            infos.append(info.resolve())
        used = agent._prefetched is not None
        return agent, infos, used

    a0, i0, used0 = run(False)
    a1, i1, used1 = run(True)
    assert used1 and not used0
    for leaf in ("critic/w1", "critic/head/kernel", "actor/w2", "enc/0/dense/kernel", "enc/proprio/dense/kernel", "temp/lagrange"):
        assert np.array_equal(a0.core.get("params", leaf), a1.core.get("params", leaf)), leaf
    assert i0[-1]["critic"] == i1[-1]["critic"] and i0[-1]["actor"] == i1[-1]["actor"]
    # a caller that skips a batch just loses the prefetch (the crop stream then differs from the unpipelined agent's)
    env, rb, agent = _setup(B=16)
    it = rb.get_iterator(sample_args={"batch_size": 16, "pack_obs_and_next_obs": True, "lazy": True})
    agent.update_critics(next(it))
    next(it)
    _, info = agent.update_high_utd(next(it), utd_ratio=1)
    assert all(np.isfinite(v) for v in info.resolve()["critic"].values()) and agent.state.step == 3


def test_prefetched_slot_is_dropped_when_state_rng_moves(gpu):
    """The next batch's crop offsets are drawn one call ahead from the state.rng the consuming call is EXPECTED to enter with.  A
    state.replace(rng=...) in between (checkpoint restore, reseed) makes that slot stale: the reference crops with the key it enters
    the call with (drq.py:276-281), so the slot must be produced again."""
    from serl_amd import jaxrng as J
    env, rb, agent = _setup(B=16)
    it = rb.get_iterator(sample_args={"batch_size": 16, "pack_obs_and_next_obs": True, "lazy": True})
    agent.update_critics(next(it))
    assert agent._prefetched is not None and np.array_equal(agent._prefetched[3], agent.state.rng)   # expected entry key of the next call
    stale = agent._slot_crops[agent._prefetched[1]]
    new = J.split(J.prngkey(999))[1]
    agent.state.replace(rng=new)
    agent.update_critics(next(it))
    k = J.split(new, 3)
    assert np.array_equal(agent.last_draws["crop_obs"], J.crop_offsets(k[1], 16, 4))
    assert np.array_equal(agent.last_draws["crop_next"], J.crop_offsets(k[2], 16, 4))
    assert not np.array_equal(agent.last_draws["crop_obs"], stale[0])
    # undisturbed, the prefetched slot IS used and carries the draws of the entry key
    k = J.split(agent.state.rng, 3)
    want = J.crop_offsets(k[1], 16, 4)
    agent.update_critics(next(it))
    assert np.array_equal(agent.last_draws["crop_obs"], want)


def test_state_replace_loads_trees_into_hbm(gpu):
    """agent.state.replace(params=...) / agent.replace(state=<restored dict>) of the reference: the trees land in HBM."""
    env, rb, src = _setup(B=8)
    src.update_high_utd(rb.sample(8, pack_obs_and_next_obs=True, lazy=True), utd_ratio=1)
    env2, rb2, dst = _setup(B=8)
    dst.state.replace(params=src.state.params, step=src.state.step)
    assert dst.state.step == src.state.step
    assert np.array_equal(dst.core.get("params", "actor/w1"), src.core.get("params", "actor/w1"))
    assert not np.array_equal(dst.core.get("target_params", "critic/w1"), src.core.get("target_params", "critic/w1"))
    dst = dst.replace(state={"target_params": src.state.target_params, "opt_states": src.state.opt_states})
    for sec, leaf in (("target_params", "critic/w1"), ("opt/critic/mu", "critic/w2"), ("opt/actor/nu", "actor/w1")):
        assert np.array_equal(dst.core.get(sec, leaf), src.core.get(sec, leaf)), (sec, leaf)
    with pytest.raises(TypeError):
        dst.state.replace(parms={})


def test_learner_example_with_the_transport_endpoint(gpu):
    """examples/learner_drq_synthetic.py: the reference's learner loop with a TrainerServer; a mock actor thread ships
    transitions through TrainerClient.update(), requests send-stats and receives the published networks (next-row N2)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "learner_drq_synthetic.py"), "--steps", "9", "--batch_size", "32",
                        "--critic_actor_ratio", "2", "--training_starts", "300", "--steps_per_update", "3", "--log_period", "4",
                        "--port", "6712"], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    done = [ln for ln in r.stdout.splitlines() if ln.startswith("done:")]
    assert done and "18 grad-steps" in done[0] and "modules_actor" in done[0], r.stdout[-2000:]


def test_load_resnet10_params_from_a_synthetic_pickle(gpu, tmp_path):
    """N1: load_resnet10_params (train_utils.py:69-130) on a pickle shaped like resnet10_params.pkl: the trunk leaves in
    HBM (params AND target_params: sac.py:378-382 makes them one tree when the reference patches it) become the
    pickle's, the trainable leaves are untouched, and the trunk features change to those of the new weights."""
    import pickle
    from serl_amd.agents.flax_tree import _trunk_paths
    from serl_amd.utils import init as pinit
    from serl_amd.utils.train_utils import load_resnet10_params
    env, rb, agent = _setup(B=8)
    new = pinit.init_trunk(seed=77)
    tree = {}
    for leaf, sub in _trunk_paths().items():
        d = tree
        for p in sub[:-1]:
            d = d.setdefault(p, {})
        d[sub[-1]] = new[leaf]
    f = tmp_path / "resnet10_params.pkl"
    pickle.dump(tree, open(f, "wb"))
    before_w1 = agent.core.get("params", "critic/w1").copy()
    frames = torch.randint(0, 256, (4, H, W, 3), dtype=torch.uint8, device="cuda")
    feat0 = agent.core.trunk_forward(frames).cpu().numpy()
    assert load_resnet10_params(agent, KEYS, file_path=str(f)) is agent
    for leaf in new:
        for sec in ("params", "target_params"):
            assert np.array_equal(agent.core.get(sec, leaf).reshape(-1), new[leaf].reshape(-1)), (sec, leaf)
    assert np.array_equal(agent.core.get("params", "critic/w1"), before_w1)
    feat1 = agent.core.trunk_forward(frames).cpu().numpy()
    assert np.abs(feat1 - feat0).max() > 1e-3
    other = _setup(B=8)[2]
    other.load_trunk_params(tree)
    assert np.array_equal(other.core.trunk_forward(frames).cpu().numpy(), feat1)


def test_pickle_with_imagenet_like_weight_ranges_through_the_f16x3_trunk(gpu, tmp_path):
    """the split-fp16 trunk scales every output channel's weights by a power of two (pack_weights_kernel,
    pack_conv_init_u8_kernel) and has only ever met kaiming-normal pickles.  A pickle shaped like resnet10_params.pkl
    (train_utils.py:69-130) whose per-output-channel kernel magnitudes span EIGHT decades (1e-7 .. 10 of the init scale), with the
    power-of-two edge cases pinned -- channels whose max |w| is exactly 2^-20, exactly 2^-24 - 1 ulp (just below a binade), an
    all-zero channel, a channel with one weight and nothing else, weights inside a channel spanning 1e-6 .. 1 -- loaded with
    load_resnet10_params and run through the DEFAULT (f16x3) trunk against the fp64 oracle and against plain fp32 on the CPU."""
    import pickle
    from oracle import drq_oracle as O
    import agent_helpers as AH
    from serl_amd.agents.flax_tree import _trunk_paths
    from serl_amd.utils import init as pinit
    from serl_amd.utils.train_utils import load_resnet10_params
    env, rb, agent = _setup(B=8)
    agent.core.set_trunk_mode("f16x3")      # (the default; named here because this test is about its weight scaling)
    rng = np.random.default_rng(11)
    new = pinit.init_trunk(seed=78)
    for leaf, v in new.items():
        if v.ndim != 4:
            if leaf.endswith("scale"):
                new[leaf] = rng.uniform(0.2, 3.0, v.shape).astype(np.float32)
            else:
                new[leaf] = rng.uniform(-1.0, 1.0, v.shape).astype(np.float32)
            continue
        co = v.shape[-1]
        g = (10.0 ** rng.uniform(-7.0, 1.0, co)).astype(np.float32)          # eight decades across output channels
        w = v * g
        flat = w.reshape(-1, co)
        flat[:, 0] *= np.float32(2.0 ** -20) / np.abs(flat[:, 0]).max()        # max |w| == 2^-20 exactly
        flat[:, 1] *= np.nextafter(np.float32(2.0 ** -24), np.float32(0)) / np.abs(flat[:, 1]).max()   # one ulp below a binade
        flat[:, 2] = 0.0                                                       # dead channel
        flat[:, 3] = 0.0
        flat[rng.integers(0, flat.shape[0]), 3] = 0.75                         # a single tap
        flat[:, 4] *= (10.0 ** rng.uniform(-6.0, 0.0, flat.shape[0])).astype(np.float32)   # six decades INSIDE one channel
        flat[:, 5] *= np.float32(3000.0) / np.abs(flat[:, 5]).max()            # a very strong filter (fp16 overflow without the scale)
        new[leaf] = flat.reshape(v.shape).astype(np.float32)
    tree = {}
    for leaf, sub in _trunk_paths().items():
        d = tree
        for q in sub[:-1]:
            d = d.setdefault(q, {})
        d[sub[-1]] = new[leaf]
    f = tmp_path / "resnet10_params.pkl"
    pickle.dump(tree, open(f, "wb"))
    load_resnet10_params(agent, KEYS, file_path=str(f))
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([rng.integers(0, 256, (H, W, 3)), rng.integers(0, 256, (H, W, 3)), np.repeat((xx * 4)[..., None], 3, -1),
                    rng.integers(0, 2, (H, W, 3)) * 255, rng.integers(100, 140, (H, W, 3)),
                    np.repeat((((yy // 4 + xx // 4) & 1) * 255)[..., None], 3, -1)]).astype(np.uint8)
    t64 = {k: torch.tensor(v, dtype=torch.float64) for k, v in new.items()}
    ref = O.trunk_forward(t64, torch.tensor(img), torch.float64).numpy()
    ref32 = O.trunk_forward({k: v.float() for k, v in t64.items()}, torch.tensor(img), torch.float32).numpy()
    got = agent.core.trunk_forward(torch.tensor(img, device="cuda")).cpu().numpy()
    assert np.isfinite(got).all() and np.abs(ref).max() > 1e-3
    for i in range(len(img)):
        err, err32 = AH.rel_err(got[i], ref[i]), AH.rel_err(ref32[i], ref[i])
        print(f"wide-range pickle, frame {i}: f16x3 rel err vs fp64 {err:.2e} (plain fp32 on the CPU {err32:.2e})")
        assert err < max(5e-6, 4.0 * err32), (i, err, err32)
    # the exact-fp32 MFMA trunk on the same pickle agrees with the split arithmetic
    agent.core.set_trunk_mode("f32")
    got32 = agent.core.trunk_forward(torch.tensor(img, device="cuda")).cpu().numpy()
    assert AH.rel_err(got32, got) < 1e-5


def test_zero_learning_rate_is_honoured(gpu):
    """optax accepts learning_rate=0.0 (e.g. to freeze the temperature, common/optimizers.py:23-30): the kernel must train
    that optimizer's leaves at 0, not fall back to the default 3e-4, and the info dict / exported hyperparams say 0 too."""
    from serl_amd.utils.launcher import make_drq_agent
    env, rb, _ = _setup(B=8)
    obs = {"front": np.zeros((1, H, W, 3), np.uint8), "wrist": np.zeros((1, H, W, 3), np.uint8), "state": np.zeros((1, S), np.float32)}
    agent = make_drq_agent(3, obs, np.zeros((A,), np.float32), image_keys=KEYS, encoder_type="resnet-pretrained", batch_size=8,
                           temperature_optimizer_kwargs={"learning_rate": 0.0})
    lam0 = agent.core.get("params", "temp/lagrange").copy()
    w0 = agent.core.get("params", "actor/w2").copy()
    for _ in range(3):
        agent, info = agent.update_high_utd(rb.sample(8, pack_obs_and_next_obs=True, lazy=True), utd_ratio=1)
    d = info.resolve()
    assert d["temperature_lr"] == 0.0 and abs(d["actor_lr"] - 3e-4) < 1e-9
    assert np.array_equal(agent.core.get("params", "temp/lagrange"), lam0)
    assert not np.array_equal(agent.core.get("params", "actor/w2"), w0)
    assert float(agent.state.opt_states["temperature"]["hyperparams"]["learning_rate"]) == 0.0


@pytest.mark.parametrize("chain_fuse", ["1", "0"])
def test_draws_inside_the_kernels_equal_the_materialised_draws(gpu, monkeypatch, chain_fuse):
    """The reference's random stream reaches the kernels as KEYS by default (serl_noise key_*: Dropout masks drawn in the
    SpatialLearnedEmbeddings kernel, normals in the policy-head epilogue -- csrc/heads.hip through csrc/jaxrng.h) or, with
    agent.noise_form = "tensors", as tensors filled by one serl_jax_fill launch (bit-exact against the oracle's jax.random in
    tests/test_jaxrng.py).  Same keys, same elements of the same arrays: parameters and info must agree TO THE BIT, for minibatch
    windows of a UTD = 2 update too, on the fused chain and on the one-launch-per-operation chain (which materialises key draws
    itself: agent.hip jax_noise_tensors)."""
    import numpy as np
    import torch
    from serl_amd.utils.launcher import make_drq_agent
    monkeypatch.setenv("SERL_CHAIN_FUSE", chain_fuse)
    keys_, H, S, A, B = ("front", "wrist"), 64, 5, 3, 8
    obs0 = {k: np.zeros((1, H, H, 3), np.uint8) for k in keys_}
    obs0["state"] = np.zeros((1, S), np.float32)
    rng = np.random.default_rng(0)

    def batch():
        t = lambda a: torch.tensor(a, device="cuda")  # noqa: E731
        obs = {k: t(rng.integers(0, 256, (B, 2, H, H, 3), dtype=np.uint8)) for k in keys_}
        obs["state"] = t(rng.standard_normal((B, 1, S)).astype(np.float32))
        return {"observations": obs, "next_observations": {"state": t(rng.standard_normal((B, 1, S)).astype(np.float32))},
                "actions": t(rng.uniform(-1, 1, (B, A)).astype(np.float32)), "rewards": t((rng.random(B) < 0.3).astype(np.float32)),
                "masks": t((rng.random(B) < 0.9).astype(np.float32))}

    batches = [batch() for _ in range(3)]
    out = []
    for form in ("keys", "tensors"):
        agent = make_drq_agent(5, obs0, np.zeros((A,), np.float32), image_keys=keys_, encoder_type="resnet-pretrained", batch_size=B)
        agent.noise_form = form
        infos = []
        agent, info = agent.update_critics(batches[0]); infos.append(dict(info["critic"]))
        agent, info = agent.update_high_utd(batches[1], utd_ratio=2); infos.append({**info["critic"], **info["actor"], **info["temperature"]})
        agent, info = agent.update(batches[2]); infos.append({**info["critic"], **info["actor"], **info["temperature"]})
        torch.cuda.synchronize()
        out.append((infos, {k: agent.core.get("params", k) for k in ("critic/w1", "actor/w2", "enc/0/dense/kernel", "enc/1/sle", "temp/lagrange")},
                    [int(v) for v in agent.state.rng]))
    assert out[0][0] == out[1][0], (out[0][0], out[1][0])
    assert out[0][2] == out[1][2]
    for k in out[0][1]:
        assert np.array_equal(out[0][1][k].view(np.uint32), out[1][1][k].view(np.uint32)), k
