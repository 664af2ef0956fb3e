"""TEST INFRASTRUCTURE ONLY.  Runs the REFERENCE's DrQAgent (serl_launcher/agents/continuous/drq.py, built by the
reference's own make_drq_agent, utils/launcher.py:79-116) unmodified under oracle/jaxshim and returns everything a
parity check needs: the inputs, the noise the run drew (recorded from the stand-in jax.random so the same crop offsets /
eps / dropout masks / REDQ indices can be injected into the oracle and the HIP path), the info dicts and the final
train state, keyed by the flat leaf names of oracle/drq_oracle.py.

Needs /root/reference (build container only).  The product never imports this.
"""
from __future__ import annotations

import os
import pickle
import tempfile

import numpy as np

from . import drq_oracle as O
from . import ref_update_shim as R


def pretrained_pickle_tree(trunk):
    """The tree of the reference's resnet10_params.pkl (utils/train_utils.py:113-127 matches its top-level keys against
    `pretrained_encoder`'s children) filled with the given flat trunk leaves."""
    t = {"conv_init": {"kernel": trunk["trunk/conv_init"]},
         "norm_init": {"scale": trunk["trunk/norm_init/scale"], "bias": trunk["trunk/norm_init/bias"]}}
    for i in range(len(O.STAGES)):
        p = f"trunk/block{i}/"
        b = {"Conv_0": {"kernel": trunk[p + "conv0"]},
             "MyGroupNorm_0": {"scale": trunk[p + "gn0/scale"], "bias": trunk[p + "gn0/bias"]},
             "Conv_1": {"kernel": trunk[p + "conv1"]},
             "MyGroupNorm_1": {"scale": trunk[p + "gn1/scale"], "bias": trunk[p + "gn1/bias"]}}
        if p + "proj" in trunk:
            b["conv_proj"] = {"kernel": trunk[p + "proj"]}
            b["norm_proj"] = {"scale": trunk[p + "gnp/scale"], "bias": trunk[p + "gnp/bias"]}
        t[f"ResNetBlock_{i}"] = b
    return t


def _product_name(name, keys):
    parts = name.split("/")
    if parts[0] == "enc" and parts[1] in keys:
        parts[1] = str(list(keys).index(parts[1]))
    return "/".join(parts)


def theta_flax_paths(cfg):
    """oracle leaf name -> flax path of the reference's parameter tree (the product's export mapping: using it here also
    checks serl_amd/agents/flax_tree.py against the tree the reference really builds)."""
    from serl_amd.agents.flax_tree import theta_paths
    prod = theta_paths(cfg.image_keys, encoder_type=cfg.encoder_type)
    return {k: prod[_product_name(k, cfg.image_keys)][0] for k in O.trainable_param_shapes(cfg)}


def _get(tree, path):
    for p in path:
        tree = tree[p]
    return tree


def _set(tree, path, value):
    for p in path[:-1]:
        tree = tree[p]
    assert path[-1] in tree, path
    tree[path[-1]] = value


def synth_packed_batch(cfg, B, seed):
    """A replay sample in the reference's packed format (memory_efficient_replay_buffer.py:126-164 with
    pack_obs_and_next_obs=True): frames u8[B, 2, H, W, 3] per camera, state f32[B, 1, S], ..."""
    rng = np.random.default_rng(seed)
    return {
        "frames": {k: rng.integers(0, 256, (B, 2, cfg.H, cfg.W, 3), dtype=np.uint8) for k in cfg.image_keys},
        "state": rng.standard_normal((B, 1, cfg.S)).astype(np.float32),
        "next_state": rng.standard_normal((B, 1, cfg.S)).astype(np.float32),
        "action": rng.uniform(-1, 1, (B, cfg.A)).astype(np.float32),
        "reward": (rng.random(B) < 0.3).astype(np.float32),
        "mask": (rng.random(B) < 0.9).astype(np.float32),
    }


def _to_reference_batch(jnp, freeze, pb, cfg):
    obs = {k: jnp.asarray(v) for k, v in pb["frames"].items()}
    obs["state"] = jnp.asarray(pb["state"])
    return freeze({"observations": obs, "next_observations": {"state": jnp.asarray(pb["next_state"])},
                   "actions": jnp.asarray(pb["action"]), "rewards": jnp.asarray(pb["reward"]),
                   "masks": jnp.asarray(pb["mask"]), "dones": jnp.asarray(1.0 - pb["mask"])})


class _Tape:
    def __init__(self, recs):
        self.recs, self.i = recs, 0

    def take(self, kind, n=1):
        out = []
        for _ in range(n):
            r = self.recs[self.i]
            assert r["kind"] == kind, (self.i, r["kind"], kind)
            out.append(r)
            self.i += 1
        return out

    def done(self):
        return self.i == len(self.recs)


def _parse_noise(cfg, B, recs, kind, utd, nets=()):
    """Order of draws in the reference (drq.py:244-281 augmentation; sac.py loss functions in sorted-key order)."""
    t = _Tape(recs)
    n_cam = len(cfg.image_keys)
    crops = {}
    for side in (("crop_obs", "crop_next") if (kind != "update" and n_cam) else ()):
        per_cam = []
        for _ in range(n_cam):   # data_augmentation_fn loops over image_keys with the SAME rng (drq.py:244-253)
            per_cam.append(np.stack([r["value"] for r in t.take("randint", B)]).astype(np.int32))
        for c in per_cam[1:]:
            assert np.array_equal(c, per_cam[0]), "the reference must give every camera the same crop offsets"
        crops[side] = per_cam[0]
    noise = dict(crops)

    def masks(rows):
        out = {}
        if cfg.small:          # SmallEncoder(pool_method="avg") has no Dropout (small_encoders.py:43-52)
            return out
        for k in cfg.image_keys:
            (r,) = t.take("bernoulli")
            assert r["context"] and r["context"][-1].endswith(f"encoder_{k}/Dropout_0"), r["context"]
            assert r["value"].shape == (rows, cfg.sle_dim) and abs(r["p"] - (1 - cfg.dropout)) < 1e-12
            out[k] = r["value"].astype(np.uint8)
        return out

    def eps(rows):
        (r,) = t.take("normal")
        assert r["value"].shape == (rows, cfg.A)
        return r["value"].astype(np.float64)

    def critic_draws(rows):      # critic_loss_fn: policy forward at next_obs (dropout), sample, REDQ subsample
        m, e = masks(rows), eps(rows)
        if cfg.subsample is None:        # sac.py:150: no subsampling, no draw
            return m, e, np.zeros((0,), np.int32)
        (r,) = t.take("randint")
        assert r["value"].shape == (cfg.subsample,) and r["maxval"] == cfg.ensemble
        return m, e, r["value"].astype(np.int32)

    if kind == "update":         # ONE apply_loss_fns: loss functions run in sorted-key order actor, critic, temperature
        if "actor" in nets:
            noise["mask_obs_pi"] = masks(B)
            noise["eps_pi"] = eps(B)
        if "critic" in nets:
            m, e, r = critic_draws(B)
            noise["mask_next"], noise["eps_next"], noise["redq_idx"] = m, e, r[None]
        if "temperature" in nets:
            noise["mask_next_temp"] = masks(B)
            noise["eps_temp"] = eps(B)
        assert t.done(), f"{len(recs) - t.i} unexpected random draws"
        return noise
    n_crit = 1 if kind == "critics" else utd
    mb = B // n_crit
    mk, ep, rq = [], [], []
    for _ in range(n_crit):
        m, e, r = critic_draws(mb)
        mk.append(m)
        ep.append(e)
        rq.append(r)
    noise["mask_next"] = {k: np.concatenate([m[k] for m in mk]) for k in (cfg.image_keys if mk[0] else ())}
    noise["eps_next"] = np.concatenate(ep)
    noise["redq_idx"] = np.stack(rq)
    if kind == "high_utd":       # loss_fns dict in sorted order: actor, (critic: zero), temperature
        noise["mask_obs_pi"] = masks(B)
        noise["eps_pi"] = eps(B)
        noise["mask_next_temp"] = masks(B)
        noise["eps_temp"] = eps(B)
    assert t.done(), f"{len(recs) - t.i} unexpected random draws"
    return noise


def _make_reference_agent(jax, jnp, cfg, trunk):
    """The reference's own factory (utils/launcher.py: make_drq_agent / make_sac_agent) with its own hyper-parameters.
    Optimizer options (cfg.opt) are not arguments of those factories: the kwargs the factory passes to
    DrQAgent.create_drq / SACAgent.create_states are intercepted and the reference's `*_optimizer_kwargs` added."""
    from serl_launcher.agents.continuous.drq import DrQAgent
    from serl_launcher.agents.continuous.sac import SACAgent
    from serl_launcher.utils.launcher import make_drq_agent, make_sac_agent
    st = O.TrainState(cfg, {}, {}, None) if False else None
    extra = {}
    if cfg.subsample != 2 or cfg.backup_entropy:   # not arguments of the launcher factories either
        extra["critic_subsample_size"] = cfg.subsample
        extra["backup_entropy"] = bool(cfg.backup_entropy)
    if cfg.opt or cfg.state_only:
        tmp = O.TrainState.__new__(O.TrainState)
        tmp.cfg = cfg
        for tx in ("actor", "critic", "temperature"):
            kw = {k: v for k, v in tmp.tx_opt(tx).items() if v is not None}
            extra[f"{tx}_optimizer_kwargs"] = kw
    if cfg.state_only:
        target, name, factory = SACAgent, "create_states", make_sac_agent
    else:
        target, name, factory = DrQAgent, "create_drq", make_drq_agent
    orig = getattr(target, name).__func__

    def patched(cls, *a, **k):
        return orig(cls, *a, **{**k, **extra})

    setattr(target, name, classmethod(patched))
    se = se_orig = None
    if cfg.small:
        # Reference defect at this commit: EncodingWrapper calls every encoder with `encode=`
        # (common/encoding.py:46) but SmallEncoder.__call__(observations, train) does not accept it -> TypeError.  The
        # adapter below accepts and ignores the kwarg; everything else is the reference's own SmallEncoder.
        import serl_launcher.vision.small_encoders as se
        se_orig = se.SmallEncoder

        class SmallEncoder(se_orig):   # same class name: flax auto-names would not change either
            def __call__(self, observations, train=False, encode=True):
                return se_orig.__call__(self, observations, train)

        se.SmallEncoder = SmallEncoder
    try:
        if cfg.state_only:
            return make_sac_agent(0, jnp.asarray(np.zeros((cfg.S,), np.float32)), jnp.asarray(np.zeros((cfg.A,), np.float32)),
                                  discount=cfg.discount)
        home = tempfile.mkdtemp(prefix="serl_ref_home_")
        os.makedirs(os.path.join(home, ".serl"))
        if not cfg.small:
            with open(os.path.join(home, ".serl", "resnet10_params.pkl"), "wb") as f:
                pickle.dump(pretrained_pickle_tree(trunk), f)
        old_home = os.environ.get("HOME")
        os.environ["HOME"] = home            # train_utils.load_resnet10_params reads ~/.serl/resnet10_params.pkl
        try:
            sample_obs = {k: jnp.asarray(np.zeros((1, cfg.H, cfg.W, 3), np.uint8)) for k in cfg.image_keys}
            sample_obs["state"] = jnp.asarray(np.zeros((1, cfg.S), np.float32))
            return make_drq_agent(0, sample_obs, jnp.asarray(np.zeros((cfg.A,), np.float32)), image_keys=cfg.image_keys,
                                  encoder_type=cfg.encoder_type, discount=cfg.discount)
        finally:
            if old_home is not None:
                os.environ["HOME"] = old_home
    finally:
        setattr(target, name, classmethod(orig))
        if se is not None:
            se.SmallEncoder = se_orig


def synth_flat_batch(cfg, B, seed):
    """A plain ReplayBuffer sample of flat observations (state-only SAC)."""
    rng = np.random.default_rng(seed)
    return {"frames": {}, "state": rng.standard_normal((B, 1, cfg.S)).astype(np.float32),
            "next_state": rng.standard_normal((B, 1, cfg.S)).astype(np.float32),
            "action": rng.uniform(-1, 1, (B, cfg.A)).astype(np.float32),
            "reward": (rng.random(B) < 0.3).astype(np.float32), "mask": (rng.random(B) < 0.9).astype(np.float32)}


def run_reference(cfg: O.Config, B: int, schedule, param_seed=42, batch_seed=100, float64=True):
    """schedule: list of ("critics",), ("high_utd", utd_ratio) or ("update", (network names...)) -- the last one is
    SACAgent.update on an un-augmented batch.  cfg.image_keys == (): the state-only agent of make_sac_agent.
    Returns dict(steps=[...], final=...)."""
    assert R.reference_available(), "/root/reference is not present"
    jax = R.install(float64)
    import jax.numpy as jnp
    from flax.core.frozen_dict import freeze

    trunk, theta = O.init_params(cfg, param_seed)
    agent = _make_reference_agent(jax, jnp, cfg, trunk)
    assert agent.config["critic_ensemble_size"] == cfg.ensemble and agent.config["critic_subsample_size"] == cfg.subsample
    assert abs(agent.config["soft_target_update_rate"] - cfg.tau) < 1e-12 and abs(agent.config["target_entropy"] - cfg.target_entropy) < 1e-12
    assert abs(agent.config["discount"] - cfg.discount) < 1e-12

    # the trunk came in through the reference's own loader; check it landed, then overwrite the trainable leaves
    paths = theta_flax_paths(cfg)
    params = jax.tree_map(lambda a: a, agent.state.params)     # fresh containers, shared leaves
    first = sorted(cfg.image_keys)[0] if (cfg.image_keys and not cfg.small) else None
    if first is not None:
        pe = params["modules_actor"]["encoder"][f"encoder_{first}"]["pretrained_encoder"]
        assert np.array_equal(np.asarray(pe["conv_init"]["kernel"]), trunk["trunk/conv_init"]), "pretrained weights were not patched in"
    for name, path in paths.items():
        cur = _get(params, path)
        val = np.asarray(theta[name], np.float64 if float64 else np.float32).reshape(tuple(cur.shape))
        _set(params, path, jnp.asarray(val))
    params = jax.tree_map(lambda a: jnp.asarray(np.asarray(a)), params)
    agent = agent.replace(state=agent.state.replace(params=params, target_params=params))

    rng0 = [int(v) & 0xFFFFFFFF for v in np.asarray(agent.state.rng).reshape(-1)]   # state.rng as the reference's factory left it
    steps = []
    for i, item in enumerate(schedule):
        kind = item[0]
        utd = item[1] if kind == "high_utd" else 1
        nets = tuple(item[1]) if kind == "update" else ()
        pb = synth_flat_batch(cfg, B, batch_seed + i) if cfg.state_only else synth_packed_batch(cfg, B, batch_seed + i)
        if cfg.state_only:
            batch = freeze({"observations": jnp.asarray(pb["state"][:, 0]), "next_observations": jnp.asarray(pb["next_state"][:, 0]),
                            "actions": jnp.asarray(pb["action"]), "rewards": jnp.asarray(pb["reward"]),
                            "masks": jnp.asarray(pb["mask"])})
        elif kind == "update":   # SACAgent.update takes an already unpacked (and, in the learner, augmented) batch
            obs = {k: jnp.asarray(v[:, :1]) for k, v in pb["frames"].items()}
            nobs = {k: jnp.asarray(v[:, 1:]) for k, v in pb["frames"].items()}
            obs["state"], nobs["state"] = jnp.asarray(pb["state"]), jnp.asarray(pb["next_state"])
            batch = freeze({"observations": obs, "next_observations": nobs, "actions": jnp.asarray(pb["action"]),
                            "rewards": jnp.asarray(pb["reward"]), "masks": jnp.asarray(pb["mask"])})
        else:
            batch = _to_reference_batch(jnp, freeze, pb, cfg)
        tape = jax.random.start_tape()
        if kind == "critics":
            agent, info = agent.update_critics(batch)
        elif kind == "high_utd":
            agent, info = agent.update_high_utd(batch, utd_ratio=utd)
        else:
            agent, info = agent.update(batch, networks_to_update=frozenset(nets))
        jax.random.stop_tape()
        noise = _parse_noise(cfg, B, tape, kind, utd, nets)
        flat_info = {}
        for k, v in info.items():
            if isinstance(v, dict):
                for kk, vv in v.items():
                    flat_info[kk] = float(np.asarray(vv))
            else:
                flat_info[k] = float(np.asarray(v))
        steps.append({"kind": kind, "utd": utd, "nets": nets, "batch": pb, "noise": noise, "info": flat_info})

    st = agent.state
    final = {"step": int(np.asarray(st.step)), "params": {}, "target": {}, "mu": {}, "nu": {}, "count": {}}
    for name, path in paths.items():
        final["params"][name] = np.asarray(_get(st.params, path), np.float64).reshape(-1)
        final["target"][name] = np.asarray(_get(st.target_params, path), np.float64).reshape(-1)
    for tx in ("actor", "critic", "temperature"):
        s = st.opt_states[tx]
        adam = s.inner_state[-1][0]       # chain([clip_by_global_norm], adam | adamw)[-1] = (ScaleByAdamState, ...)
        final["count"][tx] = (int(np.asarray(s.count)), int(np.asarray(adam.count)))
        final["mu"][tx] = {n: np.asarray(_get(adam.mu, p), np.float64).reshape(-1) for n, p in paths.items()}
        final["nu"][tx] = {n: np.asarray(_get(adam.nu, p), np.float64).reshape(-1) for n, p in paths.items()}
    if first is not None:
        # the frozen trunk: parameters must not have moved (without weight decay), the target copy follows the EMA
        tpe = st.target_params["modules_actor"]["encoder"][f"encoder_{first}"]["pretrained_encoder"]
        ppe = st.params["modules_actor"]["encoder"][f"encoder_{first}"]["pretrained_encoder"]
        final["trunk_conv_init"] = np.asarray(ppe["conv_init"]["kernel"], np.float64).reshape(-1)
        final["trunk_conv_init_target"] = np.asarray(tpe["conv_init"]["kernel"], np.float64).reshape(-1)
    final["param_tree"] = jax.tree_map(lambda a: tuple(np.shape(a)), st.params)
    final["opt_state_tree"] = jax.tree_map(lambda a: tuple(np.shape(a)), {k: _state_dict(v) for k, v in st.opt_states.items()})
    final["rng"] = [int(v) & 0xFFFFFFFF for v in np.asarray(agent.state.rng).reshape(-1)]
    return {"cfg": cfg, "B": B, "schedule": list(schedule), "steps": steps, "final": final, "rng0": rng0}


def _state_dict(x):
    """flax.serialization.to_state_dict restated for the optimizer states: NamedTuples -> dicts of their fields,
    tuples / lists -> dicts keyed '0', '1', ... (what a flax checkpoint of agent.state.opt_states contains)."""
    if isinstance(x, tuple) and hasattr(x, "_fields"):
        return {f: _state_dict(getattr(x, f)) for f in x._fields}
    if isinstance(x, (tuple, list)):
        return {str(i): _state_dict(v) for i, v in enumerate(x)}
    if isinstance(x, dict):
        return {k: _state_dict(v) for k, v in x.items()}
    return x


def oracle_batch_and_noise(cfg, step, dtype):
    """The oracle's (cropped-frame) batch and torch noise for one recorded reference step."""
    import torch
    from .replay_oracle import random_shift
    pb, n = step["batch"], step["noise"]
    ident = np.full((pb["reward"].shape[0], 2), 4, np.int32)     # SACAgent.update: no augmentation = the centre shift
    b = {"obs": {k: torch.from_numpy(random_shift(pb["frames"][k][:, 0], n.get("crop_obs", ident))) for k in cfg.image_keys},
         "next": {k: torch.from_numpy(random_shift(pb["frames"][k][:, 1], n.get("crop_next", ident))) for k in cfg.image_keys},
         "state": torch.tensor(pb["state"][:, 0], dtype=dtype), "next_state": torch.tensor(pb["next_state"][:, 0], dtype=dtype),
         "action": torch.tensor(pb["action"], dtype=dtype), "reward": torch.tensor(pb["reward"], dtype=dtype),
         "mask": torch.tensor(pb["mask"], dtype=dtype)}
    return b, O.noise_to_torch(n, dtype)
