"""DrQAgent / SACAgent with the reference's Python API, backed by libserl_mi355.so.

Mirrors (same names, argument meaning, error behaviour):
  serl_launcher/agents/continuous/drq.py:24-328   DrQAgent.create_drq / update_high_utd / update_critics
  serl_launcher/agents/continuous/sac.py:243-320  SACAgent.update / sample_actions
  serl_launcher/common/common.py:81-114           JaxRLTrainState fields (step, params, target_params,
                                                   opt_states, rng) exposed through `agent.state`
The reference is functional (returns a new agent); this agent mutates its device state and
returns `self`, which is what the learner loop (`agent, info = agent.update_critics(batch)`) needs.
All numerics run in HIP kernels; there is no CPU fallback.
"""
from __future__ import annotations

from typing import Dict, FrozenSet, Iterable, Optional

import numpy as np
import torch

from ..data.data_store import LazyBatch, gather_crop
from ..utils import init as pinit
from .. import _lib
from .. import jaxrng as J
from .batch import DeviceBatch
from .core import APPLY_ACTOR_TEMP, APPLY_CRITIC, TX_NAMES, AgentCore
from .flax_tree import export_tree


def lr_schedule(kw: dict, count: int) -> float:
    """make_optimizer's learning-rate schedule (common/optimizers.py:14-30) at `count`."""
    import math
    peak, warm = float(kw.get("learning_rate", 3e-4)), int(kw.get("warmup_steps", 0) or 0)
    if count < warm:
        return peak * count / warm
    cos = kw.get("cosine_decay_steps")
    if cos is not None:
        T = max(int(cos) - warm, 1)
        return peak * 0.5 * (1.0 + math.cos(math.pi * min(count - warm, T) / T))
    return peak


class PendingInfo:
    """Info dict of the LAST update call; reading it synchronises the stream (the reference's jitted
    update returns device arrays that are also only materialised when logged)."""

    def __init__(self, agent, kind, serial):
        self._agent, self._kind, self._serial, self._val = agent, kind, serial, None

    def resolve(self) -> dict:
        if self._val is None:
            if self._agent._update_serial != self._serial:
                raise RuntimeError("info of an older update was overwritten; read it before the next update")
            r = self._agent.core.read_info()
            lr = {f"{n}_lr": r[f"{n}_lr"] for n in TX_NAMES}
            if isinstance(self._kind, frozenset):   # SACAgent.update: an info dict per network, {} for the skipped ones
                nets = self._kind
                out = {"critic": {k: r[k] for k in ("critic_loss", "predicted_qs", "target_qs")} if "critic" in nets else {},
                       "actor": {k: r[k] for k in ("actor_loss", "temperature", "entropy")} if "actor" in nets else {},
                       "temperature": {"temperature_loss": r["temperature_loss"]} if "temperature" in nets else {}}
            else:
                out = {"critic": {k: r[k] for k in ("critic_loss", "predicted_qs", "target_qs")}}
                if self._kind == "high_utd":
                    out["actor"] = {k: r[k] for k in ("actor_loss", "temperature", "entropy")}
                    out["temperature"] = {"temperature_loss": r["temperature_loss"]}
            out.update(lr)
            self._val = out
        return self._val

    def __getitem__(self, k):
        return self.resolve()[k]

    def items(self):
        return self.resolve().items()

    def keys(self):
        return self.resolve().keys()

    def __repr__(self):
        return repr(self.resolve())


class TrainStateView:
    """agent.state: JaxRLTrainState-shaped, materialised from HBM on access (common.py:81-114)."""

    def __init__(self, agent):
        self._a = agent

    @property
    def step(self):
        return self._a.core.step

    @property
    def params(self):
        return export_tree(self._a.core, "params", self._a.image_keys)

    @property
    def target_params(self):
        return export_tree(self._a.core, "target_params", self._a.image_keys)

    @property
    def opt_states(self):
        """{tx: InjectHyperparamsState} in flax's state-dict form (flax.serialization.to_state_dict: NamedTuples become
        dicts of their fields, tuples dicts keyed '0', '1', ...).  Each reference tx is
        optax.inject_hyperparams(chain([clip_by_global_norm], adam | adamw)) (common/optimizers.py:32-56):
          {count, hyperparams: {learning_rate}, inner_state: {'0': <clip: {}>?, '<k>': {'0': {count, mu, nu}, '1': {} ...}}}
        (non-numeric hyper-parameters such as weight_decay=None are not stored by optax)."""
        out = {}
        step = np.int32(self._a.core.step)
        for tx in TX_NAMES:
            kw = self._a._opts.get(tx, {})
            adam = {"count": step, "mu": export_tree(self._a.core, f"opt/{tx}/mu", self._a.image_keys),
                    "nu": export_tree(self._a.core, f"opt/{tx}/nu", self._a.image_keys)}
            # optax.adam = chain(scale_by_adam, scale_by_learning_rate); adamw adds add_decayed_weights in between
            inner_adam = {"0": adam, "1": {}} if kw.get("weight_decay") is None else {"0": adam, "1": {}, "2": {}}
            stages = ([{}] if kw.get("clip_grad_norm") is not None else []) + [inner_adam]
            hp = {"learning_rate": np.float32(self._a.lr_at(int(step), tx))}
            if kw.get("weight_decay") is not None:
                hp["weight_decay"] = np.float32(kw["weight_decay"])
            out[tx] = {"count": step, "hyperparams": hp, "inner_state": {str(i): s for i, s in enumerate(stages)}}
        return out

    @property
    def rng(self):
        return self._a._rng_key.copy()

    def replace(self, **kw):
        """JaxRLTrainState.replace(params=..., target_params=..., opt_states=..., step=...): loads the given trees
        into HBM (in place) and returns the view."""
        bad = set(kw) - {"params", "target_params", "opt_states", "step", "rng"}
        if bad:
            raise TypeError(f"unknown TrainState fields: {sorted(bad)}")
        from ..utils.checkpoint import load_state_dict
        load_state_dict(self._a, {k: v for k, v in kw.items() if k != "rng"})
        if kw.get("rng") is not None:
            self._a._rng_key = np.asarray(kw["rng"], np.uint32).copy()
        return self


class DrQAgent:
    _DRQ_AUG = True     # update_critics / update_high_utd open with `rng, obs_rng, next_obs_rng = split(rng, 3)` (drq.py:276-277,307-308)

    def __init__(self, core: AgentCore, image_keys, config: dict, seed: int):
        self.core = core
        self.image_keys = tuple(image_keys)
        self.config = config
        self._np_rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, 0xD1CE])))
        # RANDOMNESS.  "threefry" (default): the reference's own stream -- crop offsets, REDQ indices, policy noise and Dropout
        # masks are drawn from jax.random's threefry2x32 with the reference's key schedule (serl_amd/jaxrng.py), and `state.rng`
        # advances as common.py:197-209 / sac.py:287-289 / drq.py:276-318 do: a learner started from the same seed consumes the
        # numbers a JAX learner consumes (integers bit-exact).  "hash": crop offsets from a numpy PCG64 stream, noise hashed on
        # the device from cfg.seed inside the kernels that use it (no noise tensors; what the bench's DataParallelLearner runs).
        self.rng_impl = "threefry"
        # how the threefry draws reach the kernels: "keys" = the kernels draw them in place from the call's keys (no noise tensors,
        # no extra launch); "tensors" = one serl_jax_fill launch per call materialises them (bit-identical results:
        # tests/test_drq_agent_gpu.py; the tensors can then be inspected -- agent._noise_bufs)
        self.noise_form = "keys"
        # state.rng as create_drq / create leave it (drq.py:69-84, sac.py:355-372): rng = PRNGKey(seed); rng, init_rng = split(rng);
        # rng, create_rng = split(rng); JaxRLTrainState.create(rng=create_rng)
        self._rng_key = J.split(J.split(J.prngkey(seed))[0])[1]
        self.last_draws = {}     # what the last update call drew from its keys (crop offsets, REDQ indices): parity tests read it
        self._noise_bufs = {}
        self._batch: Optional[DeviceBatch] = None
        self._update_serial = 0
        self.state = TrainStateView(self)
        # transparent software pipelining for batches that come from `get_iterator(..., lazy=True)`: the iterator has
        # already sampled the NEXT batch, so its gather + crop + frozen trunk run on a second stream under this update
        self.prefetch = True
        self._sched = None
        self._slot_batches = [None, None, None]   # one device batch per pipeline slot (TorchPipelineSchedule.slots)
        self._slot_crops = [None, None, None]     # the crop offsets drawn for the batch in each slot
        self._prefetched = None   # (key of the lazy batch, slot)

    # ------------------------------------------------------------------ construction
    @classmethod
    def create_drq(cls, rng, observations, actions, encoder_type: str = "small", shared_encoder: bool = True,
                   use_proprio: bool = False, critic_network_kwargs: dict = None,
                   policy_network_kwargs: dict = None, policy_kwargs: dict = None,
                   critic_ensemble_size: int = 2, critic_subsample_size: Optional[int] = None,
                   temperature_init: float = 1.0, image_keys: Iterable[str] = ("image",),
                   discount: float = 0.95, soft_target_update_rate: float = 0.005,
                   target_entropy: Optional[float] = None, backup_entropy: bool = False,
                   batch_size: int = 256, device: int = 0, learning_rate: float = 3e-4,
                   actor_optimizer_kwargs: dict = None, critic_optimizer_kwargs: dict = None,
                   temperature_optimizer_kwargs: dict = None, **kwargs):
        """drq.py:104-242.  Only the configuration the reference's examples run is built natively:
        encoder_type="resnet-pretrained", use_proprio=True, REDQ subsample 2, tanh-squashed
        exp-parameterised policy, LayerNorm+tanh 256x256 MLPs (utils/launcher.py:79-116)."""
        if encoder_type not in ("resnet-pretrained", "small"):
            # drq.py:155-167 also has "resnet" (trainable ResNet-10, never selected by an example: not built)
            raise NotImplementedError(f"Unknown encoder type: {encoder_type}")
        if not use_proprio:
            raise NotImplementedError("only use_proprio=True")
        pk = policy_kwargs or {}
        if pk.get("std_parameterization", "exp") != "exp" or not pk.get("tanh_squash_distribution", True):
            raise NotImplementedError("policy must be tanh-squashed with std_parameterization='exp'")
        for nk in (critic_network_kwargs or {}, policy_network_kwargs or {}):
            if list(nk.get("hidden_dims", [256, 256])) != [256, 256] or not nk.get("use_layer_norm", True):
                raise NotImplementedError("MLPs must be [256,256] with LayerNorm")
        seed = int(np.asarray(rng).reshape(-1)[-1]) if not isinstance(rng, int) else rng
        image_keys = tuple(image_keys)
        img = np.asarray(observations[image_keys[0]])
        H, W = img.shape[-3], img.shape[-2]
        S = int(np.asarray(observations["state"]).shape[-1])
        A = int(np.asarray(actions).shape[-1])
        if target_entropy is None:
            target_entropy = -A / 2  # drq.py:88-89
        # drq.py:35-43: each optimizer = make_optimizer(learning_rate=3e-4) unless overridden (optimizers.py:6-13)
        opts = {"actor": {"learning_rate": learning_rate, **(actor_optimizer_kwargs or {})},
                "critic": {"learning_rate": learning_rate, **(critic_optimizer_kwargs or {})},
                "temperature": {"learning_rate": learning_rate, **(temperature_optimizer_kwargs or {})}}
        core = AgentCore(device=device, n_cam=len(image_keys), H=H, W=W, state_dim=S, act_dim=A,
                         batch=batch_size, ensemble=critic_ensemble_size, discount=discount,
                         tau=soft_target_update_rate, lr=learning_rate, std_min=pk.get("std_min", 1e-5),
                         std_max=pk.get("std_max", 10.0), target_entropy=target_entropy, seed=seed,
                         optimizers={k: {"warmup_steps": 0, **v} for k, v in opts.items()}, encoder_type=encoder_type,
                         critic_subsample_size=critic_subsample_size, backup_entropy=backup_entropy)
        theta = pinit.init_theta(len(image_keys), H, W, S, A, seed=seed, temperature_init=temperature_init,
                                 ensemble=critic_ensemble_size, encoder_type=encoder_type)
        trunk = pinit.init_trunk(seed=seed) if encoder_type == "resnet-pretrained" else {}
        for sec in ("params", "target_params"):  # JaxRLTrainState.create(target_params=params)
            core.load_flat(sec, theta)
            core.load_flat(sec, trunk)
        config = dict(critic_ensemble_size=critic_ensemble_size, critic_subsample_size=critic_subsample_size,
                      discount=discount, soft_target_update_rate=soft_target_update_rate,
                      target_entropy=target_entropy, backup_entropy=backup_entropy, image_keys=image_keys)
        agent = cls(core, image_keys, config, seed)
        agent._opts = {k: {"warmup_steps": 0, **v} for k, v in opts.items()}
        return agent

    def lr_at(self, count, tx="critic"):
        """optimizers.py:14-30: warm-up -> constant, or warm-up -> cosine decay."""
        return lr_schedule(self._opts[tx], count)

    def load_trunk_params(self, pretrained: Dict[str, dict]):
        """utils/train_utils.py:69-130: patch the frozen ResNet-10 trunk from the pretrained pickle's
        tree {conv_init: {kernel}, norm_init: {scale,bias}, ResNetBlock_i: {...}} (params and target)."""
        from .flax_tree import trunk_from_flax
        flat = trunk_from_flax(pretrained)
        for sec in ("params", "target_params"):
            self.core.load_flat(sec, flat)
        return self

    def replace(self, **kw):
        """agent.replace(state=...) of the reference (e.g. `agent.replace(state=restored_ckpt)`).  The state is
        device-resident here: a flax-layout dict (or this agent's own state view) is loaded into HBM in place."""
        if kw and set(kw) - {"state"}:
            raise NotImplementedError(list(kw))
        st = kw.get("state")
        if isinstance(st, dict):
            from ..utils.checkpoint import load_state_dict
            load_state_dict(self, st)
        elif st is not None and st is not self.state:
            raise TypeError("state must be a flax-layout dict (params / target_params / opt_states / step) or agent.state")
        return self

    # ------------------------------------------------------------------ batches
    def _device_batch(self, B):
        c = self.core.cfg
        if self._batch is None or self._batch.batch != B:
            self._batch = DeviceBatch(B, c.n_cam, c.H, c.W, 3, c.state_dim, c.act_dim, c.device)
        return self._batch

    def _draw_crops(self, B, rng=None):
        """data_augmentation_fn (drq.py:244-253): one (dy, dx) in [0, 8] per frame, the SAME for every camera; obs and next_obs
        use independent keys (drq.py:279-281).  `rng`: state.rng at the entry of the call that consumes the batch (None: the
        current one) -- rng, obs_rng, next_obs_rng = split(rng, 3); offsets = batched_random_crop's (data_augmentations.py:22-36)."""
        if self.rng_impl != "threefry":
            co = self._np_rng.integers(0, 9, size=(B, 2)).astype(np.int32)
            cn = self._np_rng.integers(0, 9, size=(B, 2)).astype(np.int32)
            return co, cn
        k = J.split(self._rng_key if rng is None else rng, 3)
        return J.crop_offsets(k[1], B, 4), J.crop_offsets(k[2], B, 4)

    def _jax_keys(self, keys, n_cam, want_critic, want_actor):
        """serl_noise with KEYS only: every draw happens inside the kernel that consumes it (heads.hip: Dropout mask in the
        SpatialLearnedEmbeddings kernel, normals in the policy-head epilogue); REDQ indices are host integers."""
        c, noise = self.core.cfg, {}
        cam_keys = lambda k: np.stack([J.flax_make_rng(k, J.dropout_path(cam), 1) for cam in self.image_keys[:n_cam]])  # noqa: E731
        if want_critic and keys.n_critic:
            m = int(c.critic_subsample_size)
            noise["key_eps_next"] = np.stack(keys.k_next_action)
            if n_cam:
                noise["key_mask_next"] = np.stack([cam_keys(k) for k in keys.k_next_action])
            if m > 0:
                noise["redq_idx"] = np.stack([J.randint(k, m, 0, c.ensemble) for k in keys.k_subsample]).astype(np.int32)
                self.last_draws["redq_idx"] = noise["redq_idx"].copy()
        if want_actor and keys.has_actor_temp:
            noise["key_eps_pi"], noise["key_eps_temp"] = keys.k_sample, keys.k_temp
            if n_cam:
                noise["key_mask_obs_pi"], noise["key_mask_next_temp"] = cam_keys(keys.k_policy), cam_keys(keys.k_temp)
        return noise

    # ------------------------------------------------------------------ the reference's random stream for one call
    def _call_keys(self, n_critic, has_actor_temp, drq_aug=None, combined=False):
        """Keys of the learner call about to run, derived from state.rng (None when the device-hashed stream is selected)."""
        if self.rng_impl != "threefry":
            return None
        return J.UpdateKeys(self._rng_key, self._DRQ_AUG if drq_aug is None else drq_aug, n_critic, has_actor_temp, combined)

    def _finish_call(self, keys):
        if keys is not None:
            self._rng_key = keys.rng_out.copy()

    def _jax_noise(self, keys, B, want_critic=True, want_actor=True):
        """The noise tensors of one call, filled on the device from the call's keys (ONE launch): for critic update i over the
        minibatch rows [i*mb, (i+1)*mb) the next-action sample and the policy encoder's Dropout masks of sac.py:118-132, the
        REDQ subsample (host integers, sac.py:150-157); for the actor + temperature update the samples / masks of sac.py:197-201 and
        :224-227.  Shapes are the reference's ((mb, A) normals, one (mb, 4096) mask per camera from that camera's Dropout key)."""
        c = self.core.cfg
        dev, A = self.core.device, c.act_dim
        n_cam = c.n_cam if c.encoder_type == 0 else 0          # (the SmallEncoder path has no Dropout: pooling "avg")
        D = 512 * c.sle_features
        if self.noise_form == "keys":
            return self._jax_keys(keys, n_cam, want_critic, want_actor)
        nb = self._noise_bufs.get(B)
        if nb is None:
            nb = {k: torch.empty((B, A), dtype=torch.float32, device=dev) for k in ("eps_next", "eps_pi", "eps_temp")}
            if n_cam:
                nb.update({k: torch.empty((n_cam, B, D), dtype=torch.uint8, device=dev) for k in ("mask_next", "mask_obs_pi", "mask_next_temp")})
            self._noise_bufs = {B: nb}
        keep = 1.0 - float(c.dropout)
        jobs, noise = [], {}

        def masks(name, key, row0, rows):
            for ci, cam in enumerate(self.image_keys[:n_cam]):
                jobs.append(J.job(J.BERNOULLI_U8, J.flax_make_rng(key, J.dropout_path(cam), 1), rows * D,
                                  nb[name].data_ptr() + (ci * B + row0) * D, p=keep))
            if n_cam:
                noise[name] = nb[name]

        if want_critic and keys.n_critic:
            mb = B // keys.n_critic
            m = int(c.critic_subsample_size)
            redq = np.zeros((keys.n_critic, max(m, 1)), np.int32)
            for i in range(keys.n_critic):
                jobs.append(J.job(J.NORMAL, keys.k_next_action[i], mb * A, nb["eps_next"].data_ptr() + i * mb * A * 4))
                masks("mask_next", keys.k_next_action[i], i * mb, mb)
                if m > 0:
                    redq[i] = J.randint(keys.k_subsample[i], m, 0, c.ensemble)
            noise["eps_next"] = nb["eps_next"]
            if m > 0:
                noise["redq_idx"] = redq
                self.last_draws["redq_idx"] = redq.copy()
        if want_actor and keys.has_actor_temp:
            jobs.append(J.job(J.NORMAL, keys.k_sample, B * A, nb["eps_pi"].data_ptr()))
            masks("mask_obs_pi", keys.k_policy, 0, B)
            jobs.append(J.job(J.NORMAL, keys.k_temp, B * A, nb["eps_temp"].data_ptr()))
            masks("mask_next_temp", keys.k_temp, 0, B)
            noise["eps_pi"], noise["eps_temp"] = nb["eps_pi"], nb["eps_temp"]
        J.fill(c.device, jobs, self.core._stream())
        return noise

    def prepare(self, batch, crops=None) -> DeviceBatch:
        """sample-gather [+ concat_batches] + _unpack + random-shift crop -> DeviceBatch."""
        if isinstance(batch, DeviceBatch):
            return batch
        if isinstance(batch, LazyBatch):
            B = batch.batch_size
            out = self._device_batch(B)
            co, cn = crops if crops is not None else self._draw_crops(B)
            self.last_draws["crop_obs"], self.last_draws["crop_next"] = co, cn
            gather_crop(batch.parts, co, cn, out)
            return out
        # reference-format dict of device tensors (packed or unpacked frames)
        import ctypes as C
        obs, nobs = batch["observations"], batch["next_observations"]
        B = int(batch["rewards"].shape[0])
        out = self._device_batch(B)
        co, cn = crops if crops is not None else self._draw_crops(B)
        self.last_draws["crop_obs"], self.last_draws["crop_next"] = co, cn
        keep, ptrs = [], (C.c_void_p * len(self.image_keys))()
        for i, k in enumerate(self.image_keys):
            if k in nobs:  # unpacked: re-pack [B,2,H,W,C] (train_utils._unpack inverse)
                p = torch.cat([obs[k], nobs[k]], dim=1).contiguous()
            else:
                p = obs[k].contiguous()
            assert p.shape[1] == 2, "only num_stack == 1 (T=1) is supported"
            keep.append(p)
            ptrs[i] = p.data_ptr()
        c = self.core.cfg
        s = torch.cuda.current_stream(self.core.device).cuda_stream
        _lib.check(_lib.lib().serl_crop_packed(c.device, ptrs, c.n_cam, B, c.H, c.W, 3, co.ctypes.data,
                                               cn.ctypes.data, out.frames.data_ptr(), C.c_void_p(s)))
        out.state[0].copy_(obs["state"].reshape(B, -1))
        out.state[1].copy_(nobs["state"].reshape(B, -1))
        out.action.copy_(batch["actions"])
        out.reward.copy_(batch["rewards"])
        out.mask.copy_(batch["masks"])
        out._keep = keep
        return out

    # ------------------------------------------------------------------ pipelined acquisition of lazy batches
    @staticmethod
    def _lazy_key(batch: LazyBatch):
        return tuple((id(b), id(ix)) for b, ix in batch.parts)

    def _can_prefetch(self, batch, crops):
        # (only the FROZEN trunk can run ahead of the update: a trainable encoder depends on the parameters)
        return (self.prefetch and crops is None and isinstance(batch, LazyBatch) and self.core.cfg.n_cam > 0
                and self.core.cfg.encoder_type == 0 and batch.batch_size <= self.core.cfg.batch)

    def _slot_batch(self, slot, B):
        c = self.core.cfg
        if self._slot_batches[slot] is None or self._slot_batches[slot].batch != B:
            if self._sched is not None:     # the side / gather streams may still be using the buffers about to be freed
                self._sched.side_stream.synchronize()
                if getattr(self._sched, "gather_stream", None) is not None:
                    self._sched.gather_stream.synchronize()
            self._slot_batches[slot] = DeviceBatch(B, c.n_cam, c.H, c.W, 3, c.state_dim, c.act_dim, c.device)
        return self._slot_batches[slot]

    def _produce(self, batch: LazyBatch, slot, db, rng=None):
        """gather + crop on the schedule's gather stream, the frozen trunk on the side stream.  (Round 2 gathered on the CALLER's
        stream at the moment the pass needed it and made the side stream wait: a dependency that crosses streams and actually
        blocks costs 60-100 us on this stack, at every pass.  Since round 5 the gather of the NEXT batch runs on a third stream while
        the previous pass is still computing -- the side stream's wait is satisfied long before it is reached -- and the pass starts
        at conv_init: parallel.py TorchPipelineSchedule.gathered.  An actor-side insert that overwrites a slot waits ON THE COPY
        STREAM for the gathers in flight, never on the host.)"""
        sch = self._sched
        co, cn = self._draw_crops(db.batch, rng)      # rng: state.rng at the entry of the call that will consume this batch
        self._slot_crops[slot] = (co, cn)
        sch.wait_consumed(slot)     # host side: the update that used this slot ended two passes ago
        with sch.side():
            sch.gathered(slot, lambda: gather_crop(batch.parts, co, cn, db))
            self.core.encode_slot(db, slot)
        sch.produced(slot)

    def _acquire(self, batch: LazyBatch, next_rng=None):
        """-> (slot, DeviceBatch) with the frozen-trunk features of `batch` ready in `slot` (prefetched during the
        previous update, or encoded now), and starts the same work for the iterator's next batch on the side stream.
        `next_rng`: state.rng after the call in progress = at the entry of the call that consumes the next batch (the key
        schedule is pure host arithmetic, so the next batch's crop keys are known before this update has run)."""
        from ..parallel import TorchPipelineSchedule
        if self._sched is None:
            self._sched = TorchPipelineSchedule(self.core.device, prioritise_update=False)
            if self.core.cfg.batch >= 128:
                self.core.set_chain_budget(256)      # the chain now runs beside the trunk pass (see parallel.py)
        sch, B = self._sched, batch.batch_size
        key = self._lazy_key(batch)
        hit = self._prefetched is not None and self._prefetched[0] == key
        if hit and self.rng_impl == "threefry" and self._prefetched[3] is not None \
                and not np.array_equal(self._prefetched[3], self._rng_key):
            # The slot's crop offsets were drawn from the state.rng this call was EXPECTED to enter with.  state.rng moved in
            # between (state.replace(rng=...) after a checkpoint restore / reseed, or an update on a non-lazy batch): the
            # reference would crop with the current key, so the slot is stale -- produce it again.
            hit = False
        if hit:
            slot = self._prefetched[1]
            sch.wait_produced(slot)
            db = self._slot_batches[slot]
        else:
            # not prefetched (first call, or the caller skipped a batch).  All trunk work goes through the side stream,
            # in order: the feature slots share one trunk workspace.
            slot = 0 if self._prefetched is None else (self._prefetched[1] + 1) % sch.slots
            db = self._slot_batch(slot, B)
            self._produce(batch, slot, db)
            sch.wait_produced(slot)
        self.last_draws["crop_obs"], self.last_draws["crop_next"] = self._slot_crops[slot]
        self._prefetched = None
        nxt = batch.peek_next() if batch.peek_next is not None else None
        if nxt is not None and nxt.batch_size == B:
            s2 = (slot + 1) % sch.slots
            self._produce(nxt, s2, self._slot_batch(s2, B), next_rng)
            # (the parts are kept alive with the key: a freed index array's address could otherwise be reused by a
            # later sample and false-match)
            self._prefetched = (self._lazy_key(nxt), s2, list(nxt.parts), None if next_rng is None else np.array(next_rng, np.uint32))
        self.core.select_slot(slot)
        return slot, db

    def _sync_side_stream(self):
        """The non-pipelined paths run the trunk on the caller's stream and share the trunk workspace with a prefetched
        encode_slot that may still be in flight on the side stream: order behind it and drop the prefetch (it would
        otherwise be consumed after its workspace was overwritten)."""
        if self._sched is not None:
            torch.cuda.current_stream(self.core.device).wait_stream(self._sched.side_stream)
            self._prefetched = None

    # ------------------------------------------------------------------ updates
    def update_critics(self, batch, *, pmap_axis: Optional[str] = None, noise=None, crops=None):
        """drq.py:296-328.  `noise` / `crops` inject explicit draws (parity tests); None = drawn from state.rng."""
        keys = self._call_keys(1, False)
        if self._can_prefetch(batch, crops):
            slot, db = self._acquire(batch, None if keys is None else keys.rng_out)
            if noise is None and keys is not None:
                noise = self._jax_noise(keys, db.batch)
            self.core.begin_update()
            self.core.critic_grads(0, db.batch, db.batch, noise)
            self.core.apply(APPLY_CRITIC)
            self._sched.consumed(slot)
        else:
            db = self.prepare(batch, crops)
            self._sync_side_stream()
            if noise is None and keys is not None:
                noise = self._jax_noise(keys, db.batch)
            self.core.update_critics(db, noise)
        self._finish_call(keys)
        self._update_serial += 1
        return self, PendingInfo(self, "critics", self._update_serial)

    def update_high_utd(self, batch, *, utd_ratio: int, pmap_axis: Optional[str] = None, noise=None, crops=None):
        """drq.py:255-294 -> sac.py:544-596."""
        keys = self._call_keys(utd_ratio, True)
        if self._can_prefetch(batch, crops):
            B = batch.batch_size
            assert B % utd_ratio == 0, f"Batch size {B} must be divisible by UTD ratio {utd_ratio}"  # sac.py:561-563
            slot, db = self._acquire(batch, None if keys is None else keys.rng_out)
            if noise is None and keys is not None:
                noise = self._jax_noise(keys, B)
            mb = B // utd_ratio
            self.core.begin_update()
            for i in range(utd_ratio):
                self.core.critic_grads(i * mb, mb, mb, noise, i)
                self.core.apply(APPLY_CRITIC, 1.0 / utd_ratio)
            self.core.actor_grads(B, noise)
            self.core.apply(APPLY_ACTOR_TEMP)
            self._sched.consumed(slot)
        else:
            db = self.prepare(batch, crops)
            assert db.batch % utd_ratio == 0, \
                f"Batch size {db.batch} must be divisible by UTD ratio {utd_ratio}"  # sac.py:561-563
            self._sync_side_stream()
            if noise is None and keys is not None:
                noise = self._jax_noise(keys, db.batch)
            self.core.update_high_utd(db, utd_ratio, noise)
        self._finish_call(keys)
        self._update_serial += 1
        return self, PendingInfo(self, "high_utd", self._update_serial)

    def update(self, batch, *, pmap_axis: str = None,
               networks_to_update: FrozenSet[str] = frozenset({"actor", "critic", "temperature"}), noise=None):
        """sac.py:243-299 on an already augmented batch: every loss in `networks_to_update` is evaluated at the same
        parameters, all three optimizers step once (zero gradients for the others), target EMA iff "critic" is in the
        set.  Any non-empty subset works, the default (all three) included."""
        loss_keys = {"actor", "critic", "temperature"}
        assert set(networks_to_update).issubset(loss_keys), f"Invalid gradient steps: {networks_to_update}"
        assert len(networks_to_update) > 0, "networks_to_update is empty"
        if isinstance(batch, DeviceBatch):
            db = batch
        else:  # identity crop (offset 4 = centre of the 9 shifts): the batch is taken as already augmented
            n = batch.batch_size if isinstance(batch, LazyBatch) else int(batch["rewards"].shape[0])
            db = self.prepare(batch, crops=(np.full((n, 2), 4, np.int32),) * 2)
        self._sync_side_stream()
        # SACAgent.update (sac.py:243-299): no augmentation split; every selected loss takes its key from ONE 4-way split
        nets = set(networks_to_update)
        crit, act = "critic" in nets, bool(nets & {"actor", "temperature"})
        keys = self._call_keys(1 if crit else 0, act, drq_aug=False, combined=crit and act)
        if noise is None and keys is not None:
            noise = self._jax_noise(keys, db.batch, want_critic=crit, want_actor=act)
        self.core.update(db, tuple(networks_to_update), noise)
        self._finish_call(keys)
        self._update_serial += 1
        return self, PendingInfo(self, frozenset(networks_to_update), self._update_serial)

    # ------------------------------------------------------------------ acting
    def sample_actions(self, observations, *, seed=None, argmax: bool = False, **kwargs):
        """sac.py:301-320: policy forward with train=False; sample (external seed) or mode."""
        if argmax:
            assert seed is None, "Cannot specify seed when sampling deterministically"
        c = self.core.cfg
        st = np.asarray(observations["state"], np.float32)
        batched = st.ndim == 3
        n = st.shape[0] if batched else 1
        frames = np.stack([np.asarray(observations[k], np.uint8).reshape(n, c.H, c.W, 3) for k in self.image_keys])
        self._sync_side_stream()
        f = torch.from_numpy(frames).to(self.core.device)
        s = torch.from_numpy(st.reshape(n, -1)).to(self.core.device)
        eps = None if argmax else self._action_noise(seed, n)
        a = self.core.sample_actions(f, s, eps).cpu().numpy()
        return a if batched else a[0]

    def _action_noise(self, seed, n):
        """dist.sample(seed=seed) of sac.py:316-320: distrax draws jax.random.normal(seed, (1,) + batch + (A,))."""
        assert seed is not None, "Must specify rng when sampling"
        c = self.core.cfg
        eps = torch.empty((n, c.act_dim), dtype=torch.float32, device=self.core.device)
        if self.rng_impl == "threefry" and J.is_key(seed):
            J.fill(c.device, [J.job(J.NORMAL, seed, n * c.act_dim, eps.data_ptr())], self.core._stream())
            return eps
        g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(np.asarray(seed).reshape(-1).tolist())))
        return torch.from_numpy(g.standard_normal((n, c.act_dim)).astype(np.float32)).to(self.core.device)
