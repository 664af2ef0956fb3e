"""State-only SAC agent with the reference's surface (serl_launcher/agents/continuous/sac.py):
  SACAgent.create_states (:486-542), update_high_utd (:544-596), update (:243-299), sample_actions (:301-320),
  agent.state / agent.config -- BASELINE.json configs[0] `async_sac_state_sim`.

No encoder: observations are flat vectors, the ensemblized Critic has one Dense(1) head per member, actor and
critic optimizers warm up over `warmup_steps` (sac.py:333-343 defaults: 2000) and the temperature optimizer does not.
All numerics run in the same HIP kernels as the DrQ agent (libserl_mi355.so with n_cam = 0); no CPU fallback.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ..data.data_store import LazyBatch, gather_crop
from ..utils import init as pinit
from .batch import DeviceBatch
from .core import AgentCore
from .drq import DrQAgent


class SACAgent(DrQAgent):
    _DRQ_AUG = False    # SACAgent.update_high_utd (sac.py:544-596) has no augmentation split

    @classmethod
    def create_states(cls, rng, observations, actions, critic_network_kwargs: dict = None,
                      critic_ensemble_size: int = 2, critic_subsample_size: Optional[int] = None,
                      policy_network_kwargs: dict = None, policy_kwargs: dict = None, temperature_init: float = 1.0,
                      discount: float = 0.95, soft_target_update_rate: float = 0.005,
                      target_entropy: Optional[float] = None, backup_entropy: bool = False,
                      actor_optimizer_kwargs: dict = None, critic_optimizer_kwargs: dict = None,
                      temperature_optimizer_kwargs: dict = None, batch_size: int = 256, device: int = 0, **kwargs):
        """sac.py:486-542 -> create (:323-400).  Built natively: the configuration of utils/launcher.py:50-76
        (REDQ subsample 2, tanh-squashed exp-parameterised policy, LayerNorm+tanh 256x256 MLPs)."""
        pk = policy_kwargs or {}
        if pk.get("std_parameterization", "exp") != "exp" or not pk.get("tanh_squash_distribution", True):
            raise NotImplementedError("policy must be tanh-squashed with std_parameterization='exp'")
        for nk in (critic_network_kwargs or {}, policy_network_kwargs or {}):
            if list(nk.get("hidden_dims", [256, 256])) != [256, 256] or not nk.get("use_layer_norm", True):
                raise NotImplementedError("MLPs must be [256,256] with LayerNorm")
        ao = {"learning_rate": 3e-4, "warmup_steps": 2000, **(actor_optimizer_kwargs or {})}     # sac.py:333-336
        co = {"learning_rate": 3e-4, "warmup_steps": 2000, **(critic_optimizer_kwargs or {})}    # sac.py:337-340
        to = {"learning_rate": 3e-4, **(temperature_optimizer_kwargs or {})}                     # sac.py:341-343
        seed = int(np.asarray(rng).reshape(-1)[-1]) if not isinstance(rng, int) else rng
        S = int(np.asarray(observations).shape[-1])
        A = int(np.asarray(actions).shape[-1])
        if target_entropy is None:
            target_entropy = -A / 2   # sac.py:387-388
        core = AgentCore(device=device, n_cam=0, H=0, W=0, state_dim=S, act_dim=A, batch=batch_size,
                         ensemble=critic_ensemble_size, discount=discount, tau=soft_target_update_rate,
                         lr=ao["learning_rate"], warmup_steps=int(ao["warmup_steps"]),
                         temp_warmup_steps=int(to.get("warmup_steps", 0)), std_min=pk.get("std_min", 1e-5),
                         std_max=pk.get("std_max", 10.0), target_entropy=target_entropy, seed=seed,
                         optimizers={"actor": ao, "critic": co, "temperature": {"warmup_steps": 0, **to}},
                         critic_subsample_size=critic_subsample_size, backup_entropy=backup_entropy)
        theta = pinit.init_theta(0, 0, 0, S, A, seed=seed, temperature_init=temperature_init, ensemble=critic_ensemble_size)
        for sec in ("params", "target_params"):
            core.load_flat(sec, theta)
        config = dict(critic_ensemble_size=critic_ensemble_size, critic_subsample_size=critic_subsample_size,
                      discount=discount, soft_target_update_rate=soft_target_update_rate,
                      target_entropy=target_entropy, backup_entropy=backup_entropy)
        agent = cls(core, (), config, seed)
        agent._opts = {"actor": ao, "critic": co, "temperature": {"warmup_steps": 0, **to}}
        return agent

    # ------------------------------------------------------------------ batches (flat observations, no augmentation)
    def prepare(self, batch, crops=None) -> DeviceBatch:
        if isinstance(batch, DeviceBatch):
            return batch
        if isinstance(batch, LazyBatch):
            out = self._device_batch(batch.batch_size)
            gather_crop(batch.parts, None, None, out)
            return out
        B = int(batch["rewards"].shape[0])
        out = self._device_batch(B)
        dev = self.core.device

        def t(x):
            return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x), device=dev)
        out.state[0].copy_(t(batch["observations"]).reshape(B, -1))
        out.state[1].copy_(t(batch["next_observations"]).reshape(B, -1))
        out.action.copy_(t(batch["actions"]))
        out.reward.copy_(t(batch["rewards"]))
        out.mask.copy_(t(batch["masks"]))
        return out

    def update_critics(self, batch, **kw):
        raise AttributeError("SACAgent has no update_critics (drq.py:296-328 is DrQ-only); use update(networks_to_update={'critic'})")

    def sample_actions(self, observations, *, seed=None, argmax: bool = False, **kwargs):
        """sac.py:301-320."""
        if argmax:
            assert seed is None, "Cannot specify seed when sampling deterministically"
        c = self.core.cfg
        st = np.asarray(observations, np.float32)
        batched = st.ndim == 2
        n = st.shape[0] if batched else 1
        s = torch.from_numpy(st.reshape(n, -1)).to(self.core.device)
        eps = None if argmax else self._action_noise(seed, n)
        a = self.core.sample_actions(None, s, eps).cpu().numpy()
        return a if batched else a[0]
