"""Factories with the reference's names and kwargs (serl_launcher/utils/launcher.py:79-116,201-271)."""
from __future__ import annotations

from typing import Optional

from ..agents.drq import DrQAgent
from ..agents.sac import SACAgent
from ..data.data_store import MemoryEfficientReplayBufferDataStore, ReplayBufferDataStore
from ..transport.endpoint import make_trainer_config  # noqa: F401  (launcher.py:171-177: the reference exports it from here)


def make_drq_agent(seed, sample_obs, sample_action, image_keys=("image",), encoder_type="small",
                   discount=0.96, batch_size=256, device=0, **create_kwargs):
    """launcher.py:79-116 (hyper-parameters copied from there).  `create_kwargs` (e.g. critic_optimizer_kwargs) are
    passed on to DrQAgent.create_drq, whose signature has them in the reference too (drq.py:34-43)."""
    kw = dict(backup_entropy=False, critic_ensemble_size=10, critic_subsample_size=2)   # launcher.py:111-114
    kw.update(create_kwargs)
    return DrQAgent.create_drq(
        seed, sample_obs, sample_action, encoder_type=encoder_type, use_proprio=True, image_keys=image_keys,
        policy_kwargs={"tanh_squash_distribution": True, "std_parameterization": "exp", "std_min": 1e-5, "std_max": 5},
        critic_network_kwargs={"activations": "tanh", "use_layer_norm": True, "hidden_dims": [256, 256]},
        policy_network_kwargs={"activations": "tanh", "use_layer_norm": True, "hidden_dims": [256, 256]},
        temperature_init=1e-2, discount=discount, batch_size=batch_size, device=device, **kw)


def make_sac_agent(seed, sample_obs, sample_action, discount=0.99, batch_size=256, device=0, **create_kwargs):
    """launcher.py:50-76 (hyper-parameters copied from there; optimizer defaults from sac.py:333-343)."""
    kw = dict(backup_entropy=False, critic_ensemble_size=10, critic_subsample_size=2)   # launcher.py:70-73
    kw.update(create_kwargs)
    return SACAgent.create_states(
        seed, sample_obs, sample_action,
        policy_kwargs={"tanh_squash_distribution": True, "std_parameterization": "exp", "std_min": 1e-5, "std_max": 5},
        critic_network_kwargs={"activations": "tanh", "use_layer_norm": True, "hidden_dims": [256, 256]},
        policy_network_kwargs={"activations": "tanh", "use_layer_norm": True, "hidden_dims": [256, 256]},
        temperature_init=1e-2, discount=discount, batch_size=batch_size, device=device, **kw)


def make_replay_buffer(env, capacity: int = 1000000, rlds_logger_path: Optional[str] = None,
                       type: str = "replay_buffer", image_keys: list = [], preload_rlds_path: Optional[str] = None,
                       preload_data_transform: Optional[callable] = None, device: int = 0):
    """launcher.py:201-271.  Only the memory-efficient pixel buffer lives in HBM."""
    if rlds_logger_path or preload_rlds_path:
        raise NotImplementedError("RLDS logging / preload are outside the MI355X hot path")
    if type == "replay_buffer":   # launcher.py:236-243
        return ReplayBufferDataStore(env.observation_space, env.action_space, capacity=capacity, device=device)
    if type != "memory_efficient_replay_buffer":
        raise ValueError(f"Unsupported replay_buffer_type: {type}")
    return MemoryEfficientReplayBufferDataStore(env.observation_space, env.action_space, capacity=capacity,
                                                image_keys=image_keys, device=device)
