"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (serl_amd/).

Loads the reference's NumPy replay buffer *unmodified* from /root/reference under
import stubs, so that it can pin the restatement in
``oracle/replay_oracle.py`` and generate the golden fixtures under ``tests/golden/``.

/root/reference only exists in the build container, not on the GPU box: callers must
check ``reference_available()`` and skip otherwise.  Nothing here copies reference
source; the modules are imported from where they lie.

Stubbed third-party modules (absent in this image): gym, flax.core.frozen_dict, jax.
The stubs provide exactly the attributes the reference replay code touches:
  * serl_launcher/data/dataset.py:9  ``from gym.utils import seeding`` -> np_random
  * serl_launcher/data/replay_buffer.py:4-5 ``import gym``/``import jax``
  * serl_launcher/data/memory_efficient_replay_buffer.py:8-9 frozen_dict, gym.spaces.Box
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference/serl_launcher"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "serl_launcher", "data"))


class _Space:
    pass


class Box(_Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()


class Dict(_Space):
    def __init__(self, spaces):
        # gym 0.26 sorts plain-dict keys alphabetically
        self.spaces = {k: spaces[k] for k in sorted(spaces.keys())}

    def keys(self):
        return self.spaces.keys()

    def __getitem__(self, k):
        return self.spaces[k]


class FrozenDict(dict):
    def unfreeze(self):
        return {k: (v.unfreeze() if isinstance(v, FrozenDict) else v) for k, v in self.items()}

    def copy(self, add_or_replace=None):  # flax API used by train_utils._unpack
        d = FrozenDict(self)
        if add_or_replace:
            d.update(add_or_replace)
        return d


def _freeze(d):
    return FrozenDict({k: (_freeze(v) if isinstance(v, dict) else v) for k, v in d.items()})


def _np_random(seed=None):
    # gym.utils.seeding.np_random (gym 0.26): Generator(PCG64(SeedSequence(seed)))
    ss = np.random.SeedSequence(seed)
    return np.random.Generator(np.random.PCG64(ss)), ss.entropy


def install_stubs():
    if "gym" in sys.modules and getattr(sys.modules["gym"], "_serl_stub", False):
        return
    gym = types.ModuleType("gym")
    gym._serl_stub = True
    spaces = types.ModuleType("gym.spaces")
    spaces.Box, spaces.Dict, spaces.Space = Box, Dict, _Space
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    seeding.np_random = _np_random
    utils.seeding = seeding
    gym.spaces, gym.utils, gym.Space = spaces, utils, _Space
    flax = types.ModuleType("flax")
    core = types.ModuleType("flax.core")
    fd = types.ModuleType("flax.core.frozen_dict")
    fd.freeze, fd.FrozenDict = _freeze, FrozenDict
    core.frozen_dict = fd
    flax.core = core
    jax = types.ModuleType("jax")
    jax.device_put = lambda x, device=None: x
    jnp = types.ModuleType("jax.numpy")
    jnp.concatenate = np.concatenate
    jax.numpy = jnp
    for name, mod in {
        "gym": gym, "gym.spaces": spaces, "gym.utils": utils, "gym.utils.seeding": seeding,
        "flax": flax, "flax.core": core, "flax.core.frozen_dict": fd,
        "jax": jax, "jax.numpy": jnp,
    }.items():
        sys.modules.setdefault(name, mod)


def load_reference_buffer_cls():
    """Returns the reference's MemoryEfficientReplayBuffer class (unmodified)."""
    if not reference_available():
        raise RuntimeError("/root/reference not present (GPU box?) -- use the golden fixtures")
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from serl_launcher.data.memory_efficient_replay_buffer import MemoryEfficientReplayBuffer
    return MemoryEfficientReplayBuffer


def make_spaces(image_keys, H, W, C, T, S, A):
    obs = {"state": Box(-np.inf, np.inf, (T, S), np.float32)}
    for k in image_keys:
        obs[k] = Box(0, 255, (T, H, W, C), np.uint8)
    return Dict(obs), Box(-1.0, 1.0, (A,), np.float32)


def load_reference_plain_buffer_cls():
    """Returns the reference's plain ReplayBuffer class (replay_buffer.py:40-75, unmodified)."""
    if not reference_available():
        raise RuntimeError("/root/reference not present (GPU box?) -- use the golden fixtures")
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from serl_launcher.data.replay_buffer import ReplayBuffer
    return ReplayBuffer
