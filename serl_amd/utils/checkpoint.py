"""Checkpoint save / restore of `agent.state` (next-row N1).

The reference calls `flax.training.checkpoints.save_checkpoint(path, agent.state, step=, keep=)`
(examples/async_drq_sim/async_drq_sim.py:303-307), which writes `<path>/checkpoint_<step>` containing
`flax.serialization.to_bytes(state)`: a msgpack map of the state's pytree fields (`step, params,
target_params, opt_states, rng`; common/common.py:108-114) in which every ndarray is a msgpack ExtType(1)
holding msgpack((shape, dtype.name, raw bytes)).  flax is not installable here, so this writer/reader
restates that published format; the TREE inside (parameter paths, the optax InjectHyperparamsState / chain /
ScaleByAdamState nesting of `opt_states`) is checked against the state the reference's own code builds under the
stand-ins of oracle/jaxshim (tests/test_reference_update.py).  `restore_checkpoint` below reads such files back into
the HIP agent (params, target_params, Adam moments, step).
"""
from __future__ import annotations

import os
import re
from typing import Optional

import msgpack
import numpy as np

from ..agents.core import TX_NAMES
from ..agents.flax_tree import theta_paths, trunk_owner, _trunk_paths

_EXT_NDARRAY = 1


def _pack_default(x):
    if isinstance(x, (np.ndarray, np.generic)):
        a = np.asarray(x)
        return msgpack.ExtType(_EXT_NDARRAY, msgpack.packb((list(a.shape), a.dtype.name, a.tobytes("C")), use_bin_type=True))
    raise TypeError(f"cannot serialise {type(x)}")


def _unpack_ext(code, data):
    if code == _EXT_NDARRAY:
        shape, dtype, buf = msgpack.unpackb(data, raw=False)
        return np.frombuffer(buf, dtype=np.dtype(dtype)).reshape(shape).copy()
    return msgpack.ExtType(code, data)


def state_dict(agent) -> dict:
    st = agent.state
    return {"step": np.int32(st.step), "params": st.params, "target_params": st.target_params,
            "opt_states": st.opt_states, "rng": st.rng}


def save_checkpoint(ckpt_dir: str, agent, step: int, prefix: str = "checkpoint_", keep: int = 1, overwrite: bool = False) -> str:
    os.makedirs(ckpt_dir, exist_ok=True)
    path = os.path.join(ckpt_dir, f"{prefix}{step}")
    if os.path.exists(path) and not overwrite:
        raise ValueError(f"checkpoint {path} exists (overwrite=False)")
    blob = msgpack.packb(state_dict(agent), default=_pack_default, strict_types=True, use_bin_type=True)
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(blob)
    os.replace(tmp, path)
    # keep the `keep` most recent checkpoints (flax semantics)
    steps = sorted(int(m.group(1)) for m in (re.fullmatch(re.escape(prefix) + r"(\d+)", n) for n in os.listdir(ckpt_dir)) if m)
    for s in steps[:-keep] if keep > 0 else []:
        os.remove(os.path.join(ckpt_dir, f"{prefix}{s}"))
    return path


def latest_checkpoint(ckpt_dir: str, prefix: str = "checkpoint_") -> Optional[str]:
    if not os.path.isdir(ckpt_dir):
        return None
    steps = [int(m.group(1)) for m in (re.fullmatch(re.escape(prefix) + r"(\d+)", n) for n in os.listdir(ckpt_dir)) if m]
    return os.path.join(ckpt_dir, f"{prefix}{max(steps)}") if steps else None


def _walk(tree, path):
    for p in path:
        tree = tree[p]
    return tree


def restore_checkpoint(ckpt_dir_or_file: str, agent, step: Optional[int] = None, prefix: str = "checkpoint_"):
    """Loads params / target_params / Adam moments / step back into the agent's HBM arena."""
    path = ckpt_dir_or_file
    if os.path.isdir(path):
        path = os.path.join(path, f"{prefix}{step}") if step is not None else latest_checkpoint(path, prefix)
        if path is None:
            return agent  # flax returns the target unchanged when there is nothing to restore
    with open(path, "rb") as f:
        sd = msgpack.unpackb(f.read(), ext_hook=_unpack_ext, raw=False, strict_map_key=False)
    return load_state_dict(agent, sd)


def read_checkpoint_tree(ckpt_dir_or_file: str, step: Optional[int] = None, prefix: str = "checkpoint_") -> dict:
    """The state dict stored in a flax checkpoint file (or the latest / the given step of a directory of them)."""
    path = ckpt_dir_or_file
    if os.path.isdir(path):
        path = os.path.join(path, f"{prefix}{step}") if step is not None else latest_checkpoint(path, prefix)
        if path is None:
            raise FileNotFoundError(f"no {prefix}<step> file in {ckpt_dir_or_file}")
    with open(path, "rb") as f:
        return msgpack.unpackb(f.read(), ext_hook=_unpack_ext, raw=False, strict_map_key=False)


def write_checkpoint_tree(ckpt_dir: str, tree: dict, step: int, prefix: str = "checkpoint_") -> str:
    """Writes `tree` (nested dicts of ndarrays / scalars) as <ckpt_dir>/<prefix><step> in the same msgpack layout."""
    os.makedirs(ckpt_dir, exist_ok=True)
    path = os.path.join(ckpt_dir, f"{prefix}{step}")
    with open(path, "wb") as f:
        f.write(msgpack.packb(tree, default=_pack_default, strict_types=True, use_bin_type=True))
    return path


def _find_adam_state(node):
    """The ScaleByAdamState {count, mu, nu} inside an InjectHyperparamsState / chain state dict (any nesting), or the
    node itself for the flat {count, mu, nu} layout written by earlier versions of this module."""
    if isinstance(node, dict):
        if "mu" in node and "nu" in node:
            return node
        for v in node.values():
            r = _find_adam_state(v)
            if r is not None:
                return r
    return None


def load_state_dict(agent, sd: dict):
    """Loads any of {params, target_params, opt_states, step} (flax-layout trees, e.g. a restored checkpoint or
    `agent.state.replace(...)` arguments) into the agent's HBM arena."""
    core, keys = agent.core, agent.image_keys
    etype = "small" if core.cfg.encoder_type == 1 else "resnet-pretrained"
    tp = theta_paths(keys, encoder_type=etype)
    trunk = _trunk_paths() if (keys and etype != "small") else {}   # state-only / SmallEncoder agents have no frozen trunk
    for section in ("params", "target_params"):
        tree = sd.get(section)
        if tree is None:
            continue
        for leaf, paths in tp.items():
            core.set(section, leaf, _walk(tree, paths[0]))
        if trunk:
            # Where flax puts the ONE shared frozen trunk is derived from its adoption rule (first camera in sorted-key
            # order) and unverified against a real flax install: accept it under any camera, like
            # reward_classifier.load_params and the reference's own `if "pretrained_encoder" in ...` guard do
            enc = _walk(tree, ("modules_actor", "encoder"))
            owners = [k for k in [trunk_owner(keys)] + sorted(keys) if "pretrained_encoder" in enc.get(f"encoder_{k}", {})]
            if not owners:
                raise KeyError(f"'{section}' holds no pretrained_encoder under any of encoder_{{{', '.join(sorted(keys))}}}")
            root = enc[f"encoder_{owners[0]}"]["pretrained_encoder"]
            for leaf, sub in trunk.items():
                core.set(section, leaf, _walk(root, sub))
    if sd.get("opt_states") is not None:
        for tx in TX_NAMES:
            adam = _find_adam_state(sd["opt_states"][tx])
            if adam is None:
                raise KeyError(f"opt_states['{tx}'] holds no ScaleByAdamState (mu / nu)")
            for mom in ("mu", "nu"):
                tree = adam[mom]
                for leaf, paths in tp.items():
                    # leaves outside the optimizer's support are exact zeros; the C ABI accepts (and checks) them
                    core.set(f"opt/{tx}/{mom}", leaf, np.asarray(_walk(tree, paths[0]), np.float32))
    if sd.get("step") is not None:
        core.step = int(np.asarray(sd["step"]))
    return agent
