"""Helpers for the agent parity tests: build the same agent in the oracle and in the HIP library."""
import numpy as np
import torch

from oracle import drq_oracle as O


def product_name(name, image_keys):
    parts = name.split("/")
    if parts[0] == "enc" and parts[1] in image_keys:
        parts[1] = str(list(image_keys).index(parts[1]))
    return "/".join(parts)


def make_pair(cfg: O.Config, B, dtype=torch.float64, seed=42, agent_seed=0, trunk_mode=None):
    """-> (oracle TrainState, AgentCore) holding identical parameters."""
    from serl_amd.agents.core import AgentCore
    trunk, theta = O.init_params(cfg, seed)
    st = O.TrainState(cfg, trunk, theta, dtype)
    core = AgentCore(encoder_type=cfg.encoder_type, n_cam=cfg.n_cam, H=cfg.H, W=cfg.W, state_dim=cfg.S, act_dim=cfg.A, batch=B,
                     ensemble=cfg.ensemble, discount=cfg.discount, tau=cfg.tau, lr=cfg.lr,
                     warmup_steps=cfg.warmup, dropout=cfg.dropout, std_min=cfg.std_min, std_max=cfg.std_max,
                     target_entropy=cfg.target_entropy, seed=agent_seed,
                     temp_warmup_steps=-1 if cfg.temp_warmup is None else cfg.temp_warmup)
    if trunk_mode is not None:
        core.set_trunk_mode(trunk_mode)
    for sec in ("params", "target_params"):
        core.load_flat(sec, trunk)
        core.load_flat(sec, {product_name(k, cfg.image_keys): v for k, v in theta.items()})
    return st, core


_FRAMES = {}


def synth_batch(cfg: O.Config, B, seed=0, frames_seed=None):
    """frames_seed: take the (obs, next) frames from that seed instead of `seed` -- full-shape tests share ONE frame set so that the
    fp64 oracle's frozen-trunk pass over it (O.features, memoised) is paid once per session; everything else follows `seed`."""
    rng = np.random.default_rng(seed)
    if frames_seed is not None:
        fk = (cfg.image_keys, B, cfg.H, cfg.W, frames_seed)
        if fk not in _FRAMES:
            fr = np.random.default_rng(frames_seed)
            _FRAMES[fk] = ({k: fr.integers(0, 256, (B, cfg.H, cfg.W, 3), dtype=np.uint8) for k in cfg.image_keys},
                           {k: fr.integers(0, 256, (B, cfg.H, cfg.W, 3), dtype=np.uint8) for k in cfg.image_keys})
        for k in cfg.image_keys:       # (keep `rng` where it would be without the shared frames: scalar fields unchanged)
            rng.integers(0, 256, (1,), dtype=np.uint8)
        obs, nxt = _FRAMES[fk]
    return {
        "obs": {k: v.copy() for k, v in obs.items()} if frames_seed is not None else {k: rng.integers(0, 256, (B, cfg.H, cfg.W, 3), dtype=np.uint8) for k in cfg.image_keys},
        "next": {k: v.copy() for k, v in nxt.items()} if frames_seed is not None else {k: rng.integers(0, 256, (B, cfg.H, cfg.W, 3), dtype=np.uint8) for k in cfg.image_keys},
        "state": rng.standard_normal((B, cfg.S)).astype(np.float32),
        "next_state": rng.standard_normal((B, cfg.S)).astype(np.float32),
        "action": rng.uniform(-1, 1, (B, cfg.A)).astype(np.float32),
        "reward": (rng.random(B) < 0.3).astype(np.float32),
        "mask": (rng.random(B) < 0.9).astype(np.float32),
    }


def batch_to_torch(b, dtype):
    out = {"obs": {k: torch.tensor(v) for k, v in b["obs"].items()},
           "next": {k: torch.tensor(v) for k, v in b["next"].items()}}
    for k in ("state", "next_state", "action", "reward", "mask"):
        out[k] = torch.tensor(b[k], dtype=dtype)
    return out


def batch_to_device(cfg, b):
    from serl_amd.agents.batch import DeviceBatch
    B = b["reward"].shape[0]
    db = DeviceBatch(B, cfg.n_cam, cfg.H, cfg.W, 3, cfg.S, cfg.A, 0)
    for c, k in enumerate(cfg.image_keys):
        db.frames[0, c].copy_(torch.tensor(b["obs"][k]))
        db.frames[1, c].copy_(torch.tensor(b["next"][k]))
    db.state[0].copy_(torch.tensor(b["state"]))
    db.state[1].copy_(torch.tensor(b["next_state"]))
    db.action.copy_(torch.tensor(b["action"]))
    db.reward.copy_(torch.tensor(b["reward"]))
    db.mask.copy_(torch.tensor(b["mask"]))
    db.done.zero_()
    return db


def noise_to_device(cfg, noise):
    out = {}
    for k, v in noise.items():
        if k.startswith("eps"):
            out[k] = torch.tensor(v, device="cuda")
        elif k.startswith("mask"):
            if cfg.image_keys and v:
                out[k] = torch.tensor(np.stack([v[c] for c in cfg.image_keys]), device="cuda")
        elif k == "redq_idx":
            out[k] = np.asarray(v, np.int32)
    return out


def flat_theta(cfg, d, order):
    """dict (oracle names) -> flat vector in the product's arena order."""
    return np.concatenate([np.asarray(d[k].detach().cpu().numpy() if hasattr(d[k], "detach") else d[k],
                                      dtype=np.float64).reshape(-1) for k in order])


def leaf_slices(cfg):
    sl, off = {}, 0
    for k, shp in O.trainable_param_shapes(cfg).items():
        n = int(np.prod(shp)) if len(shp) else 1
        sl[k] = (off, off + n)
        off += n
    return sl, off


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def elem_rel_err(a, b, floor=1e-3):
    """worst |a-b|/|b| over the elements with |b| >= floor * max|b| (errors in small elements are not normalised away)."""
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    m = np.abs(b) >= floor * (np.max(np.abs(b)) + 1e-300)
    if not m.any():
        return 0.0
    return float(np.max(np.abs(a[m] - b[m]) / np.abs(b[m])))
