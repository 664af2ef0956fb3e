"""TEST INFRASTRUCTURE ONLY.  Compact on-disk form of a reference run (oracle/ref_update_runner.run_reference) for
tests/golden/update_*.npz: inputs are regenerated from their seed (a CRC guards against drift), the recorded noise is
stored (dropout masks bit-packed), and the final train state is stored as, per leaf, the full tensor when small or
else three fp64 projections (sum, sum of squares, dot with a fixed +-1 vector) plus 2048 sampled elements."""
from __future__ import annotations

import json
import zlib

import numpy as np

from . import drq_oracle as O
from .ref_update_runner import synth_flat_batch, synth_packed_batch

FULL_MAX = 4096
N_SAMPLE = 2048
SECTIONS = ("params", "target", "mu_critic", "nu_critic", "mu_actor", "nu_actor", "mu_temperature", "nu_temperature")


def _proj_vec(n, salt):
    r = np.random.Generator(np.random.PCG64(np.random.SeedSequence([n, salt])))
    return r.integers(0, 2, n).astype(np.float64) * 2.0 - 1.0


def _sample_idx(n, salt):
    r = np.random.Generator(np.random.PCG64(np.random.SeedSequence([n, salt, 7])))
    return np.sort(r.choice(n, size=N_SAMPLE, replace=False))


def _salt(name):
    return zlib.crc32(name.encode())


def leaf_record(name, v):
    v = np.asarray(v, np.float64).reshape(-1)
    if v.size <= FULL_MAX:
        return {"full": v}
    idx = _sample_idx(v.size, _salt(name))
    return {"stat": np.array([v.sum(), (v * v).sum(), (v * _proj_vec(v.size, _salt(name))).sum()]), "val": v[idx]}


def leaf_compare(name, rec, got):
    """-> (worst error relative to the leaf's scale, description).  `got`: candidate tensor (any float dtype)."""
    got = np.asarray(got, np.float64).reshape(-1)
    if "full" in rec:
        ref = rec["full"]
        assert got.size == ref.size, (name, got.size, ref.size)
        scale = np.abs(ref).max() + 1e-300
        return float(np.abs(got - ref).max() / scale), "full"
    idx = _sample_idx(got.size, _salt(name))
    ref = rec["val"]
    scale = np.abs(ref).max() + 1e-300
    e_val = float(np.abs(got[idx] - ref).max() / scale)
    st = np.array([got.sum(), (got * got).sum(), (got * _proj_vec(got.size, _salt(name))).sum()])
    # projections of N elements: compare against the natural size of such a sum, sqrt(N) * rms
    rms = np.sqrt(max(rec["stat"][1], 1e-300) / got.size)
    e_sum = abs(st[0] - rec["stat"][0]) / (np.sqrt(got.size) * rms)
    e_dot = abs(st[2] - rec["stat"][2]) / (np.sqrt(got.size) * rms)
    e_sq = abs(st[1] - rec["stat"][1]) / max(rec["stat"][1], 1e-300)
    return max(e_val, e_sum / np.sqrt(got.size), e_dot / np.sqrt(got.size), e_sq), "sampled"


def cfg_to_dict(cfg: O.Config):
    return {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.__dict__.items()}


def cfg_from_dict(d):
    d = dict(d)
    d["image_keys"] = tuple(d["image_keys"])
    return O.Config(**d)


def _jsonable(t):
    if isinstance(t, dict):
        return {k: _jsonable(v) for k, v in t.items()}
    return list(t)


def shape_tree(t):
    """nested dict of arrays / scalars -> nested dict of shape lists (the form stored in the golden's meta)"""
    if isinstance(t, dict):
        return {k: shape_tree(v) for k, v in t.items()}
    return list(np.shape(t))


def pack(res, param_seed, batch_seed):
    cfg = res["cfg"]
    out = {"meta": np.array(json.dumps({"cfg": cfg_to_dict(cfg), "B": res["B"],
                                        "schedule": [[s[0]] + [list(x) if isinstance(x, (tuple, list)) else x for x in s[1:]] for s in res["schedule"]],
                                        "param_seed": param_seed, "batch_seed": batch_seed, "final_step": res["final"]["step"],
                                        # which stream the stand-in jax.random drew from, state.rng before / after the schedule
                                        "prng": res.get("prng", "philox"), "rng0": res.get("rng0"), "rng_final": res["final"].get("rng"),
                                        # shapes of the reference's agent.state.params / opt_states trees (flax state-dict form)
                                        "param_tree": _jsonable(res["final"]["param_tree"]),
                                        "opt_state_tree": _jsonable(res["final"]["opt_state_tree"])}))}
    for i, st in enumerate(res["steps"]):
        for k, v in st["batch"]["frames"].items():
            out[f"s{i}_crc_{k}"] = np.uint32(zlib.crc32(v.tobytes()))
        n = st["noise"]
        for k, v in n.items():
            if isinstance(v, dict):
                for cam, m in v.items():
                    out[f"s{i}_{k}_{cam}"] = np.packbits(m.astype(np.uint8), axis=None)
            else:
                out[f"s{i}_{k}"] = np.asarray(v)
        keys = sorted(st["info"])
        out[f"s{i}_info_keys"] = np.array(keys)
        out[f"s{i}_info_vals"] = np.array([st["info"][k] for k in keys], np.float64)
    f = res["final"]
    secs = {"params": f["params"], "target": f["target"]}
    for tx in ("critic", "actor", "temperature"):
        secs[f"mu_{tx}"], secs[f"nu_{tx}"] = f["mu"][tx], f["nu"][tx]
    for sec, tree in secs.items():
        for name, v in tree.items():
            for kind, arr in leaf_record(f"{sec}/{name}", v).items():
                out[f"f_{sec}|{name}|{kind}"] = arr
    return out


def unpack(npz):
    meta = json.loads(str(npz["meta"]))
    cfg = cfg_from_dict(meta["cfg"])
    B, D = meta["B"], cfg.sle_dim
    steps = []
    for i, item in enumerate(meta["schedule"]):
        kind = item[0]
        utd = item[1] if kind == "high_utd" else 1
        nets = tuple(item[1]) if kind == "update" else ()
        pb = (synth_flat_batch if cfg.state_only else synth_packed_batch)(cfg, B, meta["batch_seed"] + i)
        for k, v in pb["frames"].items():
            assert np.uint32(zlib.crc32(v.tobytes())) == npz[f"s{i}_crc_{k}"], "synthetic inputs drifted from the golden run's"
        noise = {}
        for key in npz.files:
            if not key.startswith(f"s{i}_") or "_crc_" in key or "_info_" in key:
                continue
            nm = key[len(f"s{i}_"):]
            if nm.startswith("mask_"):
                for cam in cfg.image_keys:
                    if nm.endswith("_" + cam):
                        base = nm[:-len(cam) - 1]
                        noise.setdefault(base, {})[cam] = np.unpackbits(npz[key])[:B * D].reshape(B, D)
            else:
                noise[nm] = npz[key]
        for eps_name, mask_name in (("eps_next", "mask_next"), ("eps_pi", "mask_obs_pi"), ("eps_temp", "mask_next_temp")):
            if eps_name in noise:
                noise.setdefault(mask_name, {})      # state-only agent: no cameras, no dropout masks
        info = dict(zip([str(k) for k in npz[f"s{i}_info_keys"]], npz[f"s{i}_info_vals"]))
        steps.append({"kind": kind, "utd": utd, "nets": nets, "batch": pb, "noise": noise, "info": info})
    final = {}
    for key in npz.files:
        if key.startswith("f_"):
            sec, name, kind = key[2:].split("|")
            final.setdefault(sec, {}).setdefault(name, {})[kind] = npz[key]
    return {"cfg": cfg, "B": B, "meta": meta, "steps": steps, "final": final}


def leaf_errors(name, rec, got):
    """-> (errors of the stored elements relative to the leaf's scale [array], scale).  For fp32 candidates whose
    comparison needs a percentile (Adam's sign-like update) instead of the max."""
    got = np.asarray(got, np.float64).reshape(-1)
    if "full" in rec:
        ref, g = rec["full"], got
    else:
        ref, g = rec["val"], got[_sample_idx(got.size, _salt(name))]
    scale = np.abs(ref).max() + 1e-300
    return np.abs(g - ref), scale
