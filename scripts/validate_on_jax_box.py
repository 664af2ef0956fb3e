#!/usr/bin/env python
"""Pins what this build could not pin (no jax / flax / optax / agentlace in the build image) -- run it ONCE on
a box that has the reference's own environment (serl_launcher/requirements.txt: jax 0.4.35, flax, optax, distrax; agentlace@cf2c337,
serl_launcher/setup.py:16), an MI355X and this repository built (`python __graft_entry__.py`):

    SERL_REFERENCE=/path/to/serl python scripts/validate_on_jax_box.py            # every stage
    python scripts/validate_on_jax_box.py --list                                  # the stages and what each needs

Every stage prints PASS / FAIL / SKIP(reason); the exit status is 1 if any stage FAILED.  Nothing here is product code and nothing
in the product or in the GPU suite depends on it; tests/test_validate_on_jax_box.py runs the stages whose packages are importable
and skips, with the reason, otherwise.

Stages (reference call sites in brackets):
  flax_checkpoint   a checkpoint written by serl_amd/utils/checkpoint.py is restored by flax.training.checkpoints.restore_checkpoint
                    -- raw, and INTO the state the reference's DrQAgent.create_drq builds (same tree, shapes, dtypes)
                    [examples/async_drq_sim/async_drq_sim.py:303-307, async_peg_insert_drq/async_drq_randomized.py:100-105]
  param_tree        agent.state.params of the HIP agent vs the reference agent: paths / shapes / dtypes equal
                    [agents/continuous/drq.py:105-242]
  init_from_seed    both agents created from the same seed: per-leaf max |difference| (REPORTED; a difference is the known hole of
                    DESIGN.md section 2 -- flax's initialisers are not restated in the product -- and does not fail the run)
                    [agents/continuous/drq.py:70, sac.py:369]
  update_parity     the reference's parameters copied into the HIP agent, ONE update_high_utd(utd_ratio=1) on the same batch with the
                    same state.rng: every parameter leaf, target leaf and info scalar within 1e-4 (this is the check of flax's
                    Dropout key derivation, optax.adam and distrax under the REAL libraries, not the stand-ins of oracle/jaxshim)
                    [agents/continuous/drq.py:255-294, sac.py:243-299, common/common.py:136-221]
  agentlace_wire    agentlace's own TrainerClient against serl_amd.transport.TrainerServer: connect, update() with a queued data
                    store, request("send-stats"), publish_network -> recv_network_callback
                    [utils/launcher.py:171-177, examples/async_drq_sim/async_drq_sim.py:95-108,202-229,297]
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KEYS, H, W, S, A, B = ("front", "wrist"), 128, 128, 24, 6, 16
STAGES = {
    "flax_checkpoint": ("flax", "jax", "serl_launcher", "gpu"),
    "param_tree": ("flax", "jax", "serl_launcher", "gpu"),
    "init_from_seed": ("flax", "jax", "serl_launcher", "gpu"),
    "update_parity": ("flax", "jax", "optax", "distrax", "serl_launcher", "gpu"),
    "agentlace_wire": ("agentlace",),
}


class Skip(Exception):
    pass


def missing(needs):
    """-> the first requirement of `needs` this box does not meet, or None"""
    for n in needs:
        if n == "gpu":
            import torch
            if not torch.cuda.is_available():
                return "no GPU visible"
            continue
        if n == "serl_launcher":
            ref = os.environ.get("SERL_REFERENCE")
            if ref and os.path.isdir(os.path.join(ref, "serl_launcher")):
                p = os.path.join(ref, "serl_launcher")
                if p not in sys.path:
                    sys.path.insert(0, p)
        try:
            m = importlib.import_module(n)
        except Exception as e:   # noqa: BLE001 (a broken install is a reason to skip, and to say why)
            return f"cannot import {n} ({type(e).__name__}: {e})"
        if "jaxshim" in (getattr(m, "__file__", "") or ""):     # (the test suite puts oracle/jaxshim on sys.path: jax, flax, agentlace stubs ...)
            return f"{n} resolves to the oracle's stand-in ({m.__file__}), not the real package"
    return None


def flatten(tree, pre=()):
    out = {}
    for k, v in tree.items():
        if hasattr(v, "items"):
            out.update(flatten(v, pre + (str(k),)))
        else:
            out[pre + (str(k),)] = v
    return out


def sample_inputs():
    obs = {k: np.zeros((1, H, W, 3), np.uint8) for k in KEYS}
    obs["state"] = np.zeros((1, S), np.float32)
    return obs, np.zeros((A,), np.float32)


def hip_agent(seed=7):
    from serl_amd.utils.launcher import make_drq_agent
    obs, act = sample_inputs()
    return make_drq_agent(seed, obs, act, image_keys=KEYS, encoder_type="resnet-pretrained", batch_size=B)


def ref_agent(seed=7):
    """The reference's own factory (utils/launcher.py:79-116).  `resnet-pretrained` builds the frozen ResNet-10 with random
    weights until load_resnet10_params patches it (train_utils.py:69-130 needs the network); the tree is the same either way."""
    from serl_launcher.utils.launcher import make_drq_agent
    obs, act = sample_inputs()
    return make_drq_agent(seed=seed, sample_obs=obs, sample_action=act, image_keys=KEYS, encoder_type="resnet-pretrained")


def batch(seed=3):
    rng = np.random.default_rng(seed)
    fr = lambda: {k: rng.integers(0, 256, (B, 1, H, W, 3), dtype=np.uint8) for k in KEYS}   # noqa: E731
    o, n = fr(), fr()
    o["state"], n["state"] = rng.standard_normal((B, 1, S)).astype(np.float32), rng.standard_normal((B, 1, S)).astype(np.float32)
    return {"observations": o, "next_observations": n, "actions": rng.uniform(-1, 1, (B, A)).astype(np.float32),
            "rewards": (rng.random(B) < 0.3).astype(np.float32), "masks": (rng.random(B) < 0.9).astype(np.float32),
            "dones": np.zeros(B, bool)}


def stage_flax_checkpoint():
    from flax.serialization import to_state_dict
    from flax.training import checkpoints
    from serl_amd.utils import checkpoint as ck
    a, r = hip_agent(), ref_agent()
    with tempfile.TemporaryDirectory() as d:
        ck.save_checkpoint(d, a, step=5)
        raw = checkpoints.restore_checkpoint(d, target=None)
        assert raw is not None, "flax found no checkpoint in the directory this library wrote"
        assert set(raw) == {"step", "params", "target_params", "opt_states", "rng"}, sorted(raw)
        mine, want = flatten(raw), flatten(to_state_dict(r.state))
        assert set(mine) == set(want), f"paths differ: {sorted(set(mine) ^ set(want))[:8]}"
        for p, v in want.items():
            g = np.asarray(mine[p])
            assert g.shape == np.shape(v) and g.dtype == np.asarray(v).dtype, (p, g.shape, np.shape(v), g.dtype, np.asarray(v).dtype)
        restored = checkpoints.restore_checkpoint(d, target=r.state)      # flax checks the structure against the reference's TrainState
        got = flatten(to_state_dict(restored))
        hp = flatten(a.state.params)
        for p, v in hp.items():
            assert np.array_equal(np.asarray(got[("params",) + p]), v), p
        # and the other direction: a checkpoint flax writes from the reference's state is read by this library
        checkpoints.save_checkpoint(os.path.join(d, "ref"), r.state, step=9, keep=1)
        ck.restore_checkpoint(os.path.join(d, "ref"), a)
        back = flatten(a.state.params)
        rp = flatten(to_state_dict(r.state)["params"])
        for p, v in rp.items():
            assert np.allclose(back[p], np.asarray(v), rtol=0, atol=0), p
    return f"{len(want)} leaves, both directions"


def stage_param_tree():
    from flax.serialization import to_state_dict
    a, r = hip_agent(), ref_agent()
    mine, want = flatten(a.state.params), flatten(to_state_dict(r.state)["params"])
    assert set(mine) == set(want), f"paths differ: {sorted(set(mine) ^ set(want))[:8]}"
    for p, v in want.items():
        assert np.shape(mine[p]) == np.shape(v) and np.asarray(mine[p]).dtype == np.asarray(v).dtype, p
    return f"{len(want)} leaves"


def stage_init_from_seed():
    from flax.serialization import to_state_dict
    a, r = hip_agent(seed=11), ref_agent(seed=11)
    mine, want = flatten(a.state.params), flatten(to_state_dict(r.state)["params"])
    worst = sorted(((float(np.max(np.abs(np.asarray(mine[p], np.float64) - np.asarray(v, np.float64)))), "/".join(p)) for p, v in want.items()),
                   reverse=True)
    same_rng = np.array_equal(np.asarray(a.state.rng, np.uint32), np.asarray(r.state.rng, np.uint32).reshape(-1))
    rep = f"state.rng equal: {same_rng}; largest per-leaf |difference|: " + ", ".join(f"{n} {e:.3g}" for e, n in worst[:4])
    assert same_rng, "state.rng after create differs (the key chain of create_drq / create, drq.py:69-84) -- " + rep
    return rep + ("  [parameters identical]" if worst[0][0] == 0.0 else "  [parameters differ: the documented hole, not a failure]")


def stage_update_parity():
    import jax
    from flax.serialization import to_state_dict
    a, r = hip_agent(seed=5), ref_agent(seed=5)
    sd = to_state_dict(r.state)
    a.state.replace(params=sd["params"], target_params=sd["target_params"], rng=np.asarray(sd["rng"], np.uint32).reshape(-1))
    b = batch()
    packed = {**b, "observations": {k: (np.concatenate([b["observations"][k], b["next_observations"][k]], axis=1) if k in KEYS
                                        else b["observations"][k]) for k in b["observations"]},
              "next_observations": {"state": b["next_observations"]["state"]}}
    r2, rinfo = r.update_high_utd(jax.tree_util.tree_map(np.asarray, packed), utd_ratio=1)
    import torch
    dev = {k: ({kk: torch.as_tensor(vv).cuda() for kk, vv in v.items()} if isinstance(v, dict) else torch.as_tensor(v).cuda())
           for k, v in packed.items()}
    a2, ainfo = a.update_high_utd(dev, utd_ratio=1)
    ai = ainfo.resolve()
    worst = 0.0
    for sec in ("params", "target_params"):
        want = flatten(to_state_dict(r2.state)[sec])
        got = flatten(getattr(a2.state, sec))
        for p, v in want.items():
            v = np.asarray(v, np.float64)
            e = float(np.max(np.abs(np.asarray(got[p], np.float64) - v)) / (np.max(np.abs(v)) + 1e-30))
            worst = max(worst, e)
            assert e < 1e-4, (sec, "/".join(p), e)
    for grp in ("critic", "actor", "temperature"):
        for k, v in rinfo[grp].items():
            assert abs(float(ai[grp][k]) - float(v)) <= 1e-4 * max(1.0, abs(float(v))), (grp, k, float(ai[grp][k]), float(v))
    assert np.array_equal(np.asarray(a2.state.rng, np.uint32), np.asarray(r2.state.rng, np.uint32).reshape(-1)), "state.rng after the update"
    return f"worst leaf-normalised parameter error {worst:.2e}; info scalars and state.rng equal"


def stage_agentlace_wire():
    from agentlace.data.data_store import QueuedDataStore as RefStore
    from agentlace.trainer import TrainerClient as RefClient, TrainerConfig as RefConfig
    from serl_amd.transport import QueuedDataStore, TrainerServer, make_trainer_config
    cfg = make_trainer_config(port_number=5588, broadcast_port=5589)
    stats, store = [], QueuedDataStore(1000)
    srv = TrainerServer(cfg, request_callback=lambda t, p: (stats.append((t, p)), {})[1])
    srv.register_data_store("actor_env", store)
    srv.start(threaded=True)
    try:
        rc = RefConfig(port_number=5588, broadcast_port=5589, request_types=["send-stats"])
        ds = RefStore(1000)
        cli = RefClient("actor_env", "127.0.0.1", rc, ds, wait_for_server=True)
        got = []
        cli.recv_network_callback(lambda p: got.append(p))
        for i in range(5):
            ds.insert({"observations": np.full((3,), i, np.float32), "rewards": np.float32(i)})
        assert cli.update(), "TrainerClient.update() was refused"
        deadline = time.time() + 10
        while len(store) < 5 and time.time() < deadline:
            time.sleep(0.05)
        assert len(store) == 5, f"{len(store)} of 5 transitions arrived"
        cli.request("send-stats", {"timer": {"total": 1.0}})
        assert stats and stats[-1][0] == "send-stats", stats
        tree = {"modules_actor": {"w": np.arange(6, dtype=np.float32).reshape(2, 3)}}
        deadline = time.time() + 10
        while not got and time.time() < deadline:
            srv.publish_network(tree)
            time.sleep(0.2)
        assert got and np.array_equal(np.asarray(got[-1]["modules_actor"]["w"]), tree["modules_actor"]["w"]), "publish_network did not arrive"
        cli.stop()
    finally:
        srv.stop()
    return "update(), request('send-stats'), publish_network all round-tripped with agentlace's own client"


def run(names, verbose=True):
    """-> {stage: ("PASS" | "FAIL" | "SKIP", detail)}"""
    out = {}
    for n in names:
        why = missing(STAGES[n])
        if why:
            out[n] = ("SKIP", why)
        else:
            try:
                out[n] = ("PASS", globals()[f"stage_{n}"]())
            except Skip as e:
                out[n] = ("SKIP", str(e))
            except Exception as e:   # noqa: BLE001
                import traceback
                out[n] = ("FAIL", f"{type(e).__name__}: {e}\n{traceback.format_exc(limit=4)}")
        if verbose:
            print(f"[{out[n][0]}] {n}: {out[n][1]}", flush=True)
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("stages", nargs="*", default=list(STAGES))
    ap.add_argument("--list", action="store_true")
    args = ap.parse_args()
    if args.list:
        for n, needs in STAGES.items():
            print(f"{n}: needs {', '.join(needs)}; here: {missing(needs) or 'available'}")
        return 0
    res = run(args.stages)
    return 1 if any(v[0] == "FAIL" for v in res.values()) else 0


if __name__ == "__main__":
    sys.exit(main())
