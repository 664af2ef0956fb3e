"""CPU: pins oracle/drq_oracle.py against the REFERENCE's own update code.

* golden fixtures tests/golden/update_*.npz were produced by tests/golden/make_golden_update.py, which imports the
  reference's DrQAgent (agents/continuous/drq.py:255-328 -> sac.py:243-299,544-596 -> common/common.py:124-221) from
  /root/reference and runs it UNMODIFIED, in fp64, under stand-ins for its absent third-party libraries
  (oracle/jaxshim).  The oracle, fed the noise that run drew, must reproduce every info scalar and the whole final
  train state (params, target params, all three Adam states) to fp64 round-off.
* when /root/reference is present (build container) the reference is also run live and compared directly.
"""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import drq_oracle as O
from oracle import golden_update as G
from oracle import ref_update_runner as RR
from oracle import ref_update_shim as RS

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "update_*.npz")))
F64_TOL = 1e-9


def _run_oracle(cfg, steps, param_seed):
    trunk, theta = O.init_params(cfg, param_seed)
    st = O.TrainState(cfg, trunk, theta, torch.float64)
    infos = []
    for step in steps:
        b, n = RR.oracle_batch_and_noise(cfg, step, torch.float64)
        if step["kind"] == "critics":
            info, _ = O.update_critics(st, b, n)
        elif step["kind"] == "high_utd":
            info, _ = O.update_high_utd(st, b, n, step["utd"])
        else:
            info = O.update(st, b, n, step["nets"])
        infos.append(info)
    return st, infos


def _oracle_sections(st):
    secs = {"params": st.params, "target": st.target}
    for tx in O.TX_NAMES:
        secs[f"mu_{tx}"], secs[f"nu_{tx}"] = st.opt[tx]["mu"], st.opt[tx]["nu"]
    return secs


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[7:-4] for p in GOLDEN])
def test_oracle_reproduces_the_reference_golden(path):
    g = G.unpack(np.load(path))
    cfg = g["cfg"]
    st, infos = _run_oracle(cfg, g["steps"], g["meta"]["param_seed"])
    sched = O.TrainState(cfg, {}, {}, torch.float64)
    st_counts, c = [], 0
    for step in g["steps"]:       # optimizer count before the LAST optimizer step of each call
        c += {"critics": 1, "update": 1}.get(step["kind"], step["utd"] + 1)
        st_counts.append(c - 1)
    for i, (info, step) in enumerate(zip(infos, g["steps"])):
        for k, v in info.items():
            r = step["info"][k]
            assert abs(v - r) <= F64_TOL * max(1.0, abs(r)), (i, k, v, r)
        # the reference logs the (float32) learning rate each optimizer used: schedule(count before the step)
        count = st_counts[i]
        for tx in O.TX_NAMES:
            assert abs(step["info"][f"{tx}_lr"] - np.float32(sched.lr_at(count, tx))) < 1e-12, (i, tx)
    assert st.step == g["meta"]["final_step"]
    worst = 0.0
    for sec, tree in _oracle_sections(st).items():
        for name, t in tree.items():
            e, how = G.leaf_compare(f"{sec}/{name}", g["final"][sec][name], t.numpy())
            assert e < F64_TOL, (sec, name, how, e)
            worst = max(worst, e)
    print(f"{os.path.basename(path)}: oracle vs reference golden, worst {worst:.1e}")


def test_golden_fixtures_exist():
    assert len(GOLDEN) >= 8, "tests/golden/update_*.npz missing: run tests/golden/make_golden_update.py in the build container"


needs_ref = pytest.mark.skipif(not RS.reference_available(), reason="/root/reference not present (GPU box)")


@needs_ref
def test_live_reference_matches_oracle():
    """The reference itself, imported from /root/reference and run here, against the oracle (no fixture in between)."""
    cfg = O.Config(image_keys=("wrist_1", "wrist_2"), H=64, W=64, S=19, A=7)     # the peg-insertion key names / dims
    sched = [("critics",), ("high_utd", 2), ("critics",)]
    res = RR.run_reference(cfg, 4, sched, param_seed=7, batch_seed=55)
    st, infos = _run_oracle(cfg, res["steps"], 7)
    for info, step in zip(infos, res["steps"]):
        for k, v in info.items():
            assert abs(v - step["info"][k]) <= F64_TOL * max(1.0, abs(step["info"][k])), (k, v, step["info"][k])
    f = res["final"]
    assert st.step == f["step"] == 5 and all(c == (5, 5) for c in f["count"].values())
    for k in st.params:
        for got, ref in ((st.params[k], f["params"][k]), (st.target[k], f["target"][k])):
            assert np.abs(got.numpy().reshape(-1) - ref).max() <= F64_TOL * (np.abs(ref).max() + 1e-30), k
        for tx in O.TX_NAMES:
            for got, ref in ((st.opt[tx]["mu"][k], f["mu"][tx][k]), (st.opt[tx]["nu"][k], f["nu"][tx][k])):
                assert np.abs(got.numpy().reshape(-1) - ref).max() <= F64_TOL * (np.abs(ref).max() + 1e-300), (tx, k)
    # frozen trunk: parameters untouched, target copy = EMA of identical values (drifts by round-off only)
    trunk, _ = O.init_params(cfg, 7)
    assert np.array_equal(f["trunk_conv_init"], trunk["trunk/conv_init"].astype(np.float64).reshape(-1))
    assert np.abs(f["trunk_conv_init_target"] - f["trunk_conv_init"]).max() < 1e-14


@needs_ref
def test_live_reference_on_jax_key_chain(monkeypatch):
    """The same with the stand-in jax.random drawing through JAX's own threefry2x32 key chain (oracle/jaxshim/jax/threefry.py,
    pinned in tests/test_threefry_oracle.py): the reference's split / fold_in / randint / normal / bernoulli calls then consume
    the numbers a real JAX run draws for these keys; the oracle, fed the recorded draws, still agrees."""
    monkeypatch.setenv("SERL_JAXSHIM_PRNG", "threefry")
    cfg = O.Config(image_keys=("front",), H=64, W=64, S=7, A=4)
    sched = [("critics",), ("high_utd", 1)]
    res = RR.run_reference(cfg, 4, sched, param_seed=3, batch_seed=11)
    st, infos = _run_oracle(cfg, res["steps"], 3)
    for info, step in zip(infos, res["steps"]):
        for k, v in info.items():
            assert abs(v - step["info"][k]) <= F64_TOL * max(1.0, abs(step["info"][k])), (k, v, step["info"][k])
    f = res["final"]
    for k in st.params:
        assert np.abs(st.params[k].numpy().reshape(-1) - f["params"][k]).max() <= F64_TOL * (np.abs(f["params"][k]).max() + 1e-30), k


@needs_ref
def test_reference_random_crop_equals_the_oracle_shift():
    """vision/data_augmentations.py:7-36 (edge pad 4 + dynamic_slice) run from the reference == oracle random_shift."""
    jax = RS.install(True)
    import jax.numpy as jnp
    from serl_launcher.vision.data_augmentations import batched_random_crop
    from oracle.replay_oracle import random_shift
    img = np.random.default_rng(0).integers(0, 256, (6, 1, 24, 20, 3), dtype=np.uint8)
    tape = jax.random.start_tape()
    out = np.asarray(batched_random_crop(jnp.asarray(img), jax.random.PRNGKey(3), padding=4, num_batch_dims=2))
    jax.random.stop_tape()
    offs = np.stack([r["value"] for r in tape]).astype(np.int32)
    assert offs.shape == (6, 2) and offs.min() >= 0 and offs.max() <= 8
    assert np.array_equal(out[:, 0], random_shift(img[:, 0], offs))


@needs_ref
def test_reference_parameter_tree_is_what_the_product_exports():
    """agent.state.params of the reference (built by its own make_drq_agent + load_resnet10_params on a synthetic
    pickle) has exactly the paths serl_amd/agents/flax_tree.py exports, the trunk only under the first camera."""
    from serl_amd.agents import flax_tree as FT
    cfg = O.Config(image_keys=("front", "wrist"), H=64, W=64, S=5, A=3)
    res = RR.run_reference(cfg, 2, [], param_seed=1)
    tree = res["final"]["param_tree"]

    def paths(t, pre=()):
        out = {}
        for k, v in t.items():
            if isinstance(v, dict):
                out.update(paths(v, pre + (k,)))
            else:
                out[pre + (k,)] = tuple(v)
        return out

    ref_paths = paths(tree)
    mine = {}
    shapes = FT.theta_shapes(2, 64, 64, 5, 3)
    for leaf, ps in FT.theta_paths(cfg.image_keys).items():
        mine[ps[0]] = tuple(shapes[leaf])
    tsh = FT.trunk_shapes()
    for leaf, sub in FT._trunk_paths().items():
        mine[("modules_actor", "encoder", f"encoder_{FT.trunk_owner(cfg.image_keys)}", "pretrained_encoder") + sub] = tuple(tsh[leaf])
    assert set(mine) == set(ref_paths), (sorted(set(mine) ^ set(ref_paths))[:6])
    for p, shp in mine.items():
        assert int(np.prod(shp)) == int(np.prod(ref_paths[p])), (p, shp, ref_paths[p])
