"""GPU parity ON THE SHAPES bench.py TIMES but the other suites only reach at reduced size:

* one rank's share of a data-parallel step -- `bench.py --emulate-world 8 / 2` runs an agent created with batch 32 / 128
  (a trunk pass over 128 / 512 images selects other conv kernels than the 1024-image pass: 64x64 register-staged tiles,
  conv_init with `pool_finish_split`, 128x64 LDS-DMA tiles) and normalises its losses by the GLOBAL batch
  (common/common.py:213-214 `pmean` made real: serl_agent_set_shard / critic_grads(offset, count, global) / grad_view).
  Here the bench shape (128x128, 2 cameras, S=24, A=6, B=256) is run as 8 shards of 32 and as 2 shards of 128, the
  shards' gradient views and loss scalars are SUMMED (what the all-reduce does) and compared, leaf by leaf, with the
  full-batch fp64 oracle at 1e-4; the summed gradients then go through `apply` and the parameters are compared too.
* BASELINE.json configs[0] (`async_sac_state_sim`) at the shape `bench.py --workload sac_state` times: B = 2048 = 256 x UTD 8
  (examples/async_sac_state_sim/async_sac_state_sim.py:231,296).
* a critic_actor_ratio = 4 / 8 iteration ((car - 1) x update_critics + update_high_utd; async_drq_sim.py:266-292) at 128x128 / B=256,
  the `drq_demos` / `peg` workloads' sequence: Adam moments, zero-gradient optimizer steps, EMA and step bookkeeping at the
  full shape."""
import numpy as np
import pytest
import torch

from oracle import drq_oracle as O
import agent_helpers as AH
from test_agent_gpu import SEQ_TOL, _compare_state
from test_bench_shape_gpu import _assert_grads, _grad_report

pytestmark = pytest.mark.gpu
TOL = 1e-4
KEYS = ("front", "wrist")
B = 256
APPLY_CRITIC, APPLY_ACTOR_TEMP = 1, 6


def _cfg():
    return O.Config(image_keys=KEYS, H=128, W=128, S=24, A=6)


def _slice_batch(b, lo, hi):
    return {k: ({c: v[lo:hi] for c, v in x.items()} if isinstance(x, dict) else x[lo:hi]) for k, x in b.items()}


def _slice_noise(noise, lo, hi):
    out = {}
    for k, v in noise.items():
        if k == "redq_idx":
            out[k] = v
        elif isinstance(v, dict):
            out[k] = {c: m[lo:hi] for c, m in v.items()}
        else:
            out[k] = v[lo:hi]
    return out


_ORACLE = {}


def _oracle_pair():
    """critic step, then actor + temperature step (= update_high_utd(utd_ratio=1)) of the fp64 oracle on the FULL batch;
    computed once for both shardings (~15 s of host time)."""
    if "r" not in _ORACLE:
        cfg = _cfg()
        trunk, theta = O.init_params(cfg, 42)
        st = O.TrainState(cfg, trunk, theta, torch.float64)
        b = AH.synth_batch(cfg, B, seed=61, frames_seed=70)
        noise = O.make_noise(cfg, B, seed=62, utd_ratio=1)
        tb, tn = AH.batch_to_torch(b, torch.float64), O.noise_to_torch(noise, torch.float64)
        fo, fn = O.features(st, tb["obs"]), O.features(st, tb["next"])
        n = dict(tn)
        n["redq_idx"] = np.asarray(noise["redq_idx"]).reshape(-1, cfg.subsample)[0]
        cinfo, caux = O.critic_update(st, fo, fn, tb["state"], tb["next_state"], tb["action"], tb["reward"], tb["mask"], n)
        ainfo, aaux = O.actor_temp_update(st, fo, fn, tb["state"], tb["next_state"], tn)
        _ORACLE["r"] = (cfg, st, b, noise, cinfo, caux, ainfo, aaux)
    return _ORACLE["r"]


@pytest.mark.parametrize("shards", [8, 2])
def test_shards_of_the_bench_batch_sum_to_the_full_batch_oracle(gpu, shards):
    cfg, st, b, noise, cinfo, caux, ainfo, aaux = _oracle_pair()
    Bl = B // shards
    _, core = AH.make_pair(cfg, Bl)          # the agent a rank of an N-GPU job creates: batch = B / N
    sl, _ = AH.leaf_slices(cfg)
    pc = sl["enc/proprio/ln/bias"][1]
    pa0, pa1 = sl["enc/proprio/dense/kernel"][0], sl["actor/logstd/bias"][1]
    dbs = [AH.batch_to_device(cfg, _slice_batch(b, r * Bl, (r + 1) * Bl)) for r in range(shards)]
    dns = [AH.noise_to_device(cfg, _slice_noise(noise, r * Bl, (r + 1) * Bl)) for r in range(shards)]

    # ---- critic phase: every rank's [gradients | loss scalars], summed
    g = np.zeros(pc, np.float64)
    sc = np.zeros(32, np.float64)
    core.begin_update()
    for r in range(shards):
        core.set_shard(r * Bl, B)
        core.encode(dbs[r])
        core.critic_grads(0, Bl, B, dns[r])
        g += core.debug("g_critic", pc)
        sc[:3] += core.debug("scalars", 3)
    plan = core.trunk_plan()
    print(f"{shards} x {Bl}: trunk plan {plan}")
    assert plan["images"] == 4 * Bl
    if shards == 8:      # 128 images: conv_init + pool_finish_split, 64x64 register-staged tiles in stages 2-3
        assert plan["pool"] == 1 and plan["raw_b0"] == 0
        assert plan["b0_conv0"][0] == "S" and plan["b0_conv0"][3] == 1 and plan["b1_conv1"][0] == "S"
        assert plan["b1_conv0"][:2] == ("D", 4)
        for l in ("b2_conv0", "b2_conv1", "b2_proj", "b3_conv0", "b3_conv1", "b3_proj"):
            assert plan[l][:2] == ("R", 2), (l, plan[l])
    else:                # 512 images: the full-batch front end, 128x64 LDS-DMA tiles in stage 3
        assert plan["pool"] == 2 and plan["raw_b0"] == 1
        assert plan["b2_conv1"][:2] == ("D", 0) and plan["b3_conv1"][:2] == ("D", 4)
    core.debug_set("g_critic", g)
    core.debug_set("scalars", sc)
    worst, worst_el = _grad_report(cfg, core, caux["grads"], "g_critic", 0)
    _assert_grads(worst, worst_el, f"{shards} x {Bl} shards: critic grads")
    core.apply(APPLY_CRITIC)
    got = core.read_info()
    for k in ("critic_loss", "predicted_qs", "target_qs"):
        assert abs(got[k] - cinfo[k]) < TOL * max(1.0, abs(cinfo[k])), (k, got[k], cinfo[k])

    # ---- actor + temperature phase at the updated parameters
    ga = np.zeros(pa1 - pa0, np.float64)
    sc = np.zeros(32, np.float64)
    for r in range(shards):
        core.set_shard(r * Bl, B)
        core.encode(dbs[r])
        core.actor_grads(B, dns[r])
        ga += core.debug("g_actor", pa1 - pa0)
        sc[3:6] += core.debug("scalars", 6)[3:6]
    core.debug_set("g_actor", ga)
    core.debug_set("scalars", sc)
    worst, worst_el = _grad_report(cfg, core, aaux["g_actor"], "g_actor", pa0)
    _assert_grads(worst, worst_el, f"{shards} x {Bl} shards: actor grads")
    core.apply(APPLY_ACTOR_TEMP)
    got = core.read_info()
    for k in ("actor_loss", "temperature", "entropy", "temperature_loss"):
        assert abs(got[k] - ainfo[k]) < TOL * max(1.0, abs(ainfo[k])), (k, got[k], ainfo[k])
    _compare_state(cfg, st, core, steps=2)
    assert core.step == st.step == 2


def test_state_sac_at_the_timed_shape(gpu):
    """C1: B = 2048 = 256 x UTD 8, S=10, A=4 -- two learner iterations (16 critic + 2 actor/temperature updates)."""
    cfg = O.Config(image_keys=(), S=10, A=4, discount=0.99, warmup=4, temp_warmup=0)
    Bt, utd = 2048, 8
    st, core = AH.make_pair(cfg, Bt)
    sl, _ = AH.leaf_slices(cfg)
    for it in range(2):
        b = AH.synth_batch(cfg, Bt, seed=14 + it)
        noise = O.make_noise(cfg, Bt, seed=18 + it, utd_ratio=utd)
        info, aux = O.update_high_utd(st, AH.batch_to_torch(b, torch.float64), O.noise_to_torch(noise, torch.float64), utd)
        core.update_high_utd(AH.batch_to_device(cfg, b), utd, AH.noise_to_device(cfg, noise))
        got = core.read_info()
        for k in ("critic_loss", "predicted_qs", "target_qs", "actor_loss", "temperature", "entropy", "temperature_loss"):
            assert abs(got[k] - info[k]) < TOL * max(1.0, abs(info[k])), (it, k, got[k], info[k])
        pa0 = sl["actor/w1"][0]
        g = core.debug("g_actor", sl["actor/logstd/bias"][1] - pa0)
        for k, gv in aux["g_actor"].items():
            lo, hi = sl[k]
            e = AH.rel_err(g[lo - pa0:hi - pa0], gv.numpy().reshape(-1))
            assert e < TOL, ("g_actor", k, e)
    _compare_state(cfg, st, core, tol=TOL, steps=2 * (utd + 1))
    assert core.step == st.step == 2 * (utd + 1)


@pytest.mark.parametrize("car", [4, 8])   # (8 = the drq_demos / peg workloads' ratio)
def test_car_iteration_at_bench_shape(gpu, car):
    """critic_actor_ratio = 4 and 8 at 128x128 / B=256: (car - 1) x update_critics + update_high_utd(utd_ratio=1).  The oracle's frozen-trunk
    features are computed once per frame set (two sets, alternating) -- the trunk is frozen, so that only saves host time;
    the HIP side runs the whole path (trunk + update) on every step."""
    cfg = _cfg()
    st, core = AH.make_pair(cfg, B)
    frames = [AH.synth_batch(cfg, B, seed=70 + i, frames_seed=70 + i) for i in range(2)]
    feats = []
    for f in frames:
        tb = AH.batch_to_torch(f, torch.float64)
        feats.append((O.features(st, tb["obs"]), O.features(st, tb["next"])))
    for it in range(car):
        b = AH.synth_batch(cfg, B, seed=80 + it)
        b["obs"], b["next"] = frames[it % 2]["obs"], frames[it % 2]["next"]
        last = it == car - 1
        noise = O.make_noise(cfg, B, seed=90 + it, utd_ratio=1)
        tb, tn = AH.batch_to_torch(b, torch.float64), O.noise_to_torch(noise, torch.float64)
        fo, fn = feats[it % 2]
        n = dict(tn)
        n["redq_idx"] = np.asarray(noise["redq_idx"]).reshape(-1, cfg.subsample)[0]
        cinfo, _ = O.critic_update(st, fo, fn, tb["state"], tb["next_state"], tb["action"], tb["reward"], tb["mask"], n)
        db, dn = AH.batch_to_device(cfg, b), AH.noise_to_device(cfg, noise)
        if last:
            ainfo, _ = O.actor_temp_update(st, fo, fn, tb["state"], tb["next_state"], tn)
            core.update_high_utd(db, 1, dn)
        else:
            core.update_critics(db, dn)
        got = core.read_info()
        for k in ("critic_loss", "predicted_qs", "target_qs"):
            assert abs(got[k] - cinfo[k]) < TOL * max(1.0, abs(cinfo[k])), (it, k, got[k], cinfo[k])
        if last:
            for k in ("actor_loss", "temperature", "entropy", "temperature_loss"):
                assert abs(got[k] - ainfo[k]) < TOL * max(1.0, abs(ainfo[k])), (k, got[k], ainfo[k])
    worst = _compare_state(cfg, st, core, tol=SEQ_TOL, steps=car + 1)
    print(f"CAR={car} at the bench shape, worst bulk rel err after {car + 1} optimizer steps:", worst)
    assert core.step == st.step == car + 1
