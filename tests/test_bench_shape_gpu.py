"""GPU parity AT THE BENCHMARK SHAPE (BASELINE.json headline: batch 256, 2 cameras 128x128x3, S=24, A=6, REDQ-10):
the HIP update through the C ABI against the fp64 CPU oracle, every gradient leaf, q, target_q and the info scalars at
the north star's 1e-4.  This is the configuration bench.py times: 4x4 SpatialLearnedEmbeddings, the K-split budget and
the deferred parameter-gradient launches taken at 256 rows, the 128x128 / 128x64 conv tiles inside a full update
(reference: agents/continuous/drq.py:255-328, sac.py:118-299).  The fp64 oracle needs ~10-20 s per update on the GPU
box's host cores."""
import numpy as np
import pytest
import torch

from oracle import drq_oracle as O
import agent_helpers as AH

pytestmark = pytest.mark.gpu
TOL = 1e-4
KEYS = ("front", "wrist")
B = 256


def _cfg(keys=KEYS, A=6):
    return O.Config(image_keys=keys, H=128, W=128, S=24, A=A)


def _grad_report(cfg, core, grads, tap, sl_lo):
    """max-abs relative error per leaf (normalised by the leaf's max-abs) AND the worst per-element relative error over
    the elements above 1e-3 of the leaf's max-abs (so errors in small-but-not-negligible elements cannot hide)."""
    sl, _ = AH.leaf_slices(cfg)
    pc = sl["enc/proprio/ln/bias"][1]
    n = {"g_critic": pc, "g_actor": sl["actor/logstd/bias"][1] - sl_lo}[tap]
    g = core.debug(tap, n).astype(np.float64)
    worst, worst_el = {}, {}
    for k, gv in grads.items():
        lo, hi = sl[k]
        got, ref = g[lo - sl_lo:hi - sl_lo], gv.numpy().reshape(-1)
        worst[k] = AH.rel_err(got, ref)
        worst_el[k] = AH.elem_rel_err(got, ref, floor=1e-3)
    return worst, worst_el


def _assert_grads(worst, worst_el, what):
    for k, e in worst.items():
        assert e < TOL, (what, k, e)
    for k, e in worst_el.items():
        # per-element: fp32 accumulation over up to 2560 (ensemble x batch) or 4096 (SLE) terms against fp64
        assert e < 1e-3, (what, "per-element", k, e)   # measured 1.4e-4 .. 2.6e-4
    print(f"{what}: worst leaf {max(worst.values()):.2e}, worst element (>1e-3 of max) {max(worst_el.values()):.2e}")


# C2 (headline) and the C4/C5-style keys with A = 7, both at the full shape, in every run (round 6: no longer behind SERL_SLOW)
@pytest.mark.parametrize("keys,A", [(KEYS, 6), (("wrist_1", "wrist_2"), 7)])
def test_update_critics_at_bench_shape(gpu, keys, A):
    cfg = _cfg(keys, A)
    st, core = AH.make_pair(cfg, B)
    b = AH.synth_batch(cfg, B, seed=21, frames_seed=70 if keys == KEYS else None)
    noise = O.make_noise(cfg, B, seed=22)
    torch.set_num_threads(max(1, torch.get_num_threads()))
    info, aux = O.update_critics(st, AH.batch_to_torch(b, torch.float64), O.noise_to_torch(noise, torch.float64))
    db = AH.batch_to_device(cfg, b)
    core.update_critics(db, AH.noise_to_device(cfg, noise))
    got = core.read_info()
    for k in ("critic_loss", "predicted_qs", "target_qs"):
        assert abs(got[k] - info[k]) < TOL * max(1.0, abs(info[k])), (k, got[k], info[k])
    q = core.debug("q", cfg.ensemble * B).reshape(cfg.ensemble, B)
    assert AH.rel_err(q, aux["q"].numpy()) < TOL
    assert AH.elem_rel_err(q, aux["q"].numpy(), floor=1e-2) < 1e-3
    assert AH.rel_err(core.debug("target_q", B), aux["target_q"].numpy()) < TOL
    # the [enc | action] critic input: 2 x 256 image codes (4x4 SLE -> Dense -> LN -> tanh) + 64 proprio + actions
    x = core.debug("x", B * (cfg.enc_dim + cfg.A)).reshape(B, -1)
    e_enc = AH.rel_err(x[:, :cfg.enc_dim], aux["enc_obs"].numpy())
    print(f"encoder output {keys}: rel err vs fp64 = {e_enc:.2e}")
    assert e_enc < TOL
    worst, worst_el = _grad_report(cfg, core, aux["grads"], "g_critic", 0)
    _assert_grads(worst, worst_el, f"critic grads {keys}")
    assert core.step == st.step == 1


def test_update_high_utd_at_bench_shape(gpu):
    cfg = _cfg()
    st, core = AH.make_pair(cfg, B)
    b = AH.synth_batch(cfg, B, seed=23, frames_seed=71)
    noise = O.make_noise(cfg, B, seed=24, utd_ratio=1)
    info, aux = O.update_high_utd(st, AH.batch_to_torch(b, torch.float64), O.noise_to_torch(noise, torch.float64), 1)
    db = AH.batch_to_device(cfg, b)
    core.update_high_utd(db, 1, AH.noise_to_device(cfg, noise))
    got = core.read_info()
    for k in ("critic_loss", "predicted_qs", "target_qs", "actor_loss", "temperature", "entropy", "temperature_loss"):
        assert abs(got[k] - info[k]) < TOL * max(1.0, abs(info[k])), (k, got[k], info[k])
    sl, _ = AH.leaf_slices(cfg)
    worst, worst_el = _grad_report(cfg, core, aux["g_actor"], "g_actor", sl["enc/proprio/dense/kernel"][0])
    _assert_grads(worst, worst_el, "actor grads")
    assert AH.rel_err(core.debug("logp", B), aux["logp"].numpy()) < TOL
    assert core.step == st.step == 2


def test_trunk_f16x3_bound_at_bench_shape(gpu):
    """The split-fp16 trunk is an fp32-class computation: bound it at 5e-6 of the fp64 oracle (measured 0.6-1.0e-6),
    not at the update's 1e-4 -- a 10x regression of the split arithmetic must fail."""
    cfg = O.Config(image_keys=("a",), H=128, W=128, S=4, A=2)
    st, core = AH.make_pair(cfg, 64, trunk_mode="f16x3")
    img = np.random.default_rng(5).integers(0, 256, (64, 128, 128, 3), dtype=np.uint8)
    ref = O.trunk_forward(st.trunk, torch.tensor(img), torch.float64).numpy()
    got = core.trunk_forward(torch.tensor(img, device="cuda")).cpu().numpy()
    err = AH.rel_err(got, ref)
    el = AH.elem_rel_err(got, ref, floor=1e-2)
    print(f"f16x3 trunk 128x128 n=64: {err:.2e} of max, worst element (>1e-2 of max) {el:.2e}")
    assert err < 5e-6 and el < 1e-4


# ---- BASELINE.json configs[2..4] at FULL shape: two HBM buffers, RLPD 50/50 ---------------------------
# (examples/async_drq_sim/async_drq_sim.py:238 `concat_batches(batch, demo_batch, axis=0)`, async_peg_insert_drq.py:355-366,
#  async_bin_relocation_fwbw_drq.py:508-519).  The batch goes through the product's real data path: two replay buffers in HBM,
#  bit-exact index draw, ONE fused gather + concat + unpack + random-shift launch, then the update -- against the NumPy replay
#  oracle feeding the fp64 update oracle.

def _two_buffers(cfg, sizes, caps, fills):
    import itertools
    from helpers import make_spaces
    from oracle.replay_oracle import ReplayOracle
    from serl_amd.data.data_store import MemoryEfficientReplayBufferDataStore
    from serl_amd.utils.synthetic import transition_stream
    osp, asp = make_spaces(cfg.image_keys, cfg.H, cfg.W, 3, 1, cfg.S, cfg.A)
    hip, ora = [], []
    for i, (cap, fill) in enumerate(zip(caps, fills)):
        rb = MemoryEfficientReplayBufferDataStore(osp, asp, cap, image_keys=cfg.image_keys)
        ro = ReplayOracle(cfg.image_keys, cfg.H, cfg.W, 3, 1, cfg.S, cfg.A, cap)
        rb.seed(i)        # online seed(0), demo seed(1)
        ro.seed(i)
        for tr in itertools.islice(transition_stream(cfg.image_keys, cfg.H, cfg.W, 3, 1, cfg.S, cfg.A, 100, 1234 + i), fill):
            rb.insert(tr)
            ro.insert(tr)
        hip.append(rb)
        ora.append(ro)
    return hip, ora


def _sample_two(cfg, hip, ora, sizes, crop_seed):
    """-> (DeviceBatch filled by ONE fused gather_crop launch over both buffers, the same batch from the NumPy oracle)"""
    from oracle.replay_oracle import random_shift
    from serl_amd.agents.batch import DeviceBatch
    from serl_amd.data.data_store import gather_crop
    B = sum(sizes)
    parts, ob = [], []
    for rb, ro, n in zip(hip, ora, sizes):
        idx = rb.sample_indices(n)
        assert (idx == ro.sample_indices(n)).all(), "index stream must be bit-exact"
        parts.append((rb, idx))
        ob.append(ro.gather(idx))
    rng = np.random.default_rng(crop_seed)
    co = rng.integers(0, 9, size=(B, 2)).astype(np.int32)
    cn = rng.integers(0, 9, size=(B, 2)).astype(np.int32)
    db = DeviceBatch(B, cfg.n_cam, cfg.H, cfg.W, 3, cfg.S, cfg.A, 0)
    gather_crop(parts, co, cn, db)
    cat = lambda f: np.concatenate([f(o) for o in ob], axis=0)   # noqa: E731  (concat_batches: online first)
    ref = {"obs": {k: random_shift(cat(lambda o: o["observations"][k][:, 0]), co) for k in cfg.image_keys},
           "next": {k: random_shift(cat(lambda o: o["observations"][k][:, 1]), cn) for k in cfg.image_keys},
           "state": cat(lambda o: o["observations"]["state"][:, 0]), "next_state": cat(lambda o: o["next_observations"]["state"][:, 0]),
           "action": cat(lambda o: o["actions"]), "reward": cat(lambda o: o["rewards"]), "mask": cat(lambda o: o["masks"]),
           "done": cat(lambda o: o["dones"])}
    return db, ref


def _assert_batch_bytes(cfg, db, ref):
    torch.cuda.synchronize()
    fr = db.frames.cpu().numpy()
    for c, k in enumerate(cfg.image_keys):
        assert (fr[0, c] == ref["obs"][k]).all() and (fr[1, c] == ref["next"][k]).all(), k
    assert (db.state[0].cpu().numpy() == ref["state"]).all() and (db.state[1].cpu().numpy() == ref["next_state"]).all()
    assert (db.action.cpu().numpy() == ref["action"]).all() and (db.reward.cpu().numpy() == ref["reward"]).all()
    assert (db.mask.cpu().numpy() == ref["mask"]).all() and (db.done.cpu().numpy().astype(bool) == ref["done"]).all()


@pytest.mark.parametrize("sizes", [(128, 128), (256, 256), (192, 64)])
def test_two_buffer_gather_crop_full_size_byte_exact(gpu, sizes):
    """128x128 frames, two cameras, online + demo buffers in one launch: bytes equal the NumPy oracle's
    sample -> concat_batches -> _unpack -> random shift (4 consecutive batches, so the RNG streams of both buffers advance)."""
    cfg = _cfg(("wrist_1", "wrist_2"), 6)
    hip, ora = _two_buffers(cfg, sizes, caps=(1200, 300), fills=(1000, 260))
    for trial in range(4):
        db, ref = _sample_two(cfg, hip, ora, sizes, crop_seed=100 + trial)
        _assert_batch_bytes(cfg, db, ref)


def _update_pair_checks(cfg, st, core, db, ref, Bt, what):
    """critic step then actor + temperature step on the same batch (= update_high_utd(utd_ratio=1)), every gradient leaf of
    both phases, q / target_q / logp and the info scalars at 1e-4 of the fp64 oracle."""
    noise = O.make_noise(cfg, Bt, seed=31, utd_ratio=1)
    tn = O.noise_to_torch(noise, torch.float64)
    tb = AH.batch_to_torch({k: v for k, v in ref.items() if k != "done"}, torch.float64)
    fo, fn = O.features(st, tb["obs"]), O.features(st, tb["next"])
    n = dict(tn)
    n["redq_idx"] = np.asarray(noise["redq_idx"]).reshape(-1, cfg.subsample)[0]
    cinfo, caux = O.critic_update(st, fo, fn, tb["state"], tb["next_state"], tb["action"], tb["reward"], tb["mask"], n)
    ainfo, aaux = O.actor_temp_update(st, fo, fn, tb["state"], tb["next_state"], tn)
    core.update_high_utd(db, 1, AH.noise_to_device(cfg, noise))
    got = core.read_info()
    info = dict(cinfo)
    info.update(ainfo)
    for k in ("critic_loss", "predicted_qs", "target_qs", "actor_loss", "temperature", "entropy", "temperature_loss"):
        assert abs(got[k] - info[k]) < TOL * max(1.0, abs(info[k])), (what, k, got[k], info[k])
    assert AH.rel_err(core.debug("target_q", Bt), caux["target_q"].numpy()) < TOL
    worst, worst_el = _grad_report(cfg, core, caux["grads"], "g_critic", 0)   # the actor phase leaves Gc untouched
    _assert_grads(worst, worst_el, f"{what}: critic grads")
    sl, _ = AH.leaf_slices(cfg)
    worst, worst_el = _grad_report(cfg, core, aaux["g_actor"], "g_actor", sl["enc/proprio/dense/kernel"][0])
    _assert_grads(worst, worst_el, f"{what}: actor grads")
    assert AH.rel_err(core.debug("logp", Bt), aaux["logp"].numpy()) < TOL
    assert core.step == st.step == 2


def test_two_buffer_update_at_bench_shape(gpu):
    """C3 / C4: 128 online + 128 demo samples, batch 256, 2 x 128x128 cameras."""
    cfg = _cfg(("wrist_1", "wrist_2"), 6)
    hip, ora = _two_buffers(cfg, (128, 128), caps=(1200, 300), fills=(1000, 260))
    st, core = AH.make_pair(cfg, B)
    db, ref = _sample_two(cfg, hip, ora, (128, 128), crop_seed=7)
    _assert_batch_bytes(cfg, db, ref)
    _update_pair_checks(cfg, st, core, db, ref, B, "two-buffer B=256")


def test_fwbw_batch_512_update(gpu):
    """C5 (async_bin_relocation_fwbw_drq): batch 512 = 256 online + 256 demo, keys front / wrist_1, A = 7: one trunk pass over
    2048 images, the update chain at 512 rows (5120 ensemble rows)."""
    cfg = _cfg(("front", "wrist_1"), 7)
    hip, ora = _two_buffers(cfg, (256, 256), caps=(1200, 400), fills=(1000, 350))
    st, core = AH.make_pair(cfg, 512)
    db, ref = _sample_two(cfg, hip, ora, (256, 256), crop_seed=8)
    _assert_batch_bytes(cfg, db, ref)
    _update_pair_checks(cfg, st, core, db, ref, 512, "fwbw B=512")
