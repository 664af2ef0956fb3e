"""GPU parity of the HIP SAC/DrQ update (through the C ABI) against the fp64 CPU oracle.
Tolerance: 1e-4 (north-star: "within 1e-4 fp32"), measured as max-abs error relative to the
tensor's max-abs (per leaf for gradients and parameters)."""
import numpy as np
import pytest
import torch

from oracle import drq_oracle as O
import agent_helpers as AH

pytestmark = pytest.mark.gpu
TOL = 1e-4
import os
SEQ_TOL = float(os.environ.get("SERL_TEST_SEQ_TOL", "1e-4"))   # multi-step sequences: the same 1e-4 (measured on MI355X: passes)


MODES = ["f16x3", "f32"]   # split-fp16 trunk convs (default) and exact fp32 MFMA


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("H,W,n", [(64, 64, 6), (128, 128, 5), (128, 64, 3), (128, 128, 70)])
def test_trunk_forward(gpu, H, W, n, mode):
    cfg = O.Config(image_keys=("a",), H=H, W=W, S=4, A=2)
    st, core = AH.make_pair(cfg, B=max(n, 4), trunk_mode=mode)
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
    ref = O.trunk_forward(st.trunk, torch.tensor(img), torch.float64).numpy()
    got = core.trunk_forward(torch.tensor(img, device="cuda")).cpu().numpy()
    assert got.shape == ref.shape
    err = AH.rel_err(got, ref)
    print(f"trunk {mode} {H}x{W} n={n}: rel err vs fp64 = {err:.2e}")
    # both modes are fp32-class (measured: f16x3 0.6-1.0e-6, exact-fp32 MFMA 1.2-1.7e-6 of the fp64 oracle)
    assert err < 5e-6, err


@pytest.mark.parametrize("ksplit", [2, 4])
def test_trunk_forward_with_the_opt_in_conv_k_split(gpu, ksplit, monkeypatch):
    """SERL_CONV_KSPLIT: 2 / 4 workgroups per 64x64 tile of the small-M conv kernel, partial tiles summed by the last arriver
    (trunk_f16x3.hip; a rank's share of a data-parallel batch).  Same 5e-6 bound, and the plan must show the split."""
    monkeypatch.setenv("SERL_CONV_KSPLIT", str(ksplit))
    for H, n in ((64, 6), (128, 70)):
        cfg = O.Config(image_keys=("a",), H=H, W=H, S=4, A=2)
        st, core = AH.make_pair(cfg, B=max(n, 4), trunk_mode="f16x3")
        img = np.random.default_rng(1).integers(0, 256, (n, H, H, 3), dtype=np.uint8)
        ref = O.trunk_forward(st.trunk, torch.tensor(img), torch.float64).numpy()
        got = core.trunk_forward(torch.tensor(img, device="cuda")).cpu().numpy()
        plan = core.trunk_plan()
        assert any(len(v) == 5 and v[4] >= 2 for v in plan.values() if isinstance(v, tuple)), plan
        err = AH.rel_err(got, ref)
        print(f"trunk f16x3 {H}x{H} n={n} K-split {ksplit}: rel err vs fp64 = {err:.2e}")
        assert err < 5e-6, err


@pytest.mark.parametrize("n", [1024, 128])
def test_fused_projection(gpu, n, monkeypatch):
    """Default since round 5 (SERL_PROJ_FUSE=0 = separate launch): a block's 1x1 stride-2 projection computed by conv0's workgroups (conv_dma_f16x3_kernel<.., PROJ = true>;
    resnet_v1.py:129-156 -- the projection's pixel is conv0's tap (0, 0)).  At 1024 images all three projections ride
    (plan 'F'); at 128 images (a rank's share) the stages that leave the LDS-DMA kernel keep their own launch.  Features within
    5e-6 of the fp64 oracle and within fp32 round-off of the pass with separate projection launches."""
    cfg = O.Config(image_keys=("a",), H=128, W=128, S=4, A=2)
    st, core = AH.make_pair(cfg, B=max(n // 2, 4), trunk_mode="f16x3")
    img = torch.randint(0, 256, (n, 128, 128, 3), dtype=torch.uint8, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
    monkeypatch.setenv("SERL_PROJ_FUSE", "0")
    sep = core.trunk_forward(img).clone()
    monkeypatch.setenv("SERL_PROJ_FUSE", "1")
    fus = core.trunk_forward(img).clone()
    plan = core.trunk_plan()
    rode = [i for i in (1, 2, 3) if plan.get(f"b{i}_proj", ("?",))[0] == "F"]
    print(f"fused projection n={n}: plan {[(k, v) for k, v in plan.items() if k.endswith('proj')]}")
    if n == 1024:
        assert rode == [1, 2, 3], plan
    scale = float(sep.abs().max())
    assert float((fus - sep).abs().max()) / scale < 2e-6
    sel = list(range(6)) + list(range(n - 6, n))
    ref = O.trunk_forward(st.trunk, img[sel].cpu(), torch.float64).numpy()
    err = AH.rel_err(fus[sel].cpu().numpy(), ref)
    print(f"fused projection n={n}: rel err vs fp64 = {err:.2e}")
    assert err < 5e-6, err


@pytest.mark.parametrize("n", [1024, 128, 6])
def test_row_slab_kernels_fused_and_unfused(gpu, n, monkeypatch):
    """The stride-1 3x3 convs of stage 0 and b1_conv1 run on the row-slab kernels (resnet_v1.py:129-156): b0_conv0 on
    conv3x3_rowslab_f16x3_kernel when conv_init hands over its raw pooled output (GroupNorm + ReLU + split while a slab is staged,
    weights by LDS-DMA), b0_conv1 / b1_conv1 on conv3x3_slabdma_f16x3_kernel (slab and weights by global_load_lds); a fused launch
    stores through the row-major epilogue (rowtile_epilogue_t), an unfused one (SERL_GN_FUSE=0: the reference arithmetic of the
    fused GroupNorm exchange) the raw tile.  The plan reports tile-config 9 for these layers in both modes; fused and unfused
    features agree to the order of the fp64 statistics atomics and fma contraction, and both stay within 5e-6 of the fp64 oracle.
    (Round 6 removed the register-staged split8 row-slab kernel and the C-layout fused epilogue together with their switches
    SERL_SLAB_DMA / SERL_EPI_T: both had lost their same-call A/Bs twice, profiles/README.md.)"""
    cfg = O.Config(image_keys=("a",), H=128, W=128, S=4, A=2)
    st, core = AH.make_pair(cfg, B=max(n // 2, 4), trunk_mode="f16x3")
    img = torch.randint(0, 256, (n, 128, 128, 3), dtype=torch.uint8, device="cuda", generator=torch.Generator("cuda").manual_seed(6))
    monkeypatch.setenv("SERL_GN_FUSE", "0")
    plain = core.trunk_forward(img).clone()
    plan = core.trunk_plan()
    assert plan["b0_conv1"][:2] == ("S", 9) and plan["b1_conv1"][:2] == ("S", 9) and plan["b0_conv1"][3] == 0, plan
    monkeypatch.delenv("SERL_GN_FUSE")
    scale = float(plain.abs().max())
    for rep in range(3):
        fused = core.trunk_forward(img).clone()
        plan = core.trunk_plan()
        assert plan["b0_conv1"][:2] == ("S", 9) and plan["b1_conv1"][:2] == ("S", 9) and plan["b0_conv0"][0] == "S", plan
        if plan["raw_b0"]:
            assert plan["b0_conv0"][:2] == ("S", 9), plan
        if n >= 128:
            assert plan["b0_conv1"][3] == 1, plan      # the fused (row-major) epilogue ran
        assert float((fused - plain).abs().max()) / scale < 2e-6, rep
    sel = list(range(min(n, 6))) + list(range(max(n - 6, 0), n))
    ref = O.trunk_forward(st.trunk, img[sel].cpu(), torch.float64).numpy()
    for name, got in (("fused", fused), ("unfused", plain)):
        err = AH.rel_err(got[sel].cpu().numpy(), ref)
        print(f"row-slab kernels n={n} {name}: rel err vs fp64 = {err:.2e}; plan b0_conv0 {plan['b0_conv0']}, b0_conv1 {plan['b0_conv1']}, b1_conv1 {plan['b1_conv1']}")
        assert err < 5e-6, (name, err)


def _pretrained_like_trunk(trunk, seed=3):
    """Weight statistics a trained ImageNet ResNet with GroupNorm shows and kaiming-normal init does not: a wide
    per-output-channel spread of kernel magnitudes (nearly dead channels and a few very strong ones), first-layer
    filters of order 1, GroupNorm scales from slightly negative to several units, biases of a few units."""
    rng = np.random.default_rng(seed)
    out = {}
    for k, v in trunk.items():
        v = (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)).astype(np.float32)
        if v.ndim == 4:   # HWIO kernel
            co = v.shape[-1]
            g = np.exp(rng.normal(0.0, 1.2, co)).astype(np.float32)
            g[rng.random(co) < 0.05] *= 1e-3
            g[rng.random(co) < 0.02] *= 20.0
            if k.endswith("conv_init"):
                g *= 8.0
            v = v * g
        elif k.endswith("scale"):
            v = rng.uniform(-0.5, 4.0, v.shape).astype(np.float32)
            v[rng.random(v.shape) < 0.03] = 0.0
        else:
            v = rng.uniform(-2.0, 2.0, v.shape).astype(np.float32)
        out[k] = v
    return out


@pytest.mark.parametrize("mode", MODES)
def test_trunk_forward_with_pretrained_like_statistics(gpu, mode):
    """The split-fp16 arithmetic (hi + 2^-11 lo', no clamp on activations) on adversarial ranges: trained-network
    weight statistics and saturated / constant / high-contrast frames, against the fp64 oracle.  (Without the
    per-output-channel power-of-two weight scales of pack_weights_kernel this test fails with errors of order 1: strong
    first-layer filters overflow the folded fp16 weights, weak channels sink into fp16 subnormals and GroupNorm blows
    their error up.  Measured with the scales: <= 1.3e-6 on every frame.)"""
    H = W = 128
    cfg = O.Config(image_keys=("a",), H=H, W=W, S=4, A=2)
    st, core = AH.make_pair(cfg, B=8, trunk_mode=mode)
    hard = _pretrained_like_trunk(st.trunk)
    hard_t = {k: torch.tensor(v, dtype=torch.float64) for k, v in hard.items()}
    for sec in ("params", "target_params"):
        core.load_flat(sec, hard)
    rng = np.random.default_rng(2)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([
        np.zeros((H, W, 3)), np.full((H, W, 3), 255), np.repeat((((yy + xx) & 1) * 255)[..., None], 3, -1),
        np.repeat((xx * 2)[..., None], 3, -1), rng.integers(0, 256, (H, W, 3)), rng.integers(0, 2, (H, W, 3)) * 255,
        np.where(rng.random((H, W, 1)) < 0.01, 255, 0) * np.ones((1, 1, 3)), rng.integers(120, 124, (H, W, 3)),
    ]).astype(np.uint8)
    ref = O.trunk_forward(hard_t, torch.tensor(img), torch.float64).numpy()
    # the yardstick for badly conditioned frames: the same algorithm in plain fp32 (what the reference's XLA program computes in)
    ref32 = O.trunk_forward({k: v.float() for k, v in hard_t.items()}, torch.tensor(img), torch.float32).numpy()
    got = core.trunk_forward(torch.tensor(img, device="cuda")).cpu().numpy()
    assert np.isfinite(got).all()
    bad = []
    for i in range(len(img)):
        err, err32 = AH.rel_err(got[i], ref[i]), AH.rel_err(ref32[i], ref[i])
        print(f"trunk {mode} adversarial frame {i}: rel err vs fp64 = {err:.2e} (plain fp32 on the CPU: {err32:.2e}), "
              f"max |feature| = {np.abs(ref[i]).max():.3g}")
        # constant frames have near-zero GroupNorm variance in the first layers (E[x^2]-E[x]^2 cancels, rstd -> 1/sqrt(eps)):
        # every fp32 evaluation of the reference's algorithm is ill-conditioned there, so the bound is relative to plain fp32
        if err > max(5e-6, 4.0 * err32):
            bad.append((i, err, err32))
    assert not bad, bad


def _compare_state(cfg, st, core, tol=TOL, steps=1):
    """End-to-end parameter parity.  Adam's update -lr*m/(sqrt(v)+1e-8) is sign-like, hence
    ill-conditioned wherever |g| ~ 1e-8 (fp32 vs fp64 gradient noise flips it by up to 2*lr): the
    bulk (99.9th percentile) must match within `tol`, every element within the Adam bound, and
    the optimizer itself is checked exactly with injected gradients in test_adam_ema_injected."""
    worst = 0.0
    for k in st.params:
        for sec, tree in (("params", st.params), ("target_params", st.target)):
            got = core.get(sec, AH.product_name(k, cfg.image_keys)).astype(np.float64)
            ref = tree[k].numpy().reshape(-1)
            err = np.abs(got - ref)
            scale = max(np.max(np.abs(ref)), 1e-30)
            bulk = float(np.quantile(err, 0.999)) / scale
            worst = max(worst, bulk)
            assert bulk < tol, (sec, k, bulk)
            bound = 2.1 * cfg.lr * steps * (cfg.tau * steps if sec == "target_params" else 1.0) + tol * scale
            assert err.max() <= bound, (sec, k, err.max(), bound)
    return worst


def test_adam_ema_injected(gpu):
    """3x Adam (zero-gradient momentum steps included) + summed update + target EMA, elementwise
    against the oracle with the SAME injected gradients (common.py:124-168)."""
    cfg = O.Config(image_keys=("front", "wrist"), H=64, W=64, S=5, A=3, warmup=3)
    st, core = AH.make_pair(cfg, 4, dtype=torch.float64)
    sl, P = AH.leaf_slices(cfg)
    pc, pa0, pa1 = sl["enc/proprio/ln/bias"][1], sl["enc/proprio/dense/kernel"][0], sl["actor/logstd/bias"][1]
    rng = np.random.default_rng(0)
    H_target = cfg.target_entropy
    for it in range(6):
        crit = it % 3 != 2
        if crit:
            g = (rng.standard_normal(pc) * 10.0 ** rng.uniform(-9, -2, pc)).astype(np.float32)
            grads = {"critic": {k: torch.tensor(g[lo:hi].astype(np.float64)).reshape(st.params[k].shape)
                                for k, (lo, hi) in sl.items() if hi <= pc}}
            core.debug_set("g_critic", g)
            O.apply_gradients(st, grads)
            O.target_update(st)
            core.apply(1)
        else:
            g = (rng.standard_normal(pa1 - pa0) * 10.0 ** rng.uniform(-9, -2, pa1 - pa0)).astype(np.float32)
            sum_logp_next = np.float32(rng.standard_normal() * 3)
            lam = st.params["temp/lagrange"]
            H = -float(sum_logp_next) / 1.0
            gt = torch.sigmoid(lam) * (H - H_target)
            grads = {"actor": {k: torch.tensor(g[lo - pa0:hi - pa0].astype(np.float64)).reshape(st.params[k].shape)
                               for k, (lo, hi) in sl.items() if lo >= pa0 and hi <= pa1},
                     "temperature": {"temp/lagrange": gt}}
            sc = np.zeros(32, np.float32)
            sc[5] = sum_logp_next
            core.debug_set("g_actor", g)
            core.debug_set("scalars", sc)
            O.apply_gradients(st, grads)
            core.apply(6)   # SERL_NET_ACTOR | SERL_NET_TEMPERATURE
        for k in st.params:
            for sec, tree in (("params", st.params), ("target_params", st.target)):
                got = core.get(sec, AH.product_name(k, cfg.image_keys))
                ref = tree[k].numpy().reshape(-1)
                assert np.allclose(got, ref, rtol=2e-5, atol=2e-7), (it, sec, k, np.abs(got - ref).max())
    assert core.step == st.step == 6


def _check_grads(cfg, core, grads, tap, sl_lo, tol=TOL):
    sl, _ = AH.leaf_slices(cfg)
    pc = sl.get("enc/proprio/ln/bias", sl["critic/head/bias"])[1]   # end of the critic optimizer's support
    n = {"g_critic": pc, "g_actor": sl["actor/logstd/bias"][1] - sl_lo}[tap]
    g = core.debug(tap, n)
    for k, gv in grads.items():
        lo, hi = sl[k]
        e = AH.rel_err(g[lo - sl_lo:hi - sl_lo], gv.numpy().reshape(-1))
        assert e < tol, (tap, k, e)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("B", [16, 40])
def test_update_critics_matches_oracle(gpu, B, mode):
    cfg = O.Config(image_keys=("front", "wrist"), H=64, W=64, S=5, A=3)
    st, core = AH.make_pair(cfg, B, trunk_mode=mode)
    b = AH.synth_batch(cfg, B, seed=3)
    noise = O.make_noise(cfg, B, seed=7)
    info, aux = O.update_critics(st, AH.batch_to_torch(b, torch.float64), O.noise_to_torch(noise, torch.float64))
    db = AH.batch_to_device(cfg, b)
    core.update_critics(db, AH.noise_to_device(cfg, noise))
    got = core.read_info()
    for k in ("critic_loss", "predicted_qs", "target_qs"):
        assert abs(got[k] - info[k]) < TOL * max(1.0, abs(info[k])), (k, got[k], info[k])
    q = core.debug("q", cfg.ensemble * B).reshape(cfg.ensemble, B)
    assert AH.rel_err(q, aux["q"].numpy()) < TOL
    assert AH.rel_err(core.debug("target_q", B), aux["target_q"].numpy()) < TOL
    _check_grads(cfg, core, aux["grads"], "g_critic", 0)
    _compare_state(cfg, st, core)
    assert core.step == st.step == 1


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("utd", [1, 2])
def test_update_high_utd_matches_oracle(gpu, utd, mode):
    cfg = O.Config(image_keys=("front", "wrist"), H=64, W=64, S=5, A=3)
    B = 16
    st, core = AH.make_pair(cfg, B, trunk_mode=mode)
    b = AH.synth_batch(cfg, B, seed=4)
    noise = O.make_noise(cfg, B, seed=8, utd_ratio=utd)
    info, aux = O.update_high_utd(st, AH.batch_to_torch(b, torch.float64), O.noise_to_torch(noise, torch.float64), utd)
    db = AH.batch_to_device(cfg, b)
    core.update_high_utd(db, utd, AH.noise_to_device(cfg, noise))
    got = core.read_info()
    for k in ("critic_loss", "predicted_qs", "target_qs", "actor_loss", "temperature", "entropy", "temperature_loss"):
        assert abs(got[k] - info[k]) < TOL * max(1.0, abs(info[k])), (k, got[k], info[k])
    sl, _ = AH.leaf_slices(cfg)
    _check_grads(cfg, core, aux["g_actor"], "g_actor", sl["enc/proprio/dense/kernel"][0])
    _compare_state(cfg, st, core, steps=utd + 1)
    assert core.step == st.step == utd + 1


def test_multi_step_sequence(gpu):
    """CAR=4 style sequence (3x update_critics + 1x update_high_utd), twice: exercises the
    zero-gradient Adam momentum steps, the EMA and step/bias-correction bookkeeping."""
    cfg = O.Config(image_keys=("front", "wrist"), H=64, W=64, S=5, A=3)
    B = 8
    st, core = AH.make_pair(cfg, B)
    for it in range(8):
        b = AH.synth_batch(cfg, B, seed=100 + it)
        noise = O.make_noise(cfg, B, seed=200 + it)
        tb, tn = AH.batch_to_torch(b, torch.float64), O.noise_to_torch(noise, torch.float64)
        db, dn = AH.batch_to_device(cfg, b), AH.noise_to_device(cfg, noise)
        if it % 4 == 3:
            O.update_high_utd(st, tb, tn, 1)
            core.update_high_utd(db, 1, dn)
        else:
            O.update_critics(st, tb, tn)
            core.update_critics(db, dn)
    worst = _compare_state(cfg, st, core, tol=SEQ_TOL, steps=10)
    assert core.step == st.step == 10
    print("worst rel err after 10 steps:", worst)


def test_dp_split_equals_full_batch(gpu):
    """Batch-sharded data parallelism (common.py:213-214 pmean): gradients of two half batches
    (normalised by the global count) sum to the full-batch gradient."""
    cfg = O.Config(image_keys=("front", "wrist"), H=64, W=64, S=5, A=3)
    B = 16
    _, core = AH.make_pair(cfg, B)
    b = AH.synth_batch(cfg, B, seed=5)
    noise = AH.noise_to_device(cfg, O.make_noise(cfg, B, seed=9))
    db = AH.batch_to_device(cfg, b)
    sl, _ = AH.leaf_slices(cfg)
    n = sl["enc/proprio/ln/bias"][1]
    core.begin_update()
    core.encode(db)
    core.critic_grads(0, B, B, noise)
    full = core.debug("g_critic", n).astype(np.float64)
    sc_full = core.debug("scalars", 3).astype(np.float64)
    parts, scs = [], []
    for r in range(2):
        core.critic_grads(r * 8, 8, B, noise)
        parts.append(core.debug("g_critic", n).astype(np.float64))
        scs.append(core.debug("scalars", 3).astype(np.float64))
    assert AH.rel_err(parts[0] + parts[1], full) < 1e-5
    assert AH.rel_err(scs[0] + scs[1], sc_full) < 1e-5


def test_sample_actions(gpu):
    cfg = O.Config(image_keys=("front", "wrist"), H=64, W=64, S=5, A=3)
    st, core = AH.make_pair(cfg, 8)
    b = AH.synth_batch(cfg, 4, seed=6)
    frames = torch.tensor(np.stack([b["obs"][k] for k in cfg.image_keys]), device="cuda")
    state = torch.tensor(b["state"], device="cuda")
    feats = O.features(st, {k: torch.tensor(v) for k, v in b["obs"].items()})
    enc = O.encode(st.params, cfg, feats, torch.tensor(b["state"], dtype=torch.float64))
    mean, std = O.policy_head(st.params, cfg, enc)
    mode = core.sample_actions(frames, state, None).cpu().numpy()
    assert AH.rel_err(mode, torch.tanh(mean).numpy()) < TOL
    eps = np.random.default_rng(0).standard_normal((4, cfg.A)).astype(np.float32)
    a, _ = O.sample_and_log_prob(mean, std, torch.tensor(eps, dtype=torch.float64))
    got = core.sample_actions(frames, state, torch.tensor(eps, device="cuda")).cpu().numpy()
    assert AH.rel_err(got, a.numpy()) < TOL


def test_error_behaviour(gpu):
    from serl_amd._lib import SerlError
    cfg = O.Config(image_keys=("front",), H=64, W=64, S=5, A=3)
    _, core = AH.make_pair(cfg, 6)
    db = AH.batch_to_device(cfg, AH.synth_batch(cfg, 6))
    with pytest.raises(SerlError, match="divisible by UTD"):
        core.update_high_utd(db, 4)  # sac.py:561-563
    with pytest.raises(SerlError):
        core.set("params", "no/such/leaf", np.zeros(3))
    with pytest.raises(SerlError):
        core.set("opt/critic/mu", "actor/w1", np.ones(core.leaves["actor/w1"]))  # outside the support


def test_production_noise_runs(gpu):
    """noise=None: device RNG path (no parity claim, just finite results and moving params)."""
    cfg = O.Config(image_keys=("front", "wrist"), H=64, W=64, S=5, A=3)
    _, core = AH.make_pair(cfg, 8)
    db = AH.batch_to_device(cfg, AH.synth_batch(cfg, 8))
    before = core.get("params", "actor/w1").copy()
    core.update_critics(db)
    core.update_high_utd(db, 1)
    info = core.read_info()
    assert all(np.isfinite(v) for v in info.values()), info
    assert not np.array_equal(before, core.get("params", "actor/w1"))


def test_grad_view_aliases_library_memory(gpu):
    """The zero-copy torch view used for the RCCL all-reduce must alias the library's gradient buffer."""
    cfg = O.Config(image_keys=("front", "wrist"), H=64, W=64, S=5, A=3)
    _, core = AH.make_pair(cfg, 8)
    sl, _ = AH.leaf_slices(cfg)
    pc = sl["enc/proprio/ln/bias"][1]
    v = core.grad_view(1)
    assert v.is_cuda and v.dtype == torch.float32 and v.numel() == pc + 32
    g = np.random.default_rng(0).standard_normal(pc).astype(np.float32)
    core.debug_set("g_critic", g)
    assert np.array_equal(v[:pc].cpu().numpy(), g)
    v.mul_(2.0)                                    # what all_reduce(SUM) over 2 identical ranks would do
    torch.cuda.synchronize()
    assert np.array_equal(core.debug("g_critic", pc), 2 * g)
    va = core.grad_view(6)
    assert va.data_ptr() == v.data_ptr() + 4 * pc  # [scalars | actor grads] starts at the scalars


def test_pipelined_schedule_matches_serial(gpu):
    """trunk(i+1) overlapped with update(i) on a second stream gives bit-identical parameters -- also when the trunk pass
    is issued in two pieces and the update waits for the first piece of the NEXT pass (update_after_stage)."""
    import itertools
    from helpers import make_spaces
    from serl_amd.agents.batch import DeviceBatch
    from serl_amd.data.data_store import MemoryEfficientReplayBufferDataStore, gather_crop
    from serl_amd.parallel import DataParallelLearner, SerialSchedule, TorchPipelineSchedule
    from serl_amd.utils.synthetic import transition_stream
    cfg = O.Config(image_keys=("front", "wrist"), H=64, W=64, S=5, A=3)
    outs = []
    for sched_cls in (SerialSchedule, TorchPipelineSchedule, "split"):
        _, core = AH.make_pair(cfg, 8, agent_seed=5)
        osp, asp = make_spaces(cfg.image_keys, 64, 64, 3, 1, 5, 3)
        rb = MemoryEfficientReplayBufferDataStore(osp, asp, 200, image_keys=cfg.image_keys)
        rb.seed(0)
        for tr in itertools.islice(transition_stream(cfg.image_keys, 64, 64, 3, 1, 5, 3, 20, 1), 120):
            rb.insert(tr)
        dbs = [DeviceBatch(8, 2, 64, 64, 3, 5, 3, 0) for _ in range(3)]   # one per pipeline slot

        def gather(parts, co, cn, slot):
            gather_crop(parts, co, cn, dbs[slot])
            return dbs[slot]

        sched = sched_cls() if sched_cls is SerialSchedule else (
            TorchPipelineSchedule(torch.device("cuda", 0), update_after_stage=1) if sched_cls == "split" else sched_cls(torch.device("cuda", 0)))
        lr = DataParallelLearner(core, gather, [rb], [8], schedule=sched, seed=3)
        for _ in range(4):
            lr.iteration(critic_actor_ratio=2)
        torch.cuda.synchronize()
        outs.append({k: core.get("params", k) for k in ("critic/w1", "actor/w2", "enc/0/dense/kernel", "temp/lagrange")})
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k
        assert np.array_equal(outs[0][k], outs[2][k]), ("split pass", k)


def test_full_size_properties(gpu):
    """BASELINE.json's bench configuration (batch 256, 2 x 128x128x3, 24-dim state): the fp64 oracle is too
    slow here, so check size-independent properties -- split-fp16 and exact-fp32 trunks agree, shard
    gradients (2 and 8 ranks) sum to the full-batch gradient, and a learner iteration stays finite."""
    cfg = O.Config(image_keys=("wrist_1", "wrist_2"), H=128, W=128, S=24, A=7)
    B = 256
    _, core = AH.make_pair(cfg, B)
    b = AH.synth_batch(cfg, B, seed=11)
    noise = AH.noise_to_device(cfg, O.make_noise(cfg, B, seed=3))
    db = AH.batch_to_device(cfg, b)
    frames = db.frames.reshape(-1, cfg.H, cfg.W, 3)
    core.set_trunk_mode("f32")
    f32 = core.trunk_forward(frames).cpu().numpy()
    core.set_trunk_mode("f16x3")
    f16 = core.trunk_forward(frames).cpu().numpy()
    assert f32.shape == (4 * B, 4, 4, 512) and np.isfinite(f32).all()
    assert AH.rel_err(f16, f32) < 2e-5

    sl, _ = AH.leaf_slices(cfg)
    n = sl["enc/proprio/ln/bias"][1]
    core.begin_update()
    core.encode(db)
    core.critic_grads(0, B, B, noise)
    full = core.debug("g_critic", n).astype(np.float64)
    sc_full = core.debug("scalars", 3).astype(np.float64)
    assert np.isfinite(full).all() and np.abs(full).max() > 0
    for world in (2, 8):
        per = B // world
        acc, sc = np.zeros_like(full), np.zeros_like(sc_full)
        for r in range(world):
            core.critic_grads(r * per, per, B, noise)
            acc += core.debug("g_critic", n)
            sc += core.debug("scalars", 3)
        assert AH.rel_err(acc, full) < 1e-5, world
        assert AH.rel_err(sc, sc_full) < 1e-5, world

    w0 = core.get("params", "critic/w1").copy()
    core.update_critics(db, noise)
    core.update_high_utd(db, 1, noise)
    info = core.read_info()
    assert all(np.isfinite(v) for v in info.values()), info
    assert core.step == 3 and not np.array_equal(w0, core.get("params", "critic/w1"))


def test_rccl_all_reduce_on_the_zero_copy_gradient_view(gpu):
    """The data-parallel step all-reduces the library's gradient memory in place through a torch view that the
    caching allocator does not own.  A one-rank RCCL group runs the real ProcessGroupNCCL path (stream
    bookkeeping on foreign memory included) on this single GPU; N > 1 is covered on CPU with gloo."""
    import os
    import socket
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cfg = O.Config(image_keys=("front", "wrist"), H=64, W=64, S=5, A=3)
        _, core = AH.make_pair(cfg, 8)
        sl, _ = AH.leaf_slices(cfg)
        pc = sl["enc/proprio/ln/bias"][1]
        g = np.random.default_rng(0).standard_normal(pc).astype(np.float32)
        core.debug_set("g_critic", g)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for which in (1, 6):
                dist.all_reduce(core.grad_view(which))
        torch.cuda.synchronize()
        assert np.array_equal(core.debug("g_critic", pc), g)
    finally:
        dist.destroy_process_group()


def test_device_noise_is_indexed_by_the_global_sample(gpu):
    """Production noise (noise=None) in a batch-sharded job: with set_shard every rank draws, for each of its
    samples, the eps / dropout mask the single-GPU run draws for that sample, so the shard gradients still sum
    to the full-batch gradient (results independent of the world size)."""
    cfg = O.Config(image_keys=("front", "wrist"), H=64, W=64, S=5, A=3)
    B = 16
    sl, _ = AH.leaf_slices(cfg)
    n = sl["enc/proprio/ln/bias"][1]
    b = AH.synth_batch(cfg, B, seed=5)
    redq = {"redq_idx": np.array([[1, 7]], np.int32)}

    def run(parts):
        g = np.zeros(n, np.float64)
        per = B // parts
        for r in range(parts):
            _, core = AH.make_pair(cfg, per)     # a fresh agent per shard = one rank: same seed, same noise counter
            sub = {k: ({c: v[r * per:(r + 1) * per] for c, v in val.items()} if isinstance(val, dict) else val[r * per:(r + 1) * per])
                   for k, val in b.items()}
            db = AH.batch_to_device(cfg, sub)
            core.set_shard(r * per, B if parts > 1 else 0)
            core.begin_update()
            core.encode(db)
            core.critic_grads(0, per, B, redq)
            g += core.debug("g_critic", n)
        return g

    full, halves, quarters = run(1), run(2), run(4)
    assert np.abs(full).max() > 0
    assert AH.rel_err(halves, full) < 1e-5
    assert AH.rel_err(quarters, full) < 1e-5


# The shapes of BASELINE.json's other configurations at parity-test size:
#   C3 async_drq_sim + demos: 2 cams, S=7, A=4, CAR=8 (7x update_critics + 1x update_high_utd)
#   C4 peg insertion: wrist_1/wrist_2, S=19, A=6          C5 fwbw: front/wrist_1, S=19, A=7
#   literal "1 camera" variant of C2
@pytest.mark.parametrize("name,keys,S,A,car", [
    ("C2_one_cam", ("front",), 7, 4, 1),
    ("C3_demos_car8", ("front", "wrist"), 7, 4, 8),
    ("C4_peg", ("wrist_1", "wrist_2"), 19, 6, 2),
    ("C5_fwbw", ("front", "wrist_1"), 19, 7, 2),
])
def test_baseline_config_shapes_match_oracle(gpu, name, keys, S, A, car):
    cfg = O.Config(image_keys=keys, H=64, W=64, S=S, A=A)
    B = 8
    st, core = AH.make_pair(cfg, B)
    for it in range(car):
        b = AH.synth_batch(cfg, B, seed=50 + it)
        noise = O.make_noise(cfg, B, seed=60 + it)
        tb, tn = AH.batch_to_torch(b, torch.float64), O.noise_to_torch(noise, torch.float64)
        db, dn = AH.batch_to_device(cfg, b), AH.noise_to_device(cfg, noise)
        if it < car - 1:
            info, _ = O.update_critics(st, tb, tn)
            core.update_critics(db, dn)
            keys_ = ("critic_loss", "predicted_qs", "target_qs")
        else:
            info, _ = O.update_high_utd(st, tb, tn, 1)
            core.update_high_utd(db, 1, dn)
            keys_ = ("critic_loss", "predicted_qs", "target_qs", "actor_loss", "temperature", "entropy", "temperature_loss")
        got = core.read_info()
        for k in keys_:
            assert abs(got[k] - info[k]) < 2 * TOL * max(1.0, abs(info[k])), (name, it, k, got[k], info[k])
    _compare_state(cfg, st, core, tol=SEQ_TOL if car > 2 else TOL, steps=car + 1)
    assert core.step == st.step == car + 1


@pytest.mark.parametrize("mode", MODES)
def test_trunk_with_negative_and_zero_groupnorm_scales(gpu, mode):
    """The fused conv_init max-pool picks max or min of the raw conv output by the sign of the channel's GroupNorm
    scale (max_pool(relu(GN(x))) = relu(GN(extreme(x)))): exercise negative, zero and positive scales."""
    cfg = O.Config(image_keys=("a",), H=64, W=128, S=4, A=2)
    trunk, theta = O.init_params(cfg, 42)
    rng = np.random.default_rng(7)
    g = trunk["trunk/norm_init/scale"].copy()
    g[rng.random(64) < 0.4] *= -1.0
    g[:3] = 0.0
    trunk["trunk/norm_init/scale"] = g
    from serl_amd.agents.core import AgentCore
    core = AgentCore(n_cam=1, H=cfg.H, W=cfg.W, state_dim=cfg.S, act_dim=cfg.A, batch=8)
    core.set_trunk_mode(mode)
    for sec in ("params", "target_params"):
        core.load_flat(sec, trunk)
    st = O.TrainState(cfg, trunk, theta)
    img = rng.integers(0, 256, (6, cfg.H, cfg.W, 3), dtype=np.uint8)
    ref = O.trunk_forward(st.trunk, torch.tensor(img), torch.float64).numpy()
    got = core.trunk_forward(torch.tensor(img, device="cuda")).cpu().numpy()
    assert (g < 0).sum() > 10 and AH.rel_err(got, ref) < (TOL if mode == "f16x3" else 2e-5)


def test_fused_groupnorm_epilogue_is_race_free_at_full_batch(gpu, monkeypatch):
    """Stages 0-1 apply GroupNorm (+ residual) + ReLU + the split8 re-layout in the producing conv's epilogue: the 2..4
    workgroups of an image exchange statistics through atomics and an arrival counter while the kernel runs.  At the
    benchmark's 1024 images per pass (4096 workgroups, tiles handed out by tickets in completion order) the features must
    equal those of the separate elementwise passes (SERL_GN_FUSE=0) on every repetition -- a visibility race shows up as
    a 1e-4 error in a few images, intermittently."""
    cfg = O.Config(image_keys=("a",), H=128, W=128, S=4, A=2)
    st, core = AH.make_pair(cfg, B=512, trunk_mode="f16x3")
    n = 1024
    img = torch.randint(0, 256, (n, 128, 128, 3), dtype=torch.uint8, device="cuda")
    monkeypatch.setenv("SERL_GN_FUSE", "0")
    ref = core.trunk_forward(img).clone()
    scale = float(ref.abs().max())
    monkeypatch.setenv("SERL_GN_FUSE", "1")
    worst = 0.0
    for it in range(40):
        got = core.trunk_forward(img)
        worst = max(worst, float((got - ref).abs().max()) / scale)
    print(f"fused vs elementwise GroupNorm over 40 passes of {n} images: worst rel diff {worst:.2e}")
    assert worst < 2e-6, worst


def test_fused_epilogue_is_race_free_over_2000_passes_with_a_corunning_update(gpu, monkeypatch):
    """the cross-workgroup GroupNorm exchange (per-XCD tickets, memory-side statistics atomics, arrival
    counters, system-scope zeroing kernel) under the conditions of the pipelined learner -- 2000 consecutive trunk passes of
    1024 images on one stream while a second agent's update chain (its own small trunk passes, GEMMs, Adam) runs on another
    stream and competes for CUs, LDS and memory queues.  Every pass must reproduce the features of the separate
    elementwise GroupNorm passes (SERL_GN_FUSE=0); the worst deviation is accumulated on the device, no sync inside the loop."""
    cfg = O.Config(image_keys=("a",), H=128, W=128, S=4, A=2)
    st, core = AH.make_pair(cfg, B=512, trunk_mode="f16x3")
    n = 1024
    img = torch.randint(0, 256, (n, 128, 128, 3), dtype=torch.uint8, device="cuda")
    monkeypatch.setenv("SERL_GN_FUSE", "0")
    ref = core.trunk_forward(img).clone()
    scale = float(ref.abs().max())
    monkeypatch.setenv("SERL_GN_FUSE", "1")
    cfg2 = O.Config(image_keys=("front", "wrist"), H=64, W=64, S=5, A=3)
    _, core2 = AH.make_pair(cfg2, 32)
    db2 = AH.batch_to_device(cfg2, AH.synth_batch(cfg2, 32, seed=1))
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    worst = torch.zeros((), device="cuda")
    for it in range(2000):
        with torch.cuda.stream(sa):
            got = core.trunk_forward(img)
            worst = torch.maximum(worst, (got - ref).abs().max())
        if it % 2 == 0:
            with torch.cuda.stream(sb):
                core2.update_high_utd(db2, 1, None)
    torch.cuda.synchronize()
    w = float(worst) / scale
    print(f"fused vs elementwise GroupNorm over 2000 co-running passes of {n} images: worst rel diff {w:.2e}")
    assert np.isfinite(w) and w < 2e-6, w
    assert all(np.isfinite(v) for v in core2.read_info().values())


@pytest.mark.parametrize("H,n", [(64, 512), (128, 1024)])
def test_raw_stage0_input_path(gpu, monkeypatch, H, n):
    """With >= 512 images conv_init completes the 3x3/2 max-pool itself (whole-image chunks, neighbour rows / columns read back
    by the same workgroup) and block 0 consumes the RAW pooled tensor: GroupNorm + ReLU + split while the row-slab kernel stages
    its slabs, the residual rebuilt in b0_conv1's fused epilogue.  Checked against (a) the materialised path (SERL_GN_FUSE=0:
    completed pooling + elementwise GroupNorm pass + separate block-output pass) and (b) the fp64 oracle on the first and last
    images (resnet_v1.py:249-262)."""
    cfg = O.Config(image_keys=("a",), H=H, W=H, S=4, A=2)
    st, core = AH.make_pair(cfg, B=n // 2, trunk_mode="f16x3")
    img = torch.randint(0, 256, (n, H, H, 3), dtype=torch.uint8, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
    monkeypatch.setenv("SERL_GN_FUSE", "0")
    slow = core.trunk_forward(img).clone()
    monkeypatch.setenv("SERL_GN_FUSE", "1")
    fast = core.trunk_forward(img).clone()
    scale = float(slow.abs().max())
    d = float((fast - slow).abs().max()) / scale
    sel = list(range(6)) + list(range(n - 6, n))
    ref = O.trunk_forward(st.trunk, img[sel].cpu(), torch.float64).numpy()
    err = AH.rel_err(fast[sel].cpu().numpy(), ref)
    print(f"raw stage-0 path {H}x{H} n={n}: vs materialised path {d:.2e}, vs fp64 oracle {err:.2e}")
    assert d < 2e-6 and err < 5e-6
