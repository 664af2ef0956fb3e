"""The last-arriver fusion of the update chain (serl_amd/csrc/heads.hip "last-arriver epilogues"; reference math: agents/continuous/
sac.py:134-234, common/common.py:136-221, networks/mlp.py:10-32) against the SAME chain with one launch per operation
(SERL_CHAIN_FUSE=0, the round-3 schedule):

* the fused epilogues sum slabs in the order the separate kernels did, so on identical (injected) noise every gradient, loss
  scalar, parameter and Adam moment must agree TO THE BIT -- any visibility bug of the slab hand-off (a stale or torn slab)
  shows up as a non-zero difference, not as a tolerance question;
* 2000 consecutive updates of both chains in lockstep while a third agent's update chain and trunk passes keep the GPU busy on
  other streams (uneven load is where a broken hand-off goes stale), compared bit by bit every 50 steps;
* the launch count of a critic + actor update pair is pinned (48, was 63; the 38-launch form with LayerNorm epilogues was
  bit-identical and slower in every schedule: removed in round 5)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import drq_oracle as O
import agent_helpers as AH

pytestmark = pytest.mark.gpu


def _core(cfg, B, fuse, **kw):
    """an agent with the given chain variant (the switch is read when an agent is created)"""
    old = {k: os.environ.get(k) for k in ("SERL_CHAIN_FUSE",)}
    try:
        os.environ["SERL_CHAIN_FUSE"] = "1" if fuse else "0"
        return AH.make_pair(cfg, B, **kw)[1]
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _pair(cfg, B, **kw):
    """-> (fused core, un-fused core) with identical parameters"""
    return _core(cfg, B, True, **kw), _core(cfg, B, False, **kw)


def _launches():
    from serl_amd import _lib
    return int(_lib.lib().serl_debug_chain_launches())


def _bits_equal(a, b):
    return np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))


def _assert_state_bits(cfg, fused, plain, what, leaves=None):
    for k in (leaves or list(fused.leaves)):
        if k.startswith("trunk/"):
            continue
        for sec in ("params", "target_params", "opt/critic/mu", "opt/critic/nu", "opt/actor/mu", "opt/actor/nu"):
            assert _bits_equal(fused.get(sec, k), plain.get(sec, k)), (what, sec, k)


@pytest.mark.parametrize("shape", ["bench", "small", "odd"])
def test_fused_chain_is_bit_identical_to_the_unfused_chain(gpu, shape):
    if shape == "bench":
        cfg, B = O.Config(image_keys=("front", "wrist"), H=128, W=128, S=24, A=6), 256
    elif shape == "small":     # a rank's share of an 8-GPU job: half-empty 64-row tiles, deep K-splits
        cfg, B = O.Config(image_keys=("front", "wrist"), H=128, W=128, S=24, A=6), 32
    else:                       # 2x2 SLE, one camera, row counts that are no multiple of anything, A = 7
        cfg, B = O.Config(image_keys=("wrist_1",), H=64, W=64, S=19, A=7), 40
    fused, plain = _pair(cfg, B)
    sl, _ = AH.leaf_slices(cfg)
    pc = sl["enc/proprio/ln/bias"][1]
    pa0, pa1 = sl["enc/proprio/dense/kernel"][0], sl["actor/logstd/bias"][1]
    n_pair = {}
    for it in range(3):
        b = AH.synth_batch(cfg, B, seed=300 + it)
        noise = O.make_noise(cfg, B, seed=400 + it, utd_ratio=1)
        for name, core in (("fused", fused), ("plain", plain)):
            db, dn = AH.batch_to_device(cfg, b), AH.noise_to_device(cfg, noise)
            torch.cuda.synchronize()
            l0 = _launches()
            core.update_critics(db, dn)
            core.update_high_utd(db, 1, dn)
            torch.cuda.synchronize()
            n_pair[name] = _launches() - l0
        for tap, n in (("g_critic", pc), ("g_actor", pa1 - pa0), ("scalars", 8), ("q", cfg.ensemble * B), ("target_q", B), ("logp", B),
                       ("dx", B * (cfg.enc_dim + cfg.A))):
            assert _bits_equal(fused.debug(tap, n), plain.debug(tap, n)), (shape, it, tap)
        fi, pi = fused.read_info(), plain.read_info()
        assert fi == pi, (shape, it, fi, pi)
        _assert_state_bits(cfg, fused, plain, (shape, it))
    # update_critics (critic step) + update_high_utd(1) (critic step + actor/temperature step): 2 critic phases + 1 actor phase
    print(f"{shape}: chain launches per update_critics + update_high_utd: fused {n_pair['fused']}, one-per-operation {n_pair['plain']}")
    # (one-per-operation chain: 31 + 31 + 32 with device noise; two gen_noise launches fewer with injected noise, one reduce_slabs
    #  fewer per critic phase when the head gradient's K is short)
    assert n_pair["plain"] >= 29 + 29 + 31, n_pair
    assert n_pair["fused"] <= 22 + 22 + 26, n_pair     # default: a critic + actor pair = 22 + 26 = 48 (was 63)
    l0 = _launches()
    fused.update_critics(db)                            # device noise (production mode): hashed where it is used, no extra launch
    fused.update_high_utd(db, 1)
    torch.cuda.synchronize()
    assert _launches() - l0 == n_pair["fused"]
    assert all(np.isfinite(v) for v in fused.read_info().values())


def test_fused_chain_state_only_and_utd(gpu):
    """state-only SAC (per-member heads, K-split head gradient at 2048 rows) and UTD > 1 minibatches."""
    cfg = O.Config(image_keys=(), S=10, A=4, discount=0.99, warmup=4, temp_warmup=0)
    for B, utd in ((2048, 8), (48, 2)):
        fused, plain = _pair(cfg, B)
        for it in range(2):
            b = AH.synth_batch(cfg, B, seed=500 + it)
            noise = O.make_noise(cfg, B, seed=600 + it, utd_ratio=utd)
            for core in (fused, plain):
                core.update_high_utd(AH.batch_to_device(cfg, b), utd, AH.noise_to_device(cfg, noise))
            assert fused.read_info() == plain.read_info(), (B, utd, it)
        _assert_state_bits(cfg, fused, plain, ("state", B, utd))


def test_fused_chain_hand_off_survives_2000_steps_under_load(gpu):
    """Both chains in lockstep for 2000 update steps on the production schedule's shapes (B = 256 and a rank's 32), injected
    noise, while another agent runs update chains + trunk passes on two other streams; parameters compared bit by bit."""
    cfg = O.Config(image_keys=("front", "wrist"), H=128, W=128, S=24, A=6)
    steps = int(os.environ.get("SERL_FUSE_STRESS_STEPS", "2000"))
    for B, n_steps in ((256, steps // 2), (32, steps // 2)):
        fused, plain = _pair(cfg, B)
        _, loader = AH.make_pair(cfg, B, agent_seed=3)
        _, loader2 = AH.make_pair(cfg, B, agent_seed=4)      # (one agent per stream: an agent's workspaces are not re-entrant)
        batches = [AH.batch_to_device(cfg, AH.synth_batch(cfg, B, seed=700 + i)) for i in range(3)]
        noises = [AH.noise_to_device(cfg, O.make_noise(cfg, B, seed=800 + i, utd_ratio=1)) for i in range(3)]
        side = [torch.cuda.Stream(), torch.cuda.Stream()]
        check = ["critic/w1", "critic/head/kernel", "enc/0/sle", "enc/1/dense/kernel", "enc/proprio/dense/kernel", "actor/w2",
                 "actor/mean/kernel", "temp/lagrange"]
        for core in (fused, plain, loader):
            core.encode_slot(batches[0], 0)
            core.select_slot(0)
        # an agent's trunk workspace is not re-entrant: `loader` runs its next passes on side[0], which is not ordered behind the
        # default stream -- without this wait its first pass there could overlap the encode_slot above (two passes of ONE agent
        # sharing statistics, arrival counters and tile tickets: an intermittent device trap in the first process of a fresh box,
        # round 5)
        torch.cuda.synchronize()
        for it in range(n_steps):
            k = it % 3
            # load: the third agent's whole update (trunk + chain) on one stream, a bare trunk pass on another
            with torch.cuda.stream(side[0]):
                loader.update_critics(batches[k], noises[k])
            if it % 2 == 0:
                with torch.cuda.stream(side[1]):
                    loader2.encode_slot(batches[(k + 1) % 3], 1)
            for core in (fused, plain):   # the chain alone on the features of slot 0 (no trunk pass: 4x more steps per second)
                core.begin_update()
                if it % 4 == 3:
                    core.critic_grads(0, B, B, noises[k])
                    core.apply(1)
                    core.actor_grads(B, noises[k])
                    core.apply(6)
                else:
                    core.critic_grads(0, B, B, noises[k])
                    core.apply(1)
            if it % 50 == 49 or it == n_steps - 1:
                torch.cuda.synchronize()
                _assert_state_bits(cfg, fused, plain, (B, it), leaves=check)
                # the invariant every last-arriver epilogue starts from: all arrival counters back at zero (ADVICE r4 / VERDICT r5 item 8)
                assert fused.debug("ctr_nonzero", 1)[0] == 0 and loader.debug("ctr_nonzero", 1)[0] == 0, (B, it)
        torch.cuda.synchronize()
        _assert_state_bits(cfg, fused, plain, (B, "final"))


def test_two_agents_trunk_passes_on_two_streams_neither_deadlock_nor_differ(gpu):
    """Two agents in one process, their frozen-trunk passes issued concurrently on two streams (fw / bw learners sharing a GPU).
    The fused GroupNorm epilogues spin-wait for workgroups of their own launch; with two such launches resident at once the CUs
    can fill with waiters of both (the 2000-step test above found the trap).  trunk_f16x3.hip now lets only ONE pass at a time
    take the fused path (claim_fused_pass); the other runs the separate elementwise passes.  Features must equal the ones of
    passes issued alone."""
    cfg = O.Config(image_keys=("front", "wrist"), H=128, W=128, S=24, A=6)
    B = 256
    _, a0 = AH.make_pair(cfg, B, agent_seed=1)
    _, a1 = AH.make_pair(cfg, B, agent_seed=2)
    batches = [AH.batch_to_device(cfg, AH.synth_batch(cfg, B, seed=900 + i)) for i in range(2)]
    n = 2 * cfg.n_cam * B * 16 * 512
    ref = []
    for ag, db in ((a0, batches[0]), (a1, batches[1])):      # alone, one after the other
        ag.encode_slot(db, 0)
        ag.select_slot(0)
        torch.cuda.synchronize()
        ref.append(ag.debug("feats", n).copy())
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for it in range(60):
        for ag, db, st in ((a0, batches[0], streams[0]), (a1, batches[1], streams[1])):
            with torch.cuda.stream(st):
                ag.encode_slot(db, 0)
    torch.cuda.synchronize()
    for k, ag in enumerate((a0, a1)):
        got = ag.debug("feats", n)
        err = float(np.max(np.abs(got - ref[k])) / np.max(np.abs(ref[k])))
        assert err < 2e-6, (k, err)     # fused and un-fused passes differ by fp32 round-off only
