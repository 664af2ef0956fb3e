#!/usr/bin/env python
"""Learner grad-steps/sec of the MI355X-native SERL hot path (BASELINE.json metric).

One "step" = one learner iteration of examples/async_drq_sim/async_drq_sim.py:266-292 with
critic_actor_ratio = --car (default 1, the "UTD=1" config): (car-1) x [sample -> gather+crop ->
update_critics] + 1 x [sample -> gather+crop -> update_high_utd(utd_ratio=1)], i.e. `car` critic
grad-steps (each on a fresh batch of 256) and one actor+temperature update.  Inputs (the replay
buffer) are resident in HBM before the timed region.  Synthetic transitions:
2 cameras 128x128x3 u8, 24-d state, 6-d action, random-init weights of the reference architecture.

N > 1: one process per GPU (torch.distributed / RCCL); the global batch of 256 is sharded over
ranks (strong scaling), every rank holds a replica of the replay buffer and the identical index
stream, gradients (+ loss scalars) are all-reduced once per update (common.py:213-214 pmean).
"""
import argparse
import itertools
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KEYS = ("front", "wrist")
H = W = 128
S, A, B = 24, 6, 256
# BASELINE.json configs restated as synthetic workloads.  buffers: (capacity, fill, seed,
# samples per batch) in concat order -- online first, then demos (examples/async_drq_sim/async_drq_sim.py:238
# `concat_batches(batch, demo_batch, axis=0)`).  The state is the bench's 24-d vector for every config (8(d): "bench: 24").
WORKLOADS = {
    # C2 async_drq_sim, the headline: one online buffer, batch 256, CAR=1 ("UTD=1")
    "drq": dict(name="async_drq_sim (DrQ, ResNet-10 frozen trunk, REDQ-10 critic)", keys=("front", "wrist"), A=6, car=1,
                buffers=[(200000, 20000, 0, 256)]),
    # C3 async_drq_sim + 20 demo trajectories (RLPD 50/50), CAR=8: 7 x update_critics + 1 x update_high_utd(1)
    "drq_demos": dict(name="async_drq_sim + 20 demo trajectories (RLPD 50/50 online+demo)", keys=("front", "wrist"), A=4, car=8,
                      buffers=[(200000, 20000, 0, 128), (2000, 2000, 1, 128)]),
    # C4 async_peg_insert_drq (examples/async_peg_insert_drq/async_peg_insert_drq.py:355-366): online 200k + demo 10k
    "peg": dict(name="async_peg_insert_drq (2 wrist cameras + proprio, RLPD 50/50)", keys=("wrist_1", "wrist_2"), A=6, car=8,
                buffers=[(200000, 20000, 0, 128), (10000, 2000, 1, 128)]),
    # C5 async_bin_relocation_fwbw_drq (:508-519): ONE of the two independent fw/bw learners, batch 512 = 256 + 256, CAR=4
    "fwbw": dict(name="async_bin_relocation_fwbw_drq (one of the two fw/bw learners, batch 512)", keys=("front", "wrist_1"), A=7,
                 car=4, buffers=[(200000, 20000, 0, 256), (5000, 2000, 1, 256)]),
}
PEAK_F32_MFMA = 157.3  # TFLOP/s, MI355X_MICROARCH.md (256 CU x 256 FLOP/clk x 2.4 GHz)
PEAK_F16_MFMA = 2500.0  # TFLOP/s dense fp16/bf16 MFMA (spec, MI355X_MICROARCH.md)
PEAK_HBM = 8.0         # TB/s spec


def scaling_projection():
    """The single-GPU projection of BOTH multi-GPU designs, so that one `--gpus N` line is directly comparable
    with what the pieces promised.  NOT measured by this process: profiles/scaling_pieces.json holds the pieces measured on ONE GPU
    (scripts/scaling_pieces.sh: `--emulate-world N` = one rank's share of a batch-sharded step, `--farm-role worker | updater`)."""
    path = os.path.join(ROOT, "profiles", "scaling_pieces.json")
    try:
        pc = json.load(open(path))
        one, upd, wrk = pc["one_gpu_ms"], pc["farm_updater_ms"], pc["farm_worker_ms"]
        rows = {}
        for n in (2, 4, 8):
            dp = pc["dp_share_ms"][str(n)]
            fm = max(upd, wrk / (n - 1))
            rows[str(n)] = {"dp_ms_before_collectives": dp, "dp_x": round(one / dp, 2), "farm_ms": round(fm, 4), "farm_x": round(one / fm, 2)}
        return {"by_n_gpus": rows, "one_gpu_ms": one, "commit": pc.get("commit"),
                "source": "profiles/scaling_pieces.json: pieces measured on ONE GPU (scripts/scaling_pieces.sh), not this run; "
                          "dp = one rank's share of the batch-sharded step BEFORE all-reduce time (17.6 MB + 0.9 MB per step), "
                          "farm = max(updater step, worker pass / (N - 1)); no N > 1 hardware curve exists"}
    except Exception:
        return None


def conv_macs_per_image():
    """MACs per 128x128 image of each instrumented conv launch."""
    m = {"conv_init": 64 * 64 * 64 * 147}
    hw, cin = 32, 64
    for i, (f, s) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2))):
        ho = hw // s
        m[f"conv_igemm/b{i}_conv0"] = ho * ho * f * 9 * cin
        m[f"conv_igemm/b{i}_conv1"] = ho * ho * f * 9 * f
        if s != 1 or cin != f:
            m[f"conv_igemm/b{i}_proj"] = ho * ho * f * cin
        hw, cin = ho, f
    return m


class _Sp:
    def __init__(self, shape):
        self.shape = tuple(shape)


class _DictSp:
    def __init__(self, spaces):
        self.spaces = spaces


PROFILE_EVERY = 4   # HIP events around every 4th launch of each kernel tag inside the timed region (every 16th in long runs: main())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=3,
                    help="timed repetitions of --steps iterations (each bracketed by barrier + synchronize); the MEDIAN is reported")
    ap.add_argument("--car", type=int, default=None, help="critic_actor_ratio (grad-steps per iteration); default: the workload's")
    ap.add_argument("--capacity", type=int, default=None, help="online buffer capacity (default 200000)")
    ap.add_argument("--fill", type=int, default=None, help="online buffer fill (default 20000)")
    ap.add_argument("--state-dim", type=int, default=24)
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the post-run check that the features the pipelined trunk produced (fused GroupNorm epilogues, update "
                         "chain co-running) equal a serial re-encode with the un-fused elementwise passes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--trunk", choices=["f16x3", "f32"], default="f16x3",
                    help="trunk conv arithmetic: split-fp16 MFMA (default, 1.3e-5 of fp64) or exact fp32 MFMA")
    ap.add_argument("--encoder", choices=["resnet-pretrained", "small"], default="resnet-pretrained",
                    help="resnet-pretrained: frozen ResNet-10 trunk (the official line); small: the trainable SmallEncoder "
                         "(vision/small_encoders.py:9-55; the north star's 'small ConvNet') -- side measurement with its own roofline")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="serial order: do not overlap the trunk of batch i+1 with the update of batch i")
    ap.add_argument("--workload", choices=list(WORKLOADS) + ["sac_state", "actor_latency"], default="drq",
                    help="drq: the official bench line (BASELINE configs[1]); drq_demos / peg / fwbw: configs[2..4] (two HBM "
                         "buffers, RLPD 50/50, CAR 8 / 8 / 4, batch 256 / 256 / 512); sac_state: side measurement of configs[0] "
                         "(state-only SAC); actor_latency: side measurement of the actor-side policy forward (sample_actions, batch 1)")
    ap.add_argument("--prio", choices=["auto", "update", "trunk", "none"], default="auto",
                    help="which stream gets the high-priority queue; auto: the trunk at large per-rank batches (the update "
                         "chain has slack there: 3.48 -> 3.44 ms), the latency-bound update chain at small ones")
    ap.add_argument("--update-after-stage", type=int, default=None,
                    help="start update(i) only when the trunk pass of batch i+1 has finished this residual stage (0..2; -1 = conv_init + pool); "
                         "default: off (measured slower, profiles/README.md)")
    ap.add_argument("--force-collective", action="store_true",
                    help="diagnostic: one-rank RCCL group, issue both all-reduces per step (launch-latency floor of the collectives)")
    ap.add_argument("--overlap-reduce", choices=["on", "off"], default="off",
                    help="N > 1: two gradient buckets reduced on a communication stream under the encoder-head backward (on) or "
                         "one all-reduce per update on the update stream itself (off)")
    ap.add_argument("--force-launcher", action="store_true",
                    help="go through torch.distributed.run (one rank per GPU, RCCL group) even with --gpus 1")
    ap.add_argument("--parallel", choices=["auto", "dp", "farm"], default="auto",
                    help="how --gpus N > 1 spreads the learner: dp = batch-sharded data parallelism with gradient all-reduce; "
                         "farm = step-pipelined trunk farm (serl_amd/parallel.py TrunkFarmLearner: rank 0 updates on the full batch, "
                         "ranks 1..N-1 run the frozen trunk of every (N-1)-th batch and send the features point to point); "
                         "auto (default) = dp for every N: BASELINE.json's north_star names batch-sharded DP + RCCL all-reduce as the "
                         "partition, so that is what an unlabelled `--gpus N` line measures; the farm is an explicit, labelled variant "
                         "(the JSON line's `projection` carries the single-GPU projection of both designs, DESIGN.md section 5)")
    ap.add_argument("--farm-role", choices=["updater", "worker"], default=None,
                    help="single-GPU measurement of ONE piece of the trunk farm: worker = gather + augment + trunk of every batch, no "
                         "update; updater = the update chain with nothing co-running, features arriving by a device-to-device copy")
    ap.add_argument("--noise", choices=["threefry", "hash"], default="threefry",
                    help="policy noise / Dropout masks: jax.random's threefry stream with the reference's key schedule, filled on the "
                         "device per update (default), or hashed inside the consuming kernels (no noise tensors).  Crop offsets and "
                         "REDQ indices are the reference's threefry integers either way")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="diagnostic: run ONE rank's share (B/world samples, no collective) of a world-size-N job")
    args = ap.parse_args()
    if args.workload == "sac_state":
        return sac_state_main(max(args.steps, 50))
    if args.workload == "actor_latency":
        return actor_latency_main(max(args.steps, 200))

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    launched = "WORLD_SIZE" in os.environ
    if not launched and (args.gpus > 1 or args.force_launcher):
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU over
        # RCCL), exactly as the driver's `python -m torch.distributed.run ... bench.py --gpus N` would
        return self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # never silently measure a different job than the one asked for
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, only {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local_rank)
    dist = None
    if launched or args.force_collective:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == world and dist.get_backend() == "nccl", "RCCL group does not span the job"
        # prove RCCL sees every rank: sum of (rank + 1) over the group
        probe = torch.tensor([float(rank + 1)], device="cuda")
        dist.all_reduce(probe)
        assert int(probe.item()) == world * (world + 1) // 2, f"RCCL all-reduce saw {probe.item()}, expected {world} ranks"
    wl = WORKLOADS[args.workload]
    KEYS, A, S = wl["keys"], wl["A"], args.state_dim
    car = args.car if args.car is not None else wl["car"]
    bufspec = [list(b) for b in wl["buffers"]]
    if args.capacity is not None:
        bufspec[0][0] = args.capacity
    if args.fill is not None:
        bufspec[0][1] = args.fill
    B = sum(b[3] for b in bufspec)
    farm = (args.parallel == "farm" and world > 1) or args.farm_role is not None   # (auto == dp: north_star's partition)
    assert farm or B % world == 0
    Bl = B if farm else B // world                     # a trunk farm keeps the FULL batch on every rank
    emu = args.emulate_world
    if emu:
        assert world == 1
        Bl = B // emu

    from serl_amd import _lib
    from serl_amd.agents.batch import DeviceBatch
    from serl_amd.agents.core import APPLY_ACTOR_TEMP, APPLY_CRITIC
    from serl_amd.data.data_store import MemoryEfficientReplayBufferDataStore, gather_crop
    from serl_amd.utils.launcher import make_drq_agent
    from serl_amd.utils.synthetic import transition_stream

    # ---- replay buffer resident in HBM (replicated on every rank; same content, same seed)
    spaces = {k: _Sp((1, H, W, 3)) for k in KEYS}
    spaces["state"] = _Sp((1, S))
    osp = _DictSp({k: spaces[k] for k in sorted(spaces)})
    rbs = []
    t0 = time.time()
    for bi, (cap, fill, rseed, _) in enumerate(bufspec):
        rb = MemoryEfficientReplayBufferDataStore(osp, _Sp((A,)), cap, image_keys=KEYS, device=local_rank)
        rb.seed(rseed)   # online seed(0), demo seed(1)
        for tr in itertools.islice(transition_stream(KEYS, H, W, 3, 1, S, A, 100, 1234 + bi), fill):
            rb.insert(tr)
        rbs.append(rb)
    fill_s = time.time() - t0

    sample_obs = {k: np.zeros((1, H, W, 3), np.uint8) for k in KEYS}
    sample_obs["state"] = np.zeros((1, S), np.float32)
    agent = make_drq_agent(42, sample_obs, np.zeros((A,), np.float32), image_keys=KEYS,
                           encoder_type=args.encoder, batch_size=Bl, device=local_rank)
    core = agent.core
    small = args.encoder == "small"
    if not small:
        core.set_trunk_mode(args.trunk)
    dbs = [DeviceBatch(Bl, len(KEYS), H, W, 3, S, A, local_rank) for _ in range(3)]   # one per pipeline slot

    def gather(parts, co, cn, slot):                     # fused K2+K3+K4 into the slot's device batch
        gather_crop(parts, co, cn, dbs[slot])
        return dbs[slot]

    from serl_amd.parallel import DataParallelLearner, SerialSchedule, TorchPipelineSchedule
    uas = args.update_after_stage
    if uas is not None and (uas < -1 or uas > 2 or args.trunk != "f16x3"):
        uas = None
    prio = args.prio if args.prio != "auto" else ("trunk" if Bl >= 128 else "update")
    sched = SerialSchedule() if args.no_pipeline else TorchPipelineSchedule(torch.device("cuda", local_rank), prioritise_update=prio == "update",
                                                                            prioritise_trunk=prio == "trunk",
                                                                            update_after_stage=uas)
    # gradient all-reduce (common.py:213-214 pmean made real), HIP events around every 4th call on the stream it runs on
    coll = {"calls": 0, "bytes": 0, "timed": [], "on": False}

    def all_reduce(t):
        if dist is None:
            return
        nb = t.numel() * t.element_size()
        timed = coll["on"] and coll.setdefault(nb, 0) % PROFILE_EVERY == 0   # every 4th call of each payload size
        if coll["on"]:
            coll[nb] += 1
            coll["calls"] += 1
            coll["bytes"] += nb
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(t)
            e1.record()
            coll["timed"].append((t.numel() * t.element_size(), e0, e1))
        else:
            dist.all_reduce(t)

    if farm:
        from serl_amd.parallel import TrunkFarmLearner

        class _Copied:      # emulated transfer: a device-to-device copy of one batch's features on a third stream
            def __init__(self):
                self.stream, self.ev, self.src = torch.cuda.Stream(device=local_rank), torch.cuda.Event(), None

            def __call__(self, t, peer, tag):
                if self.src is None:
                    self.src = torch.randn_like(t) * 0.1
                self.stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.stream):
                    t.copy_(self.src, non_blocking=True)
                    self.ev.record(self.stream)
                return self

            def wait(self):
                torch.cuda.current_stream().wait_event(self.ev)

        if args.farm_role is not None:
            assert world == 1
            send, recv = None, (_Copied() if args.farm_role == "updater" else None)
        else:
            send = lambda t, dst, tag: dist.isend(t, dst)       # noqa: E731  (RCCL point to point; tags are not used by the backend)
            recv = lambda t, src, tag: dist.irecv(t, src)       # noqa: E731
        learner = TrunkFarmLearner(core, gather, rbs, [b[3] for b in bufspec], rank, world, send=send, recv=recv, seed=7,
                                   schedule=sched, image_keys=KEYS, device_noise=args.noise, role=args.farm_role, chain_budget=1024)
    else:
        learner = DataParallelLearner(core, gather, rbs, [b[3] for b in bufspec], rank, emu if emu else world,
                                      all_reduce=all_reduce, seed=7, schedule=sched, overlap_reduce=args.overlap_reduce == "on",
                                      image_keys=KEYS, device_noise=args.noise)

    learner.force_reduce = args.force_collective or launched   # a launched 1-rank job still runs the RCCL path
    if os.environ.get("SERL_BENCH_DIAG") == "noproduced" and hasattr(sched, "ev_prod"):
        # TIMING-ONLY diagnostic (results are wrong: the update no longer waits for its features): the trunk stream carries no
        # event-record packet at the end of a pass -- does the idle time in front of the next pass go away?  (profiles/README.md)
        sched.produced = lambda slot: None
    hostprof = {}
    if os.environ.get("SERL_BENCH_HOSTPROF") == "1":   # diagnostic: where does the HOST spend an iteration (blocked or enqueueing)?
        def _wrap(obj, name):
            f = getattr(obj, name)
            def g(*a, **k):
                t = time.perf_counter()
                try:
                    return f(*a, **k)
                finally:
                    e = hostprof.setdefault(name, [0.0, 0])
                    e[0] += time.perf_counter() - t; e[1] += 1
            setattr(obj, name, g)
        for nm in ("wait_consumed", "wait_produced", "produced", "consumed"):
            if hasattr(sched, nm):
                _wrap(sched, nm)
        for nm in ("encode_slot", "select_slot", "begin_update", "critic_grads", "actor_grads", "apply"):
            _wrap(core, nm)
        _wrap(learner, "gather")
        _q = {"ready": 0, "n": 0}
        if hasattr(sched, "ev_cons"):
            _wc = sched.wait_consumed
            def _wc2(slot):
                ev = sched.ev_cons[slot]
                if ev is not None:
                    _q["n"] += 1; _q["ready"] += int(ev.query())
                return _wc(slot)
            sched.wait_consumed = _wc2
        hostprof["_cons_event_ready"] = _q

    def iteration():
        learner.iteration(car)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        iteration()
    barrier()
    # (SERL_BENCH_NOPROF=1: diagnostic run without the per-kernel HIP events -- measures their overhead)
    # The timing events are not free: an event pair around a launch serialises it against its neighbours (same call, 100 steps:
    # 2.535 ms with every 4th launch timed, 2.500 ms with none; profiles/r04_ab_step_boundary.log).  Long runs therefore sample
    # every 16th launch (>= 20 samples per kernel tag either way).
    global PROFILE_EVERY
    if args.steps * max(1, args.repeats) >= 320:
        PROFILE_EVERY = 16
    _lib.check(_lib.lib().serl_profile_enable(0 if os.environ.get("SERL_BENCH_NOPROF") == "1" else PROFILE_EVERY))
    _lib.check(_lib.lib().serl_profile_reset())
    coll["on"] = True
    dts = []
    for _ in range(max(1, args.repeats)):   # each repetition times EXACTLY --steps iterations
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            iteration()
        barrier()
        dts.append(time.perf_counter() - t0)
    coll["on"] = False
    if hostprof:
        sys.stderr.write("HOSTPROF " + json.dumps({k: (v if isinstance(v, dict) else {"ms_per_call": round(1e3 * v[0] / max(v[1], 1), 4), "calls": v[1]})
                                                   for k, v in hostprof.items()}) + "\n")
    prof = _lib.profile_read()
    _lib.check(_lib.lib().serl_profile_enable(0))
    torch.cuda.synchronize()
    # host cost of enqueueing one iteration on empty queues (diagnostic: is the chain launch-bound?)
    host_ms = []
    for _ in range(5):
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        iteration()
        host_ms.append((time.perf_counter() - h0) * 1e3)
    torch.cuda.synchronize()
    info = core.read_info()
    if not (farm and learner.role == "worker"):
        assert all(np.isfinite(v) for v in info.values()), info
    if world > 1:   # max over ranks, per repetition
        t = torch.tensor(dts, device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dts = [float(x) for x in t.tolist()]
    dt = float(np.median(dts))
    grad_steps = args.steps * car
    value = grad_steps / dt
    verify = None if (args.no_verify or args.no_pipeline or args.trunk != "f16x3" or small or farm) else verify_features(learner, core, dbs, car)

    # ---- roofline of the dominant kernel family (implicit-GEMM convs of the frozen trunk)
    macs = conv_macs_per_image()
    n_img = 2 * len(KEYS) * Bl
    per_kernel, tot_flop, tot_ms = {}, 0.0, 0.0
    # conv_init + pool finish run over sub-batches of the pass (Infinity-Cache-sized scratch): "conv_init" / "gn_relu_maxpool"
    # are then per SUB-BATCH launches, "conv_init_pool" the whole pair over the pass
    ci_parts = 1
    if "conv_init_pool" in prof and "conv_init" in prof:
        ci_parts = max(1, int(round(prof["conv_init"][1] / max(prof["conv_init_pool"][1], 1))))
    # fused projection (default; SERL_PROJ_FUSE=0 switches it off): a block's projection rides on its conv0 launch -- no b{i}_proj launch is timed, conv0's
    # duration covers both, so conv0 is credited with both FLOP counts
    rider = {f"conv_igemm/b{i}_conv0": f"conv_igemm/b{i}_proj" for i in range(1, 4)
             if f"conv_igemm/b{i}_proj" in macs and f"conv_igemm/b{i}_proj" not in prof and f"conv_igemm/b{i}_conv0" in prof}
    macs = dict(macs)
    for host, rid in rider.items():
        macs[host] = macs[host] + macs[rid]
    def parts_of(tag):
        if tag == "conv_init":
            return ci_parts
        return 1
    for tag, (ms, cnt) in sorted(prof.items()):
        ent = {"avg_us": 1e3 * ms / cnt, "timed_launches": cnt}
        if tag in rider:
            ent["includes"] = rider[tag]
        if tag in ("conv_init", "gn_relu_maxpool") and ci_parts > 1:
            ent["launches_per_pass"] = ci_parts
        if parts_of(tag) > 1 and tag != "conv_init":
            ent["launches_per_pass"] = parts_of(tag)
            ent["pass_us"] = 1e3 * ms / cnt * parts_of(tag)
        if tag in macs:
            fl = 2.0 * macs[tag] * n_img / parts_of(tag)
            ent["tflops"] = fl / (ms / cnt * 1e-3) / 1e12
            if tag.startswith("conv_igemm"):
                tot_flop += fl * cnt
                tot_ms += ms
        if tag == "gather_crop":
            by = 2 * Bl * len(KEYS) * 2 * H * W * 3 + Bl * ((2 * S + A + 2) * 4 + 1)
            ent["TBps"] = by / (ms / cnt * 1e-3) / 1e12
            ent["frac_hbm"] = ent["TBps"] / PEAK_HBM
        per_kernel[tag] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in ent.items()}
    achieved = tot_flop / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
    # HBM bytes per launch of the family from the PMC counters.  NOT measured by this process: rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE passes (scripts/r06_evidence.sh, serial schedule, possibly another box) leave them in profiles/pmc_traffic.json;
    # `traffic_source` says so in the line, with the commit the passes ran on.  The rocprof-derived fraction (profiles/r06_frac_from_stats.txt) rides along the same way.
    traffic, traffic_source, frac_rocprof = None, None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            pj = json.load(open(pmc))
            traffic = pj.get(f"conv_igemm_{args.trunk}_bytes_per_launch")
            traffic_source = (f"profiles/pmc_traffic.json (commit {pj.get('commit')}): separate rocprofv3 --pmc passes (scripts/r06_evidence.sh, serial "
                              "schedule, another box), not this run")
        except Exception:
            traffic = None
    try:
        import re as _re
        ff = next(f for f in ("r06_frac_from_stats.txt", "r05_frac_from_stats.txt") if os.path.exists(os.path.join(ROOT, "profiles", f)))
        txt = open(os.path.join(ROOT, "profiles", ff)).read()
        vals = [float(m) for m in _re.findall(r"frac ([0-9.]+)", txt)]     # first: pipelined schedule, second: serial
        frac_rocprof = dict(zip(("pipelined", "serial"), vals))
        cm = _re.search(r"# commit (\S+)", txt)
        frac_rocprof["source"] = (f"profiles/{ff} (commit {cm.group(1) if cm else 'of round 5'}; rocprofv3 --kernel-trace --stats CSVs, "
                                  "scripts/frac_from_stats.py), not this run")
    except Exception:
        frac_rocprof = None
    n_launch = max(1, sum(c for t, (m, c) in prof.items() if t.startswith("conv_igemm")))
    # executed-MFMA fraction per stage (b0: row-patch kernel, b1..b3: LDS-DMA kernel + 1x1 projection)
    frac_by_stage = {}
    for st in range(4):
        fl = sum(2.0 * macs[t] * n_img / parts_of(t) * c for t, (m, c) in prof.items() if t.startswith(f"conv_igemm/b{st}_") and t in macs)
        ms_ = sum(m for t, (m, c) in prof.items() if t.startswith(f"conv_igemm/b{st}_") and t in macs)
        if ms_ > 0:
            frac_by_stage[f"b{st}"] = round((3.0 if args.trunk == "f16x3" else 1.0) * fl / (ms_ * 1e-3) / 1e12 /
                                            (PEAK_F16_MFMA if args.trunk == "f16x3" else PEAK_F32_MFMA), 4)
    if args.trunk == "f16x3":
        # every fp32 product is three fp16 MFMA products (hi*hi, hi*lo', lo'*hi): the matrix pipe executes
        # 3x the algorithmic FLOPs, and that executed rate is what the fp16-MFMA roofline bounds
        executed = 3.0 * achieved
        roofline = {"bound": "mfma",
                    "kernel": (f"block convs of the frozen trunk, {11 - len(rider)} launches per pass: conv3x3_rowslab_f16x3_kernel (b0_conv0, raw input) + "
                               "conv3x3_slabdma_f16x3_kernel (b0_conv1, b1_conv1: the same row-slab tile staged by LDS-DMA) + conv_dma_f16x3_kernel (every other 3x3 conv; the three 1x1 projections ride on their block's conv0 launch "
                               "unless SERL_PROJ_FUSE=0)" if Bl * len(KEYS) * 2 >= 1024 else
                               f"block convs of the frozen trunk, {11 - len(rider)} launches per pass: conv3x3_rowslab_f16x3_kernel / conv3x3_slabdma_f16x3_kernel (stage 0, b1_conv1), "
                               "conv_dma_f16x3_kernel where at least 512 128-row tiles exist, conv_igemm_f16x3_kernel (64x64 tiles) otherwise") +
                              "; split-fp16 MFMA implicit GEMM, fp32 accumulate",
                    "achieved": round(executed, 3), "peak": PEAK_F16_MFMA, "unit": "TFLOP/s",
                    "frac": round(executed / PEAK_F16_MFMA, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "frac_from_rocprof_csv": frac_rocprof,
                    "algorithmic_tflops": round(achieved, 3),
                    "algorithmic_vs_f32_mfma_peak": round(achieved / PEAK_F32_MFMA, 4),
                    "note": "achieved = 3 x algorithmic fp32 conv FLOP/s (executed fp16 MFMA work); "
                            "algorithmic_tflops = 2*M*K*N per launch / HIP-event duration; "
                            f"HIP events around every {PROFILE_EVERY}th launch of each kernel inside the timed region; the durations of "
                            "the stage-0/1 convs INCLUDE the GroupNorm + residual + ReLU + split8 re-layout they apply in their epilogue "
                            "(SERL_GN_FUSE=0 restores the separate elementwise passes: higher frac, slower step)",
                    "frac_by_stage": frac_by_stage,
                    "flop_per_launch_avg": tot_flop / n_launch, "per_kernel": per_kernel}
    else:
        roofline = {"bound": "mfma",
                    "kernel": "conv_igemm_kernel (exact fp32 MFMA implicit GEMM, 11 launches per trunk pass)",
                    "achieved": round(achieved, 3), "peak": PEAK_F32_MFMA, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_F32_MFMA, 4), "traffic": traffic,
                    "flop_per_launch_avg": tot_flop / n_launch, "per_kernel": per_kernel}
    if not small:
        # the WHOLE step against the same peak: algorithmic FLOPs of one grad step (trunk 2 passes + heads / MLPs)
        # and the FLOPs the matrix pipes execute for them (3 fp16 products per block-conv FLOP, 2 per conv_init FLOP, 6 bf16
        # products per update-chain FLOP), both over the measured step time
        trunk_alg = 2.0 * sum(macs.values()) * n_img
        ci_alg = 2.0 * macs["conv_init"] * n_img
        chain_alg = 10.15e9 * (Bl / 256.0) * (len(KEYS) / 2.0)
        mult = 3.0 if args.trunk == "f16x3" else 1.0
        step_s = dt / grad_steps
        alg = (trunk_alg + chain_alg) / step_s / 1e12
        exe = ((trunk_alg - ci_alg) * mult + ci_alg * (2.0 if args.trunk == "f16x3" else 1.0) + chain_alg * 6.0) / step_s / 1e12
        roofline["whole_step"] = {"algorithmic_tflops": round(alg, 2), "executed_tflops": round(exe, 2),
                                  "frac": round(exe / (PEAK_F16_MFMA if args.trunk == "f16x3" else PEAK_F32_MFMA), 4),
                                  "note": "per grad step of this rank: (trunk 2 passes + ~10.15 GFLOP of heads / MLPs at B=256) / ms_per_step"}
    if small:
        roofline = small_encoder_roofline(prof, per_kernel, Bl, len(KEYS))
    if "gather_crop" in per_kernel:
        roofline["sample_aug_hbm"] = {"bound": "hbm", "kernel": "gather_crop_rgb_kernel", "achieved": per_kernel["gather_crop"]["TBps"],
                                      "peak": PEAK_HBM, "unit": "TB/s", "frac": per_kernel["gather_crop"]["frac_hbm"]}

    out = {
        "metric": f"learner grad-steps/sec (DrQ, bs{B}, {len(KEYS)}x128x128 img)" + (" [SmallEncoder]" if small else ""), "value": round(value, 3),
        "unit": "grad-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None,
        "dtype": ("f32 (all GEMMs incl. the SmallEncoder convs as bf16x3: exact three-way bf16 split, six piece products, fp32 accumulate)" if small
                  else "f32 (trunk convs as split-fp16 MFMA x3, fp32 accumulate; <=1e-6 vs fp64)" if args.trunk == "f16x3" else "f32"),
        "data": "synthetic",
        "config": {"workload": wl["name"].replace("ResNet-10 frozen trunk", "trainable SmallEncoder") if small else wl["name"], "workload_key": args.workload, "global_batch": B,
                   "per_gpu_batch": Bl, "cameras": len(KEYS), "image_keys": list(KEYS), "image": [H, W, 3], "state_dim": S, "act_dim": A,
                   "critic_actor_ratio": car, "utd_ratio": 1,
                   "buffers": [{"capacity": c_, "fill": f_, "seed": s_, "samples_per_batch": n_} for c_, f_, s_, n_ in bufspec],
                   "replay_capacity": bufspec[0][0], "replay_fill": bufspec[0][1], "parallelism": (f"trunk farm, {args.farm_role} role measured alone on one GPU" if args.farm_role else
                                                                                                      f"trunk farm: 1 updater + {world - 1} trunk workers, features point to point" if farm else
                                                                                                      f"dp{world}" + (f" (emulating 1 rank of dp{emu}, no collective)" if emu else "")), "grad_steps_per_step": car,
                   "random_stream": ("jax.random threefry2x32 with the reference's key schedule: crop offsets, REDQ indices"
                                     + (", policy noise, Dropout masks" if learner.device_noise == "threefry" else "; policy noise / Dropout masks hashed on the device")),
                   "encoder": args.encoder, "trunk_passes_per_grad_step": 0 if small else 2, "trunk_arithmetic": "f32" if small else args.trunk,
                   "schedule": "serial" if args.no_pipeline else (
                       "trunk(i+1) overlapped with update(i) on a 2nd stream" + ("; gather(i+2) on a 3rd" if getattr(sched, "gather_stream", None) is not None else "")),
                   # (every SERL_* switch set for this run: a variant line says so itself)
                   "env_switches": {k: v for k, v in sorted(os.environ.items()) if k.startswith("SERL_")}},
        "roofline": roofline,
        "repeats": len(dts), "ms_per_step_runs": [round(1e3 * x / args.steps, 4) for x in dts],
        "last_info": {k: round(float(v), 6) for k, v in info.items()},
        "host_enqueue_ms_per_iteration": round(min(host_ms), 4), "setup": {"replay_fill_s": round(fill_s, 2)},
    }
    if dist is not None:
        by_size = {}
        for nbytes, e0, e1 in coll["timed"]:
            by_size.setdefault(nbytes, []).append(e0.elapsed_time(e1) * 1e3)
        out["collective"] = {
            "backend": "nccl (RCCL)", "world_size": dist.get_world_size(), "launcher": "torch.distributed.run",
            "all_reduces_per_step": coll["calls"] / max(args.steps * len(dts), 1),
            "bytes_per_step": coll["bytes"] / max(args.steps * len(dts), 1),
            "avg_us_by_bytes": {str(k): round(float(np.mean(v)), 2) for k, v in sorted(by_size.items())},
            "overlap_reduce": args.overlap_reduce,
            "note": ("per critic update two overlapped all-reduce(SUM) buckets on a communication stream -- [ensemble | Q head | proprio | "
                     "loss scalars] issued while the encoder-head backward still runs, then [encoder heads] -- " if args.overlap_reduce == "on"
                     else "per critic update ONE all-reduce(SUM) of [critic grads | loss scalars] on the update stream (no stream crossing; "
                          "--overlap-reduce on = two buckets on a communication stream, measured slower at B/8) ")
                    + "and one of [scalars | actor grads] per actor update; HIP events on the stream the collective is enqueued on, every 4th call"}
    if verify is not None:
        out["verify"] = verify
    proj = scaling_projection()
    if proj is not None and args.farm_role is None and not emu:
        out["projection"] = proj
    diag = os.environ.get("SERL_BENCH_DIAG") or ("hostprof" if os.environ.get("SERL_BENCH_HOSTPROF") == "1" else "") or (
        f"farm-role-{args.farm_role}" if args.farm_role else "")
    if diag:
        # a diagnostic run is never a bench line: no `metric` / `value` / `unit` (SERL_BENCH_DIAG=noproduced runs updates that did
        # not wait for their features -- its numbers time a wrong computation)
        out = {"diagnostic": True, "diag": diag, "diagnostic_ms_per_step": out["ms_per_step"],
               **{k: v for k, v in out.items() if k not in ("metric", "value", "unit", "ms_per_step", "vs_baseline", "higher_is_better")}}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not diag:
        out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, KEYS, S, A, B)
    # the JSON line must be the last thing on stdout: librccl prints its version banner through C stdio, which is
    # block-buffered when stdout is a pipe/file and would otherwise surface after this line at exit -- every rank
    # flushes before the final barrier, rank 0 prints after it
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def small_encoder_flops_per_image(H=H, W=W):
    """SmallEncoder (vision/small_encoders.py:9-55): 4 x (3x3, stride 2, VALID) convs 3 -> 32 -> 64 -> 128 -> 256 (+ bias column).
    -> (forward FLOPs, backward FLOPs = weight gradient of every layer + input gradient of layers 1..3) per image."""
    feat, h, w = (3, 32, 64, 128, 256), H, W
    fwd = bwd = 0
    for l in range(4):
        h, w = (h - 3) // 2 + 1, (w - 3) // 2 + 1
        fl = 2 * h * w * 9 * feat[l] * feat[l + 1]
        fwd += fl
        bwd += fl * (2 if l > 0 else 1)
    return fwd, bwd


def small_encoder_roofline(prof, per_kernel, Bl, n_cam):
    """Roofline of the trainable SmallEncoder's conv stack: every launch group `small_encoder_fwd` is one pass of
    n_cam * Bl images through the four convs (im2col + fp32-MFMA GEMM + ReLU), `small_encoder_bwd` one backward pass."""
    fwd, bwd = small_encoder_flops_per_image()
    tot_fl = tot_ms = 0.0
    for tag, fl in (("small_encoder_fwd", fwd), ("small_encoder_bwd", bwd)):
        if tag in prof:
            ms, cnt = prof[tag]
            per_kernel[tag]["tflops"] = round(fl * n_cam * Bl / (ms / cnt * 1e-3) / 1e12, 3)
            tot_fl += fl * n_cam * Bl * cnt
            tot_ms += ms
    alg = tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
    ach = 6.0 * alg   # bf16x3: six bf16 piece products per fp32 product
    return {"bound": "mfma", "kernel": "SmallEncoder conv stack: gemm_bf16x3_kernel as implicit GEMM (layers 1-3: im2col rows gathered from the NHWC "
            "activations by the operand loader, ReLU in the epilogue; layer 0: explicit im2col of the u8 frames) + col2im, "
            "forward x3 per critic step and x3 per actor step, backward x1 per critic step", "achieved": round(ach, 3),
            "peak": PEAK_F16_MFMA, "unit": "TFLOP/s", "frac": round(ach / PEAK_F16_MFMA, 4), "traffic": None,
            "algorithmic_tflops": round(alg, 3), "algorithmic_vs_f32_mfma_peak": round(alg / PEAK_F32_MFMA, 4),
            "flop_per_image": {"forward": fwd, "backward": bwd}, "per_kernel": per_kernel,
            "note": "durations are HIP events around the whole pass (all launches of the four layers); achieved = executed bf16 MFMA rate "
                    "(6 piece products per fp32 product) against the dense bf16 peak; M is large and N = 32..256, K = 28..1153: these "
                    "GEMMs are bound by operand traffic, not by the matrix pipe"}


def verify_features(learner, core, dbs, car, iters=12, tol=2e-6):
    """A racy build must not produce a bench line: after the timed region, run `iters` more pipelined iterations and, after
    each, compare the features of the batch the side stream has just encoded (fused GroupNorm epilogues, update chain
    co-running on the other stream) with a serial re-encode of the same device batch through the separate elementwise
    GroupNorm passes (SERL_GN_FUSE=0).  Raises SystemExit on a mismatch above `tol` of the features' max-abs."""
    cfg = core.cfg
    hf = cfg.H
    for _ in range(5):
        hf = (hf + 1) // 2
    wf = cfg.W
    for _ in range(5):
        wf = (wf + 1) // 2
    n = 2 * cfg.n_cam * cfg.batch * hf * wf * 512
    worst, checked = 0.0, 0
    old = os.environ.get("SERL_GN_FUSE")
    for _ in range(iters):
        learner.iteration(car)
        torch.cuda.synchronize()
        slot = learner._pending
        if slot is None:
            return None
        core.select_slot(slot)
        got = core.debug("feats", n).copy()
        os.environ["SERL_GN_FUSE"] = "0"
        try:
            core.encode_slot(dbs[slot], slot)
            torch.cuda.synchronize()
        finally:
            if old is None:
                del os.environ["SERL_GN_FUSE"]
            else:
                os.environ["SERL_GN_FUSE"] = old
        ref = core.debug("feats", n)
        scale = float(np.abs(ref).max())
        worst = max(worst, float(np.abs(got - ref).max()) / max(scale, 1e-30))
        checked += 1
        if not np.isfinite(worst) or worst > tol:
            raise SystemExit(f"bench.py --verify: pipelined features differ from the serial re-encode by {worst:.3e} of max "
                             f"(tolerance {tol:.1e}) -- refusing to print a bench line")
    return {"batches_checked": checked, "worst_rel_diff": worst, "tol": tol,
            "what": "features of the side-stream trunk pass (fused GroupNorm epilogues, update chain co-running) vs a serial "
                    "re-encode with SERL_GN_FUSE=0"}


def self_launch(n):
    """Re-exec this script under torch.distributed.run with one rank per GPU (RCCL over xGMI).  Refuses, non-zero,
    when the node has fewer than `n` GPUs -- never falls back to a smaller world."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        print(f"bench.py: --gpus {n} needs {n} visible GPUs, found {have}", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--force-launcher"]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def actor_latency_main(iters):
    """Side measurement (NOT the official bench line), next-row N3: the actor's policy forward
    `agent.sample_actions(obs, seed=...)` / `argmax=True` (sac.py:301-320; call sites async_drq_sim.py:126-136 and the
    eval loop) on ONE observation (2 x 128x128x3 + 24-d state): host wall time per call incl. the H2D of the observation
    and the D2H of the action, and the device-only time of the enqueued kernels."""
    from serl_amd.utils.launcher import make_drq_agent
    obs0 = {"front": np.zeros((1, H, W, 3), np.uint8), "wrist": np.zeros((1, H, W, 3), np.uint8), "state": np.zeros((1, S), np.float32)}
    agent = make_drq_agent(42, obs0, np.zeros((A,), np.float32), image_keys=KEYS, encoder_type="resnet-pretrained", batch_size=8)
    rng = np.random.default_rng(0)
    obs = {"front": rng.integers(0, 256, (1, H, W, 3), dtype=np.uint8), "wrist": rng.integers(0, 256, (1, H, W, 3), dtype=np.uint8),
           "state": rng.standard_normal((1, S)).astype(np.float32)}
    for _ in range(20):
        agent.sample_actions(obs, argmax=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        a = agent.sample_actions(obs, seed=np.array([0, i], np.uint32))
    wall = (time.perf_counter() - t0) / iters
    # device-only: frames already in HBM, no readback
    fr = torch.tensor(np.stack([obs[k].reshape(1, H, W, 3) for k in KEYS]), device="cuda")
    st = torch.tensor(obs["state"].reshape(1, -1), device="cuda")
    eps = torch.zeros((1, A), device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        agent.core.sample_actions(fr, st, eps)
    e1.record()
    torch.cuda.synchronize()
    dev = e0.elapsed_time(e1) / iters
    print(json.dumps({"workload": "actor-side sample_actions, batch 1 (2 x 128x128x3 + 24-d state)", "calls": iters,
                      "host_ms_per_call": round(wall * 1e3, 4), "device_ms_per_call": round(dev, 4),
                      "actions_per_s": round(1.0 / wall, 1), "action": [round(float(x), 4) for x in np.asarray(a).reshape(-1)],
                      "note": "a real actor steps its env at 10-20 Hz: the policy forward is not its bottleneck"}))


def sac_state_main(iters):
    """Side measurement (NOT the official bench line): BASELINE.json configs[0] `async_sac_state_sim` -- state-only SAC,
    batch 256 x UTD 8 = 2048 sampled per iteration, `update_high_utd(utd_ratio=8)`
    (examples/async_sac_state_sim/async_sac_state_sim.py:231,296), plain replay buffer in HBM -- next to the CPU port
    (oracle, PyTorch-CPU fp32) on the box's host cores."""
    S, A, B, UTD = 10, 4, 2048, 8

    class _Box:
        def __init__(self, shape):
            self.shape = shape

    class _Env:
        observation_space, action_space = _Box((S,)), _Box((A,))

    from serl_amd.utils.launcher import make_replay_buffer, make_sac_agent
    from serl_amd.utils.synthetic import flat_stream
    rb = make_replay_buffer(_Env(), capacity=1_000_000, type="replay_buffer")
    rb.seed(0)
    for tr in itertools.islice(flat_stream(S, A, 100, 1234), 20000):
        rb.insert(tr)
    agent = make_sac_agent(42, np.zeros((S,), np.float32), np.zeros((A,), np.float32), batch_size=B)
    it = rb.get_iterator(sample_args={"batch_size": B, "lazy": True})
    for _ in range(20):
        agent.update_high_utd(next(it), utd_ratio=UTD)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        agent.update_high_utd(next(it), utd_ratio=UTD)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"workload": "async_sac_state_sim (state-only SAC, 2048 = 256 x UTD 8 per iteration)", "iterations": iters,
           "ms_per_iteration": round(1e3 * dt / iters, 4), "critic_grad_steps_per_s": round(UTD * iters / dt, 2),
           "kernels": "same HIP kernels as the DrQ update chain (latency-bound: ~75 dependent launches per grad-step pair)"}
    # CPU port: the oracle in fp32 on the host cores
    from oracle import drq_oracle as O
    ncpu = _effective_cpus()
    torch.set_num_threads(ncpu)
    cfg = O.Config(image_keys=(), S=S, A=A, discount=0.99, warmup=2000, temp_warmup=0)
    _, theta = O.init_params(cfg, 42)
    st = O.TrainState(cfg, {}, theta, torch.float32)
    rng = np.random.default_rng(0)

    def cpu_iter():
        b = {"obs": {}, "next": {}, "state": torch.tensor(rng.standard_normal((B, S)), dtype=torch.float32),
             "next_state": torch.tensor(rng.standard_normal((B, S)), dtype=torch.float32),
             "action": torch.tensor(rng.uniform(-1, 1, (B, A)), dtype=torch.float32),
             "reward": torch.tensor(rng.random(B), dtype=torch.float32), "mask": torch.ones(B)}
        n = O.noise_to_torch(O.make_noise(cfg, B, seed=int(rng.integers(1 << 30)), utd_ratio=UTD), torch.float32)
        O.update_high_utd(st, b, n, UTD)
    cpu_iter()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 8.0:
        cpu_iter()
        n += 1
    cdt = time.perf_counter() - t0
    out["cpu_port"] = {"critic_grad_steps_per_s": round(UTD * n / cdt, 2), "cores": ncpu, "iterations": n,
                       "kind": "port (oracle, PyTorch-CPU fp32)"}
    out["speedup"] = round(out["critic_grad_steps_per_s"] / out["cpu_port"]["critic_grad_steps_per_s"], 1)
    print(json.dumps(out))




def cpu_baseline(budget_s, KEYS=KEYS, S=S, A=A, B=B):
    """The oracle (CPU restatement of the reference, PyTorch-CPU fp32, all host cores) timed on a
    bounded sample of the SAME workload: full-size update_high_utd(utd_ratio=1) steps incl. NumPy
    replay sampling.  kind = "port": jax/flax are not installable, so the reference itself cannot run."""
    from oracle import drq_oracle as O
    from oracle.replay_oracle import ReplayOracle, random_shift
    from serl_amd.utils.synthetic import transition_stream
    torch.set_num_threads(_effective_cpus())
    cfg = O.Config(image_keys=KEYS, H=H, W=W, S=S, A=A)
    trunk, theta = O.init_params(cfg, 42)
    st = O.TrainState(cfg, trunk, theta, torch.float32)
    ro = ReplayOracle(KEYS, H, W, 3, 1, S, A, 2000)
    ro.seed(0)
    for tr in itertools.islice(transition_stream(KEYS, H, W, 3, 1, S, A, 100, 1234), 600):
        ro.insert(tr)
    noise_np = O.make_noise(cfg, B, 7)

    def step():
        b = ro.sample(B)
        bt = {"obs": {k: torch.from_numpy(random_shift(b["observations"][k][:, 0], noise_np["crop_obs"])) for k in KEYS},
              "next": {k: torch.from_numpy(random_shift(b["observations"][k][:, 1], noise_np["crop_next"])) for k in KEYS},
              "state": torch.from_numpy(b["observations"]["state"][:, 0]),
              "next_state": torch.from_numpy(b["next_observations"]["state"][:, 0]),
              "action": torch.from_numpy(b["actions"]), "reward": torch.from_numpy(b["rewards"]),
              "mask": torch.from_numpy(b["masks"])}
        O.update_high_utd(st, bt, O.noise_to_torch(noise_np, torch.float32), 1)

    t0 = time.perf_counter()
    step()  # warm-up (also the whole sample if the host is so slow that one step exceeds the budget)
    n, el = 1, time.perf_counter() - t0
    if el < budget_s:
        n, t0 = 0, time.perf_counter()
        while True:
            step()
            n += 1
            el = time.perf_counter() - t0
            if el >= budget_s or n >= 8:
                break
    return {"value": round(n / el, 4), "unit": "grad-steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} full-size update_high_utd(utd=1) steps (B={B}, {len(KEYS)}x128x128x3; 2 trunk passes per update = algorithmic minimum, the reference does 3-5) "
                      f"incl. NumPy replay sampling, PyTorch-CPU fp32, {el:.1f} s",
            "cpu": _cpu_name()}


def _effective_cpus():
    """Host cores this process may actually use: min(affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def _cpu_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


if __name__ == "__main__":
    sys.exit(main() or 0)
