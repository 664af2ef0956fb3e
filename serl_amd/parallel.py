"""Batch-sharded data parallelism + software pipelining for the learner: one process per GPU,
gradients all-reduced over RCCL/xGMI (torch.distributed backend "nccl").  This makes the reference's
dormant `jax.lax.pmean(grads_and_aux, pmap_axis)` (serl_launcher/common/common.py:213-214) real.

Sharding contract (DESIGN.md "multi-GPU"):
  * every rank holds a full replica of the replay buffer(s) with the same contents and the same
    seed, so all ranks draw the IDENTICAL global index stream (bit-exact with 1 GPU); transitions that
    arrive from the actor while training are replicated by serl_amd/data/replicated.py (rank 0 fans them
    out, every rank applies them at the same batch boundary);
  * rank r owns samples [r*B/P, (r+1)*B/P) of the (concatenated online+demo) global batch;
  * losses are normalised by the GLOBAL batch, so all-reduce(SUM) of [gradients | loss scalars]
    equals the single-device gradient; every rank then applies the identical Adam/EMA update
    (no parameter broadcast needed).

Pipelining: the frozen trunk depends only on the batch's pixels, never on the trainable parameters,
so batch i+1 is sampled, gathered, augmented and pushed through the trunk on a second stream while
the heads / backward / optimizer of batch i run on the main stream (the reference's iterator also
samples two batches ahead, data/replay_buffer.py:77-90).  Results are identical to the serial order.

The class only touches its collaborators through small duck-typed interfaces, so the sharding and
collective plumbing is testable on CPU with the gloo backend (tests/test_parallel_cpu.py).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

APPLY_CRITIC, APPLY_ACTOR_TEMP = 1, 6   # SERL_NET_CRITIC, SERL_NET_ACTOR | SERL_NET_TEMPERATURE


def shard_parts(parts: Sequence[Tuple[object, np.ndarray]], rank: int, world: int):
    """parts: [(buffer, idx[n_i])] in concat order (global batch = sum n_i).  Returns this rank's
    slice as parts plus the (lo, hi) global sample range."""
    total = sum(len(ix) for _, ix in parts)
    assert total % world == 0, f"global batch {total} not divisible by world size {world}"
    per = total // world
    lo, hi = rank * per, (rank + 1) * per
    out, start = [], 0
    for buf, ix in parts:
        a, b = max(lo, start), min(hi, start + len(ix))
        if a < b:
            out.append((buf, ix[a - start:b - start]))
        start += len(ix)
    return out, (lo, hi)


class SerialSchedule:
    """No overlap: everything on the caller's stream."""
    slots = 1

    def side(self):
        import contextlib
        return contextlib.nullcontext()

    def main(self):
        import contextlib
        return contextlib.nullcontext()

    def gathered(self, slot, fn):
        return fn()

    def produced(self, slot):
        pass

    def wait_produced(self, slot):
        pass

    def consumed(self, slot):
        pass

    def wait_consumed(self, slot):
        pass


class TorchPipelineSchedule:
    """HIP streams + events: producer (trunk) on `side`, its gather one batch ahead on `gather_stream`, consumer on the current stream.

    THREE batch slots: the pass of batch i+2 reuses the slot of batch i-1, and the host confirms that update(i-1) is done
    with a host-side event wait (normally already complete: the host then runs at most two updates ahead of the device)
    instead of making the side stream wait for update(i) on the device -- a dependency that crosses streams costs 60-100 us
    on this stack and sat on the trunk stream's critical path at every pass boundary."""
    slots = 3

    def __init__(self, device, prioritise_update=True, prioritise_trunk=False, update_after_stage=None):
        """update_after_stage = s (0..2): the update of batch i starts only when the trunk pass of batch i+1 has finished
        its stage s (the pass is issued in two pieces with an event between them), so the update chain's small kernels
        co-run with the later, register-lighter conv kernels instead of conv_init / stage 0."""
        import torch
        self.torch = torch
        self.device = device
        self.update_after_stage = update_after_stage
        self.ev_mid = [torch.cuda.Event() for _ in range(self.slots)]
        # the update's long chain of small dependent kernels gets the high-priority queue so that it is
        # not starved by the (throughput-bound) trunk kernels of the next batch
        self.side_stream = torch.cuda.Stream(device=device, priority=-1 if prioritise_trunk else 0)
        self.main_stream = torch.cuda.Stream(device=device, priority=-1) if prioritise_update else None
        # (round 5: the update stream confined to a CU mask -- hipExtStreamCreateWithCUMask, first n bits -- so that the chain's
        #  workgroups stretch fewer of the trunk's: 128 / 96 / 64 / 32 CUs -> 2.546 / 2.649 / 2.947 / 3.719 ms vs 2.409 / 2.434 unmasked,
        #  same call; the stage-0 convs do get faster (b0_conv1 390 -> 355 us) but the chain becomes the critical path: removed)
        # (round 5: events created with hipEventDisableSystemFence -- these and every event of the library -- left the step and the
        #  idle time at the pass boundary unchanged: 2.4702 / 2.4698 -> 2.4790 / 2.4534 ms, profiles/README.md)
        self.ev_prod = [torch.cuda.Event() for _ in range(self.slots)]
        self.ev_cons = [None] * self.slots
        # sample -> gather -> augment of batch i+2 runs on a THIRD stream: the host issues it while the trunk pass of batch i+1 is still
        # running, so the pass of batch i+2 starts at conv_init instead of behind a 22 us (37 us co-running) gather, and its index /
        # offset upload is off the trunk stream too.  Same-call A/Bs on three boxes (profiles/r05_ab_gather_stream.txt): -2.9 % (2.422 / 2.404 -> 2.339 /
        # 2.347 ms per step), +0.25 % (noise) and -0.4 %: never slower beyond noise; features verified against a serial re-encode.  The slot's buffers are free by then: the host has waited for
        # update(i-1), the last reader of that slot.
        self.gather_stream = torch.cuda.Stream(device=device)
        self.ev_gather = [torch.cuda.Event() for _ in range(self.slots)]

    def side(self):
        return self.torch.cuda.stream(self.side_stream)

    def gathered(self, slot, fn):
        """run the gather `fn` (called with the side stream current) -- on the gather stream when there is one, the side stream then
        waits for it"""
        if self.gather_stream is None:
            return fn()
        with self.torch.cuda.stream(self.gather_stream):
            out = fn()
            self.ev_gather[slot].record(self.gather_stream)
        self.side_stream.wait_event(self.ev_gather[slot])
        return out

    def main(self):
        import contextlib
        return self.torch.cuda.stream(self.main_stream) if self.main_stream is not None else contextlib.nullcontext()

    def produced(self, slot):
        self.ev_prod[slot].record(self.side_stream)

    def mid_produced(self, slot):
        self.ev_mid[slot].record(self.side_stream)

    def wait_mid(self, slot):
        self.torch.cuda.current_stream(self.device).wait_event(self.ev_mid[slot])

    def wait_produced(self, slot):
        self.torch.cuda.current_stream(self.device).wait_event(self.ev_prod[slot])

    def consumed(self, slot):
        ev = self.torch.cuda.Event()
        ev.record(self.torch.cuda.current_stream(self.device))
        self.ev_cons[slot] = ev

    def wait_consumed(self, slot):
        if self.ev_cons[slot] is not None:
            self.ev_cons[slot].synchronize()   # host side; the update that used this slot ended two passes ago


class DataParallelLearner:
    """core: object with begin_update/encode_slot/select_slot/critic_grads/actor_grads/apply/grad_view
    (AgentCore); gather: callable(parts, crop_obs, crop_next, slot) -> device batch for that slot."""

    def __init__(self, core, gather, buffers: List[object], batch_sizes: List[int], rank: int = 0,
                 world: int = 1, all_reduce=None, seed: int = 0, ensemble: int = 10, schedule=None,
                 overlap_reduce: bool = False, image_keys=None, device_noise: str = "hash"):
        """seed: the reference's `make_drq_agent(seed, ...)` -- state.rng starts where DrQAgent.create_drq leaves it and advances
        with the reference's key schedule (serl_amd/jaxrng.py); crop offsets and REDQ indices of every step are the integers a JAX
        learner with that seed draws.  device_noise = "threefry": policy noise and Dropout masks are jax.random's too (this rank's
        rows of the global (B, A) normals / (B, 4096) masks, filled by one launch per update; needs image_keys) -- "hash": hashed
        inside the kernels that consume them from cfg.seed (no noise tensors)."""
        self.core, self.gather, self.buffers, self.batch_sizes = core, gather, buffers, batch_sizes
        self.rank, self.world = rank, world
        self.B = sum(batch_sizes)
        assert self.B % world == 0
        self.Bl = self.B // world
        self.all_reduce = all_reduce
        self.ensemble = ensemble
        self.sched = schedule or SerialSchedule()
        # the reference's random stream, identical on all ranks (same seed): state.rng as create_drq leaves it (drq.py:69-84)
        from . import jaxrng as J
        self._J = J
        self._rng = J.split(J.split(J.prngkey(seed))[0])[1]
        self._keys = None            # keys of the call in progress
        self.image_keys = tuple(image_keys) if image_keys is not None else None
        assert device_noise in ("hash", "threefry")
        self.device_noise = device_noise if (self.image_keys is not None and hasattr(core, "cfg")) else "hash"
        self._nbuf = None
        self.noise_form = "keys"     # "keys": drawn inside the consuming kernels; "tensors": one serl_jax_fill launch per update
        self.last_draws = {}
        self._gv = {}
        self._pending = None   # slot of the prefetched (sampled + gathered + encoded) batch
        self.force_reduce = False   # diagnostic: issue the collectives even with one rank
        if world > 1 and hasattr(core, "set_shard"):
            core.set_shard(rank * self.Bl, self.B)   # device noise indexed by the global sample id
        self._next_slot = 0
        # the update chain co-runs with the next batch's trunk pass: at a large per-rank batch the trunk is the critical path and
        # the chain has slack, so its GEMM launches use fewer workgroups (profiles/README.md round 4: 2.605 -> 2.570 ms)
        if self.sched.slots > 1 and self.Bl >= 128 and hasattr(core, "set_chain_budget"):
            core.set_chain_budget(256)
        # overlap_reduce=True: bucketed, overlapped gradient all-reduce (DDP-style): the critic phase publishes [ensemble | head |
        # proprio | scalars] before it starts the encoder-head backward; that bucket is reduced on a communication stream while
        # the encoder heads' weight gradients are still being computed, the second bucket follows, and only `apply` waits.
        # OPT-IN since round 3's measurement (profiles/README.md): a dependency that crosses HIP streams costs 60-100 us on this
        # stack, the update stream crosses twice per critic update (comm waits for the bucket event, `apply` waits for comm),
        # and at one rank's share of an 8-GPU batch that is more (0.772 -> 0.841 ms per step, unstable with RCCL actually
        # issued: 0.86 / 1.18 ms) than the ~0.1 ms of encoder-head backward a bucket can hide behind.  Default: ONE all-reduce
        # of the critic gradients on the update stream itself, no stream crossing.
        self._overlap = bool(overlap_reduce) and hasattr(core, "critic_grads_bucketed") and hasattr(core, "grad_bucket")
        self._comm = None

    def _view(self, which):
        if which not in self._gv:
            self._gv[which] = self.core.grad_view(which)
        return self._gv[which]

    def _reduce(self, which):
        if self.world > 1 or self.force_reduce:
            self.all_reduce(self._view(which))

    def _produce(self, rng):
        """Sample one global batch (identical index/crop streams on all ranks), materialise this rank's
        slice and run the frozen trunk on it -- on the side stream when pipelining.  `rng`: state.rng at the entry of the call
        that will consume this batch (rng, obs_rng, next_obs_rng = split(rng, 3), drq.py:276-277,307-308)."""
        slot = self._next_slot
        self._next_slot = (slot + 1) % self.sched.slots
        # replicated stores (serl_amd/data/replicated.py): every rank applies the same actor transitions here, i.e. at
        # the same point of its insert / index-draw sequence, so the global index stream stays identical on all ranks
        for buf in self.buffers:
            if hasattr(buf, "step_barrier"):
                buf.step_barrier()
        parts = [(b, b.sample_indices(n)) for b, n in zip(self.buffers, self.batch_sizes)]
        k3 = self._J.split(rng, 3)
        co, cn = self._J.crop_offsets(k3[1], self.B, 4), self._J.crop_offsets(k3[2], self.B, 4)
        self.last_draws["crops"] = (co, cn)
        local, (lo, hi) = shard_parts(parts, self.rank, self.world)
        self.sched.wait_consumed(slot)
        split = getattr(self.sched, "update_after_stage", None)
        with self.sched.side():
            db = self.sched.gathered(slot, lambda: self.gather(local, co[lo:hi], cn[lo:hi], slot))
            if split is None or not hasattr(self.core, "encode_slot_range"):
                self.core.encode_slot(db, slot)
            else:   # two pieces with an event between them (see TorchPipelineSchedule)
                self.core.encode_slot_range(db, slot, -1, split)
                self.sched.mid_produced(slot)
                self.core.encode_slot_range(db, slot, split + 1, 3)
        self.sched.produced(slot)
        return slot

    def _acquire(self):
        slot = self._pending if self._pending is not None else self._produce(self._keys.rng_in)
        self._pending = None
        if self.sched.slots > 1:
            self._pending = self._produce(self._keys.rng_out)      # batch i+1 overlaps the update of batch i; its call starts at rng_out
            if getattr(self.sched, "update_after_stage", None) is not None:
                self.sched.wait_mid(self._pending)
        self.sched.wait_produced(slot)
        self.core.select_slot(slot)
        return slot

    def _noise(self, critic: bool):
        """this rank's rows of the call's jax.random draws (device_noise = "threefry"), REDQ indices always"""
        J, keys = self._J, self._keys
        noise = {}
        if critic:
            noise["redq_idx"] = J.randint(keys.k_subsample[0], 2, 0, self.ensemble).reshape(1, 2)
            self.last_draws["redq_idx"] = noise["redq_idx"]
        if self.device_noise != "threefry":
            return noise if critic else None
        if self.noise_form == "keys":      # the kernels draw this rank's rows of the global arrays themselves (serl_agent_set_shard)
            c = self.core.cfg
            n_cam = c.n_cam if c.encoder_type == 0 else 0
            cam = lambda k: np.stack([J.flax_make_rng(k, J.dropout_path(x), 1) for x in self.image_keys[:n_cam]])  # noqa: E731
            if critic:
                noise["key_eps_next"] = keys.k_next_action[0]
                if n_cam:
                    noise["key_mask_next"] = cam(keys.k_next_action[0])
            else:
                noise["key_eps_pi"], noise["key_eps_temp"] = keys.k_sample, keys.k_temp
                if n_cam:
                    noise["key_mask_obs_pi"], noise["key_mask_next_temp"] = cam(keys.k_policy), cam(keys.k_temp)
            return noise
        import torch
        c = self.core.cfg
        A, D, Bl, B, lo = c.act_dim, 512 * c.sle_features, self.Bl, self.B, self.rank * self.Bl
        n_cam = c.n_cam if c.encoder_type == 0 else 0
        if self._nbuf is None:
            dev = self.core.device
            self._nbuf = {k: torch.empty((Bl, A), dtype=torch.float32, device=dev) for k in ("eps_next", "eps_pi", "eps_temp")}
            if n_cam:
                self._nbuf.update({k: torch.empty((n_cam, Bl, D), dtype=torch.uint8, device=dev)
                                   for k in ("mask_next", "mask_obs_pi", "mask_next_temp")})
        nb, jobs, keep = self._nbuf, [], 1.0 - float(c.dropout)

        def draws(eps_name, eps_key, mask_name, mask_key):
            jobs.append(J.job(J.NORMAL, eps_key, B * A, nb[eps_name].data_ptr(), first=lo * A, count=Bl * A))
            noise[eps_name] = nb[eps_name]
            for ci, cam in enumerate(self.image_keys[:n_cam]):
                jobs.append(J.job(J.BERNOULLI_U8, J.flax_make_rng(mask_key, J.dropout_path(cam), 1), B * D,
                                  nb[mask_name].data_ptr() + ci * Bl * D, first=lo * D, count=Bl * D, p=keep))
            if n_cam:
                noise[mask_name] = nb[mask_name]

        if critic:
            draws("eps_next", keys.k_next_action[0], "mask_next", keys.k_next_action[0])
        else:
            draws("eps_pi", keys.k_sample, "mask_obs_pi", keys.k_policy)
            draws("eps_temp", keys.k_temp, "mask_next_temp", keys.k_temp)
        J.fill(c.device, jobs, self.core._stream())
        return noise

    def _critic(self):
        noise = self._noise(True)
        self.core.begin_update()
        if self._overlap and (self.world > 1 or self.force_reduce):
            self._critic_overlapped(noise)
        else:
            self.core.critic_grads(0, self.Bl, self.B, noise)
            self._reduce(APPLY_CRITIC)
        self.core.apply(APPLY_CRITIC)

    def _critic_overlapped(self, noise):
        import torch
        dev = self.core.device
        if self._comm is None:
            self._comm = dict(stream=torch.cuda.Stream(device=dev, priority=-1), ev0=torch.cuda.Event(), ev1=torch.cuda.Event(),
                              b0=self.core.grad_bucket(0), b1=self.core.grad_bucket(1))
        c = self._comm
        cur = torch.cuda.current_stream(dev)
        self.core.critic_grads_bucketed(0, self.Bl, self.B, noise, 0, c["ev0"])   # ev0: bucket 0 final, encoder heads still running
        c["stream"].wait_event(c["ev0"])
        with torch.cuda.stream(c["stream"]):
            self.all_reduce(c["b0"])
        if c["b1"].numel():
            c["ev1"].record(cur)                                                   # the whole critic phase
            c["stream"].wait_event(c["ev1"])
            with torch.cuda.stream(c["stream"]):
                self.all_reduce(c["b1"])
        cur.wait_stream(c["stream"])

    def update_critics(self):
        """DrQAgent.update_critics over the global batch (one grad-step)."""
        self._keys = self._J.UpdateKeys(self._rng, True, 1, False)
        with self.sched.main():
            slot = self._acquire()
            self._critic()
            self.sched.consumed(slot)
        self._rng = self._keys.rng_out

    def update_high_utd(self):
        """DrQAgent.update_high_utd(utd_ratio=1): critic step, then actor+temperature on the same batch."""
        self._keys = self._J.UpdateKeys(self._rng, True, 1, True)
        with self.sched.main():
            slot = self._acquire()
            self._critic()
            self.core.actor_grads(self.B, self._noise(False))
            self._reduce(APPLY_ACTOR_TEMP)
            self.core.apply(APPLY_ACTOR_TEMP)
            self.sched.consumed(slot)
        self._rng = self._keys.rng_out

    def iteration(self, critic_actor_ratio: int = 1):
        """One learner-loop iteration (examples/async_drq_sim/async_drq_sim.py:266-292)."""
        for _ in range(critic_actor_ratio - 1):
            self.update_critics()
        self.update_high_utd()


class TrunkFarmLearner(DataParallelLearner):
    """The second way to spread the learner over P GPUs: a STEP-PIPELINED TRUNK FARM.

    The ResNet trunk is frozen and cut off by a stop_gradient (vision/resnet_v1.py:286), so the features of a batch depend on its
    pixels only -- not on the parameters being trained, not on any other batch.  Rank 0 (the updater) is the only rank that holds
    live parameters: it runs every update (sac.py:243-299) on the FULL batch and never runs the trunk.  Ranks 1 .. P-1 (trunk
    workers) hold the replay replica and the frozen trunk: worker w gathers, augments and encodes every (P-1)-th batch at full
    batch size -- the efficient 1024-image kernels, not the 1/P-size kernels of batch-sharded data parallelism -- and ships the
    features (33.5 MB at B = 256, two cameras) to rank 0 point to point.  No gradient all-reduce exists: per step one P2P
    transfer, off the updater's critical path.  Every rank draws the identical index / key streams (same seed, replicated
    buffers), so the updater's results are BIT-IDENTICAL to the single-GPU learner's, whatever P.

    Rate = min(update chain alone on rank 0, (P - 1) x one trunk pass): the chain bounds it from P = 4 on.

    send(tensor, dst, tag) / recv(tensor, src, tag) -> handle with .wait() (or None): torch.distributed isend / irecv on RCCL, a
    host-staged gloo pair in the single-GPU test.  role: "updater" / "worker" override the rank's role for single-GPU emulation
    (bench.py --farm-role: the pieces of the projection are measured one at a time)."""

    def __init__(self, core, gather, buffers, batch_sizes, rank=0, world=2, send=None, recv=None, seed=0, ensemble=10,
                 schedule=None, image_keys=None, device_noise="hash", role=None, n_workers=None, chain_budget=None):
        super().__init__(core, gather, buffers, batch_sizes, 0, 1, all_reduce=None, seed=seed, ensemble=ensemble,
                         schedule=schedule, image_keys=image_keys, device_noise=device_noise)
        assert world >= 2 or role is not None, "a trunk farm needs an updater and at least one worker"
        self.farm_rank, self.farm_world = rank, world
        self.role = role or ("updater" if rank == 0 else "worker")
        self.n_workers = n_workers if n_workers is not None else max(world - 1, 1)
        self.send, self.recv = send, recv
        if self.role == "updater" and hasattr(core, "set_chain_budget"):
            # chain_budget: K-split budget of the update chain's GEMMs on the updater (None = the library default, 512: results
            # bit-identical to a single-GPU learner that runs with the same budget -- the K-split depth fixes the fp32 summation
            # order).  Nothing co-runs with the chain on this rank, so deeper splits pay: same call, updater alone, 512 -> 0.7406,
            # 1024 -> 0.7054, 2048 -> 0.7099, 4096 -> 0.7236 ms per step (profiles/r05_scaling_pieces.txt); bench.py passes 1024.
            core.set_chain_budget(0 if chain_budget is None else int(chain_budget))
        self._t = 0                  # global batch counter (identical on every rank)
        self._recv_pending = {}      # slot -> handle

    def owner(self, t):
        return 1 + (t % self.n_workers)

    def _reduce(self, which):
        pass        # one rank holds the parameters and sees the whole batch: there is no gradient to reduce

    # every rank walks the same sequence of batches; what it does with batch t depends on its role
    def _produce(self, rng):
        slot = self._next_slot
        self._next_slot = (slot + 1) % self.sched.slots
        t = self._t
        self._t += 1
        for buf in self.buffers:
            if hasattr(buf, "step_barrier"):
                buf.step_barrier()
        parts = [(b, b.sample_indices(n)) for b, n in zip(self.buffers, self.batch_sizes)]     # (keeps every rank's index stream in step)
        k3 = self._J.split(rng, 3)
        if self.role == "worker" and self.owner(t) != self.farm_rank and self.farm_world > 1:
            return slot                                                       # another worker's batch
        co, cn = self._J.crop_offsets(k3[1], self.B, 4), self._J.crop_offsets(k3[2], self.B, 4)
        self.last_draws["crops"] = (co, cn)
        self.sched.wait_consumed(slot)
        with self.sched.side():
            db = self.sched.gathered(slot, lambda: self.gather(parts, co, cn, slot))
            if self.role == "worker":
                self._wait_transfer(slot)      # (this stream: the previous send out of this slot's feature buffer has completed)
                self.core.encode_slot(db, slot)
                if self.send is not None:
                    h = self.send(self.core.slot_features(slot), 0, t)
                    if h is not None:
                        self._recv_pending[slot] = h                            # (completion of the send frees the slot)
            else:
                self.core.bind_slot(db, slot)
                if self.recv is not None:
                    self._recv_pending[slot] = self.recv(self.core.slot_features(slot), self.owner(t), t)
        self.sched.produced(slot)
        return slot

    def _wait_transfer(self, slot):
        h = self._recv_pending.pop(slot, None)
        if h is not None:
            h.wait()

    def _acquire(self):
        slot = super()._acquire()
        if self.role == "updater":
            self._wait_transfer(slot)
        return slot

    def _worker_step(self, has_actor):
        """a worker's share of one learner call: the key bookkeeping of the call and, if it owns the batch, its pass"""
        self._keys = self._J.UpdateKeys(self._rng, True, 1, has_actor)
        slot = self._pending if self._pending is not None else self._produce(self._keys.rng_in)
        self._pending = None
        if self.sched.slots > 1:
            self._pending = self._produce(self._keys.rng_out)
        self._wait_transfer(slot)          # the send of this slot's features (if it was ours) has completed
        self.sched.consumed(slot)
        self._rng = self._keys.rng_out

    def update_critics(self):
        if self.role == "worker":
            return self._worker_step(False)
        return super().update_critics()

    def update_high_utd(self):
        if self.role == "worker":
            return self._worker_step(True)
        return super().update_high_utd()
