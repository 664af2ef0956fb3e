"""CPU (gloo, world_size 2): the data-parallel plumbing of serl_amd/parallel.py -- identical index
streams on all ranks, sample sharding over a concatenated (online+demo) batch, all-reduce of the
[gradients | scalars] view and identical parameter updates -- with a NumPy stand-in for the HIP core."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from serl_amd.parallel import APPLY_ACTOR_TEMP, APPLY_CRITIC, DataParallelLearner, shard_parts


class FakeBuffer:
    def __init__(self, n, seed):
        self.n = n
        self.rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

    def sample_indices(self, b):
        return self.rng.integers(self.n, size=b)


class FakeCore:
    """grad = sum over LOCAL samples of phi(sample id) / global_count; params -= grad on apply."""

    def __init__(self, dim=7):
        self.g = {APPLY_CRITIC: torch.zeros(dim + 3, dtype=torch.float64),
                  APPLY_ACTOR_TEMP: torch.zeros(3 + dim, dtype=torch.float64)}
        self.params = torch.zeros(dim, dtype=torch.float64)
        self.dim, self.batch, self.log, self.slots = dim, None, [], {}

    @staticmethod
    def phi(ids, dim):
        return np.stack([np.sin(ids * (k + 1) * 0.37) for k in range(dim)], axis=1)

    def set_shard(self, global_offset, global_batch):   # device noise is indexed by the global sample id
        self.shard = (global_offset, global_batch)

    def begin_update(self):
        pass

    def encode_slot(self, batch, slot):
        self.slots[slot] = batch

    def select_slot(self, slot):
        self.batch = self.slots[slot]

    def critic_grads(self, off, cnt, global_count, noise, redq_row=0):
        ids = self.batch["ids"][off:off + cnt].astype(np.float64)
        self.g[APPLY_CRITIC][:self.dim] = torch.tensor(self.phi(ids, self.dim).sum(0) / global_count)
        self.g[APPLY_CRITIC][self.dim:] = torch.tensor([ids.sum(), len(ids), float(noise["redq_idx"].sum()) * len(ids) / global_count])

    def actor_grads(self, global_count, noise):
        ids = self.batch["ids"].astype(np.float64)
        self.g[APPLY_ACTOR_TEMP][3:] = torch.tensor(self.phi(ids + 0.5, self.dim).sum(0) / global_count)
        self.g[APPLY_ACTOR_TEMP][:3] = torch.tensor([ids.sum(), len(ids), 0.0])

    def apply(self, which, w=1.0):
        g = self.g[which][:self.dim] if which == APPLY_CRITIC else self.g[which][3:]
        self.params -= g
        self.log.append(self.g[which].clone())

    def grad_view(self, which):
        return self.g[which]


def _gather(parts, co, cn, slot=0):
    ids = np.concatenate([ix + 1000 * k for k, (b, ix) in enumerate(parts)]) if parts else np.zeros(0)
    assert len(co) == len(ids) == len(cn)
    return {"ids": ids, "co": co, "cn": cn}


def _tagged_gather(bufs):
    def g(parts, co, cn, slot=0):
        ids = np.concatenate([ix + 1000 * bufs.index(b) for b, ix in parts])
        return {"ids": ids, "co": co, "cn": cn}
    return g


def _run(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bufs = [FakeBuffer(500, 0), FakeBuffer(60, 1)]
    core = FakeCore()
    lr = DataParallelLearner(core, _tagged_gather(bufs), bufs, [12, 12], rank, world,
                             all_reduce=lambda t: dist.all_reduce(t), seed=3)
    for _ in range(3):
        lr.iteration(critic_actor_ratio=2)
    out[rank] = (core.params.numpy().copy(), [x.numpy().copy() for x in core.log])
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_parts_covers_batch_exactly():
    a, b = object(), object()
    parts = [(a, np.arange(10)), (b, np.arange(100, 106))]
    seen = []
    for r in range(4):
        loc, (lo, hi) = shard_parts(parts, r, 4)
        assert hi - lo == 4 and sum(len(ix) for _, ix in loc) == 4
        seen += [int(v) for _, ix in loc for v in ix]
    assert seen == list(range(10)) + list(range(100, 106))
    with pytest.raises(AssertionError):
        shard_parts(parts, 0, 3)


def test_two_rank_dp_matches_single_process():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_run, args=(world, port, out), nprocs=world, join=True)
    # single-process reference
    bufs = [FakeBuffer(500, 0), FakeBuffer(60, 1)]
    core = FakeCore()
    lr = DataParallelLearner(core, _tagged_gather(bufs), bufs, [12, 12], 0, 1, seed=3)
    for _ in range(3):
        lr.iteration(critic_actor_ratio=2)
    p0, log0 = out[0]
    p1, log1 = out[1]
    assert np.array_equal(p0, p1), "ranks must apply identical updates"
    assert np.allclose(p0, core.params.numpy(), rtol=0, atol=1e-12)
    for x, y, z in zip(log0, log1, core.log):
        assert np.array_equal(x, y)
        assert np.allclose(x, z.numpy(), rtol=0, atol=1e-9)  # all-reduced sums == full-batch values


def test_learner_declares_its_shard():
    """A rank of a batch-sharded job tells the agent which global rows its batches hold (serl_agent_set_shard), so
    that device-generated noise is a function of the global sample id; a single rank does not shard."""
    from serl_amd.parallel import DataParallelLearner

    class Buf:
        def sample_indices(self, n):
            return np.arange(n)
    for world, rank in ((1, 0), (4, 2)):
        core = FakeCore()
        DataParallelLearner(core, lambda parts, co, cn, slot: None, [Buf()], [16], rank, world, all_reduce=lambda t: None)
        assert getattr(core, "shard", None) == ((rank * 4, 16) if world > 1 else None)


# ---- trunk farm (serl_amd/parallel.py TrunkFarmLearner): rank 0 updates, ranks 1.. encode every (P-1)-th batch ------------------
class FarmCore(FakeCore):
    """"features" of a batch = a tensor derived from the gathered sample ids; the update reads ONLY the received features"""

    def __init__(self, dim=7, B=24):
        super().__init__(dim)
        self.feat = {s: torch.zeros(B, dtype=torch.float64) for s in range(3)}
        self.encoded = 0

    def encode_slot(self, batch, slot):
        self.slots[slot] = batch
        self.feat[slot].copy_(torch.tensor(batch["ids"].astype(np.float64) * 3.0 + 1.0))
        self.encoded += 1

    def bind_slot(self, batch, slot):
        self.slots[slot] = {k: v for k, v in batch.items() if k != "ids"}          # the updater never sees pixels
        self.feat[slot].fill_(float("nan"))                                         # must be overwritten by the transfer

    def slot_features(self, slot):
        return self.feat[slot]

    def select_slot(self, slot):
        self.batch = dict(self.slots[slot], ids=((self.feat[slot] - 1.0) / 3.0).numpy().copy())


def _run_farm(rank, world, port, out):
    from serl_amd.parallel import TrunkFarmLearner
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bufs = [FakeBuffer(500, 0), FakeBuffer(60, 1)]
    core = FarmCore()

    def send(t, dst, tag):
        dist.send(t, dst)

    def recv(t, src, tag):
        dist.recv(t, src)

    lr = TrunkFarmLearner(core, _tagged_gather(bufs), bufs, [12, 12], rank, world, send=send, recv=recv, seed=3)
    for _ in range(4):
        lr.iteration(critic_actor_ratio=3)
    out[rank] = (core.params.numpy().copy(), core.encoded, [x.numpy().copy() for x in core.log])
    dist.destroy_process_group()


def test_trunk_farm_world3_equals_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_run_farm, args=(3, port, out), nprocs=3, join=True)
    bufs = [FakeBuffer(500, 0), FakeBuffer(60, 1)]
    core = FakeCore()
    lr = DataParallelLearner(core, _tagged_gather(bufs), bufs, [12, 12], 0, 1, seed=3)
    for _ in range(4):
        lr.iteration(critic_actor_ratio=3)
    p0, enc0, log0 = out[0]
    assert enc0 == 0 and np.array_equal(p0, core.params.numpy()), "the updater must reproduce the single process bit for bit"
    assert len(log0) == len(core.log) and all(np.array_equal(a, b.numpy()) for a, b in zip(log0, core.log))
    # 12 batches, two workers: six passes each, no parameters ever applied
    assert out[1][1] == 6 and out[2][1] == 6 and not out[1][2] and not out[2][2]
    assert not np.any(out[1][0]) and not np.any(out[2][0])
