"""CPU (gloo, world_size 2): cross-rank replication of replay inserts (serl_amd/data/replicated.py; SURVEY.md 8(e)
"inserts broadcast").  Rank 0 receives transitions from a concurrent "actor" thread while both ranks run the
batch-sharded learner loop; every rank must end up with the identical buffer (valid mask, insert index, size, slot
contents) and must have drawn the identical index stream at every step -- the reference's single-process contract
(data/data_store.py:96-136: insert under the lock, sample under the lock) carried over to P replicas.  The replica is
the NumPy replay oracle (bit-exact restatement of the reference buffer), so the draw really depends on the validity
mask the inserts produce."""
import itertools
import os
import socket
import threading
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.replay_oracle import ReplayOracle
from serl_amd.data.replicated import ReplicatedDataStore
from serl_amd.parallel import DataParallelLearner
from serl_amd.utils.synthetic import transition_stream

KEYS, H, W, S, A = ("front", "wrist"), 8, 8, 3, 2
CAP = 120          # small: the ring wraps during the run, so invalidation and the rejection loop are exercised


class _Core:
    """Records what the learner hands it; gradients are the batch's sample ids so the all-reduce is checkable."""

    def __init__(self):
        self.g = torch.zeros(4, dtype=torch.float64)
        self.seen = []

    def begin_update(self): pass
    def encode_slot(self, batch, slot): self.batch = batch
    def select_slot(self, slot): pass
    def set_shard(self, off, glob): pass

    def critic_grads(self, off, cnt, glob, noise, redq_row=0):
        self.g[:] = float(self.batch["idx_sum"])

    def actor_grads(self, glob, noise): pass
    def apply(self, which, w=1.0): self.seen.append(float(self.g[0]))
    def grad_view(self, which): return self.g


def _mk_store(rank, world, lag):
    ro = ReplayOracle(KEYS, H, W, 3, 1, S, A, CAP)
    ro.seed(0)
    return ReplicatedDataStore(ro, rank, world, lag=lag)


def _run(rank, world, port, lag, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    store = _mk_store(rank, world, lag)
    stream = transition_stream(KEYS, H, W, 3, 1, S, A, 10, 99)
    if rank == 0:                                   # initial fill arrives through the same door as live data
        for tr in itertools.islice(stream, 60):
            store.insert(tr)
    store.wait_until(60)
    assert len(store) >= 60
    drawn = []

    def gather(parts, co, cn, slot):
        drawn.append(np.concatenate([ix for _, ix in parts]))
        return {"idx_sum": sum(int(ix.sum()) for _, ix in parts)}

    stop = threading.Event()

    def actor():                                     # rank 0 only: transitions keep arriving while the learner runs
        for tr in stream:
            if stop.is_set():
                return
            store.insert(tr)
            time.sleep(0.0007)

    th = threading.Thread(target=actor, daemon=True)
    if rank == 0:
        th.start()
    core = _Core()
    lr = DataParallelLearner(core, gather, [store], [16], rank, world, all_reduce=lambda t: dist.all_reduce(t), seed=3)
    full_idx = []
    orig = store.replica.sample_indices
    store.replica.sample_indices = lambda n: (full_idx.append(orig(n)) or full_idx[-1])
    for _ in range(150):
        lr.iteration(2)
        time.sleep(0.001)
    stop.set()
    if rank == 0:
        th.join()
    store.flush()                                    # collective: everything rank 0 accepted is in every replica
    ro = store.replica
    out[rank] = dict(valid=ro.valid.copy(), insert_index=ro.insert_index, size=ro.size,
                     idx=np.stack(full_idx), frames=ro.frames["front"].copy(), state=ro.state.copy(),
                     rewards=ro.rewards.copy(), seen=np.array(core.seen), local=np.stack(drawn),
                     rng=ro.rng.bit_generator.state["state"]["state"])
    store.close()
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _check(out, world):
    a = out[0]
    assert a["size"] == CAP and a["idx"].shape == (300, 16)
    for r in range(1, world):
        b = out[r]
        assert np.array_equal(a["valid"], b["valid"]) and a["insert_index"] == b["insert_index"] and a["size"] == b["size"]
        assert np.array_equal(a["idx"], b["idx"]), "every rank must draw the identical global index stream"
        assert a["rng"] == b["rng"]
        assert np.array_equal(a["frames"], b["frames"]) and np.array_equal(a["state"], b["state"])
        assert np.array_equal(a["rewards"], b["rewards"])
        assert np.array_equal(a["seen"], b["seen"])
        # each rank gathered only its own slice of the global draw
        per = 16 // world
        assert np.array_equal(b["local"], a["idx"][:, r * per:(r + 1) * per])
    # the actor really was concurrent: the buffer wrapped, so inserts landed between draws
    assert a["insert_index"] != 60 % CAP


def test_inserts_replicate_identically_across_two_ranks():
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_run, args=(world, port, 2, out), nprocs=world, join=True)
    _check(out, world)


def test_lag_zero_is_synchronous_and_also_identical():
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_run, args=(world, port, 0, out), nprocs=world, join=True)
    _check(out, world)


def test_single_rank_applies_at_batch_boundaries():
    """world 1: no threads, no group -- inserts are still deferred to step_barrier (lag batches later) or flush."""
    st = ReplicatedDataStore(ReplayOracle(KEYS, H, W, 3, 1, S, A, CAP), 0, 1, lag=1)
    trs = list(itertools.islice(transition_stream(KEYS, H, W, 3, 1, S, A, 10, 5), 7))
    for tr in trs[:3]:
        st.insert(tr)
    assert len(st) == 0 and st.pending() == 3
    st.step_barrier()            # call 0 emits message 0; needs message -1: nothing applied yet
    assert len(st) == 0
    st.insert(trs[3])
    st.step_barrier()            # call 1 applies message 0 (3 transitions + the episode's first-frame slot)
    assert len(st) == 4 and st.pending() == 0
    st.flush()                   # message 1 (1 transition) + the flush message
    assert len(st) == 5
    st.step_barrier()
    st.step_barrier()            # already applied by the flush: nothing to wait for
    assert len(st) == 5
    with np.testing.assert_raises(RuntimeError):
        ReplicatedDataStore(ReplayOracle(KEYS, H, W, 3, 1, S, A, CAP), 1, 1).insert(trs[0])
