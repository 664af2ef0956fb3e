"""Cross-rank replication of replay inserts for the batch-sharded data-parallel learner (SURVEY.md 8(e):
"the replay buffer replicated in each GPU's HBM; inserts broadcast").

Reference contract being kept (single process there): the agentlace TrainerServer thread calls
`MemoryEfficientReplayBufferDataStore.insert` under the store's lock while the learner thread samples
(serl_launcher/data/data_store.py:96-136, utils/launcher.py:171-177).  With P learner ranks every rank must
draw the IDENTICAL index stream (serl_amd/parallel.py), and the draw depends on the buffer's size and validity
mask -- so it is not enough that every rank eventually receives every transition: every rank has to apply the
same transitions at the same point of its insert / sample sequence.

Protocol
  * only rank 0 owns the actor-facing endpoint.  `insert()` there (any thread) appends to a pending list and
    returns; it does not touch the HBM buffer.
  * the learner thread of EVERY rank calls `step_barrier()` once per sampled batch, before the index draw
    (DataParallelLearner._produce does).  On rank 0 call k packs the pending transitions into message k (possibly
    empty) and hands it to a sender thread, which broadcasts it over a gloo group (host bytes; ~98 KB per
    transition at 10-20 Hz).  On every rank -- rank 0 included -- call k then applies, in message order, every
    message up to message k - lag.  The lag (default 2 batches) is what hides the broadcast: the message a rank
    needs at call k left rank 0 `lag` batches earlier, and a receiver thread has normally delivered it already;
    if not, the call blocks until it arrives (never skips), so the applied prefix at call k is a pure function of k.
  * `flush()` is the lag-free collective variant (initial fill, demos, "wait until the buffer holds N"): rank 0
    emits a flush message, every rank applies everything up to and including it.
Every rank therefore executes the same insert sequence between the same two index draws: `valid_mask`,
`insert_index`, `len()` and the PCG64 streams stay bit-identical across ranks (tests/test_replicated_cpu.py on gloo,
tests/test_dp_two_process_gpu.py on the HIP path).
"""
from __future__ import annotations

import pickle
import queue
import threading
from typing import List, Optional

import numpy as np

_REG, _FLUSH, _CLOSE = 0, 1, 2


class ReplicatedDataStore:
    """Wraps one rank's replica `store` (anything with insert(transition); the HBM stores of data_store.py in
    production).  Everything except insert / step_barrier / flush / close is forwarded to the replica."""

    def __init__(self, store, rank: int, world: int, group=None, lag: int = 2, timeout_s: float = 120.0):
        self._store, self.rank, self.world = store, int(rank), int(world)
        self._lag = max(0, int(lag))
        self._timeout = timeout_s
        self._pending: List[dict] = []
        self._plock = threading.Lock()
        self._inbox: List[tuple] = []       # (kind, index, [transitions]) in message order
        self._applied = 0                   # inbox entries [0, _applied) have been applied to the replica
        self._done = {_REG: -1, _FLUSH: -1}  # highest message index of each kind already applied
        self._cv = threading.Condition()
        self._calls = 0                     # step_barrier calls so far (== regular messages emitted on rank 0)
        self._flushes = 0
        self._closed = False
        self._err: Optional[BaseException] = None
        self._group = group
        self._thread = None
        if self.world > 1:
            import torch.distributed as dist
            assert dist.is_initialized(), "ReplicatedDataStore needs an initialised torch.distributed job"
            if self._group is None:   # host-side byte broadcast: its own gloo group, whatever the gradient backend is
                self._group = dist.new_group(backend="gloo")
            if self.rank == 0:
                self._outq: "queue.Queue" = queue.Queue()
                self._thread = threading.Thread(target=self._send_loop, name="serl-replay-bcast", daemon=True)
            else:
                self._thread = threading.Thread(target=self._recv_loop, name="serl-replay-recv", daemon=True)
            self._thread.start()

    # ---- actor-facing side (rank 0; thread-safe like data_store.py:104-106) -----------------------------------------
    def insert(self, data_dict):
        if self.rank != 0:
            raise RuntimeError("only rank 0 receives transitions from the actor; the other ranks get them replicated")
        with self._plock:
            self._pending.append(data_dict)

    def batch_insert(self, batch_data):
        for d in batch_data:
            self.insert(d)

    def pending(self) -> int:
        with self._plock:
            return len(self._pending)

    # ---- learner-thread side (collective in program order: every rank makes the same calls) -------------------------
    def step_barrier(self):
        """Call once per sampled batch, before the index draw, on every rank."""
        k = self._calls
        self._calls += 1
        if self.rank == 0:
            self._emit(_REG, k)
        need = k - self._lag
        if need >= 0:
            self._apply_through(_REG, need)

    def flush(self):
        """Collective: every transition rank 0 has accepted so far is in every replica when this returns."""
        f = self._flushes
        self._flushes += 1
        if self.rank == 0:
            self._emit(_FLUSH, f)
        self._apply_through(_FLUSH, f)

    def wait_until(self, n: int, poll_s: float = 0.05):
        """Collective replacement for the learner's `while len(replay_buffer) < training_starts: sleep` loop
        (examples/async_drq_sim/async_drq_sim.py:219-227): flushes until every replica holds at least n transitions."""
        import time
        while True:
            self.flush()
            if len(self._store) >= n:
                return
            time.sleep(poll_s)

    def close(self):
        if self._closed:
            return
        self._closed = True
        if self.world > 1 and self.rank == 0:
            self._outq.put((_CLOSE, 0, b""))
        if self._thread is not None:
            self._thread.join(timeout=10.0)

    # ---- internals ---------------------------------------------------------------------------------------------------
    def _emit(self, kind, index):
        with self._plock:
            trs, self._pending = self._pending, []
        with self._cv:
            self._inbox.append((kind, index, trs))
            self._cv.notify_all()
        if self.world > 1:
            self._outq.put((kind, index, pickle.dumps(trs, protocol=5) if trs else b""))

    def _apply_through(self, kind, index):
        """Apply inbox entries in order up to and including message (kind, index); blocks until it has arrived.  A
        regular message that an earlier flush already applied is simply behind the pointer."""
        with self._cv:
            if index <= self._done[kind]:
                return
            while True:
                if self._err is not None:
                    raise RuntimeError("replay replication thread failed") from self._err
                pos = next((i for i, m in enumerate(self._inbox) if m[0] == kind and m[1] == index), None)
                if pos is not None:
                    break
                if not self._cv.wait(timeout=self._timeout):
                    raise TimeoutError(f"rank {self.rank}: replay message ({kind}, {index}) did not arrive in {self._timeout} s")
            todo = self._inbox[self._applied:pos + 1]
            self._applied = pos + 1
            for k_, i_, _ in todo:
                self._done[k_] = max(self._done[k_], i_)
        for _, _, trs in todo:
            for tr in trs:
                self._store.insert(tr)
        # applied messages are dropped so the inbox does not grow with the run
        with self._cv:
            drop = self._applied
            if drop > 64:
                del self._inbox[:drop]
                self._applied = 0

    def _send_loop(self):
        import torch
        import torch.distributed as dist
        try:
            while True:
                kind, index, blob = self._outq.get()
                hdr = torch.tensor([kind, index, len(blob)], dtype=torch.int64)
                dist.broadcast(hdr, src=0, group=self._group)
                if len(blob):
                    dist.broadcast(torch.frombuffer(bytearray(blob), dtype=torch.uint8), src=0, group=self._group)
                if kind == _CLOSE:
                    return
        except BaseException as e:   # surfaced by the next collective call of the learner thread
            with self._cv:
                self._err = e
                self._cv.notify_all()

    def _recv_loop(self):
        import torch
        import torch.distributed as dist
        try:
            while True:
                hdr = torch.zeros(3, dtype=torch.int64)
                dist.broadcast(hdr, src=0, group=self._group)
                kind, index, n = (int(x) for x in hdr.tolist())
                trs = []
                if n:
                    buf = torch.empty(n, dtype=torch.uint8)
                    dist.broadcast(buf, src=0, group=self._group)
                    trs = pickle.loads(buf.numpy().tobytes())
                if kind == _CLOSE:
                    return
                with self._cv:
                    self._inbox.append((kind, index, trs))
                    self._cv.notify_all()
        except BaseException as e:
            with self._cv:
                self._err = e
                self._cv.notify_all()

    # ---- everything else is the replica's --------------------------------------------------------------------------
    def __len__(self):
        return len(self._store)

    def __getattr__(self, name):
        return getattr(self._store, name)

    @property
    def replica(self):
        return self._store
