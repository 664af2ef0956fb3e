"""GPU: the threading contract of the data store (reference data/data_store.py:96-136): the actor-facing server thread
calls insert() while the learner thread samples, gathers and updates.  A second Python thread inserts at full speed
(ctypes releases the GIL; the insert holds the buffer mutex for a host memcpy only) while the learner runs
sample -> gather(+crop) -> update iterations on the pipelined path; every gathered sample must be an intact, consistent
transition (no torn frame, obs frame = previous next frame), and the final bookkeeping must equal the oracle's."""
import threading
import time

import numpy as np
import pytest
import torch

from helpers import make_spaces
from oracle.replay_oracle import ReplayOracle

pytestmark = pytest.mark.gpu
KEYS, H, W, S, A = ("front", "wrist"), 64, 64, 5, 3
EP = 17


def _frame(k, cam):
    """content of frame number k of camera cam: every byte depends on k and on its position (a torn copy is visible)"""
    base = (int(k) * 97 + cam * 31) % 251
    return ((np.arange(H * W * 3, dtype=np.int64) * 7 + base) % 256).astype(np.uint8).reshape(1, H, W, 3)


def _transition(k):
    """transition k: obs frame k, next frame k+1 (consecutive env steps share a frame); state[0] carries k"""
    done = (k % EP) == EP - 1
    st = np.zeros((1, S), np.float32)
    st[0, 0] = k
    nst = st.copy()
    nst[0, 1] = 1.0
    obs = {"state": st, **{c: _frame(k, i) for i, c in enumerate(KEYS)}}
    nobs = {"state": nst, **{c: _frame(k + 1, i) for i, c in enumerate(KEYS)}}
    return {"observations": obs, "next_observations": nobs, "actions": np.full((A,), 0.1, np.float32),
            "rewards": np.float32(done), "masks": np.float32(1.0 - done), "dones": bool(done)}


def _check_packed(b, idx=None):
    st = b["observations"]["state"].cpu().numpy()[:, 0, 0].astype(np.int64)
    for i, c in enumerate(KEYS):
        fr = b["observations"][c].cpu().numpy()
        for j, k in enumerate(st):
            if idx is not None and idx[j] == 0:
                # index 0 is only valid for an episode's first transition, and the reference's negative window
                # (memory_efficient_replay_buffer.py:149-153, reproduced bit for bit) pairs its record with the frames of
                # slots (cap-2, cap-1), which the look-ahead invalidation does not protect: once the write head passes
                # cap-2 the reference, too, returns the NEW frames there with the old record of slot 0.  Not a torn copy.
                assert k % EP == 0
                continue
            assert np.array_equal(fr[j, 0], _frame(k, i)[0]), f"sample {j}: obs frame of transition {k} is torn or stale"
            ok = np.array_equal(fr[j, 1], _frame(k + 1, i)[0])
            if not ok and k % EP == 0:
                # the reference's negative-window quirk (memory_efficient_replay_buffer.py:149-153, reproduced on purpose,
                # DESIGN.md): an episode's first transition that lands in slot 0 is returned with the frame pair of
                # slots (cap-2, cap-1) = (previous episode's last next-frame, this episode's first-frame slot) -- with
                # consecutively numbered frames both are frame k
                ok = np.array_equal(fr[j, 1], _frame(k, i)[0])
            assert ok, f"sample {j}: next frame of transition {k} is torn or stale"
    return len(st)


def test_insert_thread_vs_learner_thread(gpu):
    from serl_amd.data.data_store import MemoryEfficientReplayBufferDataStore
    from serl_amd.utils.launcher import make_drq_agent
    cap, B = 301, 16
    osp, asp = make_spaces(KEYS, H, W, 3, 1, S, A)
    rb = MemoryEfficientReplayBufferDataStore(osp, asp, cap, image_keys=KEYS)
    rb.seed(0)
    for k in range(200):
        rb.insert(_transition(k))
    obs = {"front": np.zeros((1, H, W, 3), np.uint8), "wrist": np.zeros((1, H, W, 3), np.uint8), "state": np.zeros((1, S), np.float32)}
    agent = make_drq_agent(1, obs, np.zeros((A,), np.float32), image_keys=KEYS, encoder_type="resnet-pretrained", batch_size=B)
    learner_done, err, inserted = threading.Event(), [], [200]

    def inserter():      # inserts for as long as the learner runs (the ring wraps many times)
        try:
            t0 = time.perf_counter()
            while not learner_done.is_set() and inserted[0] < 200000:
                rb.insert(_transition(inserted[0]))
                inserted[0] += 1
            inserter.rate = (inserted[0] - 200) / (time.perf_counter() - t0)
        except Exception as e:  # noqa: BLE001
            err.append(e)

    th = threading.Thread(target=inserter)
    it = rb.get_iterator(sample_args={"batch_size": B, "pack_obs_and_next_obs": True, "lazy": True})
    th.start()
    checked = 0
    for iters in range(1, 201):
        batch = next(it)
        if iters % 4 == 0:
            agent, _ = agent.update_high_utd(batch, utd_ratio=1)
        else:
            agent, _ = agent.update_critics(batch)
        if iters % 5 == 0:       # an eager reference-format sample, verified byte for byte (the library re-draws stale
            idx = rb.sample_indices(B)   # indices in place, so `idx` describes the batch that was gathered)
            checked += _check_packed(rb.gather(idx), idx)
    torch.cuda.synchronize()
    learner_done.set()
    th.join(timeout=60)
    assert not th.is_alive() and not err, err
    torch.cuda.synchronize()
    n_total = inserted[0]
    assert n_total > 200 + 2 * cap and checked >= 40 * B, (n_total, checked)   # the ring wrapped under the learner
    info = agent.core.read_info()
    assert all(np.isfinite(v) for v in info.values())
    # bookkeeping after the race == the oracle fed the same transitions sequentially
    o = ReplayOracle(KEYS, H, W, 3, 1, S, A, cap)
    for k in range(n_total):
        o.insert(_transition(k))
    assert len(rb) == len(o) and rb.latest_data_id() == o.insert_index
    assert np.array_equal(rb.valid_mask(), o.valid)
    # and the final content is intact: every valid slot gathers a consistent transition
    valid = np.flatnonzero(o.valid[:len(o)])
    v = valid[:64].astype(np.int64).copy()
    _check_packed(rb.gather(v), v)
    print(f"insert thread: {n_total - 200} transitions ({inserter.rate:.0f}/s) while the learner ran 200 updates; {checked} samples verified")


def test_insert_does_not_block_on_the_gpu(gpu):
    """an insert is a host memcpy into the pinned ring + asynchronous copies: far below the old ~100 us/insert that a
    stream synchronisation under the mutex cost"""
    from serl_amd.data.data_store import MemoryEfficientReplayBufferDataStore
    osp, asp = make_spaces(KEYS, 128, 128, 3, 1, 24, 6)
    rb = MemoryEfficientReplayBufferDataStore(osp, asp, 5000, image_keys=KEYS)
    rng = np.random.default_rng(0)
    fr = {c: rng.integers(0, 256, (1, 128, 128, 3), dtype=np.uint8) for c in KEYS}
    tr = {"observations": {"state": np.zeros((1, 24), np.float32), **fr}, "next_observations": {"state": np.zeros((1, 24), np.float32), **fr},
          "actions": np.zeros((6,), np.float32), "rewards": np.float32(0), "masks": np.float32(1), "dones": False}
    for _ in range(50):
        rb.insert(tr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        rb.insert(tr)
    dt = (time.perf_counter() - t0) / 2000
    torch.cuda.synchronize()
    print(f"insert: {dt * 1e6:.1f} us per transition (2 x 49 KB frames)")
    assert dt < 60e-6
