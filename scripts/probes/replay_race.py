"""Amplified version of tests/test_replay_threads_gpu.py: a tiny ring (the inserter overwrites sampled slots all the time) and a
busy stream (the gather kernel runs long after the host validated its indices), eager sample + byte-for-byte check.
usage: python scripts/probes/replay_race.py [iters] [cap]"""
import sys, threading, time
import numpy as np, torch
import os; _R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
from helpers import make_spaces
from serl_amd.data.data_store import MemoryEfficientReplayBufferDataStore

KEYS, H, W, S, A, EP = ("front", "wrist"), 64, 64, 5, 3, 17
def _frame(k, cam):
    base = (int(k) * 97 + cam * 31) % 251
    return ((np.arange(H * W * 3, dtype=np.int64) * 7 + base) % 256).astype(np.uint8).reshape(1, H, W, 3)
def _transition(k):
    done = (k % EP) == EP - 1
    st = np.zeros((1, S), np.float32); st[0, 0] = k
    nst = st.copy(); nst[0, 1] = 1.0
    obs = {"state": st, **{c: _frame(k, i) for i, c in enumerate(KEYS)}}
    nobs = {"state": nst, **{c: _frame(k + 1, i) for i, c in enumerate(KEYS)}}
    return {"observations": obs, "next_observations": nobs, "actions": np.full((A,), 0.1, np.float32),
            "rewards": np.float32(done), "masks": np.float32(1.0 - done), "dones": bool(done)}

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 48
B = 16
osp, asp = make_spaces(KEYS, H, W, 3, 1, S, A)
rb = MemoryEfficientReplayBufferDataStore(osp, asp, cap, image_keys=KEYS)
rb.seed(0)
for k in range(cap - 8):
    rb.insert(_transition(k))
stop, n_ins = threading.Event(), [cap - 8]
def inserter():
    while not stop.is_set():
        rb.insert(_transition(n_ins[0])); n_ins[0] += 1
th = threading.Thread(target=inserter); th.start()
big = torch.randn(4096, 4096, device="cuda")
bad = checked = 0
first = None
t0 = time.time()
for it in range(iters):
    y = big @ big                      # ~1 ms of work ahead of the gather on the same stream
    idx = rb.sample_indices(B)
    b = rb.gather(idx)
    st = b["observations"]["state"].cpu().numpy()[:, 0, 0].astype(np.int64)
    for i, c in enumerate(KEYS):
        fr = b["observations"][c].cpu().numpy()
        for j, k in enumerate(st):
            if idx[j] == 0:   # the reference's negative-window quirk (see tests/test_replay_threads_gpu.py)
                continue
            checked += 1
            if not np.array_equal(fr[j, 0], _frame(k, i)[0]):
                bad += 1
                if first is None:
                    first = (it, j, int(k), int(fr[j, 0].reshape(-1)[0]), int(_frame(k, i).reshape(-1)[0]))
stop.set(); th.join()
print(f"replay_race: cap {cap}, {iters} eager samples, {n_ins[0]} inserts, {checked} frames checked, {bad} torn/stale; first {first}; {time.time()-t0:.1f} s")
sys.exit(1 if bad else 0)
