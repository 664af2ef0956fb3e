"""Step rate of the REFERENCE-NAMED API (make_drq_agent + MemoryEfficientReplayBufferDataStore.get_iterator + agent.update_high_utd
with lazy batches: the path a serl example script takes), to compare with bench.py's DataParallelLearner loop.
usage: python scripts/probes/agent_api_rate.py [steps] [stream]"""
import sys, time
import numpy as np, torch
import os; _R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
from helpers import make_spaces
from serl_amd.data.data_store import MemoryEfficientReplayBufferDataStore
from serl_amd.utils.launcher import make_drq_agent
from serl_amd.utils.synthetic import transition_stream
import itertools

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
KEYS, H, W, S, A, B = ("front", "wrist"), 128, 128, 24, 6, 256
osp, asp = make_spaces(KEYS, H, W, 3, 1, S, A)
rb = MemoryEfficientReplayBufferDataStore(osp, asp, 5000, image_keys=KEYS)
rb.seed(0)
for tr in itertools.islice(transition_stream(KEYS, H, W, 3, 1, S, A, 100, 1), 3000):
    rb.insert(tr)
obs = {k: np.zeros((1, H, W, 3), np.uint8) for k in KEYS}; obs["state"] = np.zeros((1, S), np.float32)
agent = make_drq_agent(1, obs, np.zeros((A,), np.float32), image_keys=KEYS, encoder_type="resnet-pretrained", batch_size=B)
it = rb.get_iterator(sample_args={"batch_size": B, "pack_obs_and_next_obs": True, "lazy": True})
import contextlib
own = len(sys.argv) > 2 and sys.argv[2] == "stream"     # run the loop under a non-default stream
ctx = torch.cuda.stream(torch.cuda.Stream()) if own else contextlib.nullcontext()
with ctx:
    for _ in range(10):
        agent, _ = agent.update_high_utd(next(it), utd_ratio=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        agent, info = agent.update_high_utd(next(it), utd_ratio=1)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"agent API ({'own stream' if own else 'default stream'}): {steps / dt:.1f} grad-steps/s ({1e3 * dt / steps:.3f} ms per update_high_utd; host enqueue "
      f"{1e3 * th / steps:.3f} ms per call), B={B}, 2x{H}x{W}, {steps} steps")
