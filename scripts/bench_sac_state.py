"""Side measurement (not the official bench line): BASELINE.json configs[0] `async_sac_state_sim` on one MI355X --
state-only SAC, batch 256 x UTD 8 = 2048 sampled per iteration, `update_high_utd(utd_ratio=8)`
(examples/async_sac_state_sim/async_sac_state_sim.py:231,296), plain replay buffer in HBM -- next to the CPU port
(oracle, PyTorch-CPU fp32) on the box's host cores.  Prints one JSON line."""
import itertools
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S, A, B, UTD = 10, 4, 2048, 8


class _Box:
    def __init__(self, shape):
        self.shape = shape


class _Env:
    observation_space, action_space = _Box((S,)), _Box((A,))


def main():
    from serl_amd.utils.launcher import make_replay_buffer, make_sac_agent
    from serl_amd.utils.synthetic import flat_stream
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
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
    try:
        q = os.sched_getaffinity(0)
        ncpu = len(q)
        cq = open("/sys/fs/cgroup/cpu.max").read().split()
        if cq[0] != "max":
            ncpu = max(1, min(ncpu, int(int(cq[0]) / int(cq[1]))))
    except Exception:
        ncpu = os.cpu_count() or 1
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


if __name__ == "__main__":
    main()
