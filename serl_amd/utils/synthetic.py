"""Synthetic SERL transitions (SURVEY.md 8(d) distribution).

frames u8 iid uniform[0,255]; state,next_state ~ N(0,1) f32 [T,S]; action ~ U(-1,1) f32;
episodes of `episode_len` steps (done at t % episode_len == episode_len-1, mask = 1-done);
reward = 1.0 on done with p=0.5 else 0.  Shapes are what the actor sends on the wire
(reference examples/async_drq_sim/async_drq_sim.py:145-152).
"""
from __future__ import annotations

import numpy as np


def transition_stream(image_keys, H=128, W=128, C=3, T=1, S=24, A=6, episode_len=100, seed=1234):
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

    def new_obs():
        o = {"state": rng.standard_normal((T, S)).astype(np.float32)}
        for k in image_keys:
            o[k] = rng.integers(0, 256, size=(T, H, W, C), dtype=np.uint8)
        return o

    t = 0
    obs = new_obs()
    while True:
        nobs = new_obs()
        if T > 1:  # frame stacking: next stack = obs stack shifted by one frame
            for k in image_keys:
                nobs[k][:-1] = obs[k][1:]
        done = (t % episode_len) == episode_len - 1
        reward = 1.0 if (done and rng.random() < 0.5) else 0.0
        yield {
            "observations": obs,
            "next_observations": nobs,
            "actions": rng.uniform(-1.0, 1.0, size=(A,)).astype(np.float32),
            "rewards": np.float32(reward),
            "masks": np.float32(1.0 - float(done)),
            "dones": bool(done),
        }
        t += 1
        obs = new_obs() if done else nobs


def flat_stream(S, A, episode_len, seed):
    """Flat-observation transitions (async_sac_state_sim: PandaPickCube state obs)."""
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
    t = 0
    obs = rng.standard_normal(S).astype(np.float32)
    while True:
        nobs = rng.standard_normal(S).astype(np.float32)
        done = (t % episode_len) == episode_len - 1
        yield {"observations": obs, "next_observations": nobs, "actions": rng.uniform(-1, 1, A).astype(np.float32),
               "rewards": np.float32(1.0 if (done and rng.random() < 0.5) else 0.0), "masks": np.float32(1.0 - done),
               "dones": bool(done)}
        obs = rng.standard_normal(S).astype(np.float32) if done else nobs
        t += 1
