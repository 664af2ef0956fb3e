"""The learner loop of the reference's examples/async_drq_sim/async_drq_sim.py:183-311 on the MI355X path, with
synthetic transitions standing in for the agentlace actor (the transport is outside this repo's scope).

    python examples/learner_drq_synthetic.py --steps 200 --batch_size 256 --critic_actor_ratio 4

Only the import lines differ from the reference script: make_drq_agent / make_replay_buffer / concat_batches come
from serl_amd, `lazy=True` lets gather + concat + unpack + random-shift crop fuse into the update, and checkpoints
go through serl_amd.utils.checkpoint (flax's file layout).
"""
import argparse
import itertools
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from serl_amd.utils.checkpoint import save_checkpoint  # noqa: E402
from serl_amd.utils.launcher import make_drq_agent, make_replay_buffer  # noqa: E402
from serl_amd.utils.synthetic import transition_stream  # noqa: E402
from serl_amd.utils.train_utils import concat_batches  # noqa: E402

KEYS, H, W, S, A = ("front", "wrist"), 128, 128, 7, 4


class _Sp:
    def __init__(self, shape):
        self.shape = shape


class _Env:  # observation/action spaces of the PandaPickCubeVision env wrapped as in the reference (T = 1)
    class _Obs:
        spaces = {"front": _Sp((1, H, W, 3)), "state": _Sp((1, S)), "wrist": _Sp((1, H, W, 3))}
    observation_space, action_space = _Obs(), _Sp((A,))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--batch_size", type=int, default=256)
    ap.add_argument("--critic_actor_ratio", type=int, default=4)
    ap.add_argument("--training_starts", type=int, default=1000)
    ap.add_argument("--checkpoint_path", default=None)
    ap.add_argument("--checkpoint_period", type=int, default=0)
    ap.add_argument("--log_period", type=int, default=20)
    a = ap.parse_args()

    env = _Env()
    agent = make_drq_agent(seed=42, sample_obs={"front": np.zeros((1, H, W, 3), np.uint8), "wrist": np.zeros((1, H, W, 3), np.uint8),
                                                "state": np.zeros((1, S), np.float32)},
                           sample_action=np.zeros((A,), np.float32), image_keys=KEYS, encoder_type="resnet-pretrained",
                           batch_size=a.batch_size)
    replay_buffer = make_replay_buffer(env, capacity=200000, type="memory_efficient_replay_buffer", image_keys=KEYS)
    demo_buffer = make_replay_buffer(env, capacity=10000, type="memory_efficient_replay_buffer", image_keys=KEYS)
    actor = transition_stream(KEYS, H, W, 3, 1, S, A, 100, 1234)        # stands in for the agentlace data stream
    for tr in itertools.islice(transition_stream(KEYS, H, W, 3, 1, S, A, 100, 99), 2000):   # 20 demo trajectories
        demo_buffer.insert(tr)
    for tr in itertools.islice(actor, a.training_starts):
        replay_buffer.insert(tr)

    half = {"batch_size": a.batch_size // 2, "pack_obs_and_next_obs": True, "lazy": True}
    replay_iterator, demo_iterator = replay_buffer.get_iterator(sample_args=half), demo_buffer.get_iterator(sample_args=half)
    t0, update_steps = time.time(), 0
    for step in range(a.steps):
        for _ in range(a.critic_actor_ratio - 1):                        # async_drq_sim.py:266-281
            batch = concat_batches(next(replay_iterator), next(demo_iterator), axis=0)
            agent, critics_info = agent.update_critics(batch)
            update_steps += 1
        batch = concat_batches(next(replay_iterator), next(demo_iterator), axis=0)
        agent, update_info = agent.update_high_utd(batch, utd_ratio=1)   # :283-292
        update_steps += 1
        if step % 4 == 0:
            replay_buffer.insert(next(actor))                            # the actor keeps sending transitions (~20 Hz)
        if step % a.log_period == 0:
            info = update_info.resolve()                                 # synchronises; the reference logs to wandb here
            print(f"step {step:5d} updates {update_steps:6d} critic_loss {info['critic']['critic_loss']:.4f} "
                  f"actor_loss {info['actor']['actor_loss']:.4f} temperature {info['actor']['temperature']:.4f} "
                  f"{update_steps / (time.time() - t0):.1f} grad-steps/s", flush=True)
            params = agent.state.params                                  # what server.publish_network(...) would send
            assert "modules_actor" in params
        if a.checkpoint_path and a.checkpoint_period and step and step % a.checkpoint_period == 0:
            save_checkpoint(a.checkpoint_path, agent, step=update_steps, keep=20)   # :303-307


if __name__ == "__main__":
    main()
