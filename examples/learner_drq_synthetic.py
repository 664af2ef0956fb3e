"""The learner loop of the reference's examples/async_drq_sim/async_drq_sim.py:183-311 on the MI355X path.  The actor
side is an in-process mock (a thread with a TrainerClient + QueuedDataStore that ships synthetic transitions, requests
"send-stats" and receives every published network) talking to the learner's TrainerServer through serl_amd.transport
-- the agentlace call pattern of the reference script (server.register_data_store / start / publish_network,
client.update / request / recv_network_callback); real ZeroMQ is used when pyzmq + lz4 are installed.

    python examples/learner_drq_synthetic.py --steps 200 --batch_size 256 --critic_actor_ratio 4

Only the import lines differ from the reference script: make_drq_agent / make_replay_buffer / concat_batches come
from serl_amd, `lazy=True` lets gather + concat + unpack + random-shift crop fuse into the update, and checkpoints
go through serl_amd.utils.checkpoint (flax's file layout).
"""
import argparse
import itertools
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from serl_amd.transport import QueuedDataStore, TrainerClient, TrainerServer, make_trainer_config  # noqa: E402
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
    ap.add_argument("--steps_per_update", type=int, default=30)      # async_drq_sim.py:60
    ap.add_argument("--port", type=int, default=5488)
    a = ap.parse_args()

    env = _Env()
    agent = make_drq_agent(seed=42, sample_obs={"front": np.zeros((1, H, W, 3), np.uint8), "wrist": np.zeros((1, H, W, 3), np.uint8),
                                                "state": np.zeros((1, S), np.float32)},
                           sample_action=np.zeros((A,), np.float32), image_keys=KEYS, encoder_type="resnet-pretrained",
                           batch_size=a.batch_size)
    replay_buffer = make_replay_buffer(env, capacity=200000, type="memory_efficient_replay_buffer", image_keys=KEYS)
    demo_buffer = make_replay_buffer(env, capacity=10000, type="memory_efficient_replay_buffer", image_keys=KEYS)
    for tr in itertools.islice(transition_stream(KEYS, H, W, 3, 1, S, A, 100, 99), 2000):   # 20 demo trajectories
        demo_buffer.insert(tr)

    # ---- learner endpoint (async_drq_sim.py:202-212)
    stats_log = []

    def stats_callback(type: str, payload: dict) -> dict:   # noqa: A002
        assert type == "send-stats", f"Invalid request type: {type}"
        stats_log.append(payload)
        return {}

    cfg = make_trainer_config(port_number=a.port, broadcast_port=a.port + 1)
    server = TrainerServer(cfg, request_callback=stats_callback)
    server.register_data_store("actor_env", replay_buffer)
    server.start(threaded=True)

    # ---- mock actor (async_drq_sim.py:91-177 without the env / policy): its own thread, like the actor process
    stop_actor, networks = threading.Event(), []

    def actor_loop():
        data_store = QueuedDataStore(2000)
        client = TrainerClient("actor_env", "localhost", make_trainer_config(a.port, a.port + 1), data_store, wait_for_server=True)
        client.recv_network_callback(lambda params: networks.append(sorted(params.keys())))
        for step, tr in enumerate(transition_stream(KEYS, H, W, 3, 1, S, A, 100, 1234)):
            if stop_actor.is_set():
                break
            data_store.insert(tr)
            if step % a.steps_per_update == 0:
                client.update()
            if step % 200 == 0:
                client.request("send-stats", {"timer": {"total": time.time()}})
            if step >= a.training_starts:
                time.sleep(0.005)                                      # a real actor steps its env at 10-20 Hz
        client.stop()

    actor_thread = threading.Thread(target=actor_loop, name="mock-actor", daemon=True)
    actor_thread.start()
    while len(replay_buffer) < a.training_starts:                      # :215-226 "Filling up replay buffer"
        time.sleep(0.05)
    server.publish_network(agent.state.params)                          # :229 initial network

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
        if step > 0 and step % a.steps_per_update == 0:                  # :295-297
            server.publish_network(agent.state.params)
        if step % a.log_period == 0:
            info = update_info.resolve()                                 # synchronises; the reference logs to wandb here
            print(f"step {step:5d} updates {update_steps:6d} critic_loss {info['critic']['critic_loss']:.4f} "
                  f"actor_loss {info['actor']['actor_loss']:.4f} temperature {info['actor']['temperature']:.4f} "
                  f"{update_steps / (time.time() - t0):.1f} grad-steps/s", flush=True)
        if a.checkpoint_path and a.checkpoint_period and step and step % a.checkpoint_period == 0:
            save_checkpoint(a.checkpoint_path, agent, step=update_steps, keep=20)   # :303-307
    stop_actor.set()
    actor_thread.join(timeout=10)
    time.sleep(0.2)
    server.stop()
    print(f"done: {update_steps} grad-steps; transport: {server.transport}; server stats {server.stats}; "
          f"actor received {len(networks)} networks {networks[-1] if networks else None}; replay size {len(replay_buffer)}", flush=True)
    assert server.stats["transitions"] >= a.training_starts and len(networks) >= 1 and stats_log


if __name__ == "__main__":
    main()
