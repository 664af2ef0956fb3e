"""Worker of tests/test_dp_two_process_gpu.py: one rank of a batch-sharded data-parallel learner on the REAL HIP path.
Both ranks share the single GPU of the test box; the collective is gloo (RCCL needs one GPU per rank), staged through
host memory -- everything else (replicated replay buffers, identical index/crop/REDQ streams, per-rank shard, device
noise indexed by the global sample, gradient all-reduce between *_grads and apply) is the production code path."""
import itertools
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KEYS, H, W, S, A, B = ("front", "wrist"), 64, 64, 7, 4, 16


class _Sp:
    def __init__(self, shape):
        self.shape = shape


class _Obs:
    spaces = {"front": _Sp((1, H, W, 3)), "state": _Sp((1, S)), "wrist": _Sp((1, H, W, 3))}


def main():
    out_path, iters = sys.argv[1], int(sys.argv[2])
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from serl_amd.agents.batch import DeviceBatch
    from serl_amd.data.data_store import MemoryEfficientReplayBufferDataStore, gather_crop
    from serl_amd.parallel import DataParallelLearner, SerialSchedule
    from serl_amd.utils.launcher import make_drq_agent
    from serl_amd.utils.synthetic import transition_stream
    rb = MemoryEfficientReplayBufferDataStore(_Obs(), _Sp((A,)), 300, image_keys=KEYS)
    rb.seed(0)
    for tr in itertools.islice(transition_stream(KEYS, H, W, 3, 1, S, A, 20, 5), 150):
        rb.insert(tr)
    Bl = B // world
    obs = {"front": np.zeros((1, H, W, 3), np.uint8), "wrist": np.zeros((1, H, W, 3), np.uint8), "state": np.zeros((1, S), np.float32)}
    agent = make_drq_agent(3, obs, np.zeros((A,), np.float32), image_keys=KEYS, encoder_type="resnet-pretrained", batch_size=Bl)
    core = agent.core
    db = DeviceBatch(Bl, len(KEYS), H, W, 3, S, A, 0)

    def gather(parts, co, cn, slot):
        gather_crop(parts, co, cn, db)
        return db

    def all_reduce(t):                      # gloo on the host copy of the zero-copy gradient view
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)

    learner = DataParallelLearner(core, gather, [rb], [B], rank, world, all_reduce=all_reduce, seed=7, schedule=SerialSchedule())
    for _ in range(iters):
        learner.iteration(2)                # update_critics, then update_high_utd: production (device) noise
    torch.cuda.synchronize()
    if rank == 0:
        leaves = ["critic/w1", "critic/head/kernel", "actor/w2", "actor/mean/kernel", "enc/0/dense/kernel", "enc/proprio/dense/kernel",
                  "temp/lagrange"]
        np.savez(out_path, step=core.step, info=np.array(list(core.read_info().values()), np.float64),
                 **{k.replace("/", "__"): core.get("params", k) for k in leaves})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
