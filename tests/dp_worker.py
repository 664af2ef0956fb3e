"""Worker of tests/test_dp_two_process_gpu.py: one rank of a batch-sharded data-parallel learner on the REAL HIP path.
Both ranks share the single GPU of the test box; the collective is gloo (RCCL needs one GPU per rank), staged through
host memory -- everything else (replicated replay buffers, identical index/crop/REDQ streams, per-rank shard, device
noise indexed by the global sample, gradient all-reduce between *_grads and apply) is the production code path.
Transitions keep arriving at rank 0 between the steps (ReplicatedDataStore fans them out): every rank must apply them at
the same batch boundary, or the index streams -- and with them the parameters -- diverge."""
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
    from serl_amd.parallel import DataParallelLearner, SerialSchedule, TrunkFarmLearner
    farm = os.environ.get("SERL_TEST_PARALLEL", "dp") == "farm" and world > 1
    from serl_amd.utils.launcher import make_drq_agent
    from serl_amd.utils.synthetic import transition_stream
    from serl_amd.data.replicated import ReplicatedDataStore
    rb = ReplicatedDataStore(MemoryEfficientReplayBufferDataStore(_Obs(), _Sp((A,)), 300, image_keys=KEYS), rank, world, lag=2)
    rb.seed(0)
    stream = transition_stream(KEYS, H, W, 3, 1, S, A, 20, 5)
    if rank == 0:                           # the actor talks to rank 0 only
        for tr in itertools.islice(stream, 150):
            rb.insert(tr)
    rb.flush()                              # collective: the initial fill is in every replica
    assert len(rb) > 150
    Bl = B if farm else B // world          # a trunk farm keeps the full batch on every rank
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

    def send(t, dst, tag):                  # features of one batch, worker -> updater (gloo on the host copy)
        torch.cuda.synchronize()
        dist.send(t.cpu(), dst)

    def recv(t, src, tag):
        h = torch.empty(t.shape, dtype=t.dtype)
        dist.recv(h, src)
        t.copy_(h)

    if farm:
        learner = TrunkFarmLearner(core, gather, [rb], [B], rank, world, send=send, recv=recv, seed=7, schedule=SerialSchedule())
    else:
        learner = DataParallelLearner(core, gather, [rb], [B], rank, world, all_reduce=all_reduce, seed=7, schedule=SerialSchedule(),
                                      overlap_reduce=os.environ.get("SERL_TEST_OVERLAP", "1") == "1")   # (the opt-in bucketed path stays covered)
    drawn = []
    orig = rb.replica.sample_indices
    rb.replica.sample_indices = lambda n: (drawn.append(orig(n)) or drawn[-1])
    for _ in range(iters):
        if rank == 0:                       # 7 new transitions per iteration (an episode boundary every 20)
            for tr in itertools.islice(stream, 7):
                rb.insert(tr)
        learner.iteration(2)                # update_critics, then update_high_utd: production (device) noise
    rb.flush()
    torch.cuda.synchronize()
    leaves = ["critic/w1", "critic/head/kernel", "actor/w2", "actor/mean/kernel", "enc/0/dense/kernel", "enc/proprio/dense/kernel",
              "temp/lagrange"]
    np.savez(out_path if rank == 0 else out_path.replace(".npz", f".rank{rank}.npz"), step=core.step,
             info=np.array(list(core.read_info().values()), np.float64), valid=rb.valid_mask(), insert_index=rb.latest_data_id(),
             size=len(rb), idx=np.stack(drawn), **{k.replace("/", "__"): core.get("params", k) for k in leaves})
    rb.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
