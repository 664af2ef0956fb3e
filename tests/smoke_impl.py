"""One small invocation of the hot path on cuda:0, checked against the oracle (test infra)."""
import itertools

import numpy as np
import torch


def run():
    assert torch.cuda.is_available(), "smoke() needs a GPU"
    from oracle.replay_oracle import ReplayOracle, random_shift
    from serl_amd.agents.batch import DeviceBatch
    from serl_amd.data.data_store import MemoryEfficientReplayBufferDataStore, gather_crop
    from serl_amd.utils.synthetic import transition_stream

    class Sp:
        def __init__(self, shape):
            self.shape = shape

    class D:
        def __init__(self, s):
            self.spaces = s

    keys, H, W, C, S, A = ("front", "wrist"), 32, 32, 3, 8, 4
    osp = D({"front": Sp((1, H, W, C)), "state": Sp((1, S)), "wrist": Sp((1, H, W, C))})
    rb = MemoryEfficientReplayBufferDataStore(osp, Sp((A,)), 100, image_keys=keys)
    o = ReplayOracle(keys, H, W, C, 1, S, A, 100)
    rb.seed(0)
    o.seed(0)
    for tr in itertools.islice(transition_stream(keys, H, W, C, 1, S, A, 10, 3), 60):
        rb.insert(tr)
        o.insert(tr)
    idx = rb.sample_indices(16)
    assert (idx == o.sample_indices(16)).all()
    rng = np.random.default_rng(0)
    co, cn = rng.integers(0, 9, (16, 2)).astype(np.int32), rng.integers(0, 9, (16, 2)).astype(np.int32)
    out = DeviceBatch(16, 2, H, W, C, S, A, 0)
    gather_crop([(rb, idx)], co, cn, out)
    torch.cuda.synchronize()
    ob = o.gather(idx)
    fr = out.frames.cpu().numpy()
    for c, k in enumerate(keys):
        assert (fr[0, c] == random_shift(ob["observations"][k][:, 0], co)).all()
        assert (fr[1, c] == random_shift(ob["observations"][k][:, 1], cn)).all()
    print("smoke: replay gather+crop parity OK")
