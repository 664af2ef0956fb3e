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
    # one tiny critic + actor update of the HIP agent against the fp64 oracle
    import agent_helpers as AH
    from oracle import drq_oracle as O
    cfg = O.Config(image_keys=("front", "wrist"), H=64, W=64, S=5, A=3)
    st, core = AH.make_pair(cfg, 8)
    b = AH.synth_batch(cfg, 8, seed=1)
    noise = O.make_noise(cfg, 8, seed=2)
    info, _ = O.update_high_utd(st, AH.batch_to_torch(b, torch.float64), O.noise_to_torch(noise, torch.float64), 1)
    core.update_high_utd(AH.batch_to_device(cfg, b), 1, AH.noise_to_device(cfg, noise))
    got = core.read_info()
    for k in ("critic_loss", "predicted_qs", "target_qs", "actor_loss", "entropy", "temperature_loss"):
        assert abs(got[k] - info[k]) < 1e-4 * max(1.0, abs(info[k])), (k, got[k], info[k])
    print("smoke: update_high_utd parity OK", {k: round(v, 5) for k, v in got.items()})
