"""GPU parity: the HIP replay path (through the C ABI) vs the reference-generated golden fixtures
and the NumPy oracle -- bit-exact (integer / byte work)."""
import itertools

import numpy as np
import pytest
import torch

from helpers import load_case, make_spaces, stream_for
from oracle.replay_oracle import ReplayOracle, random_shift

pytestmark = pytest.mark.gpu
CASES = ["small_wrap", "small_nowrap", "one_cam", "wrap_quirk"]


def _mk(m, cap=None):
    from serl_amd.data.data_store import MemoryEfficientReplayBufferDataStore
    osp, asp = make_spaces(m["keys"], m["H"], m["W"], m["C"], m["T"], m["S"], m["A"])
    return MemoryEfficientReplayBufferDataStore(osp, asp, cap or m["cap"], image_keys=m["keys"])


@pytest.mark.parametrize("name", CASES)
def test_matches_reference_golden(gpu, name):
    z, m = load_case(name)
    rb = _mk(m)
    rb.seed(m["rseed"])
    for tr in stream_for(m):
        rb.insert(tr)
    assert len(rb) == int(z["size"]) and rb.latest_data_id() == int(z["insert_index"])
    assert (rb.valid_mask() == z["valid"]).all()
    for s in range(m["ns"]):
        idx = rb.sample_indices(m["B"])
        assert (idx == z[f"idx_{s}"]).all(), "index stream must be bit-exact"
        b = rb.gather(idx)
        torch.cuda.synchronize()
        for k in m["keys"]:
            assert (b["observations"][k].cpu().numpy() == z[f"frames_{k}_{s}"]).all()
        assert (b["observations"]["state"].cpu().numpy() == z[f"state_{s}"]).all()
        assert (b["next_observations"]["state"].cpu().numpy() == z[f"next_state_{s}"]).all()
        assert (b["actions"].cpu().numpy() == z[f"actions_{s}"]).all()
        assert (b["rewards"].cpu().numpy() == z[f"rewards_{s}"]).all()
        assert (b["masks"].cpu().numpy() == z[f"masks_{s}"]).all()
        assert (b["dones"].cpu().numpy() == z[f"dones_{s}"]).all()


def test_rng_state_tracks_numpy(gpu):
    import ctypes as C
    from serl_amd import _lib
    z, m = load_case("small_nowrap")
    rb = _mk(m)
    rb.seed(123)
    for tr in stream_for(m):
        rb.insert(tr)
    o = ReplayOracle(m["keys"], m["H"], m["W"], m["C"], m["T"], m["S"], m["A"], m["cap"])
    o.seed(123)
    for tr in stream_for(m):
        o.insert(tr)
    for _ in range(5):
        assert (rb.sample_indices(257) == o.sample_indices(257)).all()
    st = (C.c_uint64 * 4)()
    has, u = C.c_int(), C.c_uint32()
    _lib.check(_lib.lib().serl_rb_rng_state(rb.handle, st, C.byref(has), C.byref(u)))
    nst = o.rng.bit_generator.state
    assert (st[0] << 64 | st[1]) == nst["state"]["state"] and (st[2] << 64 | st[3]) == nst["state"]["inc"]
    assert has.value == nst["has_uint32"] and (not has.value or u.value == nst["uinteger"])


@pytest.mark.parametrize("name", ["small_wrap", "wrap_quirk", "one_cam"])
def test_fused_gather_crop_matches_oracle(gpu, name):
    from serl_amd.agents.batch import DeviceBatch
    from serl_amd.data.data_store import gather_crop
    z, m = load_case(name)
    rb = _mk(m)
    rb.seed(m["rseed"])
    o = ReplayOracle(m["keys"], m["H"], m["W"], m["C"], m["T"], m["S"], m["A"], m["cap"])
    o.seed(m["rseed"])
    for tr in stream_for(m):
        rb.insert(tr)
        o.insert(tr)
    rng = np.random.default_rng(7)
    B = m["B"]
    for trial in range(4):
        idx = rb.sample_indices(B)
        assert (idx == o.sample_indices(B)).all()
        co = rng.integers(0, 9, size=(B, 2)).astype(np.int32)
        cn = rng.integers(0, 9, size=(B, 2)).astype(np.int32)
        if trial == 0:
            co[:] = 0
            cn[:] = 8
        if trial == 1:
            co[:, 0], co[:, 1], cn[:, 0], cn[:, 1] = 0, 8, 8, 0
        out = DeviceBatch(B, len(m["keys"]), m["H"], m["W"], m["C"], m["S"], m["A"], 0)
        gather_crop([(rb, idx)], co, cn, out)
        torch.cuda.synchronize()
        ob = o.gather(idx)
        fr = out.frames.cpu().numpy()
        for c, k in enumerate(m["keys"]):
            packed = ob["observations"][k]
            assert (fr[0, c] == random_shift(packed[:, 0], co)).all()
            assert (fr[1, c] == random_shift(packed[:, 1], cn)).all()
        assert (out.state[0].cpu().numpy() == ob["observations"]["state"][:, 0]).all()
        assert (out.state[1].cpu().numpy() == ob["next_observations"]["state"][:, 0]).all()
        assert (out.action.cpu().numpy() == ob["actions"]).all()
        assert (out.reward.cpu().numpy() == ob["rewards"]).all()
        assert (out.mask.cpu().numpy() == ob["masks"]).all()
        assert (out.done.cpu().numpy().astype(bool) == ob["dones"]).all()


def test_two_buffer_concat_and_lazy(gpu):
    """RLPD 50/50: concat_batches(online, demo) fused into one launch."""
    from serl_amd.agents.batch import DeviceBatch
    from serl_amd.data.data_store import concat_batches, gather_crop
    z, m = load_case("small_wrap")
    a, b = _mk(m), _mk(m, cap=64)
    a.seed(0)
    b.seed(1)
    oa = ReplayOracle(m["keys"], m["H"], m["W"], m["C"], m["T"], m["S"], m["A"], m["cap"])
    ob = ReplayOracle(m["keys"], m["H"], m["W"], m["C"], m["T"], m["S"], m["A"], 64)
    trs = list(stream_for(m))
    for tr in trs:
        a.insert(tr)
        oa.insert(tr)
    for tr in trs[:40]:
        b.insert(tr)
        ob.insert(tr)
    la, lb = a.sample(8, pack_obs_and_next_obs=True, lazy=True), b.sample(8, pack_obs_and_next_obs=True, lazy=True)
    lazy = concat_batches(la, lb, axis=0)
    assert lazy.batch_size == 16
    out = DeviceBatch(16, 2, m["H"], m["W"], m["C"], m["S"], m["A"], 0)
    gather_crop(lazy.parts, None, None, out)
    eager = lazy.materialize()
    torch.cuda.synchronize()
    fr = out.frames.cpu().numpy()
    for c, k in enumerate(m["keys"]):
        ref = np.concatenate([oa.gather(la.parts[0][1])["observations"][k],
                              ob.gather(lb.parts[0][1])["observations"][k]], axis=0)
        assert (fr[0, c] == ref[:, 0]).all() and (fr[1, c] == ref[:, 1]).all()
        assert (eager["observations"][k].cpu().numpy() == ref).all()
    rew = np.concatenate([oa.rewards[la.parts[0][1]], ob.rewards[lb.parts[0][1]]])
    assert (out.reward.cpu().numpy() == rew).all()


def test_error_behaviour(gpu):
    from serl_amd._lib import SerlError
    z, m = load_case("one_cam")
    rb = _mk(m)
    rb.seed(0)
    with pytest.raises(SerlError):
        rb.sample_indices(4)  # empty buffer
    with pytest.raises(NotImplementedError):
        rb.sample(4, indx=np.arange(4))  # memory_efficient_replay_buffer.py:123-124
    with pytest.raises(NotImplementedError):
        rb.get_latest_data(0)
    for tr in itertools.islice(stream_for(m), 5):
        rb.insert(tr)
    with pytest.raises(SerlError):
        rb.gather(np.array([999], np.int64))


def test_full_size_properties(gpu):
    """BASELINE shape (2 cams 128x128x3, S=24, A=6, B=256): identity crop == packed gather,
    byte checksum of checksums, and crop-by-rows equivalence."""
    from serl_amd.agents.batch import DeviceBatch
    from serl_amd.data.data_store import MemoryEfficientReplayBufferDataStore, gather_crop
    from serl_amd.utils.synthetic import transition_stream
    keys = ("front", "wrist")
    osp, asp = make_spaces(keys, 128, 128, 3, 1, 24, 6)
    rb = MemoryEfficientReplayBufferDataStore(osp, asp, 2000, image_keys=keys)
    rb.seed(0)
    for tr in itertools.islice(transition_stream(keys, seed=1234), 1500):
        rb.insert(tr)
    B = 256
    idx = rb.sample_indices(B)
    packed = rb.gather(idx)
    out = DeviceBatch(B, 2, 128, 128, 3, 24, 6, 0)
    gather_crop([(rb, idx)], None, None, out)
    torch.cuda.synchronize()
    for c, k in enumerate(keys):
        assert torch.equal(out.frames[0, c], packed["observations"][k][:, 0])
        assert torch.equal(out.frames[1, c], packed["observations"][k][:, 1])
    rng = np.random.default_rng(3)
    co = rng.integers(0, 9, size=(B, 2)).astype(np.int32)
    cn = rng.integers(0, 9, size=(B, 2)).astype(np.int32)
    gather_crop([(rb, idx)], co, cn, out)
    torch.cuda.synchronize()
    fr = out.frames.cpu().numpy()
    for c, k in enumerate(keys):
        p = packed["observations"][k].cpu().numpy()
        assert (fr[0, c] == random_shift(p[:, 0], co)).all()
        assert (fr[1, c] == random_shift(p[:, 1], cn)).all()
