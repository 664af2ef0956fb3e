"""CPU: pins oracle/replay_oracle.py against the reference-generated golden fixtures and, when
/root/reference is present, against the reference class itself."""
import itertools
import os

import numpy as np
import pytest

from oracle import ref_shim
from oracle.replay_oracle import PCG64Py, ReplayOracle, concat_batches, random_shift, unpack
from helpers import load_case, stream_for

CASES = ["small_wrap", "small_nowrap", "one_cam", "wrap_quirk"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_golden(name):
    z, m = load_case(name)
    o = ReplayOracle(m["keys"], m["H"], m["W"], m["C"], m["T"], m["S"], m["A"], m["cap"])
    o.seed(m["rseed"])
    for tr in stream_for(m):
        o.insert(tr)
    assert len(o) == int(z["size"]) and o.insert_index == int(z["insert_index"])
    assert (o.valid == z["valid"]).all()
    for s in range(m["ns"]):
        idx = o.sample_indices(m["B"])
        assert (idx == z[f"idx_{s}"]).all()
        b = o.gather(idx)
        for k in m["keys"]:
            assert (b["observations"][k] == z[f"frames_{k}_{s}"]).all()
        assert (b["observations"]["state"] == z[f"state_{s}"]).all()
        assert (b["next_observations"]["state"] == z[f"next_state_{s}"]).all()
        for f in ("actions", "rewards", "masks", "dones"):
            assert (b[f] == z[f"{f}_{s}"]).all()


@pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference not present")
def test_oracle_matches_reference_live():
    from serl_amd.utils.synthetic import transition_stream
    keys, H, W, C, T, S, A, cap = ("a", "b"), 16, 16, 3, 1, 4, 2, 23
    Ref = ref_shim.load_reference_buffer_cls()
    osp, asp = ref_shim.make_spaces(keys, H, W, C, T, S, A)
    ref = Ref(osp, asp, cap, pixel_keys=keys)
    ref.seed(9)
    o = ReplayOracle(keys, H, W, C, T, S, A, cap)
    o.seed(9)
    for n, tr in enumerate(itertools.islice(transition_stream(keys, H, W, C, T, S, A, 5, 77), 120)):
        ref.insert(tr)
        o.insert(tr)
        assert (ref._is_correct_index == o.valid).all()
        if n % 7 == 6:
            rb = ref.sample(8, pack_obs_and_next_obs=True)
            ob = o.sample(8)
            for k in keys:
                assert (rb["observations"][k] == ob["observations"][k]).all()
            assert (np.asarray(rb["rewards"]) == ob["rewards"]).all()


def test_pcg64_restatement_matches_numpy():
    for seed in (0, 1, 42, 12345):
        g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        p = PCG64Py.from_numpy(g)
        for n in (2, 3, 7, 1000, 199999, 200000, 2**31 + 5):
            a = g.integers(n, size=33)
            b = np.array([p.bounded(n) for _ in range(33)])
            assert (a == b).all()
            assert int(g.integers(n)) == p.bounded(n)
        st = g.bit_generator.state
        assert st["state"]["state"] == p.state and st["has_uint32"] == p.has_uint32


def test_random_shift_properties():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(5, 12, 16, 3), dtype=np.uint8)
    ident = random_shift(img, np.full((5, 2), 4))
    assert (ident == img).all()
    off = np.array([[0, 0], [8, 8], [0, 8], [8, 0], [3, 6]])
    out = random_shift(img, off)
    pad = np.pad(img, ((0, 0), (4, 4), (4, 4), (0, 0)), mode="edge")
    for n in range(5):
        dy, dx = off[n]
        assert (out[n] == pad[n, dy:dy + 12, dx:dx + 16]).all()


def test_concat_and_unpack():
    a = {"observations": {"state": np.zeros((2, 1, 3)), "im": np.zeros((2, 2, 4, 4, 3), np.uint8)},
         "next_observations": {"state": np.zeros((2, 1, 3))}, "rewards": np.zeros(2)}
    b = {"observations": {"state": np.ones((3, 1, 3)), "im": np.ones((3, 2, 4, 4, 3), np.uint8)},
         "next_observations": {"state": np.ones((3, 1, 3))}, "rewards": np.ones(3)}
    c = concat_batches(a, b)
    assert c["rewards"].tolist() == [0, 0, 1, 1, 1]
    u = unpack(c, ("im",))
    assert u["observations"]["im"].shape == (5, 1, 4, 4, 3)
    assert u["next_observations"]["im"].shape == (5, 1, 4, 4, 3)


# ---- plain ReplayBuffer of flat observations (BASELINE.json configs[0], async_sac_state_sim) -------------------
PLAIN = ["plain_wrap", "plain_nowrap"]


def _plain_oracle(g):
    import itertools
    from oracle.replay_oracle import PlainReplayOracle
    from serl_amd.utils.synthetic import flat_stream
    S, A, cap, n_ins, ep, sseed, rseed, B, ns = [int(x) for x in g["meta"]]
    o = PlainReplayOracle(S, A, cap)
    o.seed(rseed)
    for tr in itertools.islice(flat_stream(S, A, ep, sseed), n_ins):
        o.insert(tr)
    return o, B, ns


@pytest.mark.parametrize("name", PLAIN)
def test_plain_oracle_matches_reference_golden(name):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"replay_{name}.npz"))
    o, B, ns = _plain_oracle(g)
    assert len(o) == int(g["size"]) and o.insert_index == int(g["insert_index"])
    for s in range(ns):
        idx = o.sample_indices(B)
        assert np.array_equal(idx, g[f"idx_{s}"])
        b = o.gather(idx)
        for f in ("observations", "next_observations", "actions", "rewards", "masks", "dones"):
            assert np.array_equal(b[f], g[f"{f}_{s}"]), (s, f)


@pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference not present")
def test_plain_oracle_matches_live_reference():
    import itertools
    from oracle.replay_oracle import PlainReplayOracle
    from serl_amd.utils.synthetic import flat_stream
    Ref = ref_shim.load_reference_plain_buffer_cls()
    S, A, cap = 7, 3, 41
    ref = Ref(ref_shim.Box(-np.inf, np.inf, (S,), np.float32), ref_shim.Box(-1, 1, (A,), np.float32), cap)
    o = PlainReplayOracle(S, A, cap)
    ref.seed(5); o.seed(5)
    for i, tr in enumerate(itertools.islice(flat_stream(S, A, 6, 99), 100)):
        ref.insert(tr); o.insert(tr)
        if i % 17 == 16:
            b = ref.sample(13)
            idx = o.sample_indices(13)
            ob = o.gather(idx)
            for f in ("observations", "next_observations", "actions", "rewards", "masks", "dones"):
                assert np.array_equal(np.asarray(b[f]), ob[f]), (i, f)
