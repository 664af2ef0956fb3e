"""CPU: host-side logic that needs no GPU (lazy batches, flax tree naming, initialisers, shims)."""
import os

import numpy as np
import pytest

from serl_amd.data.data_store import LazyBatch, concat_batches
from serl_amd.utils import init as pinit


def test_lazy_concat_order():
    a, b = object(), object()
    la, lb = LazyBatch([(a, np.arange(3))]), LazyBatch([(b, np.arange(5))])
    c = concat_batches(la, lb, axis=0)  # async_drq_sim.py:277: online first, then demo
    assert c.batch_size == 8 and c.parts[0][0] is a and c.parts[1][0] is b
    with pytest.raises(AssertionError):
        concat_batches(la, lb, axis=1)


def test_param_counts_match_survey():
    th = pinit.theta_shapes(2, 128, 128, 24, 6)
    n = sum(int(np.prod(s)) if len(s) else 1 for s in th.values())
    assert n == 4_609_998
    t = sum(int(np.prod(s)) for s in pinit.trunk_shapes().values())
    assert t == 4_905_792
    names = list(th)
    # arena order: camera heads | critic | proprio | actor | temperature (optimizer supports contiguous)
    assert names.index("critic/w1") < names.index("enc/proprio/dense/kernel") < names.index("actor/w1")
    assert names[-1] == "temp/lagrange"


def test_flax_tree_paths():
    from serl_amd.agents.flax_tree import theta_paths, _trunk_paths
    tp = theta_paths(("front", "wrist"))
    assert tp["enc/1/sle"] == [("modules_actor", "encoder", "encoder_wrist", "SpatialLearnedEmbeddings_0", "kernel")]
    assert tp["critic/head/kernel"] == [("modules_critic", "Dense_0", "kernel")]
    assert tp["temp/lagrange"] == [("modules_temperature", "lagrange")]
    assert set(tp) == set(pinit.theta_shapes(2, 128, 128, 24, 6))
    assert set(_trunk_paths()) == set(pinit.trunk_shapes())


def test_initialisers_are_seeded_and_sane():
    a = pinit.init_theta(2, 64, 64, 5, 3, seed=1)
    b = pinit.init_theta(2, 64, 64, 5, 3, seed=1)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert abs(float(np.log1p(np.exp(a["temp/lagrange"]))) - 1e-2) < 1e-6   # softplus(lambda0) = temperature_init
    assert a["critic/w1"].shape == (10, 2 * 256 + 64 + 3, 256)
    t = pinit.init_trunk(seed=1)
    assert t["trunk/block1/proj"].shape == (1, 1, 64, 128) and "trunk/block0/proj" not in t


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from serl_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.SerlError, match="no CPU fallback"):
        _lib.lib()


def test_checkpoint_file_format(tmp_path):
    """flax.serialization layout: msgpack map, ndarrays as ExtType(1, msgpack((shape, dtype, bytes)))."""
    import msgpack
    from types import SimpleNamespace
    from serl_amd.utils import checkpoint as ck
    tree = {"modules_actor": {"Dense_1": {"kernel": np.arange(6, dtype=np.float32).reshape(2, 3)}}}
    st = SimpleNamespace(step=7, params=tree, target_params=tree, rng=np.array([0, 3], np.uint32),
                         opt_states={"critic": {"count": 7, "mu": tree, "nu": tree}})
    agent = SimpleNamespace(state=st)
    for s in (5, 6, 7):
        st.step = s
        path = ck.save_checkpoint(str(tmp_path), agent, step=s, keep=2)
    assert sorted(p.name for p in tmp_path.iterdir()) == ["checkpoint_6", "checkpoint_7"]
    assert ck.latest_checkpoint(str(tmp_path)) == path
    with pytest.raises(ValueError):
        ck.save_checkpoint(str(tmp_path), agent, step=7)
    seen = []
    raw = msgpack.unpackb(open(path, "rb").read(), raw=False, ext_hook=lambda c, d: seen.append(c) or ck._unpack_ext(c, d))
    assert set(seen) == {1} and set(raw) == {"step", "params", "target_params", "opt_states", "rng"}
    k = raw["params"]["modules_actor"]["Dense_1"]["kernel"]
    assert k.dtype == np.float32 and k.shape == (2, 3) and np.array_equal(k, tree["modules_actor"]["Dense_1"]["kernel"])
    assert raw["rng"].dtype == np.uint32 and int(raw["step"]) == 7


def test_lazy_batches_know_the_next_batch():
    """get_iterator's prefetch queue already holds the next sample: lazy batches expose it (peek_next) through
    concat_batches, which is what lets the agent run the next batch's trunk under the current update."""
    from serl_amd.data.data_store import MemoryEfficientReplayBufferDataStore as Store

    class Fake:
        def __init__(self, base):
            self.n = base
        def sample(self, **kw):
            self.n += 1
            return LazyBatch([(self, np.array([self.n]))])
        get_iterator = Store.get_iterator

    a, b = Fake(0), Fake(100)
    ia, ib = a.get_iterator(sample_args={}), b.get_iterator(sample_args={})
    cur = concat_batches(next(ia), next(ib), axis=0)
    assert [int(ix[0]) for _, ix in cur.parts] == [1, 101]
    nxt = cur.peek_next()
    assert [int(ix[0]) for _, ix in nxt.parts] == [2, 102] and cur.peek_next() is nxt
    cur2 = concat_batches(next(ia), next(ib), axis=0)
    assert all(p1[1] is p2[1] for p1, p2 in zip(cur2.parts, nxt.parts))      # the very same index arrays
    assert [int(ix[0]) for _, ix in cur2.peek_next().parts] == [3, 103]
    assert LazyBatch([(a, np.array([1]))]).peek_next is None                 # plain sample(): nothing to peek


def test_bench_flop_model_matches_the_survey():
    """bench.py's roofline uses 2*M*K*Cout per conv launch: the trunk totals must equal SURVEY.md appendix D
    (580,386,816 FLOP per 128x128 image = 77.07 M for conv_init + 503.32 M for the 11 block convs)."""
    import bench
    m = bench.conv_macs_per_image()
    blocks = 2 * sum(v for k, v in m.items() if k.startswith("conv_igemm"))
    assert 2 * m["conv_init"] == 77_070_336 and blocks == 503_316_480
    assert 2 * m["conv_init"] + blocks == 580_386_816
    assert sum(1 for k in m if k.startswith("conv_igemm")) == 11


def _pickle_tree(trunk):
    """A tree shaped like resnet10_params.pkl (train_utils.py:113-127: top-level keys conv_init / norm_init /
    ResNetBlock_i matched against `pretrained_encoder`'s children)."""
    from serl_amd.agents.flax_tree import _trunk_paths
    t = {}
    for leaf, sub in _trunk_paths().items():
        d = t
        for p in sub[:-1]:
            d = d.setdefault(p, {})
        d[sub[-1]] = trunk[leaf]
    return t


def test_pretrained_pickle_maps_onto_the_trunk_leaves(tmp_path):
    """load_resnet10_params (train_utils.py:69-130) on a synthetic pickle: every leaf lands on its flat trunk leaf;
    a top-level key the pickle lacks keeps its value (train_utils.py:124-127); a missing file is an error (no download)."""
    import pickle
    from serl_amd.agents.flax_tree import trunk_from_flax
    from serl_amd.utils.train_utils import load_resnet10_params
    trunk = pinit.init_trunk(seed=5)
    tree = _pickle_tree(trunk)
    assert set(tree) == {"conv_init", "norm_init", "ResNetBlock_0", "ResNetBlock_1", "ResNetBlock_2", "ResNetBlock_3"}
    flat = trunk_from_flax(tree)
    assert set(flat) == set(trunk)
    for k in trunk:
        np.testing.assert_array_equal(flat[k], trunk[k])
    part = {k: v for k, v in tree.items() if k != "ResNetBlock_3"}
    assert not any(k.startswith("trunk/block3/") for k in trunk_from_flax(part))
    assert len(trunk_from_flax(part)) == len(trunk) - sum(k.startswith("trunk/block3/") for k in trunk)

    class FakeAgent:
        def load_trunk_params(self, t):
            self.got = trunk_from_flax(t)
            return self
    f = tmp_path / "resnet10_params.pkl"
    pickle.dump(tree, open(f, "wb"))
    a = load_resnet10_params(FakeAgent(), ("front", "wrist"), file_path=str(f))
    np.testing.assert_array_equal(a.got["trunk/block2/conv1"], trunk["trunk/block2/conv1"])
    with pytest.raises(FileNotFoundError):
        load_resnet10_params(FakeAgent(), ("front",), file_path=str(tmp_path / "absent.pkl"))


def test_restore_finds_the_trunk_under_any_camera():
    """load_state_dict (N1): the shared frozen trunk may sit under any camera of a real-flax checkpoint (layout unverified
    without a flax install) -- it is found wherever `pretrained_encoder` is, and a tree without one is a KeyError."""
    from types import SimpleNamespace
    from serl_amd.agents.flax_tree import _trunk_paths, theta_paths
    from serl_amd.utils.checkpoint import load_state_dict
    keys = ("front", "wrist")
    theta = pinit.init_theta(2, 32, 32, 4, 2, seed=2)
    trunk = pinit.init_trunk(seed=2)

    def tree(owner):
        t = {}
        def put(path, v):
            d = t
            for p in path[:-1]:
                d = d.setdefault(p, {})
            d[path[-1]] = v
        for leaf, paths in theta_paths(keys).items():
            put(paths[0], theta[leaf])
        for leaf, sub in _trunk_paths().items():
            if owner is not None:
                put(("modules_actor", "encoder", f"encoder_{owner}", "pretrained_encoder") + sub, trunk[leaf])
        return t

    class Core:
        cfg = SimpleNamespace(encoder_type=0)
        def __init__(self):
            self.got = {}
        def set(self, section, leaf, v):
            self.got[(section, leaf)] = np.asarray(v)

    for owner in keys:
        core = Core()
        load_state_dict(SimpleNamespace(core=core, image_keys=keys), {"params": tree(owner)})
        for leaf in trunk:
            np.testing.assert_array_equal(core.got[("params", leaf)], trunk[leaf])
        np.testing.assert_array_equal(core.got[("params", "enc/1/sle")], theta["enc/1/sle"])
    with pytest.raises(KeyError, match="pretrained_encoder"):
        load_state_dict(SimpleNamespace(core=Core(), image_keys=keys), {"params": tree(None)})


class _ListStore:
    def __init__(self):
        self.items = []

    def insert(self, t):
        self.items.append(t)

    def __len__(self):
        return len(self.items)


def _demo_files(tmp_path, n_files=2, per_file=3, D=14):
    import pickle
    rng = np.random.default_rng(5)
    paths = []
    for f in range(n_files):
        demo = []
        for i in range(per_file):
            demo.append({"observations": {"state": rng.normal(size=(1, D)).astype(np.float32),
                                          "wrist": rng.integers(0, 255, (1, 4, 4, 3), dtype=np.uint8)},
                         "next_observations": {"state": rng.normal(size=(1, D)).astype(np.float32),
                                               "wrist": rng.integers(0, 255, (1, 4, 4, 3), dtype=np.uint8)},
                         "actions": rng.normal(size=4).astype(np.float32), "rewards": float(i), "masks": 1.0, "dones": False})
        p = tmp_path / f"demo{f}.pkl"
        with open(p, "wb") as fh:
            pickle.dump(demo, fh)
        paths.append(str(p))
    return paths


def _reference_functions(names):
    """The named top-level functions of the reference's data/data_store.py, executed from where they lie (the module itself
    imports agentlace, which is not installed; these functions only need pickle / numpy / copy)."""
    import ast
    path = "/root/reference/serl_launcher/serl_launcher/data/data_store.py"
    if not os.path.exists(path):
        pytest.skip("/root/reference not present")
    tree = ast.parse(open(path).read())
    mod = ast.Module([n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names], type_ignores=[])
    ns = {"DataStoreBase": object}
    exec(compile(mod, path, "exec"), ns)
    return [ns[n] for n in names]


def _same_tree(a, b):
    if isinstance(a, dict):
        assert set(a) == set(b)
        for k in a:
            _same_tree(a[k], b[k])
    elif isinstance(a, np.ndarray):
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)
    else:
        assert a == b


def test_populate_data_store_matches_the_reference(tmp_path, capsys):
    from serl_amd.data.data_store import populate_data_store, populate_data_store_with_z_axis_only
    from serl_amd.utils import launcher
    paths = _demo_files(tmp_path)
    ref_plain, ref_z = _reference_functions(["populate_data_store", "populate_data_store_with_z_axis_only"])
    for ours, ref in ((populate_data_store, ref_plain), (populate_data_store_with_z_axis_only, ref_z)):
        a, b = _ListStore(), _ListStore()
        assert ours(a, paths) is a
        ref(b, paths)
        assert len(a) == len(b) == 6
        for x, y in zip(a.items, b.items):
            _same_tree(x, y)
    assert a.items[0]["observations"]["state"].shape == (1, 14 - 5)      # columns 4, 5, 7, 8, 9 dropped
    assert "Loaded 6 transitions." in capsys.readouterr().out
    one = populate_data_store(_ListStore(), paths[0])                    # a bare string is one path, not a list of characters
    assert len(one) == 3
    cfg = launcher.make_trainer_config()                                 # launcher.py:171-177 exports it too
    assert (cfg.port_number, cfg.broadcast_port, list(cfg.request_types)) == (5488, 5489, ["send-stats"])
