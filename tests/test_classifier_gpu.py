"""GPU: reward-classifier inference in libserl_mi355.so (csrc/classifier.hip) through the reference-named Python API
(serl_amd/networks/reward_classifier.py) against logits of the reference's own BinaryClassifier
(tests/golden/classifier_*.npz) and against the fp64 oracle; checkpoint -> load_classifier_func round trip."""
import os
import pickle

import numpy as np
import pytest

from oracle import classifier_oracle as CO

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-4


def _case(name):
    g = np.load(os.path.join(GOLD, f"classifier_{name}.npz"))
    keys = tuple(str(k) for k in g["image_keys"])
    return g, keys, int(g["H"]), int(g["W"]), int(g["n"]), int(g["seed"])


def _pickle_tree(params):
    from serl_amd.agents.flax_tree import _trunk_paths
    t = {}
    for leaf, sub in _trunk_paths().items():
        d = t
        for p in sub[:-1]:
            d = d.setdefault(p, {})
        d[sub[-1]] = params[leaf]
    return t


@pytest.mark.parametrize("name", ["two_cams_128", "one_cam_64"])
def test_logits_equal_the_reference_classifier(gpu, name):
    from serl_amd.networks.reward_classifier import Classifier
    g, keys, H, W, n, seed = _case(name)
    params = CO.make_params(keys, H, W, seed)
    obs = CO.make_obs(keys, H, W, n, seed + 1)
    c = Classifier(keys, H, W, max_batch=3).load_flat(params)      # max_batch < n: the batch is processed in pieces
    out = c.logits(obs)
    assert out.shape == (n, 1)
    err = np.abs(out - g["logits"]).max() / max(1.0, np.abs(g["logits"]).max())
    print(f"classifier {name}: max logit error vs the reference = {err:.2e}")
    assert err < TOL
    one = c.logits({k: v[0] for k, v in obs.items()})               # unbatched (T, H, W, C): shape (1,), as `.item()` needs
    assert one.shape == (1,) and abs(float(one[0]) - float(g["logits_unbatched"].reshape(-1)[0])) < TOL
    # a larger batch than the golden holds: against the fp64 oracle
    obs2 = CO.make_obs(keys, H, W, 9, seed + 5)
    ref2 = CO.logits(params, keys, obs2)
    assert np.abs(c.logits(obs2) - ref2).max() < TOL
    # the flax-layout export is the tree the reference builds
    ref_paths = {str(p): str(s) for p, s in zip(g["tree_paths"], g["tree_shapes"])}
    got = {}

    def walk(t, pre=()):
        for k, v in t.items():
            if isinstance(v, dict):
                walk(v, pre + (k,))
            else:
                got["/".join(pre + (k,))] = str(tuple(v.shape))
    walk(c.params)
    assert got == ref_paths


def test_create_save_and_load_classifier_func(gpu, tmp_path):
    """create_classifier (synthetic resnet10 pickle) -> checkpoint in the flax layout of the classifier TrainState ->
    load_classifier_func -> same logits; apply_fn with a foreign parameter tree loads it first."""
    from serl_amd.networks.reward_classifier import create_classifier, load_classifier_func
    from serl_amd.utils.checkpoint import write_checkpoint_tree
    keys, H, W = ("front", "wrist"), 64, 64
    params = CO.make_params(keys, H, W, 3)
    pkl = tmp_path / "resnet10_params.pkl"
    pickle.dump(_pickle_tree(params), open(pkl, "wb"))
    sample = {k: np.zeros((1, 1, H, W, 3), np.uint8) for k in keys}
    c = create_classifier(np.array([0, 5], np.uint32), sample, list(keys), pretrained_encoder_path=str(pkl))
    assert np.array_equal(c.get("trunk/block2/conv1"), params["trunk/block2/conv1"].reshape(-1))
    obs = CO.make_obs(keys, H, W, 4, 9)
    a = c.apply_fn({"params": c.params}, obs, train=False)
    tree = c.params
    flat = {k: v for k, v in params.items() if k.startswith("trunk/")}
    for k in keys:
        sub = tree["encoder_def"][f"encoder_{k}"]
        flat.update({f"enc/{k}/sle": sub["SpatialLearnedEmbeddings_0"]["kernel"], f"enc/{k}/dense/kernel": sub["Dense_0"]["kernel"],
                     f"enc/{k}/dense/bias": sub["Dense_0"]["bias"], f"enc/{k}/ln/scale": sub["LayerNorm_0"]["scale"],
                     f"enc/{k}/ln/bias": sub["LayerNorm_0"]["bias"]})
    flat.update({"head/dense0/kernel": tree["Dense_0"]["kernel"], "head/dense0/bias": tree["Dense_0"]["bias"],
                 "head/ln/scale": tree["LayerNorm_0"]["scale"], "head/ln/bias": tree["LayerNorm_0"]["bias"],
                 "head/dense1/kernel": tree["Dense_1"]["kernel"], "head/dense1/bias": tree["Dense_1"]["bias"]})
    assert np.abs(a - CO.logits(flat, keys, obs)).max() < TOL
    write_checkpoint_tree(str(tmp_path / "ckpt"), {"step": np.int32(100), "params": c.params, "opt_state": {}}, step=100)
    func = load_classifier_func(np.array([0, 6], np.uint32), sample, list(keys), str(tmp_path / "ckpt"),
                                pretrained_encoder_path=str(pkl))
    assert np.array_equal(func(obs), a)
    other = create_classifier(7, sample, list(keys), pretrained_encoder_path=str(pkl))
    assert np.abs(other.logits(obs) - a).max() > 1e-6          # different head initialisation
    assert np.array_equal(other.apply_fn({"params": c.params}, obs), a)
    with pytest.raises(NotImplementedError):
        c.apply_fn({"params": c.params}, obs, train=True)
