"""Generates tests/golden/classifier_*.npz: logits of the reference's OWN BinaryClassifier
(/root/reference/serl_launcher/serl_launcher/networks/reward_classifier.py, executed unmodified under the stand-ins of
oracle/jaxshim, fp64) on seeded observations, with its parameter tree overwritten by oracle.classifier_oracle.make_params
(numpy-seeded, so the fixture stays a few hundred bytes: the parity tests rebuild the same parameters from the seed).
Also records the parameter-tree paths and shapes the reference builds.  Run in the build container:
    python tests/golden/make_golden_classifier.py"""
import os
import pickle
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_update_shim as R   # noqa: E402

R.install(True)
import jax   # noqa: E402
import jax.numpy as jnp   # noqa: E402
from oracle import classifier_oracle as CO, ref_update_runner as RR   # noqa: E402
from serl_amd.agents.flax_tree import _trunk_paths, trunk_owner   # noqa: E402  (pure-python path tables)
from serl_amd.networks.reward_classifier import _CAM_PATHS, _HEAD_PATHS   # noqa: E402
from serl_launcher.networks.reward_classifier import create_classifier   # noqa: E402

CASES = {"two_cams_128": (("front", "wrist"), 128, 128, 5, 11), "one_cam_64": (("image",), 64, 64, 4, 12)}


def put(tree, path, v):
    d = tree
    for p in path[:-1]:
        d = d[p]
    assert tuple(d[path[-1]].shape) == tuple(np.shape(v)), (path, d[path[-1]].shape, np.shape(v))
    d[path[-1]] = jnp.asarray(np.asarray(v, np.float64))


def main():
    for name, (keys, H, W, n, seed) in CASES.items():
        params = CO.make_params(keys, H, W, seed)
        d = tempfile.mkdtemp()
        pkl = os.path.join(d, "resnet10_params.pkl")
        pickle.dump(RR.pretrained_pickle_tree({k: v for k, v in params.items() if k.startswith("trunk/")}), open(pkl, "wb"))
        sample = {k: jnp.asarray(np.zeros((1, 1, H, W, 3), np.uint8)) for k in keys}
        c = create_classifier(jax.random.PRNGKey(0), sample, list(keys), pretrained_encoder_path=pkl)
        tree = c.params.unfreeze()
        shapes = {}

        def walk(t, pre=()):
            for k, v in t.items():
                if isinstance(v, dict):
                    walk(v, pre + (k,))
                else:
                    shapes["/".join(pre + (k,))] = tuple(v.shape)
        walk(tree)
        hw = (H // 32) * (W // 32)
        side = int(round(hw ** 0.5))
        for k in keys:
            for leaf, (mod, pn) in _CAM_PATHS.items():
                v = params[f"enc/{k}/{leaf}"]
                if leaf == "sle":
                    v = v.reshape(side, hw // side, 512, 8)
                elif leaf == "dense/kernel":
                    v = v.reshape(4096, 256)
                put(tree, ("encoder_def", f"encoder_{k}", mod, pn), v)
        owner = trunk_owner(keys)
        assert "pretrained_encoder" in tree["encoder_def"][f"encoder_{owner}"]
        assert sum("pretrained_encoder" in tree["encoder_def"][f"encoder_{k}"] for k in keys) == 1   # ONE shared trunk
        for leaf, (mod, pn) in _HEAD_PATHS.items():
            put(tree, (mod, pn), params[leaf])
        obs = CO.make_obs(keys, H, W, n, seed + 1)
        out = c.apply_fn({"params": tree}, {k: jnp.asarray(v) for k, v in obs.items()}, train=False)
        logits = np.asarray(out, np.float64)
        single = np.asarray(c.apply_fn({"params": tree}, {k: jnp.asarray(v[0]) for k, v in obs.items()}, train=False), np.float64)
        ours = CO.logits(params, keys, obs)
        print(name, "reference logits", logits.reshape(-1), "oracle max diff", np.abs(ours - logits).max(), "unbatched", single.shape)
        np.savez(os.path.join(ROOT, "tests", "golden", f"classifier_{name}.npz"), logits=logits, logits_unbatched=single,
                 image_keys=np.array(keys), H=H, W=W, n=n, seed=seed, trunk_owner=owner,
                 tree_paths=np.array(sorted(shapes)), tree_shapes=np.array([str(shapes[k]) for k in sorted(shapes)]))


if __name__ == "__main__":
    main()
