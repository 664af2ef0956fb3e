"""TEST INFRASTRUCTURE ONLY -- CPU restatement (PyTorch-CPU) of the reward classifier's forward pass
(serl_launcher/networks/reward_classifier.py:16-28,31-57: BinaryClassifier over EncodingWrapper(use_proprio=False,
enable_stacking=True) over the frozen ResNet-10 trunk), built from the update oracle's primitives.  Pinned by
tests/golden/classifier_*.npz: logits of the reference's own BinaryClassifier executed unmodified under the stand-ins of
oracle/jaxshim (tests/golden/make_golden_classifier.py).  Only tests/ may import this."""
import numpy as np
import torch

from . import drq_oracle as O


def make_params(image_keys, H, W, seed):
    """Deterministic (numpy-seeded) parameters in the product's flat leaf names: trunk leaves as in drq_oracle, camera
    heads 'enc/<key>/...', classifier head 'head/...'.  Shared by the golden generator (which injects them into the
    reference's parameter tree) and the parity tests (which load them into the HIP classifier)."""
    cfg = O.Config(image_keys=tuple(image_keys), H=H, W=W, S=4, A=2)
    trunk, theta = O.init_params(cfg, seed)
    rng = np.random.default_rng(seed + 1000)
    p = dict(trunk)
    for k in image_keys:
        for leaf in ("sle", "dense/kernel", "dense/bias", "ln/scale", "ln/bias"):
            p[f"enc/{k}/{leaf}"] = np.asarray(theta[f"enc/{k}/{leaf}"], np.float32)
    E = 256 * len(image_keys)
    p["head/dense0/kernel"] = (rng.standard_normal((E, 256)) / np.sqrt(E)).astype(np.float32)
    p["head/dense0/bias"] = (0.1 * rng.standard_normal(256)).astype(np.float32)
    p["head/ln/scale"] = (1.0 + 0.1 * rng.standard_normal(256)).astype(np.float32)
    p["head/ln/bias"] = (0.1 * rng.standard_normal(256)).astype(np.float32)
    p["head/dense1/kernel"] = (rng.standard_normal((256, 1)) / 16.0).astype(np.float32)
    p["head/dense1/bias"] = (0.1 * rng.standard_normal(1)).astype(np.float32)
    return p


def make_obs(image_keys, H, W, n, seed):
    rng = np.random.default_rng(seed)
    return {k: rng.integers(0, 256, (n, 1, H, W, 3), dtype=np.uint8) for k in image_keys}   # (B, T = 1, H, W, C)


def logits(params, image_keys, obs, dtype=torch.float64):
    """reward_classifier.py:20-28 with train=False (both Dropout layers are the identity)."""
    th = {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in params.items()}
    codes = []
    for k in image_keys:
        img = torch.tensor(np.asarray(obs[k]))
        img = img.reshape(img.shape[0], *img.shape[2:])                       # encoding.py:41-44 'B T H W C -> B H W (T C)', T = 1
        f = O.sle(O.trunk_forward(th, img, dtype), th[f"enc/{k}/sle"])       # resnet_v1.py:341-349
        z = f @ th[f"enc/{k}/dense/kernel"] + th[f"enc/{k}/dense/bias"]
        codes.append(torch.tanh(O.layer_norm(z, th[f"enc/{k}/ln/scale"], th[f"enc/{k}/ln/bias"])))   # :371-374
    x = torch.cat(codes, dim=-1)                                              # encoding.py:51 (use_proprio=False)
    x = x @ th["head/dense0/kernel"] + th["head/dense0/bias"]                 # reward_classifier.py:23
    x = torch.relu(O.layer_norm(x, th["head/ln/scale"], th["head/ln/bias"]))  # :25-26
    return (x @ th["head/dense1/kernel"] + th["head/dense1/bias"]).numpy()    # :27
