"""Reward classifier, inference side, with the reference's names (serl_launcher/networks/reward_classifier.py):

    create_classifier(key, sample, image_keys, pretrained_encoder_path)   (:31-90)
    load_classifier_func(key, sample, image_keys, checkpoint_path, step)  (:93-113) -> func(obs) -> logits

The forward pass (frozen ResNet-10 trunk -> per camera SpatialLearnedEmbeddings / Dense / LayerNorm / tanh -> Dense(256)
-> LayerNorm -> ReLU -> Dense(1), Dropout = identity at train=False) runs in libserl_mi355.so (csrc/classifier.hip); no
CPU fallback.  Training the classifier (examples/.../train_reward_classifier.py) is outside the hot path: checkpoints
written by the reference's trainer are read here (flax msgpack layout, see utils/checkpoint.py).
"""
from __future__ import annotations

import ctypes as C
import os
import pickle
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from .. import _lib
from ..agents.flax_tree import _trunk_paths, trunk_from_flax, trunk_owner


class SerlClassifierCfg(C.Structure):
    _fields_ = [("device", C.c_int), ("n_cam", C.c_int), ("H", C.c_int), ("W", C.c_int), ("max_batch", C.c_int)]


def _declare(lib):
    if getattr(lib, "_serl_classifier_declared", False):
        return
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    sigs = {
        "serl_classifier_create": [C.POINTER(SerlClassifierCfg), C.POINTER(vp)],
        "serl_classifier_destroy": [vp],
        "serl_classifier_num_leaves": [vp],
        "serl_classifier_leaf_info": [vp, i32, C.c_char_p, i32, C.POINTER(i64)],
        "serl_classifier_set": [vp, C.c_char_p, vp, i64],
        "serl_classifier_get": [vp, C.c_char_p, vp, i64],
        "serl_classifier_logits": [vp, vp, i32, vp, vp],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = i32
    lib._serl_classifier_declared = True


_HEAD_PATHS = {   # flat leaf -> path in BinaryClassifier's parameter tree (flax auto-names, reward_classifier.py:20-28)
    "head/dense0/kernel": ("Dense_0", "kernel"), "head/dense0/bias": ("Dense_0", "bias"),
    "head/ln/scale": ("LayerNorm_0", "scale"), "head/ln/bias": ("LayerNorm_0", "bias"),
    "head/dense1/kernel": ("Dense_1", "kernel"), "head/dense1/bias": ("Dense_1", "bias"),
}
_CAM_PATHS = {"sle": ("SpatialLearnedEmbeddings_0", "kernel"), "dense/kernel": ("Dense_0", "kernel"),
              "dense/bias": ("Dense_0", "bias"), "ln/scale": ("LayerNorm_0", "scale"), "ln/bias": ("LayerNorm_0", "bias")}


class Classifier:
    """The role of the reference's `TrainState` for inference: `.params` (flax-layout tree) and
    `.apply_fn({"params": params}, obs, train=False)`; parameters live in HBM."""

    def __init__(self, image_keys, H, W, max_batch=64, device=0):
        self.L = _lib.lib()
        _declare(self.L)
        self.image_keys = tuple(image_keys)
        self.H, self.W, self.max_batch, self.device = H, W, max_batch, device
        cfg = SerlClassifierCfg(device, len(self.image_keys), H, W, max_batch)
        h = C.c_void_p()
        _lib.check(self.L.serl_classifier_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self._counts = {}
        name = C.create_string_buffer(128)
        cnt = C.c_int64()
        for i in range(self.L.serl_classifier_num_leaves(h)):
            _lib.check(self.L.serl_classifier_leaf_info(h, i, name, 128, C.byref(cnt)))
            self._counts[name.value.decode()] = cnt.value

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self.L.serl_classifier_destroy(h)

    # ---- flat leaves
    def set(self, leaf, value):
        a = np.ascontiguousarray(np.asarray(value, np.float32).reshape(-1))
        self._params_cache = None
        _lib.check(self.L.serl_classifier_set(self._h, leaf.encode(), a.ctypes.data_as(C.c_void_p), a.size))

    def get(self, leaf):
        out = np.empty(self._counts[leaf], np.float32)
        _lib.check(self.L.serl_classifier_get(self._h, leaf.encode(), out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def _leaf(self, k, leaf):
        return f"enc/{self.image_keys.index(k)}/{leaf}"

    def load_flat(self, flat: Dict[str, np.ndarray]):
        """flat: trunk leaves, 'enc/<image key>/...', 'head/...'."""
        for name, v in flat.items():
            if name.startswith("enc/"):
                _, k, leaf = name.split("/", 2)
                name = self._leaf(k, leaf)
            self.set(name, v)
        return self

    # ---- flax layout (reward_classifier.py:58-60: classifier_def.init(key, sample)["params"])
    @property
    def params(self):
        """The flax-layout tree (built from HBM once, cached until a leaf is set)."""
        if self._params_cache is None:
            self._params_cache = self._export()
        return self._params_cache

    def _export(self):
        from ..utils.init import trunk_shapes
        tree = {"encoder_def": {}}
        tsh = trunk_shapes()
        for k in self.image_keys:
            sub = tree["encoder_def"].setdefault(f"encoder_{k}", {})
            for leaf, (mod, name) in _CAM_PATHS.items():
                v = self.get(self._leaf(k, leaf))
                shape = {"sle": (-1, 512, 8), "dense/kernel": (4096, 256)}.get(leaf, (-1,))
                if leaf == "sle":
                    hw = v.size // (512 * 8)
                    side = int(round(hw ** 0.5))
                    shape = (side, hw // side, 512, 8)
                sub.setdefault(mod, {})[name] = v.reshape(shape)
        owner = tree["encoder_def"][f"encoder_{trunk_owner(self.image_keys)}"]   # ONE shared frozen trunk (:37-51)
        for leaf, path in _trunk_paths().items():
            d = owner.setdefault("pretrained_encoder", {})
            for p in path[:-1]:
                d = d.setdefault(p, {})
            d[path[-1]] = self.get(leaf).reshape(tsh[leaf])
        E = 256 * len(self.image_keys)
        for leaf, (mod, name) in _HEAD_PATHS.items():
            shape = {"head/dense0/kernel": (E, 256), "head/dense1/kernel": (256, 1)}.get(leaf, (-1,))
            tree.setdefault(mod, {})[name] = self.get(leaf).reshape(shape)
        return tree

    def load_params(self, tree):
        """A BinaryClassifier parameter tree (e.g. the `params` entry of a checkpoint the reference's trainer wrote)."""
        enc = tree["encoder_def"]
        for k in self.image_keys:
            sub = enc[f"encoder_{k}"]
            for leaf, (mod, name) in _CAM_PATHS.items():
                self.set(self._leaf(k, leaf), sub[mod][name])
            if "pretrained_encoder" in sub:
                for leaf, v in trunk_from_flax(sub["pretrained_encoder"]).items():
                    self.set(leaf, v)
        for leaf, (mod, name) in _HEAD_PATHS.items():
            self.set(leaf, tree[mod][name])
        return self

    def replace(self, params=None, **kw):
        if kw:
            raise NotImplementedError(list(kw))
        if params is not None:
            self.load_params(params)
        return self

    # ---- forward
    def logits(self, observations) -> np.ndarray:
        """observations: {image_key: u8 (T=1, H, W, 3) or (B, T=1, H, W, 3)} (encoding.py:39-44 stacking) -> (1,) / (B, 1)."""
        first = np.asarray(observations[self.image_keys[0]])
        batched = first.ndim == 5
        frames = []
        for k in self.image_keys:
            x = np.asarray(observations[k])
            if x.dtype != np.uint8:
                raise TypeError(f"observation '{k}' must be uint8 (got {x.dtype})")
            x = x if batched else x[None]
            if x.shape[1] != 1:
                raise NotImplementedError("frame stacking T > 1 is not supported")
            frames.append(x[:, 0])
        fr = np.stack(frames)                                  # [n_cam][n][H][W][3]
        n = fr.shape[1]
        out = np.empty((n, 1), np.float32)
        dev = torch.device("cuda", self.device)
        for lo in range(0, n, self.max_batch):
            hi = min(n, lo + self.max_batch)
            d_fr = torch.from_numpy(np.ascontiguousarray(fr[:, lo:hi])).to(dev)
            d_out = torch.empty((hi - lo,), dtype=torch.float32, device=dev)
            st = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(self.L.serl_classifier_logits(self._h, d_fr.data_ptr(), hi - lo, d_out.data_ptr(), C.c_void_p(st)))
            out[lo:hi, 0] = d_out.cpu().numpy()
        return out if batched else out[0]

    def apply_fn(self, variables, observations, train=False, **kw):
        if train:
            raise NotImplementedError("classifier training is outside the MI355X hot path")
        p = None if variables is None else variables.get("params")
        if p is not None and p is not self._params_cache:    # foreign parameters: load them first
            self.load_params(p)
        return self.logits(observations)

    _params_cache = None


def create_classifier(key, sample: Dict, image_keys: List[str], pretrained_encoder_path: str = "./resnet10_params.pkl",
                      max_batch: int = 64, device: int = 0) -> Classifier:
    """reward_classifier.py:31-90: a freshly initialised classifier whose frozen trunk holds the pretrained ResNet-10."""
    from ..utils import init as pinit
    first = np.asarray(sample[image_keys[0]])
    H, W = int(first.shape[-3]), int(first.shape[-2])
    seed = int(np.asarray(key).reshape(-1)[-1]) if not isinstance(key, int) else key
    c = Classifier(image_keys, H, W, max_batch=max_batch, device=device)
    for name, v in pinit.init_classifier(len(image_keys), H, W, seed).items():
        c.set(name, v)
    with open(pretrained_encoder_path, "rb") as f:
        encoder_params = pickle.load(f)
    for leaf, v in trunk_from_flax(encoder_params).items():   # top-level keys the pickle lacks keep their value (:76-86)
        c.set(leaf, v)
    return c


def load_classifier_func(key, sample: Dict, image_keys: List[str], checkpoint_path: str, step: Optional[int] = None,
                         pretrained_encoder_path: str = "./resnet10_params.pkl") -> Callable[[Dict], np.ndarray]:
    """reward_classifier.py:93-113: restore `checkpoint_path` (a directory of checkpoint_<step> files or one file, flax
    msgpack layout of the classifier TrainState) and return obs -> logits."""
    from ..utils.checkpoint import read_checkpoint_tree
    classifier = create_classifier(key, sample, image_keys, pretrained_encoder_path) if os.path.exists(pretrained_encoder_path) \
        else _blank_classifier(sample, image_keys)
    tree = read_checkpoint_tree(checkpoint_path, step)
    classifier.load_params(tree["params"])
    return lambda obs: classifier.logits(obs)


def _blank_classifier(sample, image_keys):
    first = np.asarray(sample[image_keys[0]])
    return Classifier(image_keys, int(first.shape[-3]), int(first.shape[-2]))
