"""Synthetic parameter initialisation (flax initialisers restated) for the DrQ agent.

The reference initialises with flax (`model_def.init`, agents/continuous/drq.py:69-75) and then
overwrites the ResNet-10 trunk with weights downloaded at run time (utils/train_utils.py:69-130).
Neither flax nor the network is available here, so the trunk gets seeded synthetic weights of the
same tree/shape; real weights are loaded with DrQAgent.load_trunk_params().
Leaf names/shapes = the flat arena of libserl_mi355.so, camera index instead of key.
"""
from __future__ import annotations

import math

import numpy as np

STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))


def trunk_shapes():
    sh = {"trunk/conv_init": (7, 7, 3, 64), "trunk/norm_init/scale": (64,), "trunk/norm_init/bias": (64,)}
    cin = 64
    for i, (f, s) in enumerate(STAGES):
        p = f"trunk/block{i}/"
        sh[p + "conv0"] = (3, 3, cin, f)
        sh[p + "gn0/scale"] = (f,)
        sh[p + "gn0/bias"] = (f,)
        sh[p + "conv1"] = (3, 3, f, f)
        sh[p + "gn1/scale"] = (f,)
        sh[p + "gn1/bias"] = (f,)
        if s != 1 or cin != f:
            sh[p + "proj"] = (1, 1, cin, f)
            sh[p + "gnp/scale"] = (f,)
            sh[p + "gnp/bias"] = (f,)
        cin = f
    return sh


def feat_hw(H, W):
    h, w = H, W
    for _ in range(5):
        h, w = (h + 1) // 2, (w + 1) // 2
    return h, w


SMALL_FEATURES = (3, 32, 64, 128, 256)   # SmallEncoder(features=(32, 64, 128, 256)) on RGB (drq.py:140-151)


def theta_shapes(n_cam, H, W, S, A, ensemble=10, hidden=256, bottleneck=256, sle_features=8, proprio_dim=64,
                 encoder_type="resnet-pretrained"):
    if n_cam == 0:   # state-only SAC (SACAgent.create_states, sac.py:486-542): no encoder, per-member Q heads
        N, Hd = ensemble, hidden
        return {
            "critic/w1": (N, S + A, Hd), "critic/b1": (N, Hd), "critic/ln1/scale": (N, Hd), "critic/ln1/bias": (N, Hd),
            "critic/w2": (N, Hd, Hd), "critic/b2": (N, Hd), "critic/ln2/scale": (N, Hd), "critic/ln2/bias": (N, Hd),
            "critic/head/kernel": (N, Hd, 1), "critic/head/bias": (N, 1),   # vmapped Dense(1): one bias per member
            "actor/w1": (S, Hd), "actor/b1": (Hd,), "actor/ln1/scale": (Hd,), "actor/ln1/bias": (Hd,),
            "actor/w2": (Hd, Hd), "actor/b2": (Hd,), "actor/ln2/scale": (Hd,), "actor/ln2/bias": (Hd,),
            "actor/mean/kernel": (Hd, A), "actor/mean/bias": (A,),
            "actor/logstd/kernel": (Hd, A), "actor/logstd/bias": (A,),
            "temp/lagrange": (),
        }
    fh, fw = feat_hw(H, W)
    E = bottleneck * n_cam + proprio_dim
    sh = {}
    for k in range(n_cam):
        if encoder_type == "small":    # 4 x Conv(3x3, stride 2, VALID) with bias, then mean-pool -> Dense(256)
            for l in range(4):
                sh[f"enc/{k}/conv{l}/kernel"] = (3, 3, SMALL_FEATURES[l], SMALL_FEATURES[l + 1])
                sh[f"enc/{k}/conv{l}/bias"] = (SMALL_FEATURES[l + 1],)
            sh[f"enc/{k}/dense/kernel"] = (SMALL_FEATURES[4], bottleneck)
        else:
            sh[f"enc/{k}/sle"] = (fh, fw, 512, sle_features)
            sh[f"enc/{k}/dense/kernel"] = (512 * sle_features, bottleneck)
        sh[f"enc/{k}/dense/bias"] = (bottleneck,)
        sh[f"enc/{k}/ln/scale"] = (bottleneck,)
        sh[f"enc/{k}/ln/bias"] = (bottleneck,)
    N, Hd = ensemble, hidden
    sh.update({
        "critic/w1": (N, E + A, Hd), "critic/b1": (N, Hd), "critic/ln1/scale": (N, Hd), "critic/ln1/bias": (N, Hd),
        "critic/w2": (N, Hd, Hd), "critic/b2": (N, Hd), "critic/ln2/scale": (N, Hd), "critic/ln2/bias": (N, Hd),
        "critic/head/kernel": (Hd, 1), "critic/head/bias": (1,),
        "enc/proprio/dense/kernel": (S, proprio_dim), "enc/proprio/dense/bias": (proprio_dim,),
        "enc/proprio/ln/scale": (proprio_dim,), "enc/proprio/ln/bias": (proprio_dim,),
        "actor/w1": (E, Hd), "actor/b1": (Hd,), "actor/ln1/scale": (Hd,), "actor/ln1/bias": (Hd,),
        "actor/w2": (Hd, Hd), "actor/b2": (Hd,), "actor/ln2/scale": (Hd,), "actor/ln2/bias": (Hd,),
        "actor/mean/kernel": (Hd, A), "actor/mean/bias": (A,),
        "actor/logstd/kernel": (Hd, A), "actor/logstd/bias": (A,),
        "temp/lagrange": (),
    })
    return sh


def init_trunk(seed=42):
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, 1])))
    out = {}
    for name, shp in trunk_shapes().items():
        if len(shp) == 4:  # nn.initializers.kaiming_normal (resnet_v1.py:228-233)
            out[name] = (rng.standard_normal(shp) * math.sqrt(2.0 / (shp[0] * shp[1] * shp[2]))).astype(np.float32)
        elif name.endswith("scale"):
            out[name] = np.ones(shp, np.float32)
        else:
            out[name] = np.zeros(shp, np.float32)
    return out


def init_theta(n_cam, H, W, S, A, seed=42, temperature_init=1e-2, **kw):
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, 2])))
    out = {}
    for name, shp in theta_shapes(n_cam, H, W, S, A, **kw).items():
        if name.endswith("/sle"):  # lecun_normal over (h,w,c) (resnet_v1.py:86)
            out[name] = (rng.standard_normal(shp) / math.sqrt(shp[0] * shp[1] * shp[2])).astype(np.float32)
        elif name.startswith("enc/") and "/conv" in name and name.endswith("kernel"):   # nn.Conv default lecun_normal
            out[name] = (rng.standard_normal(shp) / math.sqrt(shp[0] * shp[1] * shp[2])).astype(np.float32)
        elif name.startswith("enc/") and name.endswith("dense/kernel") and "proprio" not in name:
            out[name] = (rng.standard_normal(shp) / math.sqrt(shp[0])).astype(np.float32)  # nn.Dense default
        elif name in ("critic/w1", "critic/w2") or (name == "critic/head/kernel" and len(shp) == 3):
            # default_init = xavier_uniform, vmapped per member
            a = math.sqrt(6.0 / (shp[1] + shp[2]))
            out[name] = rng.uniform(-a, a, shp).astype(np.float32)
        elif name.endswith("kernel") or name in ("actor/w1", "actor/w2"):
            a = math.sqrt(6.0 / (shp[0] + shp[1]))
            out[name] = rng.uniform(-a, a, shp).astype(np.float32)
        elif name.endswith("scale"):
            out[name] = np.ones(shp, np.float32)
        elif name == "temp/lagrange":  # lagrange.py:28-29
            out[name] = np.float32(math.log(math.exp(temperature_init) - 1.0))
        else:
            out[name] = np.zeros(shp, np.float32)
    return out


def init_classifier(n_cam, H, W, seed=42):
    """Initial parameters of the reward classifier's trainable part (networks/reward_classifier.py:16-28 over
    EncodingWrapper(use_proprio=False)): camera heads as in init_theta (SpatialLearnedEmbeddings lecun-normal, Dense
    xavier-uniform, LayerNorm ones/zeros -- resnet_v1.py:94-104,363-374), Dense layers of the head with flax's default
    lecun-normal kernels and zero biases.  Host-side numpy streams, not jax's threefry (as for the agents)."""
    theta = init_theta(n_cam, H, W, 4, 2, seed=seed)
    out = {k: v for k, v in theta.items() if k.startswith("enc/") and not k.startswith("enc/proprio")}
    rng = np.random.default_rng(seed + 77)
    E = 256 * n_cam

    def lecun(fan_in, shape):   # jax.nn.initializers.lecun_normal: truncated normal, std = sqrt(1 / fan_in) / 0.8796
        std = math.sqrt(1.0 / fan_in) / 0.87962566103423978
        v = rng.standard_normal(shape)
        bad = np.abs(v) > 2.0
        while bad.any():
            v[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(v) > 2.0
        return (v * std).astype(np.float32)
    out["head/dense0/kernel"] = lecun(E, (E, 256))
    out["head/dense0/bias"] = np.zeros(256, np.float32)
    out["head/ln/scale"] = np.ones(256, np.float32)
    out["head/ln/bias"] = np.zeros(256, np.float32)
    out["head/dense1/kernel"] = lecun(256, (256, 1))
    out["head/dense1/bias"] = np.zeros(1, np.float32)
    return out
