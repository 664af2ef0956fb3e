"""TEST INFRASTRUCTURE ONLY -- CPU restatement (PyTorch-CPU, fp64 or fp32) of the reference's
SAC/DrQ update step.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this; the product path (serl_amd/) never does.

PARITY UNPINNED: the reference's arithmetic lives in un-vendored third-party packages that are
neither in /root/reference nor installable here (jax/jaxlib README pin 0.4.35; flax>=0.8.0,
optax>=0.1.5, distrax>=0.1.2, chex>=0.1.85 -- serl_launcher/requirements.txt:3-8), and the
reference has no tests or golden vectors for this path.  This file restates the
published algorithms of those libraries at the reference's own call sites; gradients come from
torch autograd in fp64, which independently checks the hand-derived backward kernels.

Follows (reference file:line, relative to serl_launcher/serl_launcher/):
  agents/continuous/drq.py:244-328        augmentation + update_critics / update_high_utd
  agents/continuous/sac.py:118-299,544-596  losses, update(), update_high_utd()
  common/common.py:124-221                target EMA, apply_gradients (3 Adam txs, summed), rng fan-out
  common/optimizers.py:6-56               adam + warmup->constant schedule
  common/encoding.py:26-72                EncodingWrapper (per-camera encode, stop_gradient, proprio)
  vision/resnet_v1.py:81-156,189-376      SpatialLearnedEmbeddings, ResNetBlock, ResNet-10 trunk
  networks/actor_critic_nets.py:49-73,156-272  Critic, ensemblize, Policy, TanhMultivariateNormalDiag
  networks/mlp.py:10-32, networks/lagrange.py:9-74
  utils/launcher.py:79-116, agents/continuous/drq.py:35-53,88-89   hyper-parameters
Randomness is injected explicitly (crop offsets, eps, dropout masks, REDQ indices): jax's
threefry / flax rng folding is not reproduced.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))  # (filters, stride) of the four ResNetBlocks
SMALL_FEATURES = (3, 32, 64, 128, 256)             # SmallEncoder(features=(32, 64, 128, 256)) on RGB (drq.py:140-151)


@dataclass
class Config:
    image_keys: tuple = ("front", "wrist")
    H: int = 128
    W: int = 128
    S: int = 24
    A: int = 6
    ensemble: int = 10
    subsample: int = 2
    hidden: int = 256
    bottleneck: int = 256
    sle_features: int = 8
    proprio_dim: int = 64
    discount: float = 0.96          # launcher.py:85
    tau: float = 0.005              # drq.py:46
    lr: float = 3e-4                # drq.py:35-43
    warmup: int = 0
    temperature_init: float = 1e-2  # launcher.py:110
    dropout: float = 0.1            # resnet_v1.py:351
    std_min: float = 1e-5           # launcher.py:97-98
    std_max: float = 5.0
    target_entropy: float = None    # drq.py:88-89: -A/2
    temp_warmup: int = None         # temperature optimizer's own warm-up (None: same as `warmup`); SACAgent.create
                                    # defaults: actor/critic warmup 2000, temperature none (sac.py:333-343)
    encoder_type: str = "resnet-pretrained"   # or "small": the trainable SmallEncoder (vision/small_encoders.py:9-55,
                                    # drq.py:137-153; the reference's call site passes it an `encode=` kwarg it does not
                                    # accept -- SURVEY.md fact 4 -- so it is restated as if the kwarg were ignored)
    opt: dict = None                # optional {"actor"|"critic"|"temperature": make_optimizer kwargs} (optimizers.py:6-13:
                                    # learning_rate, warmup_steps, cosine_decay_steps, weight_decay, clip_grad_norm)

    def __post_init__(self):
        if self.target_entropy is None:
            self.target_entropy = -self.A / 2

    @property
    def n_cam(self):
        return len(self.image_keys)

    @property
    def state_only(self):
        """SACAgent.create_states (sac.py:486-542): no encoder, observations are flat state vectors, and the
        ensemblized Critic carries one Dense(1) head PER member (actor_critic_nets.py:49-73 under ensemblize)."""
        return len(self.image_keys) == 0

    @property
    def feat_hw(self):  # trunk output spatial size: /2 (conv), /2 (pool), /2 /2 /2 (blocks 1-3)
        h = w = None
        def down(n):
            return (n + 1) // 2
        h, w = self.H, self.W
        for _ in range(5):
            h, w = down(h), down(w)
        return h, w

    @property
    def small(self):
        return self.encoder_type == "small"

    @property
    def sle_dim(self):
        return 512 * self.sle_features

    @property
    def enc_dim(self):
        if self.state_only:
            return self.S
        return self.bottleneck * self.n_cam + self.proprio_dim


# ---------------------------------------------------------------------------------------------
# parameters (synthetic, seeded; flax initialisers restated)
# ---------------------------------------------------------------------------------------------
def trunk_param_shapes():
    shapes = {"trunk/conv_init": (7, 7, 3, 64), "trunk/norm_init/scale": (64,), "trunk/norm_init/bias": (64,)}
    cin = 64
    for i, (f, s) in enumerate(STAGES):
        shapes[f"trunk/block{i}/conv0"] = (3, 3, cin, f)
        shapes[f"trunk/block{i}/gn0/scale"] = (f,)
        shapes[f"trunk/block{i}/gn0/bias"] = (f,)
        shapes[f"trunk/block{i}/conv1"] = (3, 3, f, f)
        shapes[f"trunk/block{i}/gn1/scale"] = (f,)
        shapes[f"trunk/block{i}/gn1/bias"] = (f,)
        if s != 1 or cin != f:
            shapes[f"trunk/block{i}/proj"] = (1, 1, cin, f)
            shapes[f"trunk/block{i}/gnp/scale"] = (f,)
            shapes[f"trunk/block{i}/gnp/bias"] = (f,)
        cin = f
    return shapes


def trainable_param_shapes(cfg: Config):
    """Order == the flat layout of the product's parameter arena."""
    fh, fw = cfg.feat_hw
    E, A, Hd, N = cfg.enc_dim, cfg.A, cfg.hidden, cfg.ensemble
    sh = {}
    for k in cfg.image_keys:
        if cfg.small:
            for l in range(4):
                sh[f"enc/{k}/conv{l}/kernel"] = (3, 3, SMALL_FEATURES[l], SMALL_FEATURES[l + 1])
                sh[f"enc/{k}/conv{l}/bias"] = (SMALL_FEATURES[l + 1],)
            sh[f"enc/{k}/dense/kernel"] = (SMALL_FEATURES[4], cfg.bottleneck)
        else:
            sh[f"enc/{k}/sle"] = (fh, fw, 512, cfg.sle_features)
            sh[f"enc/{k}/dense/kernel"] = (cfg.sle_dim, cfg.bottleneck)
        sh[f"enc/{k}/dense/bias"] = (cfg.bottleneck,)
        sh[f"enc/{k}/ln/scale"] = (cfg.bottleneck,)
        sh[f"enc/{k}/ln/bias"] = (cfg.bottleneck,)
    sh.update({
        "critic/w1": (N, E + A, Hd), "critic/b1": (N, Hd), "critic/ln1/scale": (N, Hd), "critic/ln1/bias": (N, Hd),
        "critic/w2": (N, Hd, Hd), "critic/b2": (N, Hd), "critic/ln2/scale": (N, Hd), "critic/ln2/bias": (N, Hd),
        "critic/head/kernel": (Hd, 1), "critic/head/bias": (1,),
    })
    if cfg.state_only:
        sh["critic/head/kernel"], sh["critic/head/bias"] = (N, Hd, 1), (N,)
    else:
        sh.update({
            "enc/proprio/dense/kernel": (cfg.S, cfg.proprio_dim), "enc/proprio/dense/bias": (cfg.proprio_dim,),
            "enc/proprio/ln/scale": (cfg.proprio_dim,), "enc/proprio/ln/bias": (cfg.proprio_dim,),
        })
    sh.update({
        "actor/w1": (E, Hd), "actor/b1": (Hd,), "actor/ln1/scale": (Hd,), "actor/ln1/bias": (Hd,),
        "actor/w2": (Hd, Hd), "actor/b2": (Hd,), "actor/ln2/scale": (Hd,), "actor/ln2/bias": (Hd,),
        "actor/mean/kernel": (Hd, A), "actor/mean/bias": (A,),
        "actor/logstd/kernel": (Hd, A), "actor/logstd/bias": (A,),
        "temp/lagrange": (),
    })
    return sh


def init_params(cfg: Config, seed: int = 42, perturb: bool = True):
    """Returns (trunk: dict[str, np.float32], theta: dict[str, np.float32]).
    kaiming-normal convs, lecun-normal Dense/SLE, xavier-uniform `default_init` layers, unit
    scales / zero biases; `perturb` jitters scales/biases so tests notice a dropped bias or scale."""
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

    def normal(shape, std):
        return (rng.standard_normal(shape) * std).astype(np.float32)

    def xavier(shape, fan_in, fan_out):
        a = math.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-a, a, size=shape).astype(np.float32)

    def scale(shape):
        return (1.0 + (0.1 * rng.standard_normal(shape) if perturb else 0.0)).astype(np.float32) * np.ones(shape, np.float32)

    def bias(shape):
        return ((0.1 * rng.standard_normal(shape)) if perturb else np.zeros(shape)).astype(np.float32)

    trunk = {}
    for name, shp in ({} if (cfg.state_only or cfg.small) else trunk_param_shapes()).items():
        if len(shp) == 4:
            trunk[name] = normal(shp, math.sqrt(2.0 / (shp[0] * shp[1] * shp[2])))
        elif name.endswith("scale"):
            trunk[name] = scale(shp)
        else:
            trunk[name] = bias(shp)
    theta = {}
    for name, shp in trainable_param_shapes(cfg).items():
        if name.endswith("/sle") or ("/conv" in name and name.endswith("kernel")):   # lecun_normal over (h, w, cin)
            theta[name] = normal(shp, math.sqrt(1.0 / (shp[0] * shp[1] * shp[2])))
        elif name.startswith("enc/") and name.endswith("dense/kernel") and "proprio" not in name:
            theta[name] = normal(shp, math.sqrt(1.0 / shp[0]))        # nn.Dense default lecun_normal
        elif name == "critic/head/kernel" and len(shp) == 3:
            theta[name] = xavier(shp, shp[1], shp[2])                 # per-member heads (state-only SAC)
        elif name.endswith("kernel") or name in ("actor/w1", "actor/w2"):
            theta[name] = xavier(shp, shp[0], shp[1])                 # default_init / xavier_uniform
        elif name in ("critic/w1", "critic/w2"):
            theta[name] = xavier(shp, shp[1], shp[2])
        elif name.endswith("scale"):
            theta[name] = scale(shp)
        elif name == "temp/lagrange":                                 # lagrange.py:28-29
            theta[name] = np.float32(math.log(math.exp(cfg.temperature_init) - 1.0))
        else:
            theta[name] = bias(shp)
    return trunk, theta


def to_torch(d, dtype):
    return {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in d.items()}


# ---------------------------------------------------------------------------------------------
# network forward (flax semantics restated)
# ---------------------------------------------------------------------------------------------
def group_norm(x, scale, bias, groups=4, eps=1e-5):
    """flax nn.GroupNorm on NHWC, fast variance E[x^2]-E[x]^2 clamped at 0 (resnet_v1.py:119-126,237)."""
    N, H, W, Cc = x.shape
    xg = x.reshape(N, H * W, groups, Cc // groups)
    mean = xg.mean(dim=(1, 3), keepdim=True)
    mean2 = (xg * xg).mean(dim=(1, 3), keepdim=True)
    var = torch.clamp(mean2 - mean * mean, min=0.0)
    y = (xg - mean) * torch.rsqrt(var + eps)
    return y.reshape(N, H, W, Cc) * scale + bias


def layer_norm(x, scale, bias, eps=1e-6):
    """flax nn.LayerNorm default (eps 1e-6, fast variance) over the last axis."""
    mean = x.mean(dim=-1, keepdim=True)
    mean2 = (x * x).mean(dim=-1, keepdim=True)
    var = torch.clamp(mean2 - mean * mean, min=0.0)
    return (x - mean) * torch.rsqrt(var + eps) * scale + bias


def same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def conv_nhwc(x, w_hwio, stride, pad):
    """x [N,H,W,Cin], w [kh,kw,Cin,Cout]; pad = ((top,bottom),(left,right)) zero padding."""
    xn = x.permute(0, 3, 1, 2)
    xn = F.pad(xn, (pad[1][0], pad[1][1], pad[0][0], pad[0][1]))
    y = F.conv2d(xn, w_hwio.permute(3, 2, 0, 1), stride=stride)
    return y.permute(0, 2, 3, 1)


def conv_same(x, w, stride):
    kh, kw = w.shape[0], w.shape[1]
    return conv_nhwc(x, w, stride, (same_pad(x.shape[1], kh, stride), same_pad(x.shape[2], kw, stride)))


def trunk_forward(tp, img_u8, dtype, return_intermediates=False):
    """Frozen ResNet-10 trunk (resnet_v1.py:189-286).  img_u8 [N,H,W,3] uint8 -> [N,h,w,512]."""
    inter = {}
    mean = torch.tensor(IMAGENET_MEAN, dtype=dtype)
    std = torch.tensor(IMAGENET_STD, dtype=dtype)
    x = (img_u8.to(dtype) / 255.0 - mean) / std                              # :221-223
    x = conv_nhwc(x, tp["trunk/conv_init"], 2, ((3, 3), (3, 3)))             # :249-255
    inter["conv_init"] = x
    x = torch.relu(group_norm(x, tp["trunk/norm_init/scale"], tp["trunk/norm_init/bias"]))
    ph, pw = same_pad(x.shape[1], 3, 2), same_pad(x.shape[2], 3, 2)          # max_pool SAME :259
    xn = F.pad(x.permute(0, 3, 1, 2), (pw[0], pw[1], ph[0], ph[1]), value=float("-inf"))
    x = F.max_pool2d(xn, 3, 2).permute(0, 2, 3, 1)
    inter["pool"] = x
    for i, (f, s) in enumerate(STAGES):                                      # ResNetBlock :143-156
        p = f"trunk/block{i}/"
        y = conv_same(x, tp[p + "conv0"], s)
        inter[f"b{i}_raw0"] = y
        y = torch.relu(group_norm(y, tp[p + "gn0/scale"], tp[p + "gn0/bias"]))
        y = conv_same(y, tp[p + "conv1"], 1)
        y = group_norm(y, tp[p + "gn1/scale"], tp[p + "gn1/bias"])
        r = x
        if (p + "proj") in tp:
            r = conv_same(x, tp[p + "proj"], s)
            r = group_norm(r, tp[p + "gnp/scale"], tp[p + "gnp/bias"])
        x = torch.relu(r + y)
        inter[f"b{i}_out"] = x
    return (x, inter) if return_intermediates else x


def sle(feats, kernel):
    """SpatialLearnedEmbeddings (resnet_v1.py:94-111): [N,h,w,C] x [h,w,C,F] -> [N, C*F] (c-major)."""
    out = torch.einsum("nhwc,hwcf->ncf", feats, kernel)
    return out.reshape(feats.shape[0], -1)


def small_encoder_trunk(th, key, img_u8):
    """SmallEncoder up to the pooling (small_encoders.py:23-41): x/255, 4 x (Conv 3x3 stride 2 VALID + bias, ReLU),
    mean over (H, W).  img_u8 [N,H,W,3] uint8 (or float already) -> [N,256]."""
    dt = th[f"enc/{key}/conv0/kernel"].dtype
    x = img_u8.to(dt) / 255.0
    for l in range(4):
        x = conv_nhwc(x, th[f"enc/{key}/conv{l}/kernel"], 2, ((0, 0), (0, 0))) + th[f"enc/{key}/conv{l}/bias"]
        x = torch.relu(x)
    return x.mean(dim=(1, 2))


def encode(th, cfg, feats, state, drop_masks=None, stop_gradient=False):
    """EncodingWrapper (encoding.py:26-72) on precomputed trunk features.
    feats: {cam: [N,h,w,512]}; state [N,S]; drop_masks: {cam: [N,4096] keep-mask} or None."""
    if cfg.state_only:
        return state        # encoder=None (actor_critic_nets.py:59-60,185-186)
    codes = []
    for k in cfg.image_keys:
        if cfg.small:                                                        # small_encoders.py:19-41 (pool_method "avg")
            f = small_encoder_trunk(th, k, feats[k])
        else:
            f = sle(feats[k], th[f"enc/{k}/sle"])
            if drop_masks is not None:                                       # nn.Dropout(0.1) :351
                f = torch.where(drop_masks[k].bool(), f / (1.0 - cfg.dropout), torch.zeros_like(f))
        z = f @ th[f"enc/{k}/dense/kernel"] + th[f"enc/{k}/dense/bias"]
        z = torch.tanh(layer_norm(z, th[f"enc/{k}/ln/scale"], th[f"enc/{k}/ln/bias"]))  # :371-374
        if stop_gradient:
            z = z.detach()                                                   # encoding.py:48-49
        codes.append(z)
    p = state @ th["enc/proprio/dense/kernel"] + th["enc/proprio/dense/bias"]
    p = torch.tanh(layer_norm(p, th["enc/proprio/ln/scale"], th["enc/proprio/ln/bias"]))
    return torch.cat(codes + [p], dim=-1)


def policy_head(th, cfg, enc):
    """Policy (actor_critic_nets.py:179-227) + MLP (mlp.py:18-32): returns mean, std."""
    h = torch.tanh(layer_norm(enc @ th["actor/w1"] + th["actor/b1"], th["actor/ln1/scale"], th["actor/ln1/bias"]))
    h = torch.tanh(layer_norm(h @ th["actor/w2"] + th["actor/b2"], th["actor/ln2/scale"], th["actor/ln2/bias"]))
    mean = h @ th["actor/mean/kernel"] + th["actor/mean/bias"]
    log_std = h @ th["actor/logstd/kernel"] + th["actor/logstd/bias"]
    std = torch.clamp(torch.exp(log_std), cfg.std_min, cfg.std_max)
    return mean, std


def sample_and_log_prob(mean, std, eps):
    """TanhMultivariateNormalDiag.sample_and_log_prob (actor_critic_nets.py:230-272; distrax
    MultivariateNormalDiag + Block(Tanh,1)): log det of tanh = 2(log2 - u - softplus(-2u))."""
    u = mean + std * eps
    a = torch.tanh(u)
    base = (-0.5 * eps * eps - torch.log(std) - 0.5 * math.log(2.0 * math.pi)).sum(-1)
    ldj = (2.0 * (math.log(2.0) - u - F.softplus(-2.0 * u))).sum(-1)
    return a, base - ldj


def critic_forward(th, cfg, enc, act):
    """Critic (actor_critic_nets.py:56-73) with the vmapped ensemble MLP and the shared head
    (drq.py:201-207): returns Q [ensemble, B]."""
    x = torch.cat([enc, act], dim=-1)
    h = torch.einsum("bi,eio->ebo", x, th["critic/w1"]) + th["critic/b1"][:, None, :]
    h = torch.tanh(layer_norm(h, th["critic/ln1/scale"][:, None, :], th["critic/ln1/bias"][:, None, :]))
    h = torch.einsum("ebi,eio->ebo", h, th["critic/w2"]) + th["critic/b2"][:, None, :]
    h = torch.tanh(layer_norm(h, th["critic/ln2/scale"][:, None, :], th["critic/ln2/bias"][:, None, :]))
    if cfg.state_only:  # ensemblize vmaps the whole Critic: every member has its own Dense(1)
        return torch.einsum("ebi,eio->ebo", h, th["critic/head/kernel"]).squeeze(-1) + th["critic/head/bias"][:, None]
    return (h @ th["critic/head/kernel"]).squeeze(-1) + th["critic/head/bias"]


# ---------------------------------------------------------------------------------------------
# train state, optimizer, update
# ---------------------------------------------------------------------------------------------
TX_NAMES = ("actor", "critic", "temperature")  # dict order = sorted keys (common.py:161-164)


class TrainState:
    """JaxRLTrainState (common.py:81-114) restated: step, params, target_params, 3 Adam states."""

    def __init__(self, cfg, trunk, theta, dtype=torch.float64):
        self.cfg, self.dtype = cfg, dtype
        self.trunk = to_torch(trunk, dtype)
        self.params = to_torch(theta, dtype)
        self.target = {k: v.clone() for k, v in self.params.items()}
        self.target_trunk = {k: v.clone() for k, v in self.trunk.items()}
        self.step = 0
        self.opt = {tx: {"count": 0,
                         "mu": {k: torch.zeros_like(v) for k, v in self.params.items()},
                         "nu": {k: torch.zeros_like(v) for k, v in self.params.items()}} for tx in TX_NAMES}

    def tx_opt(self, tx):
        c = self.cfg
        warm = c.warmup if (tx != "temperature" or c.temp_warmup is None) else c.temp_warmup
        kw = {"learning_rate": c.lr, "warmup_steps": warm, "cosine_decay_steps": None, "weight_decay": None,
              "clip_grad_norm": None}
        kw.update((c.opt or {}).get(tx, {}))
        return kw

    def lr_at(self, count, tx="critic"):
        """optimizers.py:14-30: join_schedules([linear(0, lr, warmup), constant(lr)], [warmup]) or
        optax.warmup_cosine_decay_schedule(0, lr, warmup, cosine_decay_steps, 0)."""
        kw = self.tx_opt(tx)
        lr, warm = kw["learning_rate"], int(kw["warmup_steps"] or 0)
        if count < warm:
            return lr * count / warm
        if kw["cosine_decay_steps"] is not None:
            T = max(int(kw["cosine_decay_steps"]) - warm, 1)
            return lr * 0.5 * (1.0 + math.cos(math.pi * min(count - warm, T) / T))
        return lr


def apply_gradients(st: TrainState, grads):
    """common.py:136-168 + optax.adam (b1 .9, b2 .999, eps 1e-8, eps_root 0) restated, with make_optimizer's optional
    stages (optimizers.py:32-46): clip_by_global_norm first, adamw's decoupled weight decay on the WHOLE tree.
    grads: {tx: {name: tensor or None}}; a missing tx / name means an exact-zero gradient, which is
    still a real Adam step (moment decay + momentum update)."""
    b1, b2, eps = 0.9, 0.999, 1e-8
    updates, trunk_decay = {}, 0.0
    for tx in TX_NAMES:
        o = st.opt[tx]
        kw = st.tx_opt(tx)
        # optax.inject_hyperparams keeps the scheduled learning rate as a float32 array (the value the reference logs as
        # `<tx>_lr`): 3e-4 enters the update as float32(3e-4) = 0.00030000001425, also in an fp64 run
        lr = float(np.float32(st.lr_at(o["count"], tx)))
        o["count"] += 1
        t = o["count"]
        bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
        g_tx = {k: (grads.get(tx, {}).get(k) if grads.get(tx, {}).get(k) is not None else torch.zeros_like(st.params[k]))
                for k in st.params}
        if kw["clip_grad_norm"] is not None:    # optax.clip_by_global_norm over this optimizer's whole gradient tree
            g_norm = torch.sqrt(sum((g * g).sum() for g in g_tx.values()))
            if not bool(g_norm < kw["clip_grad_norm"]):
                g_tx = {k: g / g_norm * kw["clip_grad_norm"] for k, g in g_tx.items()}
        wd = kw["weight_decay"]
        upd = {}
        for k in st.params:
            g = g_tx[k]
            o["mu"][k] = b1 * o["mu"][k] + (1.0 - b1) * g
            o["nu"][k] = b2 * o["nu"][k] + (1.0 - b2) * g * g
            d = (o["mu"][k] / bc1) / (torch.sqrt(o["nu"][k] / bc2) + eps)
            if wd is not None:                    # optax.adamw: add_decayed_weights before the -lr scaling
                d = d + wd * st.params[k]
            upd[k] = -lr * d
        if wd is not None:                        # ... and the frozen trunk leaves are part of `params` too
            trunk_decay += lr * wd
        updates[tx] = upd
    for k in st.params:
        st.params[k] = st.params[k] + ((updates["actor"][k] + updates["critic"][k]) + updates["temperature"][k])
    if trunk_decay:
        for k in st.trunk:
            st.trunk[k] = st.trunk[k] - trunk_decay * st.trunk[k]
    st.step += 1


def target_update(st: TrainState):
    """common.py:124-134: tp <- p*tau + tp*(1-tau) on every leaf (trunk copies included)."""
    tau = st.cfg.tau
    for k in st.params:
        st.target[k] = st.params[k] * tau + st.target[k] * (1.0 - tau)
    for k in st.trunk:
        st.target_trunk[k] = st.trunk[k] * tau + st.target_trunk[k] * (1.0 - tau)


def _grad_dict(loss, params):
    names = [k for k, v in params.items() if v.requires_grad]
    gs = torch.autograd.grad(loss, [params[k] for k in names], allow_unused=True)
    return {k: g for k, g in zip(names, gs) if g is not None}


def critic_update(st: TrainState, feats_obs, feats_next, state, next_state, action, reward, mask, noise, apply=True):
    """SACAgent.update(networks_to_update={"critic"}) (sac.py:243-299) on precomputed (frozen)
    trunk features of the augmented batch.  noise: eps_next [B,A], mask_next {cam:[B,4096]},
    redq_idx (2,).  Returns info dict (sac.py:185-189) and the critic gradient dict."""
    cfg = st.cfg
    with torch.no_grad():
        enc_pi = encode(st.params, cfg, feats_next, next_state, drop_masks=noise["mask_next"], stop_gradient=True)
        mean, std = policy_head(st.params, cfg, enc_pi)
        next_a, next_logp = sample_and_log_prob(mean, std, noise["eps_next"])       # sac.py:118-132
        enc_t = encode(st.target, cfg, feats_next, next_state)
        tq = critic_forward(st.target, cfg, enc_t, next_a)                           # sac.py:143-147
        i0, i1 = int(noise["redq_idx"][0]), int(noise["redq_idx"][1])
        min_q = torch.minimum(tq[i0], tq[i1])                                        # sac.py:150-161
        target_q = reward + cfg.discount * mask * min_q                              # sac.py:164-172
    p = {k: v.detach().clone().requires_grad_(True) for k, v in st.params.items()}
    enc = encode(p, cfg, feats_obs, state)
    q = critic_forward(p, cfg, enc, action)                                          # sac.py:174-176
    loss = ((q - target_q[None]) ** 2).mean()                                        # sac.py:181-183
    grads = _grad_dict(loss, p)
    info = {"critic_loss": loss.item(), "predicted_qs": q.mean().item(), "target_qs": target_q.mean().item()}
    aux = {"next_actions": next_a, "next_logp": next_logp, "target_q": target_q, "q": q.detach(),
           "grads": grads, "enc_obs": enc.detach()}
    if apply:
        apply_gradients(st, {"critic": grads})
        target_update(st)                                                            # sac.py:284-285
    return info, aux


def actor_temp_update(st: TrainState, feats_obs, feats_next, state, next_state, noise, apply=True, do_actor=True,
                      do_temp=True):
    """SACAgent.update(networks_to_update={"actor","temperature"}) (sac.py:193-234,243-299).
    noise: eps_pi, mask_obs_pi {cam}, eps_temp, mask_next_temp {cam}."""
    cfg = st.cfg
    info, grads, aux = {}, {}, {}
    if do_actor:
        p = {k: v.detach().clone().requires_grad_(True) for k, v in st.params.items()}
        # policy loss: grads through the actions and log-probs only (critic/encoder-of-critic constant)
        alpha = F.softplus(st.params["temp/lagrange"])                               # forward_temperature
        enc_pi = encode(p, cfg, feats_obs, state, drop_masks=noise["mask_obs_pi"], stop_gradient=True)
        mean, std = policy_head(p, cfg, enc_pi)
        a, logp = sample_and_log_prob(mean, std, noise["eps_pi"])
        enc_c = encode(st.params, cfg, feats_obs, state)
        q = critic_forward(st.params, cfg, enc_c, a).mean(dim=0)                     # sac.py:203-208
        actor_loss = -(q - alpha * logp).mean()                                      # sac.py:212-213
        grads["actor"] = _grad_dict(actor_loss, p)
        info.update({"actor_loss": actor_loss.item(), "temperature": alpha.item(), "entropy": (-logp.mean()).item()})
        aux.update({"g_actor": grads["actor"], "actions": a.detach(), "logp": logp.detach()})
    if do_temp:
        with torch.no_grad():
            enc_n = encode(st.params, cfg, feats_next, next_state, drop_masks=noise["mask_next_temp"], stop_gradient=True)
            m2, s2 = policy_head(st.params, cfg, enc_n)
            _, logp_n = sample_and_log_prob(m2, s2, noise["eps_temp"])
            entropy = -logp_n.mean()                                                 # sac.py:229
        lam = st.params["temp/lagrange"].detach().clone().requires_grad_(True)
        temp_loss = F.softplus(lam) * (entropy - cfg.target_entropy)                 # lagrange.py:63-72
        grads["temperature"] = {"temp/lagrange": torch.autograd.grad(temp_loss, lam)[0]}
        info["temperature_loss"] = temp_loss.item()
        aux["g_temp"] = grads["temperature"]
    if apply:
        apply_gradients(st, grads)                                                   # no EMA (sac.py:284)
    return info, aux


def update(st: TrainState, batch, noise, networks=("actor", "critic", "temperature")):
    """SACAgent.update(batch, networks_to_update) (sac.py:243-299) on an already augmented batch: every selected loss is
    evaluated at the same (pre-update) parameters, the others are `lambda params, rng: (0.0, {})` (zero gradients that
    still step their optimizer), ONE apply_gradients over all three optimizers, target EMA iff "critic" is selected."""
    nets = set(networks)
    assert nets and nets <= {"actor", "critic", "temperature"}, f"Invalid gradient steps: {networks}"
    fo, fn = features(st, batch["obs"]), features(st, batch["next"])
    n = dict(noise)
    if "redq_idx" in noise:
        n["redq_idx"] = np.asarray(noise["redq_idx"]).reshape(-1, 2)[0]
    info, grads = {}, {}
    if "critic" in nets:
        ci, caux = critic_update(st, fo, fn, batch["state"], batch["next_state"], batch["action"], batch["reward"],
                                 batch["mask"], n, apply=False)
        info.update(ci)
        grads["critic"] = caux["grads"]
    if nets & {"actor", "temperature"}:
        ai, aaux = actor_temp_update(st, fo, fn, batch["state"], batch["next_state"], n, apply=False,
                                     do_actor="actor" in nets, do_temp="temperature" in nets)
        info.update(ai)
        if "actor" in nets:
            grads["actor"] = aaux["g_actor"]
        if "temperature" in nets:
            grads["temperature"] = aaux["g_temp"]
    apply_gradients(st, grads)
    if "critic" in nets:
        target_update(st)
    return info


def features(st: TrainState, frames_u8, chunk=64):
    """frames_u8 {cam: [N,H,W,3] uint8 torch} -> {cam: [N,h,w,512]} through the frozen trunk."""
    out = {}
    if st.cfg.state_only:
        return out
    if st.cfg.small:   # the trainable encoder runs inside encode() with the parameters of the caller (online / target)
        return dict(frames_u8)
    with torch.no_grad():
        for k, v in frames_u8.items():
            parts = [trunk_forward(st.trunk, v[i:i + chunk], st.dtype) for i in range(0, v.shape[0], chunk)]
            out[k] = torch.cat(parts, dim=0)
    return out


def _slice_noise(noise, lo, hi, names_dict=("mask_next",), names_arr=("eps_next",)):
    out = dict(noise)
    for n in names_dict:
        out[n] = {k: v[lo:hi] for k, v in noise[n].items()}
    for n in names_arr:
        out[n] = noise[n][lo:hi]
    return out


def update_critics(st, batch, noise):
    """DrQAgent.update_critics (drq.py:296-328).  batch: cropped frames
    {"obs": {cam: u8[B,H,W,3]}, "next": {...}, "state", "next_state", "action", "reward", "mask"}."""
    fo, fn = features(st, batch["obs"]), features(st, batch["next"])
    n = dict(noise)
    n["redq_idx"] = np.asarray(noise["redq_idx"]).reshape(-1, 2)[0]
    return critic_update(st, fo, fn, batch["state"], batch["next_state"], batch["action"],
                         batch["reward"], batch["mask"], n)


def update_high_utd(st, batch, noise, utd_ratio=1):
    """DrQAgent.update_high_utd (drq.py:255-294) -> SACAgent.update_high_utd (sac.py:544-596):
    `utd_ratio` critic updates on consecutive minibatches, then one actor+temperature update on
    the full batch.  Returned info = mean of the critic infos + actor/temperature infos."""
    B = batch["reward"].shape[0]
    assert B % utd_ratio == 0, f"Batch size {B} must be divisible by UTD ratio {utd_ratio}"  # sac.py:561-563
    mb = B // utd_ratio
    fo, fn = features(st, batch["obs"]), features(st, batch["next"])
    infos = []
    redq = np.asarray(noise["redq_idx"]).reshape(-1, 2)
    for i in range(utd_ratio):
        lo, hi = i * mb, (i + 1) * mb
        n = _slice_noise(noise, lo, hi)
        n["redq_idx"] = redq[i]
        info, _ = critic_update(st, {k: v[lo:hi] for k, v in fo.items()}, {k: v[lo:hi] for k, v in fn.items()},
                                batch["state"][lo:hi], batch["next_state"][lo:hi], batch["action"][lo:hi],
                                batch["reward"][lo:hi], batch["mask"][lo:hi], n)
        infos.append(info)
    info = {k: float(np.mean([d[k] for d in infos])) for k in infos[0]}
    ainfo, aux = actor_temp_update(st, fo, fn, batch["state"], batch["next_state"], noise)
    info.update(ainfo)
    return info, aux


def make_noise(cfg: Config, B, seed=7, utd_ratio=1):
    """Parity-mode noise: generator seed 7."""
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
    D = cfg.sle_dim

    def masks():
        return {k: (rng.random((B, D)) < (1.0 - cfg.dropout)).astype(np.uint8) for k in cfg.image_keys}

    return {
        "crop_obs": rng.integers(0, 9, size=(B, 2)).astype(np.int32),
        "crop_next": rng.integers(0, 9, size=(B, 2)).astype(np.int32),
        "eps_next": rng.standard_normal((B, cfg.A)).astype(np.float32),
        "mask_next": masks(),
        "redq_idx": rng.integers(0, cfg.ensemble, size=(utd_ratio, 2)).astype(np.int32),
        "eps_pi": rng.standard_normal((B, cfg.A)).astype(np.float32),
        "mask_obs_pi": masks(),
        "eps_temp": rng.standard_normal((B, cfg.A)).astype(np.float32),
        "mask_next_temp": masks(),
    }


def noise_to_torch(noise, dtype):
    out = {}
    for k, v in noise.items():
        if isinstance(v, dict):
            out[k] = {c: torch.tensor(m) for c, m in v.items()}
        elif k.startswith("eps"):
            out[k] = torch.tensor(v, dtype=dtype)
        else:
            out[k] = v
    return out
