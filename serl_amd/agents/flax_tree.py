"""Flax-layout view of the agent's parameter arena (what `server.publish_network(agent.state.params)`
and flax checkpoints expect; examples/async_drq_sim/async_drq_sim.py:104-108,229,295-307).

Tree derived from flax naming rules at the reference's construction sites
(agents/continuous/drq.py:165-222, common/common.py:58-78, vision/resnet_v1.py): modules passed as attributes are
named by attachment (`modules_actor`, `network`, `encoder_<key>`), a module instance shared by several owners
(the EncodingWrapper of actor and critic, the frozen `pretrained_encoder` of all cameras) is adopted once -- the
evidence is the reference's own loader, which patches `modules_actor` only and guards
`if "pretrained_encoder" in new_encoder_params` (utils/train_utils.py:118-127).  tests/test_reference_update.py checks
these paths against the tree the reference's own code builds under the flax stand-in (oracle/jaxshim); no real flax
install is available, so the alias switches below remain for the alternative readings.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from ..utils.init import STAGES, theta_shapes, trunk_shapes


def _put(tree, path, value):
    d = tree
    for p in path[:-1]:
        d = d.setdefault(p, {})
    d[path[-1]] = value


def _trunk_paths():
    """flat trunk leaf -> path below `pretrained_encoder`."""
    m = {"trunk/conv_init": ("conv_init", "kernel"), "trunk/norm_init/scale": ("norm_init", "scale"),
         "trunk/norm_init/bias": ("norm_init", "bias")}
    cin = 64
    for i, (f, s) in enumerate(STAGES):
        p, b = f"trunk/block{i}/", f"ResNetBlock_{i}"
        m[p + "conv0"] = (b, "Conv_0", "kernel")
        m[p + "gn0/scale"] = (b, "MyGroupNorm_0", "scale")
        m[p + "gn0/bias"] = (b, "MyGroupNorm_0", "bias")
        m[p + "conv1"] = (b, "Conv_1", "kernel")
        m[p + "gn1/scale"] = (b, "MyGroupNorm_1", "scale")
        m[p + "gn1/bias"] = (b, "MyGroupNorm_1", "bias")
        if s != 1 or cin != f:
            m[p + "proj"] = (b, "conv_proj", "kernel")
            m[p + "gnp/scale"] = (b, "norm_proj", "scale")
            m[p + "gnp/bias"] = (b, "norm_proj", "bias")
        cin = f
    return m


def theta_paths(image_keys, critic_mlp_name="network", encoder_type="resnet-pretrained"):
    """flat trainable leaf -> list of flax paths (aliases)."""
    enc = ("modules_actor", "encoder")
    m = {}
    if len(image_keys) == 0:
        # SACAgent.create_states: Policy(encoder=None, network=MLP) and ensemblize(Critic(encoder=None, network=MLP))
        # -- the vmapped Critic keeps its module names, every leaf gains a leading ensemble axis (sac.py:516-524)
        return _state_paths()
    for i, k in enumerate(image_keys):
        e = enc + (f"encoder_{k}",)
        if encoder_type == "small":   # SmallEncoder (small_encoders.py:9-55): Conv_0..3 {kernel, bias}
            for l in range(4):
                m[f"enc/{i}/conv{l}/kernel"] = [e + (f"Conv_{l}", "kernel")]
                m[f"enc/{i}/conv{l}/bias"] = [e + (f"Conv_{l}", "bias")]
        else:
            m[f"enc/{i}/sle"] = [e + ("SpatialLearnedEmbeddings_0", "kernel")]
        m[f"enc/{i}/dense/kernel"] = [e + ("Dense_0", "kernel")]
        m[f"enc/{i}/dense/bias"] = [e + ("Dense_0", "bias")]
        m[f"enc/{i}/ln/scale"] = [e + ("LayerNorm_0", "scale")]
        m[f"enc/{i}/ln/bias"] = [e + ("LayerNorm_0", "bias")]
    m["enc/proprio/dense/kernel"] = [enc + ("Dense_0", "kernel")]
    m["enc/proprio/dense/bias"] = [enc + ("Dense_0", "bias")]
    m["enc/proprio/ln/scale"] = [enc + ("LayerNorm_0", "scale")]
    m["enc/proprio/ln/bias"] = [enc + ("LayerNorm_0", "bias")]
    c = ("modules_critic", critic_mlp_name)
    for j, n in ((1, 0), (2, 1)):
        m[f"critic/w{j}"] = [c + (f"Dense_{n}", "kernel")]
        m[f"critic/b{j}"] = [c + (f"Dense_{n}", "bias")]
        m[f"critic/ln{j}/scale"] = [c + (f"LayerNorm_{n}", "scale")]
        m[f"critic/ln{j}/bias"] = [c + (f"LayerNorm_{n}", "bias")]
    m["critic/head/kernel"] = [("modules_critic", "Dense_0", "kernel")]
    m["critic/head/bias"] = [("modules_critic", "Dense_0", "bias")]
    a = ("modules_actor", "network")
    for j, n in ((1, 0), (2, 1)):
        m[f"actor/w{j}"] = [a + (f"Dense_{n}", "kernel")]
        m[f"actor/b{j}"] = [a + (f"Dense_{n}", "bias")]
        m[f"actor/ln{j}/scale"] = [a + (f"LayerNorm_{n}", "scale")]
        m[f"actor/ln{j}/bias"] = [a + (f"LayerNorm_{n}", "bias")]
    m["actor/mean/kernel"] = [("modules_actor", "Dense_0", "kernel")]
    m["actor/mean/bias"] = [("modules_actor", "Dense_0", "bias")]
    m["actor/logstd/kernel"] = [("modules_actor", "Dense_1", "kernel")]
    m["actor/logstd/bias"] = [("modules_actor", "Dense_1", "bias")]
    m["temp/lagrange"] = [("modules_temperature", "lagrange")]
    return m


def _state_paths():
    m = {}
    for mod, pre in (("modules_critic", "critic"), ("modules_actor", "actor")):
        for j, n in ((1, 0), (2, 1)):
            m[f"{pre}/w{j}"] = [(mod, "network", f"Dense_{n}", "kernel")]
            m[f"{pre}/b{j}"] = [(mod, "network", f"Dense_{n}", "bias")]
            m[f"{pre}/ln{j}/scale"] = [(mod, "network", f"LayerNorm_{n}", "scale")]
            m[f"{pre}/ln{j}/bias"] = [(mod, "network", f"LayerNorm_{n}", "bias")]
    m["critic/head/kernel"] = [("modules_critic", "Dense_0", "kernel")]
    m["critic/head/bias"] = [("modules_critic", "Dense_0", "bias")]
    m["actor/mean/kernel"] = [("modules_actor", "Dense_0", "kernel")]
    m["actor/mean/bias"] = [("modules_actor", "Dense_0", "bias")]
    m["actor/logstd/kernel"] = [("modules_actor", "Dense_1", "kernel")]
    m["actor/logstd/bias"] = [("modules_actor", "Dense_1", "bias")]
    m["temp/lagrange"] = [("modules_temperature", "lagrange")]
    return m


def trunk_owner(image_keys):
    """camera whose subtree holds the shared frozen trunk (first key in sorted order)."""
    return sorted(image_keys)[0]


def export_tree(core, section: str, image_keys, duplicate_encoder_under_critic: bool = False,
                critic_mlp_name: str = "network", trunk_under_every_camera: bool = False) -> Dict:
    """Nested dict of np.float32 arrays in flax layout (HWIO convs, (in,out) dense, ensemble axis 0)."""
    cfg = core.cfg
    etype = "small" if cfg.encoder_type == 1 else "resnet-pretrained"
    shapes = theta_shapes(cfg.n_cam, cfg.H, cfg.W, cfg.state_dim, cfg.act_dim, ensemble=cfg.ensemble, encoder_type=etype)
    tree: Dict = {}
    for leaf, paths in theta_paths(image_keys, critic_mlp_name, etype).items():
        v = core.get(section, leaf).reshape(shapes[leaf])
        for p in paths:
            _put(tree, p, v)
            if duplicate_encoder_under_critic and p[:2] == ("modules_actor", "encoder"):
                _put(tree, ("modules_critic",) + p[1:], v)
    tshapes = trunk_shapes()
    for leaf, sub in (_trunk_paths() if (cfg.n_cam and etype != "small") else {}).items():
        v = core.get(section, leaf).reshape(tshapes[leaf])
        # ONE frozen trunk: drq.py:165-176 passes the same `pretrained_encoder` module to every camera's
        # PreTrainedResNetEncoder, flax adopts a shared module once -- under the first camera in sorted-key order --
        # which is why train_utils.py:118-123 guards with `if "pretrained_encoder" in new_encoder_params`
        owners = image_keys if trunk_under_every_camera else (trunk_owner(image_keys),)
        for k in owners:
            _put(tree, ("modules_actor", "encoder", f"encoder_{k}", "pretrained_encoder") + sub, v)
            if duplicate_encoder_under_critic:
                _put(tree, ("modules_critic", "encoder", f"encoder_{k}", "pretrained_encoder") + sub, v)
    return tree


def trunk_from_flax(pretrained: Dict) -> Dict[str, np.ndarray]:
    """resnet10_params.pkl tree -> flat leaves.  Top-level keys are matched by name and a top-level key the pickle
    does not hold keeps its current value (train_utils.py:124-127: `for k in new_encoder_params: if k in encoder_params`);
    a top-level key that is present must hold the whole sub-tree."""
    out = {}
    for leaf, sub in _trunk_paths().items():
        if sub[0] not in pretrained:
            continue
        d = pretrained
        for p in sub:
            d = d[p]
        out[leaf] = np.asarray(d, np.float32)
    return out
