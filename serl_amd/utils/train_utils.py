"""serl_launcher/utils/train_utils.py names: concat_batches (:16-41), _unpack (:44-66), load_resnet10_params (:69-130)."""
import os
import pickle

from ..data.data_store import LazyBatch, concat_batches  # noqa: F401


def load_resnet10_params(agent, image_keys=("image",), public=True, file_path=None):
    """train_utils.py:69-130: load `~/.serl/resnet10_params.pkl` (a pickled flax tree of the ImageNet ResNet-10) into the
    agent's frozen trunk and return the agent.  The reference downloads the file when it is absent; this box has no
    network, so a missing file is an error that says where to put it.  `image_keys` is accepted for signature parity:
    the trunk is ONE shared module (drq.py:165-176), every camera's encoder reads the same leaves.  The reference patches
    `agent.state.params` in place right after creation, when `target_params` still IS that same tree
    (sac.py:378-382 `JaxRLTrainState.create(..., target_params=params)`), so both trees receive the weights -- as here."""
    path = file_path or os.path.join(os.path.expanduser("~/.serl/"), "resnet10_params.pkl")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not found; copy resnet10_params.pkl there (the reference downloads it from its "
                                "GitHub release, train_utils.py:76-107; no network here)")
    with open(path, "rb") as f:
        encoder_params = pickle.load(f)
    replaced = [k for k in encoder_params]
    agent.load_trunk_params(encoder_params)
    for k in replaced:
        print(f"replaced {k} in pretrained_encoder")
    return agent


def _unpack(batch):
    """train_utils.py:44-66 on a dict batch of device tensors (lazy batches are unpacked inside the
    fused gather kernel)."""
    if isinstance(batch, LazyBatch):
        return batch
    obs, nobs = dict(batch["observations"]), dict(batch["next_observations"])
    for k in list(obs.keys()):
        if k not in nobs:
            packed = obs[k]
            obs[k], nobs[k] = packed[:, :-1], packed[:, 1:]
    out = dict(batch)
    out["observations"], out["next_observations"] = obs, nobs
    return out
