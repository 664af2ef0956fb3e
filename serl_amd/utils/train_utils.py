"""serl_launcher/utils/train_utils.py:16-66 names: concat_batches, _unpack."""
from ..data.data_store import LazyBatch, concat_batches  # noqa: F401


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
