"""empty stand-in for ml_collections (serl_launcher/common/wandb.py)."""


class ConfigDict(dict):
    pass
