"""flax.training.checkpoints stand-in: only enough for the reference modules that import it to load.  Reading and
writing checkpoint files is the product's own code (serl_amd/utils/checkpoint.py).  TEST INFRASTRUCTURE ONLY."""


def save_checkpoint(*a, **k):
    raise NotImplementedError("flax.training.checkpoints is not available under the stand-ins")


def restore_checkpoint(*a, **k):
    raise NotImplementedError("flax.training.checkpoints is not available under the stand-ins")
