"""flax.training stand-in (train_state, checkpoints).  TEST INFRASTRUCTURE ONLY -- see oracle/jaxshim/README.md."""
from . import checkpoints, train_state  # noqa: F401
