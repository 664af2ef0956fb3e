from . import frozen_dict  # noqa: F401
from .frozen_dict import FrozenDict, freeze, unfreeze  # noqa: F401
