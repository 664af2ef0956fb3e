"""Stand-in for the parts of flax the reference's update path touches.  TEST INFRASTRUCTURE ONLY -- see
oracle/jaxshim/README.md.  (serl_launcher/requirements.txt: flax >= 0.8.0.)"""
from . import core, struct  # noqa: F401
from . import linen  # noqa: F401

__version__ = "0.8-standin"
