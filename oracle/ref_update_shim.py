"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (serl_amd/).

Imports the reference's SAC/DrQ update code UNMODIFIED from /root/reference and runs it on PyTorch-CPU under
stand-ins for its un-vendored third-party dependencies (oracle/jaxshim/: jax, flax, optax, distrax, chex; empty
tensorflow / imageio / wandb).  Used by tests/golden/make_golden_update.py to generate tests/golden/update_*.npz and by
tests/test_reference_update.py to pin oracle/drq_oracle.py against the reference's own code.

/root/reference only exists in the build container: callers check `reference_available()`.
"""
from __future__ import annotations

import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM_DIR = os.path.join(HERE, "jaxshim")
REFERENCE_ROOT = "/root/reference/serl_launcher"
_STANDINS = ("jax", "flax", "optax", "distrax", "chex", "tensorflow", "imageio", "wandb", "tensorflow_datasets", "agentlace", "absl", "ml_collections")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "serl_launcher", "agents"))


def install(float64: bool = True):
    """Put the stand-ins first on sys.path (dropping the minimal replay-only stubs of oracle/ref_shim.py if they were
    installed) and make the reference importable.  Returns the stand-in `jax` module."""
    import torch
    for name in list(sys.modules):
        root = name.split(".")[0]
        if root in _STANDINS or root == "serl_launcher":
            m = sys.modules[name]
            f = getattr(m, "__file__", None) or ""
            if not f.startswith(SHIM_DIR) and root != "serl_launcher":
                del sys.modules[name]          # a stub module object or a foreign install: ours must win
            elif root == "serl_launcher" and "jax" not in sys.modules:
                del sys.modules[name]
    if SHIM_DIR not in sys.path:
        sys.path.insert(0, SHIM_DIR)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(1, REFERENCE_ROOT)
    import jax
    assert jax.__file__.startswith(SHIM_DIR), f"a real jax is installed at {jax.__file__}: use it instead of the stand-in"
    import chex, distrax, flax, flax.core.frozen_dict, flax.linen, optax  # noqa: F401,E401  (claim the names before any stub)
    from . import ref_shim
    ref_shim.install_stubs()     # gym (setdefault: the full stand-ins above stay in place)
    jax._core.set_float_dtype(torch.float64 if float64 else torch.float32)
    # einops (real package, used by common/encoding.py) picks its backend by type: the stand-in Array IS a torch tensor
    import einops._backends as eb
    eb._type2backend[jax.Array] = eb.TorchBackend()
    return jax
