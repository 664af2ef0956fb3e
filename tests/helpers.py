"""Shared test helpers (CPU side)."""
import itertools
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class _Sp:
    def __init__(self, shape):
        self.shape = tuple(shape)


class _DictSp:
    def __init__(self, spaces):
        self.spaces = {k: spaces[k] for k in sorted(spaces)}


def make_spaces(keys, H, W, C, T, S, A):
    d = {"state": _Sp((T, S))}
    for k in keys:
        d[k] = _Sp((T, H, W, C))
    return _DictSp(d), _Sp((A,))


def load_case(name):
    z = np.load(os.path.join(GOLDEN, f"replay_{name}.npz"))
    H, W, C, T, S, A, cap, n_ins, ep, sseed, rseed, B, ns = [int(x) for x in z["meta"]]
    keys = tuple(str(k) for k in z["keys"])
    return z, dict(keys=keys, H=H, W=W, C=C, T=T, S=S, A=A, cap=cap, n_ins=n_ins, ep=ep,
                   sseed=sseed, rseed=rseed, B=B, ns=ns)


def stream_for(m):
    from serl_amd.utils.synthetic import transition_stream
    return itertools.islice(
        transition_stream(m["keys"], m["H"], m["W"], m["C"], m["T"], m["S"], m["A"], m["ep"], m["sseed"]),
        m["n_ins"])
