"""ctypes binding of libserl_mi355.so (the C ABI declared in include/serl_mi355.h).

The product path has NO CPU fallback: if the shared library is missing this module raises at
import of the symbols, and every op raises SerlError on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SERL_MI355_LIB: load another build of the same ABI (A/B timing of kernel variants on one box)
LIB_PATH = os.environ.get("SERL_MI355_LIB") or os.path.join(_HERE, "lib", "libserl_mi355.so")

MAX_CAMS = 4
MAX_BUFFERS = 2


class SerlError(RuntimeError):
    pass


class SerlBatch(C.Structure):
    _fields_ = [
        ("batch", C.c_int), ("n_cam", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int),
        ("state_dim", C.c_int), ("act_dim", C.c_int),
        ("frames", C.c_void_p), ("state", C.c_void_p), ("action", C.c_void_p),
        ("reward", C.c_void_p), ("mask", C.c_void_p), ("done", C.c_void_p),
    ]


_lib = None


def _declare(lib):
    vp, i32, i64, u64, u32, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_uint32, C.c_float
    P = C.POINTER
    lib.serl_last_error.restype = C.c_char_p
    lib.serl_last_error.argtypes = []
    lib.serl_version.restype = i32
    lib.serl_device_count.restype = i32
    sigs = {
        "serl_rb_create": [i32, i64, i32, i32, i32, i32, i32, i32, i32, P(vp)],
        "serl_rb_destroy": [vp],
        "serl_rb_seed": [vp, u64, u64, u64, u64, i32, u32],
        "serl_rb_rng_state": [vp, P(u64), P(i32), P(u32)],
        "serl_rb_insert": [vp, P(vp), P(vp), vp, vp, vp, f32, f32, i32],
        "serl_rb_valid_mask": [vp, vp],
        "serl_rb_sample_indices": [vp, i32, vp],
        "serl_rb_gather_packed": [vp, vp, i32, P(vp), vp, vp, vp, vp, vp, vp, vp],
        "serl_rb_gather_crop": [P(vp), i32, P(vp), P(i32), vp, vp, P(SerlBatch), vp],
        "serl_crop_packed": [i32, P(vp), i32, i32, i32, i32, i32, vp, vp, vp, vp],
        "serl_profile_enable": [i32],
        "serl_profile_reset": [],
        "serl_profile_read": [i32, vp, vp, vp, P(i32)],
        # JAX's PRNG (csrc/jaxrng.hip; serl_amd/jaxrng.py)
        "serl_jax_prngkey": [u64, P(u32)],
        "serl_jax_split": [P(u32), i32, P(u32)],
        "serl_jax_fold_in": [P(u32), u32, P(u32)],
        "serl_jax_random_bits": [P(u32), i64, P(u32)],
        "serl_jax_randint": [P(u32), i64, i32, i32, P(i32)],
        "serl_jax_normal_host": [P(u32), i64, P(f32)],
        "serl_jax_crop_offsets": [P(u32), i32, i32, P(i32)],
        "serl_jax_update_keys": [P(u32), i32, i32, i32, i32, vp],
        "serl_jax_fill": [i32, vp, i32, vp],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = i32
    for name in ("serl_rb_len", "serl_rb_insert_index"):
        fn = getattr(lib, name)
        fn.argtypes = [vp]
        fn.restype = i64
    return sigs


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SerlError(
                f"{LIB_PATH} not found: build it with `python -m serl_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
        _declare_agent(_lib)
    return _lib


def _declare_agent(lib):
    """Agent entry points (declared in a second step so replay-only builds still load)."""
    try:
        from . import _lib_agent
    except ImportError:
        return
    _lib_agent.declare(lib)


def check(status: int):
    if status != 0:
        msg = lib().serl_last_error()
        raise SerlError(f"libserl_mi355 status {status}: {msg.decode() if msg else '?'}")


def profile_read(max_entries=64):
    """-> {tag: (total_ms, count)} of the instrumented kernels since the last reset."""
    import numpy as np
    names = C.create_string_buffer(max_entries * 64)
    ms = np.zeros(max_entries, np.float64)
    cnt = np.zeros(max_entries, np.int64)
    n = C.c_int()
    check(lib().serl_profile_read(max_entries, C.cast(names, C.c_void_p), ms.ctypes.data, cnt.ctypes.data, C.byref(n)))
    out = {}
    for i in range(n.value):
        tag = names.raw[i * 64:(i + 1) * 64].split(b"\0")[0].decode()
        out[tag] = (float(ms[i]), int(cnt[i]))
    return out


def exported_symbols():
    """Names declared in include/serl_mi355.h (parsed), for the symbol-presence test."""
    import re
    hdr = os.path.join(_HERE, "..", "include", "serl_mi355.h")
    txt = open(hdr).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(serl_[a-z0-9_]+)\s*\(", txt)))
