"""JAX's PRNG stream for the learner (ctypes face of csrc/jaxrng.hip; include/serl_mi355.h "JAX's PRNG").

The reference draws crop offsets, REDQ indices, policy noise and Dropout masks from `jax.random` (threefry2x32) and advances
`agent.state.rng` as common/common.py:197-209, agents/continuous/sac.py:287-289 and agents/continuous/drq.py:276-318 do; the
functions here give the product the same stream: keys and integers bit-exact, normals through XLA's float32 erf_inv polynomial.
Keys are numpy uint32[2] (`jax.random.PRNGKey(seed)` = [seed >> 32, seed & 0xffffffff]).  Nothing here computes on the CPU
what the learner's kernels compute: host code derives KEYS and small integer draws, the device fills the noise tensors."""
from __future__ import annotations

import ctypes as C
import hashlib
from typing import Iterable, Sequence

import numpy as np

from . import _lib

MAX_UTD = 32
NORMAL, BERNOULLI_U8, BITS = 0, 1, 2


class _UpdateKeys(C.Structure):
    _fields_ = [("rng_out", C.c_uint32 * 2), ("k_obs", C.c_uint32 * 2), ("k_next", C.c_uint32 * 2), ("n_critic", C.c_int32),
                ("k_next_action", (C.c_uint32 * 2) * MAX_UTD), ("k_subsample", (C.c_uint32 * 2) * MAX_UTD),
                ("k_policy", C.c_uint32 * 2), ("k_sample", C.c_uint32 * 2), ("k_temp", C.c_uint32 * 2)]


class _Job(C.Structure):
    _fields_ = [("key", C.c_uint32 * 2), ("kind", C.c_int32), ("p", C.c_float), ("n_total", C.c_int64), ("first", C.c_int64),
                ("count", C.c_int64), ("out", C.c_void_p)]


def _key(key) -> np.ndarray:
    k = np.ascontiguousarray(np.asarray(key).reshape(-1), dtype=np.uint32)
    if k.size != 2:
        raise ValueError(f"a PRNG key is two 32-bit words, got shape {np.shape(key)}")
    return k


def _kp(k: np.ndarray):
    return k.ctypes.data_as(C.POINTER(C.c_uint32))


def is_key(x) -> bool:
    try:
        a = np.asarray(x)
    except Exception:
        return False
    return a.dtype.kind in "iu" and a.size == 2 and a.ndim >= 1


def prngkey(seed: int) -> np.ndarray:
    """jax.random.PRNGKey(seed) as the reference runs it (x64 disabled: the seed is canonicalised to int32, so the key is
    [0, seed & 0xffffffff] -- PRNGKey(-1) = [0, 0xffffffff], PRNGKey(2**32 + 5) = PRNGKey(5); jax._src.prng.threefry_seed)"""
    out = np.zeros(2, np.uint32)
    _lib.check(_lib.lib().serl_jax_prngkey(C.c_uint64(int(seed) & 0xFFFFFFFF), _kp(out)))
    return out


def split(key, num: int = 2) -> np.ndarray:
    """jax.random.split(key, num) -> uint32[num][2]"""
    k, out = _key(key), np.zeros((int(num), 2), np.uint32)
    _lib.check(_lib.lib().serl_jax_split(_kp(k), int(num), _kp(out)))
    return out


def fold_in(key, data: int) -> np.ndarray:
    """jax.random.fold_in(key, data)"""
    k, out = _key(key), np.zeros(2, np.uint32)
    _lib.check(_lib.lib().serl_jax_fold_in(_kp(k), C.c_uint32(int(data) & 0xFFFFFFFF), _kp(out)))
    return out


def random_bits(key, n: int) -> np.ndarray:
    k, out = _key(key), np.zeros(int(n), np.uint32)
    _lib.check(_lib.lib().serl_jax_random_bits(_kp(k), C.c_int64(int(n)), _kp(out)))
    return out


def randint(key, n: int, minval: int, maxval: int) -> np.ndarray:
    """jax.random.randint(key, (n,), minval, maxval) -> int32[n]"""
    k, out = _key(key), np.zeros(int(n), np.int32)
    _lib.check(_lib.lib().serl_jax_randint(_kp(k), C.c_int64(int(n)), int(minval), int(maxval), out.ctypes.data_as(C.POINTER(C.c_int32))))
    return out


def normal_host(key, n: int) -> np.ndarray:
    """jax.random.normal(key, (n,)) evaluated by the library's host code (tests, single actions)"""
    k, out = _key(key), np.zeros(int(n), np.float32)
    _lib.check(_lib.lib().serl_jax_normal_host(_kp(k), C.c_int64(int(n)), out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


def crop_offsets(key, frames: int, padding: int = 4) -> np.ndarray:
    """batched_random_crop's per-frame (y, x) offsets (vision/data_augmentations.py:7-36) -> int32[frames][2]"""
    k, out = _key(key), np.zeros((int(frames), 2), np.int32)
    _lib.check(_lib.lib().serl_jax_crop_offsets(_kp(k), int(frames), int(padding), out.ctypes.data_as(C.POINTER(C.c_int32))))
    return out


class UpdateKeys:
    """Every key one learner call derives from state.rng (serl_jax_update_keys; order documented in the header)."""

    def __init__(self, rng, drq_aug: bool, n_critic: int, has_actor_temp: bool, combined: bool = False):
        # The C struct holds SERL_JAX_MAX_UTD critic slots; the schedule is a chain (every update derives its keys from the running
        # rng and advances it), so a larger utd_ratio -- the reference accepts any divisor of the batch (sac.py:544-596) -- is the
        # same call made on consecutive windows of <= MAX_UTD critic updates, the actor/temperature update riding on the last one.
        k = _key(rng)
        a = lambda f: np.array(list(f), np.uint32)
        self.rng_in = k.copy()
        self.k_next_action, self.k_subsample = [], []
        done, first, cur = 0, True, k
        while True:
            n = min(n_critic - done, MAX_UTD)
            last = done + n == n_critic
            s = _UpdateKeys()
            _lib.check(_lib.lib().serl_jax_update_keys(_kp(cur), int(bool(drq_aug) and first), int(n), int(bool(has_actor_temp) and last),
                                                       int(bool(combined)), C.byref(s)))
            if first:
                self.k_obs, self.k_next = a(s.k_obs), a(s.k_next)
            self.k_next_action += [a(s.k_next_action[i]) for i in range(n)]
            self.k_subsample += [a(s.k_subsample[i]) for i in range(n)]
            cur, done, first = a(s.rng_out), done + n, False
            if last:
                break
        self.rng_out = cur
        self.k_policy, self.k_sample, self.k_temp = a(s.k_policy), a(s.k_sample), a(s.k_temp)
        self.n_critic, self.has_actor_temp = n_critic, bool(has_actor_temp)


def flax_make_rng(rng, path: Sequence[str], counter: int = 1) -> np.ndarray:
    """The key flax's `Module.make_rng(name)` hands a module at `path` (names from the root) on its `counter`-th call
    (flax >= 0.8, serl_launcher/requirements.txt: `flax/core/scope.py` LazyRng suffix = path names + call counter, folded in by
    `_fold_in_static`: ONE fold_in of the first four bytes, big-endian, of the SHA-1 over the concatenated UTF-8 names and the
    counter's big-endian bytes; `flax_fix_rng_separator` off).  Restated from the published source -- no flax install exists in
    this image to pin it against."""
    m = hashlib.sha1()
    for x in tuple(path) + (int(counter),):
        if isinstance(x, str):
            m.update(x.encode("utf-8"))
        else:
            m.update(int(x).to_bytes((int(x).bit_length() + 7) // 8, byteorder="big"))
    return fold_in(rng, int.from_bytes(m.digest()[:4], byteorder="big"))


def dropout_path(image_key: str) -> tuple:
    """Scope path of the Dropout layer behind a camera's SpatialLearnedEmbeddings in the POLICY's encoder (the only Dropout that
    is ever active: vision/resnet_v1.py:352, networks/actor_critic_nets.py:185, agents/continuous/drq.py:155-199)."""
    return ("modules_actor", "encoder", f"encoder_{image_key}", "Dropout_0")


def job(kind: int, key, n_total: int, out_ptr: int, first: int = 0, count: int | None = None, p: float = 0.0) -> _Job:
    k = _key(key)
    j = _Job()
    j.key[0], j.key[1] = int(k[0]), int(k[1])
    j.kind, j.p = int(kind), float(p)
    j.n_total, j.first, j.count = int(n_total), int(first), int(n_total - first if count is None else count)
    j.out = C.c_void_p(int(out_ptr))
    return j


def fill(device: int, jobs: Iterable[_Job], stream) -> None:
    """One launch (per 16 jobs) writing every job's window of its jax.random array into device memory on `stream`."""
    jobs = list(jobs)
    for i in range(0, len(jobs), 16):
        part = jobs[i:i + 16]
        arr = (_Job * len(part))(*part)
        _lib.check(_lib.lib().serl_jax_fill(int(device), arr, len(part), stream))
