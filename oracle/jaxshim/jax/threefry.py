"""JAX's PRNG restated in NumPy: threefry2x32 (Random123, 20 rounds) and the jax.random functions the reference's update path
draws with (the README pins jax 0.4.35, i.e. `jax_threefry_partitionable=False`):

    PRNGKey / split / fold_in / random_bits / uniform / normal / randint / bernoulli

TEST INFRASTRUCTURE ONLY (oracle/): nothing under serl_amd/ imports this.  Sources restated (jax is not installable here):
jax/_src/prng.py (`threefry_2x32`, `_threefry_split`, `_threefry_fold_in`, `_threefry_random_bits_original`) and jax/_src/random.py
(`_uniform`, `_normal_real`, `_randint`, `_bernoulli`) as published for jax 0.4.x.  PINNED by tests/test_threefry_oracle.py against
the Random123 known-answer vectors for threefry2x32-20 and against the key / sample values printed in JAX's own documentation
("Pseudorandom numbers" tutorial: PRNGKey(42) -> split -> normal).  `erf_inv` is evaluated in float64 (scipy) and rounded to
float32, XLA uses a float32 polynomial: normals agree to ~1 ulp, not to the bit; everything integer is exact.
With SERL_JAXSHIM_PRNG=threefry the stand-in `jax.random` (random.py) draws through this module instead of its keyed Philox
streams, so the reference's code consumes the numbers a real JAX run would (crop offsets, REDQ indices, policy noise)."""
from __future__ import annotations

import hashlib

import numpy as np

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
_M32 = np.uint64(0xFFFFFFFF)


def _rotl(x, r):
    return ((x << np.uint32(r)) | (x >> np.uint32(32 - r))).astype(np.uint32)


def threefry2x32(k0, k1, x0, x1):
    """20-round Threefry-2x32 of counter words (x0, x1) (uint32 arrays) under key (k0, k1) -> (y0, y1)."""
    with np.errstate(over="ignore"):
        k0, k1 = np.uint32(k0), np.uint32(k1)
        ks = (k0, k1, np.uint32(k0 ^ k1 ^ np.uint32(0x1BD11BDA)))
        x0 = (np.asarray(x0, np.uint32) + ks[0]).astype(np.uint32)
        x1 = (np.asarray(x1, np.uint32) + ks[1]).astype(np.uint32)
        for i in range(5):
            for r in _ROT[i % 2]:
                x0 = (x0 + x1).astype(np.uint32)
                x1 = _rotl(x1, r) ^ x0
            x0 = (x0 + ks[(i + 1) % 3]).astype(np.uint32)
            x1 = (x1 + ks[(i + 2) % 3] + np.uint32(i + 1)).astype(np.uint32)
    return x0, x1


def threefry_2x32(key, count):
    """jax._src.prng.threefry_2x32: a flat uint32 counter array is hashed in two halves (odd sizes padded with one zero)."""
    key = np.asarray(key, np.uint32).reshape(2)
    count = np.asarray(count, np.uint32).reshape(-1)
    odd = count.size % 2
    if odd:
        count = np.concatenate([count, np.zeros(1, np.uint32)])
    h = count.size // 2
    y0, y1 = threefry2x32(key[0], key[1], count[:h], count[h:])
    out = np.concatenate([y0, y1])
    return out[:-1] if odd else out


def PRNGKey(seed):
    # x64 disabled (the reference's setting): the seed is canonicalised to int32 and the key's high word is 0
    return np.array([0, int(seed) & 0xFFFFFFFF], np.uint32)


def split(key, num=2):
    return threefry_2x32(key, np.arange(2 * int(num), dtype=np.uint32)).reshape(int(num), 2)


def fold_in(key, data):
    return threefry_2x32(key, PRNGKey(int(data) & 0xFFFFFFFF))


def random_bits(key, shape):
    n = int(np.prod(shape)) if len(tuple(shape)) else 1
    return threefry_2x32(key, np.arange(n, dtype=np.uint32)).reshape(tuple(shape))


def uniform(key, shape=(), minval=0.0, maxval=1.0):
    """float32 uniform in [minval, maxval): 23 mantissa bits of each 32-bit draw."""
    bits = random_bits(key, shape)
    f = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    lo, hi = np.float32(minval), np.float32(maxval)
    return np.maximum(lo, (f * (hi - lo) + lo).astype(np.float32)).astype(np.float32)


def normal(key, shape=()):
    from scipy.special import erfinv
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
    u = uniform(key, shape, lo, 1.0)
    return (np.float32(np.sqrt(2.0)) * erfinv(u.astype(np.float64)).astype(np.float32)).astype(np.float32)


def randint(key, shape, minval, maxval):
    """int32 draws in [minval, maxval): two 32-bit draws per element combined modulo the span (jax._src.random._randint)."""
    k1, k2 = split(key)
    hi_bits, lo_bits = random_bits(k1, shape).astype(np.uint64), random_bits(k2, shape).astype(np.uint64)
    span = np.uint64(max(int(maxval) - int(minval), 1))
    mult = np.uint64(1 << 16) % span
    mult = ((mult * mult) & _M32) % span      # lax.mul on uint32 wraps (matters for span > 65536)
    with np.errstate(over="ignore"):
        off = (((hi_bits % span) * mult) & _M32) + (lo_bits % span)      # uint32 arithmetic (wraps like lax.mul / lax.add)
        off = (off & _M32) % span
    return (np.int64(minval) + off.astype(np.int64)).astype(np.int32)


def bernoulli(key, p, shape):
    return uniform(key, shape) < np.float32(p)


def flax_fold_in_static(rng, path):
    """flax >= 0.8 (serl_launcher/requirements.txt): `flax/core/scope.py` `_fold_in_static` -- ONE fold_in of the first four bytes
    (big-endian) of the SHA-1 over the concatenated suffix (strings as UTF-8, integers as their big-endian bytes;
    `flax_fix_rng_separator` off).  The suffix of `Module.make_rng` is the scope path from the root plus the scope's call
    counter (starting at 1).  Restated from the published source, NOT pinned (no flax install exists to check against)."""
    m = hashlib.sha1()
    for x in path:
        if isinstance(x, str):
            m.update(x.encode("utf-8"))
        else:
            m.update(int(x).to_bytes((int(x).bit_length() + 7) // 8, byteorder="big"))
    return fold_in(rng, int.from_bytes(m.digest()[:4], "big"))


def flax_fold_in_path(rng, path):
    """flax's LazyRng suffix folding (flax/core/scope.py `_legacy_rng_fold_in`, the default up to flax 0.8): strings fold in the
    first four bytes (big-endian) of their SHA-1, integers fold in as they are.  Flax-version dependent and NOT pinned here
    (no flax install exists to check against); used only to place Dropout keys when SERL_JAXSHIM_PRNG=threefry."""
    for x in path:
        if isinstance(x, str):
            d = hashlib.sha1(x.encode("utf-8")).digest()
            rng = fold_in(rng, int.from_bytes(d[:4], "big"))
        else:
            rng = fold_in(rng, int(x))
    return rng
