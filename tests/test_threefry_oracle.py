"""CPU: JAX's PRNG restated (oracle/jaxshim/jax/threefry.py) against
* the Random123 known-answer vectors of threefry2x32 with 20 rounds (the three rows JAX's own test suite checks), and
* the keys / samples JAX's documentation prints ("Pseudorandom numbers in JAX": PRNGKey(0) / PRNGKey(42) -> split -> normal),
and the reference's consumption pattern (vision/data_augmentations.py:22-36 split + randint, agents/continuous/sac.py:151-157
randint over the ensemble): integer draws are exact, normals agree with JAX to ~1 float32 ulp (erf_inv is evaluated in float64
here, by a float32 polynomial in XLA)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle", "jaxshim"))
import importlib.util

_spec = importlib.util.spec_from_file_location("serl_threefry", os.path.join(ROOT, "oracle", "jaxshim", "jax", "threefry.py"))
tf = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(tf)


def test_random123_known_answers():
    kat = [((0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6b200159, 0x99ba4efe)),
           ((0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x1cb996fc, 0xbb002be7)),
           ((0x13198a2e, 0x03707344), (0x243f6a88, 0x85a308d3), (0xc4923a9c, 0x483df7a0))]
    for key, ctr, exp in kat:
        y0, y1 = tf.threefry2x32(key[0], key[1], np.array([ctr[0]], np.uint32), np.array([ctr[1]], np.uint32))
        assert (int(y0[0]), int(y1[0])) == exp


def test_keys_and_samples_printed_in_the_jax_documentation():
    assert tf.PRNGKey(42).tolist() == [0, 42]
    assert tf.split(tf.PRNGKey(0)).tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]
    key, sub = tf.split(tf.PRNGKey(42))
    assert key.tolist() == [2465931498, 3679230171] and sub.tolist() == [255383827, 267815257]
    assert abs(float(tf.normal(tf.PRNGKey(42))) - (-0.18471177)) < 2e-7
    assert abs(float(tf.normal(sub, (1,))[0]) - 1.3694694) < 2e-7


def test_reference_consumption_patterns():
    # batched_random_crop: keys = split(k, B); (dy, dx)_i = randint(keys[i], (2,), 0, 2 * padding + 1)
    keys = tf.split(tf.PRNGKey(7), 16)
    off = np.stack([tf.randint(k, (2,), 0, 9) for k in keys])
    assert off.shape == (16, 2) and off.dtype == np.int32 and off.min() >= 0 and off.max() <= 8
    assert len({tuple(o) for o in off.tolist()}) > 8            # independent keys: the offsets differ
    idx = tf.randint(tf.PRNGKey(3), (2,), 0, 10)                  # REDQ subsample
    assert idx.shape == (2,) and 0 <= idx.min() and idx.max() < 10
    # a large sample behaves like its distribution (uniform bits -> 23-bit mantissas; normal through erf_inv)
    u = tf.uniform(tf.PRNGKey(1), (20000,))
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.01
    z = tf.normal(tf.PRNGKey(2), (20000,))
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1.0) < 0.03
    keep = tf.bernoulli(tf.PRNGKey(5), 0.9, (20000,))
    assert abs(keep.mean() - 0.9) < 0.01
    # odd sizes are padded with one zero counter and truncated (threefry_2x32): a prefix property does NOT hold, sizes matter
    assert tf.random_bits(tf.PRNGKey(9), (5,)).shape == (5,)


def test_standin_jax_random_uses_the_key_chain_when_asked(monkeypatch):
    monkeypatch.setenv("SERL_JAXSHIM_PRNG", "threefry")
    import jax.random as jr          # the stand-in under oracle/jaxshim
    k = jr.PRNGKey(42)
    k1, k2 = jr.split(k)
    to_list = lambda a: [int(x) for x in np.asarray(a.t if hasattr(a, "t") else a).reshape(-1)]   # noqa: E731
    import jax._core as jc
    assert [int(x) for x in jc.raw(k1).reshape(-1)] == [2465931498, 3679230171]
    assert abs(float(jc.raw(jr.normal(k2, (1,))).reshape(-1)[0]) - 1.3694694) < 2e-7
