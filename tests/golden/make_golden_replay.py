"""Generates tests/golden/replay_*.npz from the REFERENCE's own MemoryEfficientReplayBuffer run
unmodified (oracle/ref_shim.py).  Run in the build container (needs /root/reference):
    python tests/golden/make_golden_replay.py
"""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
from serl_amd.utils.synthetic import flat_stream, transition_stream  # noqa: E402

CASES = {
    # name: (keys, H, W, C, T, S, A, capacity, n_insert, episode_len, stream_seed, rb_seed, batch, n_samples)
    "small_wrap": (("front", "wrist"), 16, 16, 3, 1, 5, 3, 37, 150, 7, 5, 3, 16, 6),
    "small_nowrap": (("wrist_1", "wrist_2"), 32, 16, 3, 1, 24, 6, 400, 306, 51, 11, 0, 64, 4),
    # hits the reference's negative-window quirk: a valid transition lands in slot 0
    "wrap_quirk": (("a", "b"), 16, 16, 3, 1, 4, 2, 23, 56, 5, 77, 9, 16, 6),
    "one_cam": (("image",), 16, 32, 3, 1, 7, 4, 64, 90, 10, 2, 1, 32, 3),
}


def main():
    Ref = ref_shim.load_reference_buffer_cls()
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, (keys, H, W, C, T, S, A, cap, n_ins, ep, sseed, rseed, B, ns) in CASES.items():
        ospace, aspace = ref_shim.make_spaces(keys, H, W, C, T, S, A)
        ref = Ref(ospace, aspace, cap, pixel_keys=keys)
        ref.seed(rseed)
        for tr in itertools.islice(transition_stream(keys, H, W, C, T, S, A, ep, sseed), n_ins):
            ref.insert(tr)
        rec = {"meta": np.array([H, W, C, T, S, A, cap, n_ins, ep, sseed, rseed, B, ns], np.int64),
               "keys": np.array(keys), "valid": ref._is_correct_index.copy(),
               "size": np.int64(len(ref)), "insert_index": np.int64(ref._insert_index)}
        for s in range(ns):
            # replicate sample() while also recording the indices (same RNG stream)
            st = ref.np_random.bit_generator.state
            b = ref.sample(B, pack_obs_and_next_obs=True)
            ref.np_random.bit_generator.state = st
            idx = ref.np_random.integers(len(ref), size=B)
            for i in range(B):
                while not ref._is_correct_index[idx[i]]:
                    idx[i] = ref.np_random.integers(len(ref))
            rec[f"idx_{s}"] = idx
            for k in keys:
                rec[f"frames_{k}_{s}"] = np.ascontiguousarray(b["observations"][k])
            rec[f"state_{s}"] = b["observations"]["state"]
            rec[f"next_state_{s}"] = b["next_observations"]["state"]
            for f in ("actions", "rewards", "masks", "dones"):
                rec[f"{f}_{s}"] = np.asarray(b[f])
        if name == "wrap_quirk":
            assert rec["valid"][0] and any((rec[f"idx_{s}"] == 0).any() for s in range(ns))
        np.savez_compressed(os.path.join(out_dir, f"replay_{name}.npz"), **rec)
        print(name, "ok", {k: v.shape for k, v in rec.items() if k.endswith("_0")})


PLAIN_CASES = {  # name: (S, A, capacity, n_insert, episode_len, stream_seed, rb_seed, batch, n_samples)
    "plain_wrap": (10, 4, 50, 130, 9, 21, 4, 32, 4),
    "plain_nowrap": (24, 6, 400, 123, 30, 22, 0, 64, 3),
}


def main_plain():
    Ref = ref_shim.load_reference_plain_buffer_cls()
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, (S, A, cap, n_ins, ep, sseed, rseed, B, ns) in PLAIN_CASES.items():
        ref = Ref(ref_shim.Box(-np.inf, np.inf, (S,), np.float32), ref_shim.Box(-1, 1, (A,), np.float32), cap)
        ref.seed(rseed)
        for tr in itertools.islice(flat_stream(S, A, ep, sseed), n_ins):
            ref.insert(tr)
        rec = {"meta": np.array([S, A, cap, n_ins, ep, sseed, rseed, B, ns], np.int64), "size": np.int64(len(ref)),
               "insert_index": np.int64(ref._insert_index)}
        for s_ in range(ns):
            st = ref.np_random.bit_generator.state
            b = ref.sample(B)
            ref.np_random.bit_generator.state = st
            rec[f"idx_{s_}"] = ref.np_random.integers(len(ref), size=B)     # dataset.py:85-89, same stream
            for f in ("observations", "next_observations", "actions", "rewards", "masks", "dones"):
                rec[f"{f}_{s_}"] = np.asarray(b[f])
        np.savez_compressed(os.path.join(out_dir, f"replay_{name}.npz"), **rec)
        print(name, "ok", {k: v.shape for k, v in rec.items() if k.endswith("_0")})


if __name__ == "__main__":
    main()
    main_plain()
