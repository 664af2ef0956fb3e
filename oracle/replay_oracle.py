"""TEST INFRASTRUCTURE ONLY -- CPU restatement (NumPy) of the reference replay path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product path (serl_amd/) never does and fails loudly without its HIP library.

Pinned: this restatement is checked bit-for-bit against the reference's own
``MemoryEfficientReplayBuffer`` executed unmodified (oracle/ref_shim.py) in
tests/test_replay_oracle.py, and against fixtures in tests/golden/replay_*.npz that were
generated from the reference by tests/golden/make_golden_replay.py.

Follows (reference file:line, relative to serl_launcher/serl_launcher/):
  data/replay_buffer.py:41-75          storage allocation, ring insert
  data/memory_efficient_replay_buffer.py:13-51   frame-per-slot layout, validity mask
  data/memory_efficient_replay_buffer.py:53-89   insert bookkeeping (wrap, first-frame slot)
  data/memory_efficient_replay_buffer.py:111-122 index draw + rejection loop
  data/memory_efficient_replay_buffer.py:126-164 gather of packed frame pairs
  data/dataset.py:66-74                seeding: Generator(PCG64(SeedSequence(seed)))
  utils/train_utils.py:16-31           concat_batches (online first, then demo)
  utils/train_utils.py:44-66           _unpack (obs=[:, :-1], next=[:, 1:])
  vision/data_augmentations.py:7-36    random shift crop = edge pad 4 + dynamic_slice
"""
from __future__ import annotations

import numpy as np


class ReplayOracle:
    def __init__(self, image_keys, H, W, C, T, S, A, capacity):
        self.image_keys = tuple(image_keys)
        self.T, self.S, self.A, self.cap = T, S, A, capacity
        self.frames = {k: np.zeros((capacity, H, W, C), np.uint8) for k in self.image_keys}
        self.state = np.zeros((capacity, T, S), np.float32)
        self.next_state = np.zeros((capacity, T, S), np.float32)
        self.actions = np.zeros((capacity, A), np.float32)
        self.rewards = np.zeros((capacity,), np.float32)
        self.masks = np.zeros((capacity,), np.float32)
        self.dones = np.zeros((capacity,), bool)
        self.valid = np.zeros((capacity,), bool)
        self.size = 0
        self.insert_index = 0
        self.first = True
        self.rng = None

    def __len__(self):
        return self.size

    def seed(self, seed):
        self.rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

    # -- replay_buffer.py:71-75
    def _raw_insert(self, frames, state, next_state, action, reward, mask, done):
        i = self.insert_index
        for k in self.image_keys:
            self.frames[k][i] = frames[k]
        self.state[i], self.next_state[i] = state, next_state
        self.actions[i], self.rewards[i], self.masks[i], self.dones[i] = action, reward, mask, done
        self.insert_index = (i + 1) % self.cap
        self.size = min(self.size + 1, self.cap)

    def _copy_slot_to_head(self, src):
        frames = {k: self.frames[k][src].copy() for k in self.image_keys}
        self._raw_insert(frames, self.state[src].copy(), self.next_state[src].copy(),
                         self.actions[src].copy(), self.rewards[src], self.masks[src], self.dones[src])

    # -- memory_efficient_replay_buffer.py:53-89
    def insert(self, tr):
        T = self.T
        if self.insert_index == 0 and self.cap == self.size and not self.first:
            for src in range(self.size - T, self.size):
                self.valid[self.insert_index] = False
                self._copy_slot_to_head(src)
        obs, nobs = tr["observations"], tr["next_observations"]
        common = (obs["state"], nobs["state"], tr["actions"], tr["rewards"], tr["masks"], tr["dones"])
        if self.first:
            for i in range(T):
                self.valid[self.insert_index] = False
                self._raw_insert({k: obs[k][i] for k in self.image_keys}, *common)
        self.first = bool(tr["dones"])
        self.valid[self.insert_index] = True
        self._raw_insert({k: nobs[k][-1] for k in self.image_keys}, *common)
        for i in range(T):
            self.valid[(self.insert_index + i) % self.size] = False

    # -- memory_efficient_replay_buffer.py:111-122
    def sample_indices(self, batch_size):
        n = self.size
        indx = self.rng.integers(n, size=batch_size)
        for i in range(batch_size):
            while not self.valid[indx[i]]:
                indx[i] = self.rng.integers(n)
        return indx

    # -- memory_efficient_replay_buffer.py:126-164 (pack_obs_and_next_obs=True)
    def gather(self, indx):
        T = self.T
        out = {
            "observations": {"state": self.state[indx]},
            "next_observations": {"state": self.next_state[indx]},
            "actions": self.actions[indx], "rewards": self.rewards[indx],
            "masks": self.masks[indx], "dones": self.dones[indx],
        }
        for k in self.image_keys:
            # window j covers slots j..j+T (cap-T windows over the WHOLE capacity array); the
            # sample for slot i is window i-T.  Reference quirk kept bit-for-bit: a valid slot
            # i < T (episode whose first-frame slot was the last slot of the ring) indexes window
            # i-T < 0, which numpy wraps to window cap-T+(i-T), i.e. slots cap-2T+i.. (py:149-153).
            start = indx - T
            start = np.where(start < 0, start + (self.cap - T), start)
            win = np.stack([self.frames[k][start + t] for t in range(T + 1)], axis=1)
            out["observations"][k] = win  # (B, T+1, H, W, C)
        return out

    def sample(self, batch_size):
        return self.gather(self.sample_indices(batch_size))


def concat_batches(a, b):
    """train_utils.py:16-31 with axis=0 (async_drq_sim.py:277): first argument first."""
    out = {}
    for k, v in a.items():
        out[k] = concat_batches(v, b[k]) if isinstance(v, dict) else np.concatenate((v, b[k]), axis=0)
    return out


def unpack(batch, image_keys):
    """train_utils.py:44-66."""
    obs = dict(batch["observations"])
    nobs = dict(batch["next_observations"])
    for k in image_keys:
        if k not in nobs:
            packed = batch["observations"][k]
            obs[k] = packed[:, :-1]
            nobs[k] = packed[:, 1:]
    out = dict(batch)
    out["observations"], out["next_observations"] = obs, nobs
    return out


def random_shift(img, offsets, padding=4):
    """data_augmentations.py:7-36 with explicit offsets (dy,dx) in [0, 2*padding].

    img: (N, H, W, C) uint8, offsets: (N, 2) int.  out[n,h,w] = pad_edge(img[n])[h+dy, w+dx].
    """
    N, H, W, C = img.shape
    out = np.empty_like(img)
    hh = np.arange(H)[None, :]
    ww = np.arange(W)[None, :]
    ys = np.clip(hh + offsets[:, 0:1] - padding, 0, H - 1)  # (N,H)
    xs = np.clip(ww + offsets[:, 1:2] - padding, 0, W - 1)  # (N,W)
    for n in range(N):
        out[n] = img[n][ys[n]][:, xs[n]]
    return out


# ---- PCG64 / Lemire restatement -- pins the product's C++ index sampler
# independently of numpy's own implementation (tests compare all three).
_MULT = 0x2360ED051FC65DA44385DF649FCCF645
_M128 = (1 << 128) - 1


class PCG64Py:
    def __init__(self, state, inc, has_uint32=0, uinteger=0):
        self.state, self.inc, self.has_uint32, self.uinteger = state, inc, has_uint32, uinteger

    @classmethod
    def from_numpy(cls, gen):
        st = gen.bit_generator.state
        return cls(st["state"]["state"], st["state"]["inc"], st["has_uint32"], st["uinteger"])

    def next64(self):
        self.state = (self.state * _MULT + self.inc) & _M128
        hi, lo = self.state >> 64, self.state & ((1 << 64) - 1)
        x, r = hi ^ lo, self.state >> 122
        return ((x >> r) | (x << ((64 - r) & 63))) & ((1 << 64) - 1)

    def next32(self):
        if self.has_uint32:
            self.has_uint32 = 0
            return self.uinteger
        v = self.next64()
        self.has_uint32, self.uinteger = 1, v >> 32
        return v & 0xFFFFFFFF

    def bounded(self, n):
        """Generator.integers(n) for 0 < n <= 2**32 - 1 (Lemire, 32-bit path)."""
        if n == 1:
            return 0
        m = self.next32() * n
        l = m & 0xFFFFFFFF
        if l < n:
            t = (0xFFFFFFFF - (n - 1)) % n
            while l < t:
                m = self.next32() * n
                l = m & 0xFFFFFFFF
        return m >> 32


class PlainReplayOracle:
    """ReplayBuffer of flat observations (replay_buffer.py:40-75) + Dataset.sample (dataset.py:79-102):
    insert writes at the head and advances; sample draws `integers(len, size=B)` -- no validity mask."""

    def __init__(self, S, A, capacity):
        self.obs = np.zeros((capacity, S), np.float32)
        self.next_obs = np.zeros((capacity, S), np.float32)
        self.actions = np.zeros((capacity, A), np.float32)
        self.rewards = np.zeros((capacity,), np.float32)
        self.masks = np.zeros((capacity,), np.float32)
        self.dones = np.zeros((capacity,), bool)
        self.cap, self.size, self.insert_index, self.rng = capacity, 0, 0, None

    def __len__(self):
        return self.size

    def seed(self, seed):
        self.rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

    def insert(self, d):  # replay_buffer.py:71-75
        i = self.insert_index
        self.obs[i], self.next_obs[i] = d["observations"], d["next_observations"]
        self.actions[i], self.rewards[i], self.masks[i], self.dones[i] = d["actions"], d["rewards"], d["masks"], d["dones"]
        self.insert_index = (i + 1) % self.cap
        self.size = min(self.size + 1, self.cap)

    def sample_indices(self, batch_size):  # dataset.py:85-89
        return self.rng.integers(len(self), size=batch_size)

    def gather(self, idx):
        return {"observations": self.obs[idx], "next_observations": self.next_obs[idx], "actions": self.actions[idx],
                "rewards": self.rewards[idx], "masks": self.masks[idx], "dones": self.dones[idx]}
