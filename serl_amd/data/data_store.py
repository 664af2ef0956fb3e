"""HBM-resident replay buffer with the reference's data-store interface.

Mirrors (same names, argument meaning and error behaviour):
  serl_launcher/data/memory_efficient_replay_buffer.py:12-164  MemoryEfficientReplayBuffer
  serl_launcher/data/replay_buffer.py:77-90                    get_iterator (prefetch queue)
  serl_launcher/data/data_store.py:83-144                      MemoryEfficientReplayBufferDataStore
  serl_launcher/data/dataset.py:66-74                          seed / np_random
  serl_launcher/utils/train_utils.py:16-31                     concat_batches
  serl_launcher/data/data_store.py:147-196                     populate_data_store, populate_data_store_with_z_axis_only

Storage and sampling live in libserl_mi355.so (serl_amd/csrc/replay.hip); this file only adapts
Python dicts to the C ABI.  There is no CPU fallback.
"""
from __future__ import annotations

import collections
import ctypes as C
from typing import Iterable, Optional

import numpy as np
import torch

from .. import _lib
from ..transport.endpoint import DataStoreBase


def _space_shape(space):
    return tuple(space.shape)


class LazyBatch:
    """A sampled-but-not-yet-gathered batch: (buffer, slot indices) parts in concat order.

    The agent's update turns it into device tensors with ONE fused gather+concat+unpack+crop
    kernel (serl_rb_gather_crop).  `materialize()` gives the reference's packed dict instead.
    """

    def __init__(self, parts, peek_next=None):
        self.parts = list(parts)  # [(buffer, np.int64[n])]
        # peek_next() -> the LazyBatch the iterator will yield next (already sampled by its prefetch queue), or None.
        # The agent uses it to run gather + crop + frozen trunk of the NEXT batch on a second stream under the update
        # of this one (the reference's iterator prefetches two batches to the device for the same reason).
        self.peek_next = peek_next

    @property
    def batch_size(self):
        return sum(len(ix) for _, ix in self.parts)

    def materialize(self):
        out = None
        for buf, ix in self.parts:
            d = buf.gather(ix)
            out = d if out is None else concat_batches(out, d, axis=0)
        return out


def _after(queue, me):
    """The batch the iterator yields after `me` if it is already in the prefetch queue."""
    return queue[0] if queue and queue[0] is not me else (queue[1] if len(queue) > 1 and queue[0] is me else None)


def concat_batches(offline_batch, online_batch, axis=1):
    """train_utils.py:16-31 (first argument first).  Works on dict batches of torch tensors and
    on LazyBatch (axis must be 0 there)."""
    if isinstance(offline_batch, LazyBatch) or isinstance(online_batch, LazyBatch):
        assert isinstance(offline_batch, LazyBatch) and isinstance(online_batch, LazyBatch)
        assert axis == 0, "lazy batches concatenate along the batch axis only"
        a, b = offline_batch, online_batch

        def peek():
            if a.peek_next is None or b.peek_next is None:
                return None
            na, nb = a.peek_next(), b.peek_next()
            if na is None or nb is None:
                return None
            if getattr(peek, "_cache", (None, None, None))[:2] != (id(na), id(nb)):
                peek._cache = (id(na), id(nb), concat_batches(na, nb, axis=0))
            return peek._cache[2]
        return LazyBatch(a.parts + b.parts, peek)
    batch = {}
    for k, v in offline_batch.items():
        if isinstance(v, dict):
            batch[k] = concat_batches(v, online_batch[k], axis=axis)
        else:
            batch[k] = torch.cat((v, online_batch[k]), dim=axis)
    return batch


class MemoryEfficientReplayBufferDataStore(DataStoreBase):
    """Drop-in for the reference class of the same name (data_store.py:83-144), an agentlace `DataStoreBase`: a
    TrainerServer thread may call insert() / batch_insert() while the learner thread samples (the buffer mutex lives in
    libserl_mi355.so)."""

    def __init__(self, observation_space, action_space, capacity: int,
                 image_keys: Iterable[str] = ("image",), rlds_logger=None, device: int = 0):
        if rlds_logger is not None:
            raise NotImplementedError("RLDS logging is outside the MI355X hot path")
        self.pixel_keys = tuple(image_keys)
        spaces = observation_space.spaces
        self._num_stack = None
        for k in self.pixel_keys:
            shp = _space_shape(spaces[k])
            if self._num_stack is None:
                self._num_stack = shp[0]
            else:
                assert self._num_stack == shp[0]
            self._img_shape = shp[1:]
        other = [k for k in spaces.keys() if k not in self.pixel_keys]
        if other != ["state"]:
            raise NotImplementedError(f"non-pixel observation keys must be exactly ['state'], got {other}")
        sshape = _space_shape(spaces["state"])
        assert len(sshape) == 2 and sshape[0] == self._num_stack, sshape
        self._S = sshape[1]
        self._A = _space_shape(action_space)[0]
        self._capacity = int(capacity)
        DataStoreBase.__init__(self, self._capacity)
        self.device = device
        self._torch_device = torch.device("cuda", device)
        H, W, Cc = self._img_shape
        self._h = C.c_void_p()
        _lib.check(_lib.lib().serl_rb_create(device, self._capacity, len(self.pixel_keys), H, W, Cc,
                                             self._num_stack, self._S, self._A, C.byref(self._h)))
        self._np_random = None
        self._seed = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                _lib.lib().serl_rb_destroy(h)
            except Exception:
                pass
            self._h = C.c_void_p()

    # -- Dataset.seed / np_random (dataset.py:66-74)
    def seed(self, seed: Optional[int] = None) -> list:
        ss = np.random.SeedSequence(seed)
        self._np_random = np.random.Generator(np.random.PCG64(ss))
        self._seed = ss.entropy
        st = self._np_random.bit_generator.state
        s, inc = st["state"]["state"], st["state"]["inc"]
        m = (1 << 64) - 1
        _lib.check(_lib.lib().serl_rb_seed(self._h, s >> 64, s & m, inc >> 64, inc & m,
                                           st["has_uint32"], st["uinteger"]))
        return [self._seed]

    def _ensure_seeded(self):
        if self._seed is None:
            self.seed()  # lazy OS-entropy seed like the reference (dataset.py:66-70)

    def __len__(self) -> int:
        return int(_lib.lib().serl_rb_len(self._h))

    def latest_data_id(self):  # data_store.py:138-140
        return int(_lib.lib().serl_rb_insert_index(self._h))

    def get_latest_data(self, from_id: int):  # data_store.py:142-144
        raise NotImplementedError  # same as the reference

    def valid_mask(self) -> np.ndarray:
        out = np.zeros(self._capacity, np.uint8)
        _lib.check(_lib.lib().serl_rb_valid_mask(self._h, out.ctypes.data))
        return out.astype(bool)

    # -- insert (memory_efficient_replay_buffer.py:53-89; thread-safe, data_store.py:104-106)
    def insert(self, data):   # (the reference's data stores name the transition `data`, data_store.py:104)
        data_dict = data
        obs, nobs = data_dict["observations"], data_dict["next_observations"]
        n = len(self.pixel_keys)
        keep = []
        obs_p, next_p = (C.c_void_p * n)(), (C.c_void_p * n)()
        T = self._num_stack
        for i, k in enumerate(self.pixel_keys):
            a = np.ascontiguousarray(obs[k], dtype=np.uint8)
            b = np.ascontiguousarray(nobs[k], dtype=np.uint8)
            assert a.shape == (T,) + tuple(self._img_shape), (k, a.shape)
            assert b.shape == a.shape, (k, b.shape)
            keep += [a, b]
            obs_p[i], next_p[i] = a.ctypes.data, b.ctypes.data
        st = np.ascontiguousarray(obs["state"], dtype=np.float32).reshape(-1)
        nst = np.ascontiguousarray(nobs["state"], dtype=np.float32).reshape(-1)
        act = np.ascontiguousarray(data_dict["actions"], dtype=np.float32).reshape(-1)
        assert st.size == T * self._S and nst.size == T * self._S and act.size == self._A
        _lib.check(_lib.lib().serl_rb_insert(
            self._h, obs_p, next_p, st.ctypes.data, nst.ctypes.data, act.ctypes.data,
            float(data_dict["rewards"]), float(data_dict["masks"]), int(bool(data_dict["dones"]))))

    # -- index draw (memory_efficient_replay_buffer.py:111-122)
    def sample_indices(self, batch_size: int) -> np.ndarray:
        self._ensure_seeded()
        idx = np.empty(batch_size, np.int64)
        _lib.check(_lib.lib().serl_rb_sample_indices(self._h, batch_size, idx.ctypes.data))
        return idx

    # -- gather (memory_efficient_replay_buffer.py:126-164), packed frames on the device
    def gather(self, indx: np.ndarray, stream=None):
        indx = np.ascontiguousarray(indx, dtype=np.int64)   # (stale indices are re-drawn in place by the library)
        if not indx.flags.writeable:
            indx = indx.copy()
        B, T = len(indx), self._num_stack
        H, W, Cc = self._img_shape
        dev = self._torch_device
        frames = {k: torch.empty((B, T + 1, H, W, Cc), dtype=torch.uint8, device=dev) for k in self.pixel_keys}
        st = torch.empty((B, T, self._S), dtype=torch.float32, device=dev)
        nst = torch.empty_like(st)
        act = torch.empty((B, self._A), dtype=torch.float32, device=dev)
        rew = torch.empty((B,), dtype=torch.float32, device=dev)
        msk = torch.empty((B,), dtype=torch.float32, device=dev)
        done = torch.empty((B,), dtype=torch.uint8, device=dev)
        fp = (C.c_void_p * len(self.pixel_keys))(*[frames[k].data_ptr() for k in self.pixel_keys])
        s = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        _lib.check(_lib.lib().serl_rb_gather_packed(
            self._h, indx.ctypes.data, B, fp, st.data_ptr(), nst.data_ptr(), act.data_ptr(),
            rew.data_ptr(), msk.data_ptr(), done.data_ptr(), C.c_void_p(s)))
        obs = {"state": st}
        obs.update(frames)
        return {"observations": obs, "next_observations": {"state": nst}, "actions": act,
                "rewards": rew, "masks": msk, "dones": done.bool()}

    def sample(self, batch_size: int, keys=None, indx=None, pack_obs_and_next_obs: bool = False,
               lazy: bool = False):
        if indx is not None:
            raise NotImplementedError()  # memory_efficient_replay_buffer.py:123-124
        if keys is not None:
            raise NotImplementedError("key sub-selection is not used on the hot path")
        idx = self.sample_indices(batch_size)
        if lazy:
            return LazyBatch([(self, idx)])
        batch = self.gather(idx)
        if not pack_obs_and_next_obs:  # memory_efficient_replay_buffer.py:159-162
            for k in self.pixel_keys:
                packed = batch["observations"][k]
                batch["observations"][k] = packed[:, :-1]
                batch["next_observations"][k] = packed[:, 1:]
        return batch

    # -- replay_buffer.py:77-90: depth-`queue_size` prefetch; `device` is accepted and ignored
    # (data never leaves HBM)
    def get_iterator(self, queue_size: int = 2, sample_args: dict = {}, device=None):
        queue = collections.deque()

        def enqueue(n):
            for _ in range(n):
                b = self.sample(**sample_args)
                if isinstance(b, LazyBatch):
                    b.peek_next = lambda me=b: _after(queue, me)
                queue.append(b)

        enqueue(queue_size)
        while queue:
            yield queue.popleft()
            enqueue(1)

    @property
    def handle(self):
        return self._h


class ReplayBufferDataStore(MemoryEfficientReplayBufferDataStore):
    """Drop-in for the reference's plain `ReplayBufferDataStore` (data_store.py:27-80) of flat observations
    (ReplayBuffer, replay_buffer.py:41-75: every inserted slot is valid, `Dataset.sample` draws
    `integers(len, size=B)` with no rejection) -- the store of BASELINE.json configs[0] `async_sac_state_sim`."""

    def __init__(self, observation_space, action_space, capacity: int, rlds_logger=None, device: int = 0):
        if rlds_logger is not None:
            raise NotImplementedError("RLDS logging is outside the MI355X hot path")
        if hasattr(observation_space, "spaces"):
            raise TypeError("ReplayBufferDataStore holds flat (Box) observations; use the memory-efficient store for pixels")
        self.pixel_keys = ()
        self._num_stack = 1
        self._img_shape = (0, 0, 0)
        self._S = int(np.prod(_space_shape(observation_space)))
        self._A = _space_shape(action_space)[0]
        self._capacity = int(capacity)
        DataStoreBase.__init__(self, self._capacity)
        self.device = device
        self._torch_device = torch.device("cuda", device)
        self._h = C.c_void_p()
        _lib.check(_lib.lib().serl_rb_create(device, self._capacity, 0, 0, 0, 0, 1, self._S, self._A, C.byref(self._h)))
        self._np_random = None
        self._seed = None

    def insert(self, data):  # replay_buffer.py:71-75 under the data store's lock (data_store.py:44-46)
        data_dict = data
        st = np.ascontiguousarray(data_dict["observations"], dtype=np.float32).reshape(-1)
        nst = np.ascontiguousarray(data_dict["next_observations"], dtype=np.float32).reshape(-1)
        act = np.ascontiguousarray(data_dict["actions"], dtype=np.float32).reshape(-1)
        assert st.size == self._S and nst.size == self._S and act.size == self._A
        _lib.check(_lib.lib().serl_rb_insert(
            self._h, None, None, st.ctypes.data, nst.ctypes.data, act.ctypes.data,
            float(data_dict["rewards"]), float(data_dict["masks"]), int(bool(data_dict["dones"]))))

    def gather(self, indx: np.ndarray, stream=None):
        indx = np.ascontiguousarray(indx, dtype=np.int64)
        B, dev = len(indx), self._torch_device
        st = torch.empty((B, self._S), dtype=torch.float32, device=dev)
        nst = torch.empty_like(st)
        act = torch.empty((B, self._A), dtype=torch.float32, device=dev)
        rew = torch.empty((B,), dtype=torch.float32, device=dev)
        msk = torch.empty((B,), dtype=torch.float32, device=dev)
        done = torch.empty((B,), dtype=torch.uint8, device=dev)
        s = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        _lib.check(_lib.lib().serl_rb_gather_packed(
            self._h, indx.ctypes.data, B, None, st.data_ptr(), nst.data_ptr(), act.data_ptr(),
            rew.data_ptr(), msk.data_ptr(), done.data_ptr(), C.c_void_p(s)))
        return {"observations": st, "next_observations": nst, "actions": act, "rewards": rew, "masks": msk,
                "dones": done.bool()}

    def sample(self, batch_size: int, keys=None, indx=None, lazy: bool = False, **kwargs):
        """Dataset.sample (dataset.py:79-102): `indx` may be given (used by `download`)."""
        if keys is not None:
            raise NotImplementedError("key sub-selection is not used on the hot path")
        idx = self.sample_indices(batch_size) if indx is None else np.asarray(indx, np.int64)
        return LazyBatch([(self, idx)]) if lazy else self.gather(idx)


def gather_crop(parts, crop_obs: Optional[np.ndarray], crop_next: Optional[np.ndarray], out, stream=None):
    """Fused K2+K3+K4 into a DeviceBatch `out` (serl_amd.agents.batch.DeviceBatch)."""
    n = len(parts)
    assert 1 <= n <= _lib.MAX_BUFFERS
    rbs = (C.c_void_p * n)(*[p[0].handle.value for p in parts])
    # the library re-draws, IN PLACE, indices whose slot an insert invalidated since they were drawn: pass the callers'
    # own arrays whenever possible so that LazyBatch.parts keep describing the gathered batch
    idxs = [p[1] if (isinstance(p[1], np.ndarray) and p[1].dtype == np.int64 and p[1].flags.c_contiguous and p[1].flags.writeable)
            else np.ascontiguousarray(p[1], dtype=np.int64).copy() for p in parts]
    idp = (C.c_void_p * n)(*[ix.ctypes.data for ix in idxs])
    counts = (C.c_int * n)(*[len(ix) for ix in idxs])
    co = None if crop_obs is None else np.ascontiguousarray(crop_obs, dtype=np.int32)
    cn = None if crop_next is None else np.ascontiguousarray(crop_next, dtype=np.int32)
    dev = parts[0][0]._torch_device
    s = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
    _lib.check(_lib.lib().serl_rb_gather_crop(
        rbs, n, idp, counts, None if co is None else co.ctypes.data,
        None if cn is None else cn.ctypes.data, C.byref(out.cstruct), C.c_void_p(s)))


def _demo_transitions(demos_path):
    """Transitions of the pickled demonstration files (each file: a list of transition dicts, the format the reference's
    record_demo scripts write).  `demos_path` is a LIST of paths, as at the reference's call sites
    (examples/bc_policy.py:152-156); a single string is accepted as one path instead of being iterated per character."""
    import pickle
    paths = [demos_path] if isinstance(demos_path, (str, bytes)) else list(demos_path)
    for path in paths:
        with open(path, "rb") as f:
            demo = pickle.load(f)
        for transition in demo:
            yield transition


def populate_data_store(data_store, demos_path):
    """data_store.py:147-163: insert every transition of the demonstration pickles into `data_store` (any object with
    insert() and __len__: the HBM stores above or a QueuedDataStore) and return it."""
    for transition in _demo_transitions(demos_path):
        data_store.insert(transition)
    print(f"Loaded {len(data_store)} transitions.")
    return data_store


def _z_axis_only(state: np.ndarray) -> np.ndarray:
    """data_store.py:180-196: drop the x / y Cartesian components of a [T, D] proprio state -- columns 0..3, column 6 and
    columns 10.. are kept (the reference concatenates state[:, :4], state[:, 6][None] and state[:, 10:])."""
    state = np.asarray(state)
    return np.concatenate((state[:, :4], state[:, 6][None, ...], state[:, 10:]), axis=-1)


def populate_data_store_with_z_axis_only(data_store, demos_path):
    """data_store.py:166-196: as populate_data_store, with the x / y coordinates removed from both observations' state
    (the caller's transitions are not modified)."""
    for transition in _demo_transitions(demos_path):
        t = dict(transition)
        for side in ("observations", "next_observations"):
            t[side] = dict(transition[side])
            t[side]["state"] = _z_axis_only(transition[side]["state"])
        data_store.insert(t)
    print(f"Loaded {len(data_store)} transitions.")
    return data_store
