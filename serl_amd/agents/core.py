"""Thin Python handle around the C ABI's `serl_agent` (no numerics here)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from .. import _lib
from .._lib_agent import NET_BITS, TX_INDEX, SerlAgentCfg, SerlInfo, SerlNoise

APPLY_CRITIC, APPLY_ACTOR_TEMP = 1, 6   # SERL_NET_CRITIC, SERL_NET_ACTOR | SERL_NET_TEMPERATURE
TX_NAMES = ("actor", "critic", "temperature")


class AgentCore:
    def __init__(self, *, device=0, n_cam, H, W, state_dim, act_dim, batch, ensemble=10, hidden=256,
                 bottleneck=256, sle_features=8, proprio_dim=64, warmup_steps=0, discount=0.96,
                 tau=0.005, lr=3e-4, dropout=0.1, std_min=1e-5, std_max=5.0, target_entropy=None,
                 seed=0, temp_warmup_steps=-1, optimizers=None, encoder_type="resnet-pretrained",
                 critic_subsample_size=2, backup_entropy=False):
        """optimizers: optional {"actor"|"critic"|"temperature": make_optimizer kwargs (common/optimizers.py:6-13:
        learning_rate, warmup_steps, cosine_decay_steps, weight_decay, clip_grad_norm)} overriding lr / warmup_steps."""
        if target_entropy is None:
            target_entropy = -act_dim / 2
        self.cfg = SerlAgentCfg(device, n_cam, H, W, state_dim, act_dim, batch, ensemble, hidden,
                                bottleneck, sle_features, proprio_dim, warmup_steps, temp_warmup_steps, discount, tau,
                                lr, dropout, std_min, std_max, target_entropy, seed)
        self.cfg.encoder_type = {"resnet-pretrained": 0, "small": 1}[encoder_type]
        # sac.py:150-161: None = the minimum runs over the whole target ensemble
        if critic_subsample_size is not None and not (1 <= int(critic_subsample_size) <= 16):
            raise ValueError(f"critic_subsample_size must be None or in [1, 16] (got {critic_subsample_size})")
        self.cfg.critic_subsample_size = -1 if critic_subsample_size is None else int(critic_subsample_size)
        self.cfg.backup_entropy = 1 if backup_entropy else 0
        for name, kw in (optimizers or {}).items():
            i = TX_INDEX[name]
            bad = set(kw) - {"learning_rate", "warmup_steps", "cosine_decay_steps", "weight_decay", "clip_grad_norm"}
            if bad:
                raise TypeError(f"make_optimizer() got unexpected keyword arguments {sorted(bad)}")
            if kw.get("learning_rate") is not None:
                self.cfg.tx_lr[i], self.cfg.tx_lr_set[i] = float(kw["learning_rate"]), 1
            if kw.get("warmup_steps") is not None:
                self.cfg.tx_warmup[i] = int(kw["warmup_steps"]) + 1
            if kw.get("cosine_decay_steps") is not None:
                self.cfg.tx_cosine_steps[i] = int(kw["cosine_decay_steps"])
            if kw.get("weight_decay") is not None:
                self.cfg.tx_weight_decay_on[i], self.cfg.tx_weight_decay[i] = 1, float(kw["weight_decay"])
            if kw.get("clip_grad_norm") is not None:
                self.cfg.tx_clip_norm[i] = float(kw["clip_grad_norm"])
        self._h = C.c_void_p()
        self.L = _lib.lib()
        _lib.check(self.L.serl_agent_create(C.byref(self.cfg), C.byref(self._h)))
        self.device = torch.device("cuda", device)
        self.leaves = {}
        buf = C.create_string_buffer(128)
        cnt = C.c_int64()
        for i in range(self.L.serl_agent_num_leaves(self._h)):
            _lib.check(self.L.serl_agent_leaf_info(self._h, i, buf, 128, C.byref(cnt)))
            self.leaves[buf.value.decode()] = int(cnt.value)
        self._keep = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self.L.serl_agent_destroy(h)
            except Exception:
                pass
            self._h = C.c_void_p()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- parameters ------------------------------------------------------------------------
    def set(self, section: str, leaf: str, value):
        v = np.ascontiguousarray(np.asarray(value, dtype=np.float32).reshape(-1))
        _lib.check(self.L.serl_agent_set(self._h, section.encode(), leaf.encode(), v.ctypes.data, v.size))

    def get(self, section: str, leaf: str) -> np.ndarray:
        out = np.empty(self.leaves[leaf], np.float32)
        _lib.check(self.L.serl_agent_get(self._h, section.encode(), leaf.encode(), out.ctypes.data, out.size))
        return out

    def set_trunk_mode(self, mode: str):
        """'f32' = exact fp32 MFMA convs, 'f16x3' = split-fp16 convs (default)."""
        _lib.check(self.L.serl_agent_set_trunk_mode(self._h, {"f32": 0, "f16x3": 1}[mode]))

    def set_chain_budget(self, workgroups: int):
        """Scheduling hint (no effect on results): workgroup budget of the update's K-split GEMM launches, 0 = default."""
        _lib.check(self.L.serl_agent_set_chain_budget(self._h, int(workgroups)))

    def load_flat(self, section: str, tree: Dict[str, np.ndarray]):
        for k, v in tree.items():
            self.set(section, k, v)

    @property
    def step(self):
        return int(self.L.serl_agent_get_step(self._h))

    @step.setter
    def step(self, v):
        _lib.check(self.L.serl_agent_set_step(self._h, int(v)))

    # ---- updates ---------------------------------------------------------------------------
    def _noise(self, noise: Optional[dict]):
        """noise: dict of torch device tensors (eps_*, mask_*), np.int32 redq_idx and / or jax.random keys (key_eps_* uint32[..., 2],
        key_mask_* uint32[..., n_cam, 2]: draws made inside the consuming kernels where the tensor is absent), or None."""
        if noise is None:
            self._keep = None
            return None
        keep = []

        def dev(name, dtype):
            t = noise.get(name)
            if t is None:
                return None
            t = t.to(device=self.device, dtype=dtype).contiguous()
            keep.append(t)
            return t.data_ptr()

        redq = noise.get("redq_idx")
        rp = None
        if redq is not None:
            redq = np.ascontiguousarray(redq, dtype=np.int32)
            keep.append(redq)
            rp = redq.ctypes.data
        def key(name):      # jax.random keys of draws whose tensor is absent (host uint32 words, kept alive with the struct)
            k = noise.get(name)
            if k is None:
                return None
            k = np.ascontiguousarray(k, dtype=np.uint32)
            keep.append(k)
            return k.ctypes.data

        n = SerlNoise(dev("eps_next", torch.float32), dev("mask_next", torch.uint8), rp,
                      dev("eps_pi", torch.float32), dev("mask_obs_pi", torch.uint8),
                      dev("eps_temp", torch.float32), dev("mask_next_temp", torch.uint8),
                      key("key_eps_next"), key("key_mask_next"), key("key_eps_pi"), key("key_mask_obs_pi"),
                      key("key_eps_temp"), key("key_mask_next_temp"))
        self._keep = (keep, n)
        return C.byref(n)

    def update_critics(self, batch, noise=None):
        _lib.check(self.L.serl_agent_update_critics(self._h, C.byref(batch.cstruct), self._noise(noise), self._stream()))

    def update_high_utd(self, batch, utd_ratio=1, noise=None):
        _lib.check(self.L.serl_agent_update_high_utd(self._h, C.byref(batch.cstruct), utd_ratio,
                                                     self._noise(noise), self._stream()))

    def update(self, batch, networks=("actor", "critic", "temperature"), noise=None):
        """SACAgent.update (sac.py:243-299): every selected loss at the same parameters, one optimizer step."""
        bits = 0
        for n in networks:
            bits |= NET_BITS[n]
        _lib.check(self.L.serl_agent_update(self._h, C.byref(batch.cstruct), bits, self._noise(noise), self._stream()))

    def read_info(self) -> dict:
        info = SerlInfo()
        _lib.check(self.L.serl_agent_read_info(self._h, C.byref(info), self._stream()))
        return {n: getattr(info, n) for n, _ in SerlInfo._fields_}

    # ---- data-parallel phases --------------------------------------------------------------
    def set_shard(self, global_offset: int, global_batch: int):
        """This agent's batches are rows [global_offset, ...) of a global batch: device noise is indexed globally."""
        _lib.check(self.L.serl_agent_set_shard(self._h, int(global_offset), int(global_batch)))

    def begin_update(self):
        _lib.check(self.L.serl_agent_begin_update(self._h, self._stream()))

    def encode(self, batch):
        _lib.check(self.L.serl_agent_encode(self._h, C.byref(batch.cstruct), self._stream()))

    def encode_slot(self, batch, slot):
        _lib.check(self.L.serl_agent_encode_slot(self._h, C.byref(batch.cstruct), slot, self._stream()))

    def encode_slot_range(self, batch, slot, stage_begin, stage_end):
        """A piece of encode_slot: stages [stage_begin, stage_end], -1 = conv_init + max-pool, 0..3 = residual stages."""
        _lib.check(self.L.serl_agent_encode_slot_range(self._h, C.byref(batch.cstruct), slot, stage_begin, stage_end, self._stream()))

    def select_slot(self, slot):
        _lib.check(self.L.serl_agent_select_slot(self._h, slot))

    def bind_slot(self, batch, slot):
        """attach a batch to a pipeline slot without running the trunk (its features arrive from another GPU: slot_features)"""
        _lib.check(self.L.serl_agent_bind_slot(self._h, C.byref(batch.cstruct), slot))

    def slot_features(self, slot) -> torch.Tensor:
        """zero-copy view of the slot's frozen-trunk features, f32[2 * n_cam * batch * h*w * 512]"""
        p, n = C.c_void_p(), C.c_int64()
        _lib.check(self.L.serl_agent_slot_features(self._h, slot, C.byref(p), C.byref(n)))
        return _wrap_device_f32(p.value, int(n.value), self.device)

    def critic_grads(self, offset, count, global_count, noise=None, redq_row=0):
        _lib.check(self.L.serl_agent_critic_grads(self._h, offset, count, global_count, self._noise(noise),
                                                  redq_row, self._stream()))

    def critic_grads_bucketed(self, offset, count, global_count, noise=None, redq_row=0, event=None):
        """critic_grads with bucket 0 ([ensemble | head | proprio | scalars]) published early: `event` (torch.cuda.Event) is
        recorded on the current stream once that bucket is final, before the encoder-head backward is issued."""
        ev = None
        if event is not None:
            if not event.cuda_event:        # lazily created by torch: materialise the hipEvent_t handle
                event.record(torch.cuda.current_stream(self.device))
            ev = C.c_void_p(event.cuda_event)
        _lib.check(self.L.serl_agent_critic_grads_bucketed(self._h, offset, count, global_count, self._noise(noise),
                                                           redq_row, self._stream(), ev))

    def grad_bucket(self, bucket) -> torch.Tensor:
        """Zero-copy view of gradient bucket 0 / 1 (see critic_grads_bucketed); bucket 1 is empty for state-only agents."""
        p, n = C.c_void_p(), C.c_int64()
        _lib.check(self.L.serl_agent_grad_bucket(self._h, bucket, C.byref(p), C.byref(n)))
        return _wrap_device_f32(p.value, int(n.value), self.device) if n.value else torch.empty(0, device=self.device)

    def actor_grads(self, global_count, noise=None):
        _lib.check(self.L.serl_agent_actor_grads(self._h, global_count, self._noise(noise), self._stream()))

    def apply(self, which, info_weight=1.0):
        _lib.check(self.L.serl_agent_apply(self._h, which, info_weight, self._stream()))

    def grad_view(self, which) -> torch.Tensor:
        """The contiguous [grads | scalars] (critic) or [scalars | grads] (actor) device range as a
        torch tensor (zero-copy) for torch.distributed.all_reduce."""
        p, n = C.c_void_p(), C.c_int64()
        _lib.check(self.L.serl_agent_grad_view(self._h, which, C.byref(p), C.byref(n)))
        return _wrap_device_f32(p.value, int(n.value), self.device)

    # ---- misc --------------------------------------------------------------------------------
    def trunk_forward(self, frames_u8: torch.Tensor) -> torch.Tensor:
        n = frames_u8.shape[0]
        H, W = self.cfg.H, self.cfg.W
        fh = fw = None
        h, w = H, W
        for _ in range(5):
            h, w = (h + 1) // 2, (w + 1) // 2
        out = torch.empty((n, h, w, 512), dtype=torch.float32, device=self.device)
        _lib.check(self.L.serl_agent_trunk_forward(self._h, frames_u8.contiguous().data_ptr(), n,
                                                   out.data_ptr(), self._stream()))
        return out

    def sample_actions(self, frames_u8, state, eps=None):
        n = state.shape[0]
        out = torch.empty((n, self.cfg.act_dim), dtype=torch.float32, device=self.device)
        _lib.check(self.L.serl_agent_sample_actions(
            self._h, None if frames_u8 is None else frames_u8.contiguous().data_ptr(), state.contiguous().data_ptr(), n,
            None if eps is None else eps.contiguous().data_ptr(), out.data_ptr(), self._stream()))
        return out

    def debug(self, what: str, count: int) -> np.ndarray:
        out = np.empty(count, np.float32)
        _lib.check(self.L.serl_agent_debug_get(self._h, what.encode(), out.ctypes.data, count))
        return out


    def trunk_plan(self) -> dict:
        """Kernels the last split-fp16 trunk pass selected: {"images", "pool", "raw_b0", "<layer>": (kernel, cfg, pmode, fused)}."""
        buf = C.create_string_buffer(512)
        _lib.check(self.L.serl_agent_trunk_plan(self._h, buf, 512))
        out = {}
        for tok in buf.value.decode().split():
            k, v = tok.split("=")
            if "/" in v:
                kern, cfg, pm, fz = v.split("/")
                fz, _, ks = fz.partition("k")      # "f<fused>[k<K-split>]"
                out[k] = (kern, int(cfg), int(pm), int(fz[1:])) + ((int(ks),) if ks else ())
            else:
                out[k] = int(v)
        return out

    def debug_set(self, what: str, value):
        v = np.ascontiguousarray(np.asarray(value, dtype=np.float32).reshape(-1))
        _lib.check(self.L.serl_agent_debug_set(self._h, what.encode(), v.ctypes.data, v.size))


class _DevPtr:
    """Minimal __cuda_array_interface__ carrier so torch can alias library-owned HBM."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 3}


def _wrap_device_f32(ptr, n, device):
    return torch.as_tensor(_DevPtr(ptr, n), device=device)
