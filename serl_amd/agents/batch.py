"""Device-side batch (the C ABI's `serl_batch`): cropped u8 frames + small f32 fields in HBM."""
from __future__ import annotations

import torch

from .._lib import SerlBatch


class DeviceBatch:
    """frames u8[2(obs,next)][n_cam][B][H][W][C]; state f32[2][B][S]; action f32[B][A]; ..."""

    def __init__(self, batch, n_cam, H, W, C, S, A, device):
        dev = torch.device("cuda", device) if isinstance(device, int) else device
        self.frames = torch.empty((2, n_cam, batch, H, W, C), dtype=torch.uint8, device=dev)
        self.state = torch.empty((2, batch, S), dtype=torch.float32, device=dev)
        self.action = torch.empty((batch, A), dtype=torch.float32, device=dev)
        self.reward = torch.empty((batch,), dtype=torch.float32, device=dev)
        self.mask = torch.empty((batch,), dtype=torch.float32, device=dev)
        self.done = torch.empty((batch,), dtype=torch.uint8, device=dev)
        self.cstruct = SerlBatch(batch, n_cam, H, W, C, S, A, self.frames.data_ptr(),
                                 self.state.data_ptr(), self.action.data_ptr(),
                                 self.reward.data_ptr(), self.mask.data_ptr(), self.done.data_ptr())
        self.batch, self.n_cam = batch, n_cam
