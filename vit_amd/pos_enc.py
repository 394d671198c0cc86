"""Position encoders added to the patch tokens (reference: vit/pos_enc.py).

Fourier: the batch-independent (1, N, D) table is built once per step: a K=2 product and sin/cos in fp32
(host-side set-up, as in the reference with autocast disabled), then the D x D projection runs through
libvitk's GEMM (ops.linear).  The "+ pos" itself is fused into the patch-embedding GEMM epilogue."""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from . import ops
from .initialization import INIT_STD, init_linear, trunc_normal_

PositionEncoder = str  # "fourier" | "learnable" | "none" | "rope"


_GRIDS: dict = {}   # (dims, dtype, device, normalize) -> coordinate grid: constants, built once per process


def create_grid(dims: Sequence[int], dtype=torch.float32, device="cpu", normalize: bool = True) -> Tensor:
    key = (tuple(dims), dtype, str(device), normalize)
    grid = _GRIDS.get(key)
    if grid is None:
        axes = [torch.linspace(-1, 1, d, device=device, dtype=dtype) if normalize else torch.arange(d, device=device, dtype=dtype)
                for d in dims]
        grid = torch.stack(torch.meshgrid(axes, indexing="ij"), dim=-1).view(1, -1, len(dims))
        if len(_GRIDS) < 64:
            _GRIDS[key] = grid
    return grid


def fourier_position(dims: Sequence[int], w_fourier: Tensor, w_fc1: Tensor, b_fc1: Optional[Tensor]) -> Tensor:
    grid = create_grid(dims, device=w_fourier.device)
    if len(dims) <= 3 and w_fourier.dtype == torch.float32:   # other dtypes fail below exactly as they do in the reference
        # features in fp32 arithmetic, rounded once to the dtype the projection computes in (what its input cast does)
        feats = ops.FourierFeaturesFn.apply(grid, w_fourier, ops.compute_dtype(w_fc1.dtype))
    else:
        with torch.autocast(device_type=w_fourier.device.type, enabled=False):
            feats = grid @ w_fourier
            feats = torch.cat([feats.sin(), feats.cos()], dim=-1) / math.sqrt(w_fourier.shape[-1])
    return ops.linear(feats, w_fc1, b_fc1)


class FourierPosition(nn.Module):
    def __init__(self, hidden_size: int, spatial_size: Sequence[int], device=None, dtype=None):
        super().__init__()
        self.spatial_size = spatial_size
        self.w_fourier = nn.Parameter(torch.empty(len(spatial_size), hidden_size // 2, device=device, dtype=dtype))
        self.w_fc1 = nn.Linear(hidden_size, hidden_size, device=device, dtype=dtype)
        self.reset_parameters()

    def reset_parameters(self, std: float = 1.0) -> None:
        nn.init.normal_(self.w_fourier, std=std)
        init_linear(self.w_fc1)

    def forward(self, dims: Optional[Sequence[int]]) -> Tensor:
        return fourier_position(dims or self.spatial_size, self.w_fourier, self.w_fc1.weight, self.w_fc1.bias)


@torch.no_grad()
def fourier_features_(param: Tensor, std: float = 1.0) -> None:
    *dims, hidden = param.shape
    if hidden % 2:
        raise ValueError(f"Hidden size must be even, got {hidden}")
    w = param.new_empty(len(dims), hidden // 2).normal_(std=std)
    f = create_grid(dims, dtype=param.dtype, device=param.device).squeeze(0) @ w
    param.data.copy_((torch.cat([f.sin(), f.cos()], dim=-1) / math.sqrt(hidden)).view_as(param))


class LearnablePosition(nn.Module):
    """Learned table with bicubic resize; a (1, N, D) parameter-sized tensor, batch independent, so the resize and
    the table dropout stay torch ops (set-up work), exactly as the reference orders them (vit/pos_enc.py:79-90)."""

    def __init__(self, hidden_size: int, spatial_size: Sequence[int], fourier_init: bool = True, dropout: float = 0.0,
                 device=None, dtype=None):
        super().__init__()
        self.spatial_size = spatial_size
        self.positions = nn.Parameter(torch.empty(math.prod(spatial_size), hidden_size, device=device, dtype=dtype))
        self.dropout = nn.Dropout(dropout)
        self.reset_parameters(fourier_init)

    @torch.no_grad()
    def reset_parameters(self, fourier_init: bool = True) -> None:
        if fourier_init:
            fourier_features_(self.positions.view(*self.spatial_size, self.positions.shape[-1]))
            self.positions.data.sub_(self.positions.mean()).mul_(INIT_STD / self.positions.std())
        else:
            trunc_normal_(self.positions)

    def forward(self, dims: Optional[Sequence[int]]) -> Tensor:
        dims = tuple(dims or self.spatial_size)
        pos, L_ = self.positions, math.prod(dims)
        if dims != tuple(self.spatial_size):
            p = pos.view(1, *self.spatial_size, -1).movedim(-1, 1)
            pos = F.interpolate(p, size=dims, mode="bicubic", antialias=False).movedim(1, -1).reshape(1, L_, -1)
        return F.dropout(pos, p=self.dropout.p, training=self.training).view(1, L_, -1)


def create_position_encoder(pos_enc: str, hidden_size: int, spatial_size: Sequence[int], **kwargs):
    if pos_enc == "learnable":
        kwargs.setdefault("dropout", 0.1)
        return LearnablePosition(hidden_size, spatial_size, **kwargs)
    if pos_enc == "fourier":
        return FourierPosition(hidden_size, spatial_size, **kwargs)
    if pos_enc == "none":
        return None
    if pos_enc == "rope":
        raise ValueError("Please create RoPE position encoder manually")
    raise ValueError(f"Invalid position encoder: {pos_enc}")
