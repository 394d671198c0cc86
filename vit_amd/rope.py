"""Axial RoPE table (reference: vit/rope.py).  The [2, H*W, Dh] sin/cos table is O(HW*Dh) host-side setup
and is generated with the same torch ops in the same dtype as the reference, because the reference
computes it in the parameter dtype (a bf16 table differs from the fp32 one by up to 2e-2).  The rotation
itself runs in libvitk (vitk_rope_qk) on the packed q/k projection inside the attention sub-layer."""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor, nn


class RopePositionEmbedding(nn.Module):
    periods: Tensor

    def __init__(self, embed_dim: int, *, num_heads: int, base: Optional[float] = 100.0,
                 min_period: Optional[float] = None, max_period: Optional[float] = None,
                 normalize_coords: str = "separate", shift_coords: Optional[float] = None,
                 jitter_coords: Optional[float] = None, rescale_coords: Optional[float] = None, dtype=None, device=None):
        super().__init__()
        assert embed_dim % (4 * num_heads) == 0
        both = min_period is not None and max_period is not None
        if (base is None) == (not both):
            raise ValueError("Either `base` or `min_period`+`max_period` must be provided.")
        self.D_head = embed_dim // num_heads
        self.base, self.min_period, self.max_period = base, min_period, max_period
        self.normalize_coords = normalize_coords
        self.shift_coords, self.jitter_coords, self.rescale_coords = shift_coords, jitter_coords, rescale_coords
        self.dtype = dtype
        self.register_buffer("periods", torch.empty(self.D_head // 4, device=device, dtype=dtype), persistent=True)
        self._init_weights()

    def _init_weights(self) -> None:
        dev, n = self.periods.device, self.D_head // 4
        if self.base is not None:
            per = self.base ** (2 * torch.arange(n, device=dev, dtype=self.dtype) / (self.D_head // 2))
        else:
            ratio = self.max_period / self.min_period
            per = ratio ** torch.linspace(0, 1, n, device=dev, dtype=self.dtype) / ratio * self.max_period
        self.periods.data = per

    def forward(self, *, H: int, W: int, rope_seed: Optional[int] = None) -> Tensor:
        dd = {"device": self.periods.device, "dtype": self.dtype}
        if self.normalize_coords == "separate":
            den_h, den_w = H, W
        elif self.normalize_coords == "max":
            den_h = den_w = max(H, W)
        elif self.normalize_coords == "min":
            den_h = den_w = min(H, W)
        else:
            raise ValueError(f"Unknown normalize_coords: {self.normalize_coords}")
        ch = torch.arange(0.5, H, **dd) / den_h
        cw = torch.arange(0.5, W, **dd) / den_w
        coords = torch.stack(torch.meshgrid(ch, cw, indexing="ij"), dim=-1).flatten(0, 1)
        coords = 2.0 * coords - 1.0
        gen = None
        if rope_seed is not None:
            gen = torch.Generator(device=coords.device)
            gen.manual_seed(rope_seed)
        if self.training and self.shift_coords is not None:
            coords += torch.empty(2, **dd).uniform_(-self.shift_coords, self.shift_coords, generator=gen)[None, :]
        if self.training and self.jitter_coords is not None:
            j = math.log(self.jitter_coords)
            coords *= torch.empty(2, **dd).uniform_(-j, j, generator=gen).exp()[None, :]
        if self.training and self.rescale_coords is not None:
            r = math.log(self.rescale_coords)
            coords *= torch.empty(1, **dd).uniform_(-r, r, generator=gen).exp()
        ang = (2 * math.pi * coords[:, :, None] / self.periods[None, None, :]).flatten(1, 2).tile(2)
        return torch.stack([torch.sin(ang), torch.cos(ang)], dim=0)


def apply_rope(x: Tensor, rope: Tensor) -> Tensor:
    """Rotate the last ``rope.shape[-2]`` rows of x [B, H, S, Dh] with the table [2, N, Dh] (or the per-sample
    [2, B, 1, N, Dh]); rows before them (prefix tokens) pass through (vit/rope.py:139-163)."""
    from . import ops

    return ops.apply_rope(x, rope)
