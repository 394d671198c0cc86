"""Patch embedding (reference: vit/patch_embed.py:12-109).  ``self.patch`` is an nn.Conv2d used only as the
parameter holder (state_dict keys stem.patch.weight / .bias); the convolution runs as im2col + libvitk GEMM
with bias and the position table added in the GEMM epilogue."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch.nn as nn
from torch import Tensor

from . import ops
from .initialization import init_conv
from .pos_enc import create_position_encoder


class PatchEmbed2d(nn.Module):
    def __init__(self, in_channels: int, hidden_size: int, patch_size: Sequence[int], img_size: Sequence[int],
                 pos_enc: str = "fourier", device=None, dtype=None):
        super().__init__()
        ps = tuple(patch_size)
        self.patch = nn.Conv2d(in_channels, hidden_size, ps, stride=ps, device=device, dtype=dtype)
        self.reset_parameters()
        self.pos_enc = create_position_encoder(pos_enc, hidden_size, self.tokenized_size(tuple(img_size)), device=device,
                                               dtype=dtype)

    def reset_parameters(self) -> None:
        init_conv(self.patch)

    @property
    def patch_size(self):
        return tuple(self.patch.weight.shape[2:])

    def tokenized_size(self, size: Sequence[int]):
        return tuple(s // p for s, p in zip(size, self.patch_size))

    def original_size(self, size: Sequence[int]):
        return tuple(s * p for s, p in zip(size, self.patch_size))

    def forward(self, x: Tensor, prefix: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
        """``prefix`` = (cls_tokens, register_tokens): return the blocks' whole input sequence
        cat([cls, register, patches], dim=1) (vit/vit.py:500-505) instead of the patch tokens alone."""
        pos = self.pos_enc(self.tokenized_size(x.shape[2:])) if self.pos_enc is not None else None
        cls_tokens, register_tokens = prefix if prefix is not None else (None, None)
        return ops.patch_embed(x, self.patch.weight, self.patch.bias, pos, 1, cls_tokens, register_tokens)


class PatchEmbed3d(nn.Module):
    """Volumetric patches (vit/patch_embed.py:62-109): ``self.patch`` is an nn.Conv3d parameter holder.  A patch of
    (pd, ph, pw) voxels is a 2-D patch of (ph, pw) pixels over C * pd channels once the depth offsets are folded into
    the channel axis, so the volume goes through the same im2col + GEMM path, one depth slab per batch entry."""

    def __init__(self, in_channels: int, hidden_size: int, patch_size: Sequence[int], img_size: Sequence[int],
                 pos_enc: str = "fourier", device=None, dtype=None):
        super().__init__()
        ps = tuple(patch_size)
        self.patch = nn.Conv3d(in_channels, hidden_size, ps, stride=ps, device=device, dtype=dtype)
        self.reset_parameters()
        self.pos_enc = create_position_encoder(pos_enc, hidden_size, self.tokenized_size(tuple(img_size)), device=device,
                                               dtype=dtype)

    def reset_parameters(self) -> None:
        init_conv(self.patch)

    @property
    def patch_size(self):
        return tuple(self.patch.weight.shape[2:])

    def tokenized_size(self, size: Sequence[int]):
        return tuple(s // p for s, p in zip(size, self.patch_size))

    def original_size(self, size: Sequence[int]):
        return tuple(s * p for s, p in zip(size, self.patch_size))

    def forward(self, x: Tensor, prefix: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
        B, C, Dz, H, W = x.shape
        pd, ph, pw = self.patch_size
        gd = Dz // pd
        pos = self.pos_enc(self.tokenized_size(x.shape[2:])) if self.pos_enc is not None else None
        # [B, C, gd*pd, H, W] -> [B*gd, C*pd, H, W]: channel index c*pd + t, the order of the Conv3d weight's (c, t, u, v)
        slabs = x[:, :, : gd * pd].reshape(B, C, gd, pd, H, W).permute(0, 2, 1, 3, 4, 5).reshape(B * gd, C * pd, H, W)
        w = self.patch.weight.reshape(self.patch.weight.shape[0], C * pd, ph, pw)
        cls_tokens, register_tokens = prefix if prefix is not None else (None, None)
        return ops.patch_embed(slabs, w, self.patch.bias, pos, gd, cls_tokens, register_tokens)
