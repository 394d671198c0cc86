"""Pre-norm encoder block with selective stochastic depth (reference: vit/transformer.py:14-157).

Stochastic depth here is NOT the mask-and-scale DropPath: the block runs attention / MLP on a random subset of
``keep_count`` samples and scatter-adds the result scaled by B / keep_count.  The subset is drawn with
torch.randperm exactly like the reference (same generator, same order: attention first, then MLP); the gather is
fused into the norm kernel's load and the scaled scatter-add into the out_proj / fc2 GEMM epilogue."""
from __future__ import annotations

import collections
from typing import Any, Optional

import torch
import torch.nn as nn
from torch import Tensor

from .attention import CrossAttention, SelfAttention
from .fused import NormMLP
from .initialization import zero_bias_if_present
from .layer_scale import LayerScale


def drop_path_keep_count(batch_size: int, drop_path_rate: float) -> int:
    return max(1, min(batch_size, int(batch_size * (1.0 - drop_path_rate))))


_PERMS: "collections.OrderedDict" = collections.OrderedDict()   # (device, seed, offset, n) -> (permutation, offset after the draw)


def _randperm(n: int, device: torch.device) -> Tensor:
    """torch.randperm(n, device=device) on the device's default generator.  On a GPU the draw is a function of the
    generator's (seed, offset) alone, so a draw from a state seen before (activation checkpointing replays every block's
    forward from the saved state) returns the permutation made then and moves the generator by the same amount, instead of
    launching the half-dozen small kernels of a device-side randperm again."""
    if device.type != "cuda":
        return torch.randperm(n, device=device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    gen = torch.cuda.default_generators[index]
    key = (index, gen.initial_seed(), gen.get_offset(), n)
    hit = _PERMS.get(key)
    if hit is not None:
        gen.set_offset(hit[1])
        return hit[0]
    perm = torch.randperm(n, device=device)
    _PERMS[key] = (perm, gen.get_offset())
    while len(_PERMS) > 512:
        _PERMS.popitem(last=False)
    return perm


def _select_keep_indices(x: Tensor, drop_path_rate: float, training: bool):
    """Indices of the samples that go through the residual branch, and the merge scale (vit/transformer.py:14-27)."""
    B = x.shape[0]
    if not training or drop_path_rate <= 0.0 or B <= 1:
        return None, 1.0
    keep = drop_path_keep_count(B, drop_path_rate)
    if keep == B:
        return None, 1.0
    perm = _randperm(B, x.device)
    idx = perm[:keep]
    idx._vitk_dropped = perm[keep:]   # the samples that pass through unchanged (ops copies only their rows)
    return idx, float(B / keep)


def _select_residual_subset(x: Tensor, drop_path_rate: float, training: bool):
    idx, scale = _select_keep_indices(x, drop_path_rate, training)
    return (x, None, 1.0) if idx is None else (x.index_select(0, idx), idx, scale)


def _merge_residual_subset(x: Tensor, residual: Tensor, keep_indices: Optional[Tensor], residual_scale: float) -> Tensor:
    if keep_indices is None:
        return x + residual
    return x.flatten(1).index_add(0, keep_indices, residual.flatten(1), alpha=residual_scale).view_as(x)


@torch.no_grad()
def _zero_residual_outputs(*modules: nn.Linear) -> None:
    for m in modules:
        m.weight.zero_()
        zero_bias_if_present(m)


class TransformerEncoderLayer(nn.Module):
    def __init__(self, hidden_size: int, ffn_hidden_size: int, num_attention_heads: int, hidden_dropout: float = 0.1,
                 attention_dropout: float = 0.1, attention_bias: bool = True, mlp_bias: bool = True,
                 activation: str = "gelu", drop_path_rate: float = 0.0, eps: float = 1e-5,
                 layer_scale: Optional[float] = None, glu_limit: Optional[float] = None,
                 glu_extra_bias: Optional[float] = None, mlp_quantization_config: Any = None,
                 qkv_quantization_config: Any = None, attn_quantization_config: Any = None, device=None, dtype=None,
                 norm_type: str = "rmsnorm", qk_normalization: bool = False):
        super().__init__()
        kw = {"device": device, "dtype": dtype}
        self.drop_path_rate = drop_path_rate
        self.self_attention = SelfAttention(hidden_size, num_attention_heads, hidden_dropout, attention_dropout,
                                            bias=attention_bias, eps=eps, norm_type=norm_type,
                                            qk_normalization=qk_normalization, **kw)
        self.mlp = NormMLP(hidden_size, ffn_hidden_size, bias=mlp_bias, activation=activation, eps=eps,
                           dropout=hidden_dropout, limit=glu_limit, extra_bias=glu_extra_bias, norm_type=norm_type, **kw)
        if layer_scale is not None:
            self.layer_scale_attn = LayerScale(hidden_size, layer_scale, inplace=True, **kw)
            self.layer_scale_mlp = LayerScale(hidden_size, layer_scale, inplace=True, **kw)
        else:
            self.layer_scale_attn = nn.Identity()
            self.layer_scale_mlp = nn.Identity()
        _zero_residual_outputs(self.self_attention.out_proj, self.mlp.fc2)
        self.apply_quantization(mlp_quantization_config, qkv_quantization_config, attn_quantization_config)

    def apply_quantization(self, mlp_quantization_config: Any = None, qkv_quantization_config: Any = None,
                           attn_quantization_config: Any = None) -> None:
        if any(c is not None for c in (mlp_quantization_config, qkv_quantization_config, attn_quantization_config)):
            raise NotImplementedError("vit_amd: torchao quantization is outside the encoder hot path")

    def forward(self, x: Tensor, rope: Optional[Tensor] = None) -> Tensor:
        g_attn = getattr(self.layer_scale_attn, "gamma", None)
        g_mlp = getattr(self.layer_scale_mlp, "gamma", None)
        idx, scale = _select_keep_indices(x, self.drop_path_rate, self.training)
        x = self.self_attention.fused_residual(x, rope, g_attn, idx, scale)
        idx, scale = _select_keep_indices(x, self.drop_path_rate, self.training)
        return self.mlp.fused_residual(x, g_mlp, idx, scale)


def _scale_or_identity(hidden_size: int, layer_scale: Optional[float], kw: dict) -> nn.Module:
    return LayerScale(hidden_size, layer_scale, inplace=True, **kw) if layer_scale is not None else nn.Identity()


class TransformerDecoderLayer(nn.Module):
    """Self-attention, cross-attention to ``kv``, MLP; each a pre-norm residual branch with its own drop-path subset
    and layer scale (vit/transformer.py:165-284)."""

    def __init__(self, hidden_size: int, ffn_hidden_size: int, num_attention_heads: int, hidden_dropout: float = 0.1,
                 attention_dropout: float = 0.1, attention_bias: bool = True, mlp_bias: bool = True,
                 activation: str = "gelu", drop_path_rate: float = 0.0, eps: float = 1e-5,
                 layer_scale: Optional[float] = None, glu_limit: Optional[float] = None,
                 glu_extra_bias: Optional[float] = None, mlp_quantization_config: Any = None,
                 qkv_quantization_config: Any = None, attn_quantization_config: Any = None, device=None, dtype=None,
                 norm_type: str = "rmsnorm", qk_normalization: bool = False):
        super().__init__()
        kw = {"device": device, "dtype": dtype}
        self.drop_path_rate = drop_path_rate
        self.self_attention = SelfAttention(hidden_size, num_attention_heads, hidden_dropout, attention_dropout,
                                            bias=attention_bias, eps=eps, norm_type=norm_type,
                                            qk_normalization=qk_normalization, **kw)
        self.cross_attention = CrossAttention(hidden_size, num_attention_heads, hidden_dropout, attention_dropout,
                                              bias=attention_bias, eps=eps, norm_type=norm_type,
                                              qk_normalization=qk_normalization, **kw)
        self.mlp = NormMLP(hidden_size, ffn_hidden_size, bias=mlp_bias, activation=activation, eps=eps,
                           dropout=hidden_dropout, limit=glu_limit, extra_bias=glu_extra_bias, norm_type=norm_type, **kw)
        self.layer_scale_attn = _scale_or_identity(hidden_size, layer_scale, kw)
        self.layer_scale_mlp = _scale_or_identity(hidden_size, layer_scale, kw)
        self.layer_scale_cross = _scale_or_identity(hidden_size, layer_scale, kw)
        _zero_residual_outputs(self.self_attention.out_proj, self.cross_attention.out_proj, self.mlp.fc2)
        self.apply_quantization(mlp_quantization_config, qkv_quantization_config, attn_quantization_config)

    def apply_quantization(self, mlp_quantization_config: Any = None, qkv_quantization_config: Any = None,
                           attn_quantization_config: Any = None) -> None:
        if any(c is not None for c in (mlp_quantization_config, qkv_quantization_config, attn_quantization_config)):
            raise NotImplementedError("vit_amd: torchao quantization is outside the encoder hot path")

    def forward(self, x: Tensor, kv: Tensor, rope_q: Optional[Tensor] = None, rope_k: Optional[Tensor] = None) -> Tensor:
        idx, scale = _select_keep_indices(x, self.drop_path_rate, self.training)
        x = self.self_attention.fused_residual(x, rope_q, getattr(self.layer_scale_attn, "gamma", None), idx, scale)
        idx, scale = _select_keep_indices(x, self.drop_path_rate, self.training)
        x = self.cross_attention.fused_residual(x, kv, getattr(self.layer_scale_cross, "gamma", None), idx, scale,
                                                rope_q, rope_k)
        idx, scale = _select_keep_indices(x, self.drop_path_rate, self.training)
        return self.mlp.fused_residual(x, getattr(self.layer_scale_mlp, "gamma", None), idx, scale)


class CrossAttentionTransformer(nn.Module):
    """Cross-attention to ``kv``, then MLP (vit/transformer.py:290-384)."""

    def __init__(self, hidden_size: int, ffn_hidden_size: int, num_attention_heads: int, hidden_dropout: float = 0.1,
                 attention_dropout: float = 0.1, attention_bias: bool = True, mlp_bias: bool = True,
                 activation: str = "gelu", drop_path_rate: float = 0.0, eps: float = 1e-5,
                 layer_scale: Optional[float] = None, glu_limit: Optional[float] = None,
                 glu_extra_bias: Optional[float] = None, mlp_quantization_config: Any = None,
                 qkv_quantization_config: Any = None, attn_quantization_config: Any = None, device=None, dtype=None,
                 norm_type: str = "rmsnorm", qk_normalization: bool = False):
        super().__init__()
        kw = {"device": device, "dtype": dtype}
        self.drop_path_rate = drop_path_rate
        self.cross_attention = CrossAttention(hidden_size, num_attention_heads, hidden_dropout, attention_dropout,
                                              bias=attention_bias, eps=eps, norm_type=norm_type,
                                              qk_normalization=qk_normalization, **kw)
        self.mlp = NormMLP(hidden_size, ffn_hidden_size, bias=mlp_bias, activation=activation, eps=eps,
                           dropout=hidden_dropout, limit=glu_limit, extra_bias=glu_extra_bias, norm_type=norm_type, **kw)
        self.layer_scale_cross = _scale_or_identity(hidden_size, layer_scale, kw)
        self.layer_scale_mlp = _scale_or_identity(hidden_size, layer_scale, kw)
        _zero_residual_outputs(self.cross_attention.out_proj, self.mlp.fc2)
        self.apply_quantization(mlp_quantization_config, qkv_quantization_config, attn_quantization_config)

    def apply_quantization(self, mlp_quantization_config: Any = None, qkv_quantization_config: Any = None,
                           attn_quantization_config: Any = None) -> None:
        if any(c is not None for c in (mlp_quantization_config, qkv_quantization_config, attn_quantization_config)):
            raise NotImplementedError("vit_amd: torchao quantization is outside the encoder hot path")

    def forward(self, x: Tensor, kv: Tensor, rope_q: Optional[Tensor] = None, rope_k: Optional[Tensor] = None) -> Tensor:
        idx, scale = _select_keep_indices(x, self.drop_path_rate, self.training)
        x = self.cross_attention.fused_residual(x, kv, getattr(self.layer_scale_cross, "gamma", None), idx, scale,
                                                rope_q, rope_k)
        idx, scale = _select_keep_indices(x, self.drop_path_rate, self.training)
        return self.mlp.fused_residual(x, getattr(self.layer_scale_mlp, "gamma", None), idx, scale)
