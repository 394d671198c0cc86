"""The ViT encoder (reference: vit/vit.py:254-615): same constructor, attribute tree, state_dict keys and
forward signature; every arithmetic step of forward/backward runs in libvitk kernels.

Token masks (masked pre-training) drop tokens ahead of the blocks and give every sample its own RoPE rows.
Heads: the linear head and the attentive-pool head (vit_amd/head.py).
Decoder / cross-attention layers of the backbone's width: create_decoder_layer, create_cross_attention_layer.
Not provided (outside the encoder hot path): dense-prediction heads, torchao quantization.  They raise NotImplementedError rather than fall back."""
from __future__ import annotations

from typing import Any, Optional, Sequence

import torch
import torch.nn as nn
from torch import Tensor
from torch.utils.checkpoint import checkpoint

from . import ops
from .config import ViTConfig
from .features import ViTFeatures
from .head import AttentivePoolHeadConfig, HeadConfig
from .initialization import trunc_normal_
from .norm import make_norm
from .patch_embed import PatchEmbed2d, PatchEmbed3d
from .rope import RopePositionEmbedding
from .tokens import apply_mask, create_mask
from .transformer import CrossAttentionTransformer, TransformerDecoderLayer, TransformerEncoderLayer

__all__ = ["ViT", "ViTConfig", "ViTFeatures"]


class ViT(nn.Module):
    def __init__(self, config: ViTConfig, mlp_quantization_config: Any = None, qkv_quantization_config: Any = None,
                 attn_quantization_config: Any = None, device: Optional[torch.device] = None):
        super().__init__()
        if len(config.patch_size) not in (2, 3):
            raise ValueError(f"patch_size must have 2 or 3 entries, got {config.patch_size}")
        for name, head_config in config.heads.items():
            if not isinstance(head_config, (HeadConfig, AttentivePoolHeadConfig)):
                raise NotImplementedError(f"vit_amd: head {name!r} ({type(head_config).__name__}) is not built; "
                                          "HeadConfig and AttentivePoolHeadConfig are")
        if any(c is not None for c in (mlp_quantization_config, qkv_quantization_config, attn_quantization_config)):
            raise NotImplementedError("vit_amd: torchao quantization is outside the encoder hot path")
        kw = {"device": device, "dtype": config.dtype}
        self._config = config
        D = config.hidden_size
        stem_type = PatchEmbed2d if len(config.patch_size) == 2 else PatchEmbed3d
        self.stem = stem_type(config.in_channels, D, config.patch_size, config.img_size,
                                 pos_enc="none" if config.pos_enc == "rope" else config.pos_enc, **kw)
        self.patch_embed_norm = make_norm(D, config.norm_type, **kw) if config.patch_embed_normalization else None
        self.rope = None
        if config.pos_enc == "rope":
            self.rope = RopePositionEmbedding(D, base=config.rope_base, num_heads=config.num_attention_heads,
                                              rescale_coords=config.rope_rescale_coords,
                                              shift_coords=config.rope_shift_coords,
                                              jitter_coords=config.rope_jitter_coords, **kw)
        self.register_tokens = nn.Parameter(torch.empty(1, config.num_register_tokens, D, **kw),
                                            requires_grad=config.num_register_tokens > 0)
        self.cls_tokens = nn.Parameter(torch.empty(1, config.num_cls_tokens, D, **kw),
                                       requires_grad=config.num_cls_tokens > 0)
        trunc_normal_(self.register_tokens)
        trunc_normal_(self.cls_tokens)
        self.blocks = nn.ModuleList([self.create_encoder_layer(device=device) for _ in range(config.depth)])
        self.output_norm = make_norm(D, config.norm_type, **kw)
        self.mlp_requires_grad_(config.mlp_requires_grad)
        self.self_attention_requires_grad_(config.self_attention_requires_grad)
        self.heads = nn.ModuleDict({name: self.create_head(name, hc, device=device) for name, hc in config.heads.items()})

    # ---- construction helpers ------------------------------------------------------------------
    @property
    def config(self) -> ViTConfig:
        return self._config

    def _layer(self, layer_type, mlp_quantization_config, qkv_quantization_config, attn_quantization_config, device, dtype):
        c = self._config
        return layer_type(
            c.hidden_size, c.ffn_hidden_size, c.num_attention_heads, hidden_dropout=c.hidden_dropout,
            attention_dropout=c.attention_dropout, attention_bias=c.attention_bias, mlp_bias=c.mlp_bias,
            activation=c.activation, norm_type=c.norm_type, qk_normalization=c.qk_normalization,
            drop_path_rate=c.drop_path_rate, layer_scale=c.layer_scale, glu_limit=c.glu_limit,
            glu_extra_bias=c.glu_extra_bias, mlp_quantization_config=mlp_quantization_config,
            qkv_quantization_config=qkv_quantization_config, attn_quantization_config=attn_quantization_config,
            device=device, dtype=c.dtype if dtype is None else dtype)

    def create_encoder_layer(self, mlp_quantization_config: Any = None, qkv_quantization_config: Any = None,
                             attn_quantization_config: Any = None, device=None, dtype=None) -> TransformerEncoderLayer:
        return self._layer(TransformerEncoderLayer, mlp_quantization_config, qkv_quantization_config,
                           attn_quantization_config, device, dtype)

    def create_decoder_layer(self, mlp_quantization_config: Any = None, qkv_quantization_config: Any = None,
                             attn_quantization_config: Any = None, device=None, dtype=None) -> TransformerDecoderLayer:
        """A layer of the backbone's width with self-attention, cross-attention and MLP (vit/vit.py:373-402)."""
        return self._layer(TransformerDecoderLayer, mlp_quantization_config, qkv_quantization_config,
                           attn_quantization_config, device, dtype)

    def create_cross_attention_layer(self, mlp_quantization_config: Any = None, qkv_quantization_config: Any = None,
                                     attn_quantization_config: Any = None, device=None,
                                     dtype=None) -> CrossAttentionTransformer:
        """Cross-attention + MLP (vit/vit.py:404-433)."""
        return self._layer(CrossAttentionTransformer, mlp_quantization_config, qkv_quantization_config,
                           attn_quantization_config, device, dtype)

    def create_head(self, name: str, head_config: Any, device: Optional[torch.device] = None, dtype=None) -> nn.Module:
        """vit/vit.py:435-447."""
        return head_config.instantiate(self._config, device=device, dtype=dtype or self._config.dtype)

    def get_head(self, name: str) -> nn.Module:
        return self.heads[name]

    def get_block(self, i: int) -> TransformerEncoderLayer:
        block = self.blocks[i]
        assert isinstance(block, TransformerEncoderLayer)
        return block

    def apply_quantization(self, *configs: Any) -> None:
        if any(c is not None for c in configs):
            raise NotImplementedError("vit_amd: torchao quantization is outside the encoder hot path")

    def create_mask(self, input: Tensor, unmasked_ratio: float, scale: int, roll: bool = False) -> Tensor:
        """Random token mask [B, L] for a raw (un-tokenized) input; True = kept (vit/vit.py:458-494)."""
        grid = self.stem.tokenized_size(tuple(input.shape[2:]))
        return create_mask(grid, mask_ratio=1 - unmasked_ratio, batch_size=input.shape[0], scale=scale, roll=roll,
                           device=input.device)

    @property
    def prefix_length(self) -> int:
        return self._config.num_register_tokens

    def train(self, mode: bool = True):
        """A switch between training and evaluation is where weights get swapped through ``.data`` (EMA copies, loaded
        checkpoints): the kept compute-dtype copies of the parameters are dropped with it (vit_amd.ops._cast_param)."""
        if mode != self.training:
            ops.clear_param_copies()
        return super().train(mode)

    # ---- forward -----------------------------------------------------------------------------------
    def add_prefix_tokens(self, x: Tensor) -> Tensor:
        """[B, N, D] -> [B, n_cls + n_reg + N, D], order cls, register, patches; dtype promotes like torch.cat."""
        return ops.concat_prefix(self.cls_tokens, self.register_tokens, x)

    def prepare_rope(self, tokenized_size: Sequence[int], mask: Optional[Tensor] = None,
                     rope_seed: Optional[int] = None) -> Tensor:
        if self.rope is None:
            raise ValueError("RoPE is not enabled")
        if len(tokenized_size) != 2:
            raise ValueError(f"RoPE not supported for non-2D input, got {tokenized_size}")
        rope = self.rope(H=tokenized_size[0], W=tokenized_size[1], rope_seed=rope_seed)
        if mask is not None:   # every sample keeps its own rows of the table: [2, B, 1, N', Dh] (vit/vit.py:522-527)
            B = mask.shape[0]
            sin = apply_mask(mask, rope[0][None].expand(B, -1, -1))
            cos = apply_mask(mask, rope[1][None].expand(B, -1, -1))
            rope = torch.stack([sin[:, None], cos[:, None]], dim=0)
        return rope

    def normalize_patch_embeddings(self, x: Tensor) -> Tensor:
        return x if self.patch_embed_norm is None else self.patch_embed_norm(x)

    def forward(self, x: Tensor, mask: Optional[Tensor] = None, rope_seed: Optional[int] = None,
                output_norm: bool = True) -> ViTFeatures:
        c = self._config
        grid = self.stem.tokenized_size(x.shape[2:])
        if mask is None and self.patch_embed_norm is None:
            # nothing stands between the patch embedding and the concatenation: its GEMM writes the patch rows straight
            # into the sequence, behind the prefix tokens
            x = self.stem(x, prefix=(self.cls_tokens, self.register_tokens))
        else:
            x = self.stem(x)
            x = self.normalize_patch_embeddings(x)
            if mask is not None:   # dropped tokens leave the sequence before the blocks (vit/vit.py:545)
                x = apply_mask(mask, x)
            x = self.add_prefix_tokens(x)
        rope = self.prepare_rope(grid, mask, rope_seed) if self.rope is not None else None
        for block in self.blocks:
            assert isinstance(block, TransformerEncoderLayer)
            if c.activation_checkpointing and self.training:
                x = checkpoint(block, x, rope, use_reentrant=False)
            else:
                x = block(x, rope=rope)
        if output_norm:
            x = self.output_norm(x)
        return ViTFeatures(x, c.num_register_tokens, c.num_cls_tokens, grid)

    @torch.no_grad()
    def forward_attention_weights(self, x: Tensor) -> dict:
        """Per-layer attention probabilities of every token over the visual tokens, laid out on the token grid:
        {"layer_i": [B, H, S, *grid]} (vit/vit.py:575-596; like the reference, without RoPE)."""
        grid = self.stem.tokenized_size(x.shape[2:])
        x = self.stem(x)
        x = self.normalize_patch_embeddings(x)
        x = self.add_prefix_tokens(x)
        weights = {}
        for i, block in enumerate(self.blocks):
            w = block.self_attention.forward_weights(x)
            B, H, Lq, Lk = w.shape
            weights[f"layer_{i}"] = w[..., self._config.num_register_tokens:].reshape(B, H, Lq, *grid).contiguous()
            x = block(x)
        return weights

    # ---- trainability switches -----------------------------------------------------------------------
    def mlp_requires_grad_(self, requires_grad: bool = True) -> None:
        for block in self.blocks:
            block.mlp.requires_grad_(requires_grad)

    def self_attention_requires_grad_(self, requires_grad: bool = True) -> None:
        for block in self.blocks:
            block.self_attention.requires_grad_(requires_grad)

    def backbone_requires_grad_(self, requires_grad: bool = True) -> None:
        self.stem.requires_grad_(requires_grad)
        if self.patch_embed_norm is not None:
            self.patch_embed_norm.requires_grad_(requires_grad)
        self.blocks.requires_grad_(requires_grad)
        self.output_norm.requires_grad_(requires_grad)
        self.register_tokens.requires_grad_(requires_grad)
