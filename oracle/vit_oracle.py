"""TEST INFRASTRUCTURE -- CPU restatement ("oracle") of the reference ViT encoder hot path.

This file is the checker, never the product: only tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py may import it.  Nothing under vit_amd/ imports it.

It restates, as one flat functional program over a ``state_dict``, the arithmetic that the
reference expresses as nn.Modules calling ``@torch.compile`` functions.  Every function cites
the reference lines it follows (paths relative to the reference checkout).  All arithmetic is
stock ``torch`` (F.linear, F.conv2d, F.scaled_dot_product_attention, F.rms_norm, F.layer_norm),
which is also all the reference uses, so on the same machine the fp32 results are bit-identical
to the reference's eager path; ``oracle/check_against_reference.py`` asserts that in the build
container and ``oracle/gen_golden.py`` writes the reference's own outputs to tests/golden/.

Pinned by: tests/golden/*.pt (outputs of the live reference, generated in the build container).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any, Callable, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.utils.checkpoint import checkpoint


@dataclass
class OracleConfig:
    """Field-for-field the hot-path subset of the reference's ViTConfig (vit/vit.py:67-113)."""

    in_channels: int
    patch_size: Sequence[int]
    img_size: Sequence[int]
    depth: int
    hidden_size: int
    ffn_hidden_size: int
    num_attention_heads: int
    hidden_dropout: float = 0.1
    attention_dropout: float = 0.1
    attention_bias: bool = True
    mlp_bias: bool = True
    activation: str = "srelu"
    drop_path_rate: float = 0.0
    num_register_tokens: int = 0
    num_cls_tokens: int = 0
    pos_enc: str = "rope"
    layer_scale: float | None = None
    glu_limit: float | None = None
    glu_extra_bias: float | None = None
    rope_normalize_coords: str = "separate"  # dead in the reference (never forwarded, vit/vit.py:284-292)
    rope_base: float = 100
    rope_shift_coords: float | None = None
    rope_jitter_coords: float | None = None
    rope_rescale_coords: float | None = None
    mlp_requires_grad: bool = True
    self_attention_requires_grad: bool = True
    activation_checkpointing: bool = False
    dtype: torch.dtype = torch.bfloat16
    norm_type: str = "rmsnorm"
    qk_normalization: bool = False
    patch_embed_normalization: bool = False
    heads: dict = field(default_factory=dict)

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def is_glu(self) -> bool:
        return self.activation.endswith("glu")  # vit/fused.py:189


# --------------------------------------------------------------------------------------------
# elementwise pieces
# --------------------------------------------------------------------------------------------
def activation_fn(name: str) -> Callable[[Tensor], Tensor]:
    """vit/helpers.py:35-48 (srelu :9-12, openswiglu :21-23)."""
    if name in ("relu", "reglu"):
        return F.relu
    if name in ("silu", "swiglu"):
        return F.silu
    if name in ("gelu", "geglu"):
        return F.gelu
    if name == "srelu":
        return lambda t: t.relu().pow(2)
    if name == "openswiglu":
        return lambda t: t * torch.sigmoid(1.702 * t)
    raise ValueError(f"Activation {name} not supported")


def apply_norm(x: Tensor, weight: Tensor, bias: Tensor | None, eps: float | None, use_layer_norm: bool) -> Tensor:
    """vit/norm.py:40-50.  ``eps=None`` is what nn.RMSNorm(eps=None) passes (vit/vit.py:316)."""
    if use_layer_norm:
        return F.layer_norm(x, x.shape[-1:], weight=weight, bias=bias, eps=1e-5 if eps is None else eps)
    return F.rms_norm(x, x.shape[-1:], weight=weight, eps=eps)


def rope_periods(head_dim: int, base: float, dtype: torch.dtype, device=None) -> Tensor:
    """vit/rope.py:122-136 (``base`` parametrisation only; the model never passes min/max period)."""
    return base ** (2 * torch.arange(head_dim // 4, device=device, dtype=dtype) / (head_dim // 2))


def rope_table(
    H: int,
    W: int,
    periods: Tensor,
    dtype: torch.dtype,
    *,
    training: bool = False,
    shift: float | None = None,
    jitter: float | None = None,
    rescale: float | None = None,
    rope_seed: int | None = None,
) -> Tensor:
    """vit/rope.py:60-115 with normalize_coords == "separate" (the only mode the model reaches)."""
    dd = {"device": periods.device, "dtype": dtype}
    coords_h = torch.arange(0.5, H, **dd) / H
    coords_w = torch.arange(0.5, W, **dd) / W
    coords = torch.stack(torch.meshgrid(coords_h, coords_w, indexing="ij"), dim=-1).flatten(0, 1)
    coords = 2.0 * coords - 1.0
    gen = None
    if rope_seed is not None:
        gen = torch.Generator(device=coords.device)
        gen.manual_seed(rope_seed)
    if training and shift is not None:
        coords += torch.empty(2, **dd).uniform_(-shift, shift, generator=gen)[None, :]
    if training and jitter is not None:
        jm = math.log(jitter)
        coords *= torch.empty(2, **dd).uniform_(-jm, jm, generator=gen).exp()[None, :]
    if training and rescale is not None:
        rm = math.log(rescale)
        coords *= torch.empty(1, **dd).uniform_(-rm, rm, generator=gen).exp()
    angles = 2 * math.pi * coords[:, :, None] / periods[None, None, :]
    angles = angles.flatten(1, 2).tile(2)
    return torch.stack([torch.sin(angles), torch.cos(angles)], dim=0)


def apply_rope(x: Tensor, rope: Tensor) -> Tensor:
    """vit/rope.py:139-163: half-split rotation of the last ``rope.shape[-2]`` rows, in rope dtype."""
    sin, cos = rope
    orig = x.dtype
    x = x.type_as(rope)
    prefix_len = x.shape[-2] - sin.shape[-2]
    assert prefix_len >= 0
    body = x[:, :, prefix_len:, :]
    x1, x2 = body.chunk(2, dim=-1)
    body = (body * cos) + (torch.cat([-x2, x1], dim=-1) * sin)
    return torch.cat((x[:, :, :prefix_len, :], body), dim=-2).to(orig)


def fourier_position(dims: Sequence[int], w_fourier: Tensor, w_fc1: Tensor, b_fc1: Tensor | None) -> Tensor:
    """vit/pos_enc.py:137-149 with create_grid :51-76 inlined (fp32 grid, autocast disabled inside)."""
    with torch.autocast(device_type=w_fourier.device.type, enabled=False):
        lens = [torch.linspace(-1, 1, d, device=w_fourier.device, dtype=torch.float32) for d in dims]
        grid = torch.stack(torch.meshgrid(lens, indexing="ij"), dim=-1).view(1, -1, len(dims))
        feats = grid @ w_fourier
        feats = torch.cat([feats.sin(), feats.cos()], dim=-1) / math.sqrt(w_fourier.shape[-1])
    return F.linear(feats, w_fc1, b_fc1)


def learnable_position(dims: Sequence[int], positions_size: Sequence[int], positions: Tensor,
                       dropout: float, training: bool) -> Tensor:
    """vit/pos_enc.py:79-90."""
    L = math.prod(dims)
    if tuple(dims) != tuple(positions_size):
        p = positions.view(1, *positions_size, -1).movedim(-1, 1)
        p = F.interpolate(p, size=tuple(dims), mode="bicubic", antialias=False)
        positions = p.movedim(1, -1).reshape(1, L, -1)
    positions = F.dropout(positions, p=dropout, training=training)
    return positions.view(1, L, -1)


# --------------------------------------------------------------------------------------------
# sub-layers
# --------------------------------------------------------------------------------------------
def project_qkv(cfg: OracleConfig, sd: dict, pre: str, x: Tensor, rope: Tensor | None):
    """vit/attention.py:71-111."""
    ln = cfg.norm_type == "layernorm"
    xn = apply_norm(x, sd[pre + "norm.weight"], sd.get(pre + "norm.bias"), 1e-5, ln)
    q, k, v = F.linear(xn, sd[pre + "qkv_proj.weight"], sd.get(pre + "qkv_proj.bias")).chunk(3, dim=-1)
    B, S, _ = q.shape
    dh = cfg.head_dim
    q = q.view(B, S, -1, dh).transpose(1, 2)
    k = k.view(B, S, -1, dh).transpose(1, 2)
    if cfg.qk_normalization:  # vit/attention.py:25-41
        q = apply_norm(q, sd[pre + "q_norm.weight"], sd.get(pre + "q_norm.bias"), 1e-5, ln)
        k = apply_norm(k, sd[pre + "k_norm.weight"], sd.get(pre + "k_norm.bias"), 1e-5, ln)
    if rope is not None:
        q = apply_rope(q, rope)
        k = apply_rope(k, rope)
    v = v.view(B, S, -1, dh).transpose(1, 2)
    return q, k, v


def self_attention(cfg: OracleConfig, sd: dict, pre: str, x: Tensor, rope: Tensor | None, training: bool) -> Tensor:
    """vit/attention.py:163-215 (attention_qkv_packed) as called by SelfAttention.forward :422-449."""
    q, k, v = project_qkv(cfg, sd, pre, x, rope)
    p_attn = cfg.attention_dropout if training else 0.0
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=p_attn, is_causal=False, enable_gqa=True)
    B, H, S, D = o.shape
    o = o.transpose(1, 2).reshape(B, S, H * D)
    o = F.linear(o, sd[pre + "out_proj.weight"], sd.get(pre + "out_proj.bias"))
    return F.dropout(o, p=cfg.hidden_dropout, training=training)


def norm_mlp(cfg: OracleConfig, sd: dict, pre: str, x: Tensor, training: bool) -> Tensor:
    """vit/fused.py:96-119 (plain) and :131-166 (GLU), as dispatched by NormMLP.forward :217-253."""
    act = activation_fn(cfg.activation)
    x = apply_norm(x, sd[pre + "norm.weight"], sd.get(pre + "norm.bias"), 1e-5, cfg.norm_type == "layernorm")
    x = F.linear(x, sd[pre + "fc1.weight"], sd.get(pre + "fc1.bias"))
    if cfg.is_glu:
        x_linear, x_glu = x.chunk(2, dim=-1)
        if cfg.glu_limit is not None:
            x_linear = x_linear.clamp(min=-cfg.glu_limit, max=cfg.glu_limit)
            x_glu = x_glu.clamp(min=None, max=cfg.glu_limit)
        if cfg.glu_extra_bias is not None:
            x_linear = x_linear + cfg.glu_extra_bias
        x = act(x_glu) * x_linear
    else:
        x = act(x)
    x = F.dropout(x, p=cfg.hidden_dropout, training=training)
    x = F.linear(x, sd[pre + "fc2.weight"], sd.get(pre + "fc2.bias"))
    return F.dropout(x, p=cfg.hidden_dropout, training=training)


def select_residual_subset(x: Tensor, rate: float, training: bool):
    """vit/transformer.py:14-27."""
    B = x.shape[0]
    if not training or rate <= 0.0 or B <= 1:
        return x, None, 1.0
    keep = max(1, min(B, int(B * (1.0 - rate))))
    if keep == B:
        return x, None, 1.0
    idx = torch.randperm(B, device=x.device)[:keep]
    return x.index_select(0, idx), idx, float(B / keep)


def merge_residual_subset(x: Tensor, o: Tensor, idx: Tensor | None, scale: float) -> Tensor:
    """vit/transformer.py:30-38."""
    if idx is None:
        return x + o
    return x.flatten(1).index_add(0, idx, o.flatten(1), alpha=scale).view_as(x)


def subset_batched_rope(rope: Tensor | None, idx: Tensor | None, full_batch: int) -> Tensor | None:
    """vit/transformer.py:41-46: a per-sample table [2, B, 1, N, Dh] follows the drop-path subset."""
    if rope is None or idx is None:
        return rope
    if rope.ndim == 5 and rope.shape[1] == full_batch:
        return rope.index_select(1, idx)
    return rope


def apply_token_mask(mask: Tensor, x: Tensor) -> Tensor:
    """vit/tokens.py:58-94 with fill_value=None and padding 0: kept tokens (mask True) in order, samples with fewer
    kept tokens padded with zeros at the end."""
    B, L, D = x.shape
    counts = mask.sum(dim=-1)
    width = int(counts.max())
    out = x.new_zeros((B, width, D))
    for b in range(B):
        rows = x[b][mask[b]]
        out[b, : rows.shape[0]] = rows
    return out


def encoder_layer(cfg: OracleConfig, sd: dict, i: int, x: Tensor, rope: Tensor | None, training: bool) -> Tensor:
    """vit/transformer.py:146-157; layer scale = vit/layer_scale.py:8-13 (out-of-place form)."""
    pre = f"blocks.{i}."
    xr, idx, scale = select_residual_subset(x, cfg.drop_path_rate, training)
    o = self_attention(cfg, sd, pre + "self_attention.", xr, subset_batched_rope(rope, idx, x.shape[0]), training)
    if cfg.layer_scale is not None:
        o = o * sd[pre + "layer_scale_attn.gamma"]
    x = merge_residual_subset(x, o, idx, scale)
    xr, idx, scale = select_residual_subset(x, cfg.drop_path_rate, training)
    o = norm_mlp(cfg, sd, pre + "mlp.", xr, training)
    if cfg.layer_scale is not None:
        o = o * sd[pre + "layer_scale_mlp.gamma"]
    return merge_residual_subset(x, o, idx, scale)


def patch_embed(cfg: OracleConfig, sd: dict, x: Tensor, training: bool) -> Tensor:
    """vit/patch_embed.py:52-59 (2-D) and :101-109 (3-D)."""
    p = tuple(cfg.patch_size)
    conv = F.conv2d if len(p) == 2 else F.conv3d
    y = conv(x, sd["stem.patch.weight"], sd.get("stem.patch.bias"), stride=p).flatten(2).transpose(1, 2)
    dims = tuple(s // q for s, q in zip(x.shape[2:], p))
    if cfg.pos_enc == "fourier":
        return y + fourier_position(dims, sd["stem.pos_enc.w_fourier"], sd["stem.pos_enc.w_fc1.weight"],
                                    sd.get("stem.pos_enc.w_fc1.bias"))
    if cfg.pos_enc == "learnable":
        size = tuple(s // q for s, q in zip(cfg.img_size, p))
        return y + learnable_position(dims, size, sd["stem.pos_enc.positions"], 0.1, training)  # dropout 0.1: pos_enc.py:24
    return y


def vit_forward(cfg: OracleConfig, sd: dict, x: Tensor, *, training: bool = False, output_norm: bool = True,
                rope_seed: int | None = None, mask: Tensor | None = None) -> Tensor:
    """vit/vit.py:534-561; returns ``dense_features`` [B, S, D].  ``mask`` [B, L] bool (True = kept token) is the
    masked-token branch: vit/vit.py:545 and :522-527."""
    ln = cfg.norm_type == "layernorm"
    dims = tuple(s // q for s, q in zip(x.shape[2:], cfg.patch_size))
    x = patch_embed(cfg, sd, x, training)
    if cfg.patch_embed_normalization:
        x = apply_norm(x, sd["patch_embed_norm.weight"], sd.get("patch_embed_norm.bias"), None, ln)
    B = x.shape[0]
    if mask is not None:
        x = apply_token_mask(mask, x)
    x = torch.cat([sd["cls_tokens"].expand(B, -1, -1), sd["register_tokens"].expand(B, -1, -1), x], dim=1)
    rope = None
    if cfg.pos_enc == "rope":
        rope = rope_table(dims[0], dims[1], sd["rope.periods"], cfg.dtype, training=training,
                          shift=cfg.rope_shift_coords, jitter=cfg.rope_jitter_coords,
                          rescale=cfg.rope_rescale_coords, rope_seed=rope_seed)
        if mask is not None:
            sin = apply_token_mask(mask, rope[0][None].expand(B, -1, -1))
            cos = apply_token_mask(mask, rope[1][None].expand(B, -1, -1))
            rope = torch.stack([sin[:, None], cos[:, None]], dim=0)
    for i in range(cfg.depth):
        if cfg.activation_checkpointing and training:
            x = checkpoint(lambda t, r, _i=i: encoder_layer(cfg, sd, _i, t, r, training), x, rope, use_reentrant=False)
        else:
            x = encoder_layer(cfg, sd, i, x, rope, training)
    if output_norm:
        x = apply_norm(x, sd["output_norm.weight"], sd.get("output_norm.bias"), None, ln)
    return x


# --------------------------------------------------------------------------------------------
# heads (vit/head.py, vit/attention.py:112-162, :218-280, :594-667)
# --------------------------------------------------------------------------------------------
def cross_attention(cfg: OracleConfig, sd: dict, pre: str, q: Tensor, kv: Tensor, num_heads: int, training: bool = False,
                    attention_dropout: float = 0.0, hidden_dropout: float = 0.0, rope_q: Tensor | None = None,
                    rope_k: Tensor | None = None) -> Tensor:
    """attention_q_kv_packed, vit/attention.py:218-280: norm on the query side only."""
    ln = cfg.norm_type == "layernorm"
    D = q.shape[-1]
    dh = D // num_heads
    qn = apply_norm(q, sd[pre + "norm.weight"], sd.get(pre + "norm.bias"), 1e-5, ln)
    qp = F.linear(qn, sd[pre + "q_proj.weight"], sd.get(pre + "q_proj.bias"))
    k, v = F.linear(kv, sd[pre + "kv_proj.weight"], sd.get(pre + "kv_proj.bias")).chunk(2, dim=-1)
    B, Lq, _ = qp.shape
    Lk = k.shape[1]
    qp = qp.view(B, Lq, -1, dh).transpose(1, 2)
    k = k.view(B, Lk, -1, dh).transpose(1, 2)
    v = v.view(B, Lk, -1, dh).transpose(1, 2)
    if cfg.qk_normalization:
        qp = apply_norm(qp, sd[pre + "q_norm.weight"], sd.get(pre + "q_norm.bias"), 1e-5, ln)
        k = apply_norm(k, sd[pre + "k_norm.weight"], sd.get(pre + "k_norm.bias"), 1e-5, ln)
    if rope_q is not None:
        qp = apply_rope(qp, rope_q)
    if rope_k is not None:
        k = apply_rope(k, rope_k)
    o = F.scaled_dot_product_attention(qp, k, v, attn_mask=None, dropout_p=attention_dropout if training else 0.0,
                                       is_causal=False, enable_gqa=True)
    o = o.transpose(1, 2).reshape(B, Lq, D)
    o = F.linear(o, sd[pre + "out_proj.weight"], sd.get(pre + "out_proj.bias"))
    return F.dropout(o, p=hidden_dropout, training=training)


def self_attention_weights(cfg: OracleConfig, sd: dict, pre: str, x: Tensor, rope: Tensor | None = None) -> Tensor:
    """attention_weights_qkv_packed, vit/attention.py:281-321."""
    q, k, _ = project_qkv(cfg, sd, pre, x, rope)
    return (q @ k.mT * (cfg.head_dim ** -0.5)).softmax(dim=-1)


def vit_attention_weights(cfg: OracleConfig, sd: dict, x: Tensor) -> dict:
    """ViT.forward_attention_weights, vit/vit.py:575-596."""
    ln = cfg.norm_type == "layernorm"
    dims = tuple(s // q for s, q in zip(x.shape[2:], cfg.patch_size))
    x = patch_embed(cfg, sd, x, False)
    if cfg.patch_embed_normalization:
        x = apply_norm(x, sd["patch_embed_norm.weight"], sd.get("patch_embed_norm.bias"), None, ln)
    B = x.shape[0]
    x = torch.cat([sd["cls_tokens"].expand(B, -1, -1), sd["register_tokens"].expand(B, -1, -1), x], dim=1)
    out = {}
    for i in range(cfg.depth):
        w = self_attention_weights(cfg, sd, f"blocks.{i}.self_attention.", x)
        Bw, H, Lq, _ = w.shape
        out[f"layer_{i}"] = w[..., cfg.num_register_tokens:].reshape(Bw, H, Lq, *dims).contiguous()
        x = encoder_layer(cfg, sd, i, x, None, False)
    return out


def attentive_pool_weights(cfg: OracleConfig, sd: dict, pre: str, x: Tensor, num_heads: int) -> Tensor:
    """AttentivePool.forward_weights -> attention_weights_q_kv_packed, vit/attention.py:324-373, :658-659."""
    ln = cfg.norm_type == "layernorm"
    cp = pre + "cross_attention."
    D = x.shape[-1]
    dh = D // num_heads
    q = sd[pre + "query"].expand(x.shape[0], -1, -1)
    qn = apply_norm(q, sd[cp + "norm.weight"], sd.get(cp + "norm.bias"), 1e-5, ln)
    qp = F.linear(qn, sd[cp + "q_proj.weight"], sd.get(cp + "q_proj.bias"))
    k, _ = F.linear(x, sd[cp + "kv_proj.weight"], sd.get(cp + "kv_proj.bias")).chunk(2, dim=-1)
    B, Lq, _ = qp.shape
    qp = qp.view(B, Lq, -1, dh).transpose(1, 2)
    k = k.view(B, k.shape[1], -1, dh).transpose(1, 2)
    if cfg.qk_normalization:
        qp = apply_norm(qp, sd[cp + "q_norm.weight"], sd.get(cp + "q_norm.bias"), 1e-5, ln)
        k = apply_norm(k, sd[cp + "k_norm.weight"], sd.get(cp + "k_norm.bias"), 1e-5, ln)
    return (qp @ k.mT * (dh ** -0.5)).softmax(dim=-1)


def decoder_layer(cfg: OracleConfig, sd: dict, pre: str, x: Tensor, kv: Tensor, training: bool, self_attn: bool,
                  rope_q: Tensor | None = None, rope_k: Tensor | None = None) -> Tensor:
    """TransformerDecoderLayer.forward (vit/transformer.py:262-284) when ``self_attn``, else
    CrossAttentionTransformer.forward (:367-384).  ``pre`` is the prefix of the layer's keys in ``sd``."""
    def scaled(o, name):
        return o * sd[pre + name + ".gamma"] if cfg.layer_scale is not None else o

    B = x.shape[0]
    if self_attn:
        xr, idx, scale = select_residual_subset(x, cfg.drop_path_rate, training)
        o = self_attention(cfg, sd, pre + "self_attention.", xr, subset_batched_rope(rope_q, idx, B), training)
        x = merge_residual_subset(x, scaled(o, "layer_scale_attn"), idx, scale)
    xr, idx, scale = select_residual_subset(x, cfg.drop_path_rate, training)
    kvr = kv if idx is None or kv.shape[0] != B else kv.index_select(0, idx)
    o = cross_attention(cfg, sd, pre + "cross_attention.", xr, kvr, cfg.num_attention_heads, training,
                        cfg.attention_dropout, cfg.hidden_dropout, subset_batched_rope(rope_q, idx, B),
                        subset_batched_rope(rope_k, idx, B))
    x = merge_residual_subset(x, scaled(o, "layer_scale_cross"), idx, scale)
    xr, idx, scale = select_residual_subset(x, cfg.drop_path_rate, training)
    o = norm_mlp(cfg, sd, pre + "mlp.", xr, training)
    return merge_residual_subset(x, scaled(o, "layer_scale_mlp"), idx, scale)


def attentive_pool_head(cfg: OracleConfig, sd: dict, pre: str, x: Tensor, num_heads: int) -> Tensor:
    """AttentivePoolHead.forward, vit/head.py:247-248 with AttentivePool.forward vit/attention.py:648-652 and
    Head.forward vit/head.py:204-206 (dropout probabilities 0)."""
    query = sd[pre + "pool.query"].expand(x.shape[0], -1, -1)
    y = cross_attention(cfg, sd, pre + "pool.cross_attention.", query, x, num_heads)
    if y.shape[1] == 1:
        y = y.squeeze(1)
    return F.linear(y, sd[pre + "proj.proj.weight"], sd.get(pre + "proj.proj.bias"))


def linear_head(sd: dict, pre: str, x: Tensor) -> Tensor:
    """Head.forward, vit/head.py:204-206."""
    return F.linear(x, sd[pre + "proj.weight"], sd.get(pre + "proj.bias"))


# --------------------------------------------------------------------------------------------
# state construction (used by tests and the CPU baseline; the product has its own initialiser)
# --------------------------------------------------------------------------------------------
def init_state(cfg: OracleConfig, seed: int = 0, device=None, perturb_residual_outputs: bool = True) -> dict:
    """Parameter dictionary with the reference's key layout (vit/vit.py:255-323) and init rules
    (vit/initialization.py:6-36; zero residual outputs vit/transformer.py:57-66,132).  With
    ``perturb_residual_outputs`` the zero-initialised out_proj/fc2 get N(0, 0.02) so that parity checks
    are not vacuous (the reference's own tests do this: tests/test_transformer.py:33-36)."""
    g = torch.Generator().manual_seed(seed)
    dt, D, F1 = cfg.dtype, cfg.hidden_size, cfg.ffn_hidden_size

    def tn(*shape):
        t = torch.empty(*shape, dtype=torch.float32)
        torch.nn.init.trunc_normal_(t, mean=0.0, std=0.02, a=-0.04, b=0.04, generator=g)
        return t.to(dt)

    def nrm(*shape, std=0.02):
        return (torch.randn(*shape, generator=g) * std).to(dt)

    ln = cfg.norm_type == "layernorm"
    sd: dict[str, Tensor] = {}
    sd["register_tokens"] = tn(1, cfg.num_register_tokens, D)
    sd["cls_tokens"] = tn(1, cfg.num_cls_tokens, D)
    sd["stem.patch.weight"] = tn(D, cfg.in_channels, *cfg.patch_size)
    sd["stem.patch.bias"] = torch.zeros(D, dtype=dt)
    dims = tuple(s // q for s, q in zip(cfg.img_size, cfg.patch_size))
    if cfg.pos_enc == "fourier":
        sd["stem.pos_enc.w_fourier"] = nrm(len(dims), D // 2, std=1.0)
        sd["stem.pos_enc.w_fc1.weight"] = tn(D, D)
        sd["stem.pos_enc.w_fc1.bias"] = torch.zeros(D, dtype=dt)
    elif cfg.pos_enc == "learnable":
        sd["stem.pos_enc.positions"] = tn(math.prod(dims), D)
    if cfg.patch_embed_normalization:
        sd["patch_embed_norm.weight"] = torch.ones(D, dtype=dt)
        if ln:
            sd["patch_embed_norm.bias"] = torch.zeros(D, dtype=dt)
    if cfg.pos_enc == "rope":
        sd["rope.periods"] = rope_periods(cfg.head_dim, cfg.rope_base, dt)
    for i in range(cfg.depth):
        a, m = f"blocks.{i}.self_attention.", f"blocks.{i}.mlp."
        for pre in (a, m):
            sd[pre + "norm.weight"] = torch.ones(D, dtype=dt)
            if ln:
                sd[pre + "norm.bias"] = torch.zeros(D, dtype=dt)
        sd[a + "qkv_proj.weight"] = tn(3 * D, D)
        sd[a + "out_proj.weight"] = nrm(D, D) if perturb_residual_outputs else torch.zeros(D, D, dtype=dt)
        if cfg.attention_bias:
            sd[a + "qkv_proj.bias"] = torch.zeros(3 * D, dtype=dt)
            sd[a + "out_proj.bias"] = torch.zeros(D, dtype=dt)
        if cfg.qk_normalization:
            for n in ("q_norm", "k_norm"):
                sd[a + n + ".weight"] = torch.ones(cfg.head_dim, dtype=dt)
                if ln:
                    sd[a + n + ".bias"] = torch.zeros(cfg.head_dim, dtype=dt)
        sd[m + "fc1.weight"] = tn((2 if cfg.is_glu else 1) * F1, D)
        sd[m + "fc2.weight"] = nrm(D, F1) if perturb_residual_outputs else torch.zeros(D, F1, dtype=dt)
        if cfg.mlp_bias:
            sd[m + "fc1.bias"] = torch.zeros((2 if cfg.is_glu else 1) * F1, dtype=dt)
            sd[m + "fc2.bias"] = torch.zeros(D, dtype=dt)
        if cfg.layer_scale is not None:
            sd[f"blocks.{i}.layer_scale_attn.gamma"] = torch.full((D,), cfg.layer_scale, dtype=dt)
            sd[f"blocks.{i}.layer_scale_mlp.gamma"] = torch.full((D,), cfg.layer_scale, dtype=dt)
    sd["output_norm.weight"] = torch.ones(D, dtype=dt)
    if ln:
        sd["output_norm.bias"] = torch.zeros(D, dtype=dt)
    if device is not None:
        sd = {k: v.to(device) for k, v in sd.items()}
    return sd


def randomize_state(sd: dict, seed: int = 1, scale: float = 1.0) -> dict:
    """Make every parameter 'interesting' for parity: biases, norm weights, gammas get noise too."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        if k == "rope.periods" or v.numel() == 0:
            out[k] = v
            continue
        noise = torch.randn(v.shape, generator=g).to(v.dtype).to(v.device)
        if k.endswith("bias"):
            out[k] = v + 0.02 * scale * noise
        elif k.endswith("norm.weight") or k.endswith("gamma"):
            out[k] = v * (1 + 0.1 * scale * noise)
        else:
            out[k] = v
    return out


def run_fwd_bwd(cfg: OracleConfig, sd: dict, x: Tensor, *, training: bool = True, autocast: torch.dtype | None = None,
                loss: str = "square_mean", seed: int | None = None, x_requires_grad: bool = True,
                output_norm: bool = True, mask: Tensor | None = None) -> dict[str, Any]:
    """Forward + backward with the loss of the reference's own parity test (tests/test_vit.py:841-846:
    ``dense_features.float().square().mean()``) or its benchmark (benchmark/benchmark.py:203-209: ``mean``)."""
    params = {k: v.detach().clone().requires_grad_(v.is_floating_point() and k != "rope.periods" and v.numel() > 0)
              for k, v in sd.items()}
    xi = x.detach().clone().requires_grad_(x_requires_grad)
    if seed is not None:
        torch.manual_seed(seed)
    ctx = torch.autocast(x.device.type, dtype=autocast) if autocast is not None else torch.autocast(x.device.type, enabled=False)
    with ctx:
        y = vit_forward(cfg, params, xi, training=training, output_norm=output_norm, mask=mask)
    val = y.float().square().mean() if loss == "square_mean" else y.mean()
    val.backward()
    grads = {k: p.grad for k, p in params.items() if p.requires_grad and p.grad is not None}
    return {"out": y.detach(), "loss": val.detach(), "x_grad": xi.grad, "grads": grads}
