"""Self-attention of the encoder block (reference: vit/attention.py:71-111,163-215,375-449).

``SelfAttention`` holds the same parameters under the same names as the reference; its forward unpacks them
into ``attention_qkv_packed``, which runs norm -> QKV GEMM -> RoPE -> flash attention -> out_proj GEMM as one
autograd function over libvitk kernels.  ``CrossAttention`` / ``AttentivePool`` (vit/attention.py:451-667) and the
``forward_weights`` diagnostics (vit/attention.py:281-373) run through ``ops.CrossAttentionFn`` and
``vitk_attn_weights``."""
from __future__ import annotations

from typing import Any, Optional

import torch
import torch.nn as nn
from torch import Tensor

from . import ops
from .initialization import init_linear, trunc_normal_
from .norm import get_norm_bias, is_layer_norm, make_norm


def _sublayer_opts(*, head_dim, eps, layernorm, attention_dropout, dropout, training, residual, alpha=1.0,
                   qk_eps=1e-5, qk_layernorm=False):
    return dict(head_dim=head_dim, eps=float(eps), layernorm=bool(layernorm), attention_dropout=float(attention_dropout),
                dropout=float(dropout), training=bool(training), residual=residual, alpha=float(alpha),
                qk_eps=float(qk_eps), qk_layernorm=bool(qk_layernorm))


def attention_qkv_packed(x, w_in, b_in, w_norm, b_norm, use_layer_norm, head_dim, w_out, b_out, attn_mask, eps,
                         q_norm_weight, q_norm_bias, k_norm_weight, k_norm_bias, qk_use_layer_norm, qk_eps,
                         qk_normalization, attention_dropout, dropout, training, rope=None) -> Tensor:
    """Same positional signature as the reference function (vit/attention.py:164-189); returns the projected
    attention output (without the residual)."""
    if attn_mask is not None:
        raise NotImplementedError("vit_amd: attention masks are not on the ViT encoder path (vit/transformer.py:151)")
    if qk_normalization:
        assert q_norm_weight is not None and k_norm_weight is not None
    else:
        q_norm_weight = q_norm_bias = k_norm_weight = k_norm_bias = None
    opt = _sublayer_opts(head_dim=head_dim, eps=eps, layernorm=use_layer_norm, attention_dropout=attention_dropout,
                         dropout=dropout, training=training, residual=False, qk_eps=qk_eps, qk_layernorm=qk_use_layer_norm)
    return ops.AttentionSublayerFn.apply(x, w_norm, b_norm, w_in, b_in, w_out, b_out, None, rope, None, q_norm_weight,
                                         q_norm_bias, k_norm_weight, k_norm_bias, opt)


class SelfAttention(nn.Module):
    def __init__(self, hidden_size: int, num_attention_heads: int, hidden_dropout: float = 0.1,
                 attention_dropout: float = 0.1, bias: bool = True, eps: float = 1e-5,
                 qkv_quantization_config: Any = None, out_quantization_config: Any = None, device=None, dtype=None,
                 norm_type: str = "rmsnorm", qk_normalization: bool = False):
        super().__init__()
        if qkv_quantization_config is not None or out_quantization_config is not None:
            raise NotImplementedError("vit_amd: torchao quantization is outside the encoder hot path")
        kw = {"device": device, "dtype": dtype}
        self.norm = make_norm(hidden_size, norm_type, eps=eps, **kw)
        self._use_layer_norm = is_layer_norm(norm_type)
        self.qkv_proj = nn.Linear(hidden_size, 3 * hidden_size, bias=bias, **kw)
        self.out_proj = nn.Linear(hidden_size, hidden_size, bias=bias, **kw)
        self.dropout = nn.Dropout(hidden_dropout)
        self.attention_dropout = nn.Dropout(attention_dropout)
        self._head_dim = hidden_size // num_attention_heads
        self._qk_normalization = qk_normalization
        if qk_normalization:
            self.q_norm = make_norm(self._head_dim, norm_type, eps=eps, **kw)
            self.k_norm = make_norm(self._head_dim, norm_type, eps=eps, **kw)
        else:
            self.q_norm = self.k_norm = None
        self.reset_parameters()

    def reset_parameters(self) -> None:
        init_linear(self.qkv_proj)
        init_linear(self.out_proj)

    def apply_quantization(self, qkv_quantization_config: Any = None, out_quantization_config: Any = None) -> None:
        if qkv_quantization_config is not None or out_quantization_config is not None:
            raise NotImplementedError("vit_amd: torchao quantization is outside the encoder hot path")

    def fused_residual(self, x: Tensor, rope: Optional[Tensor], gamma: Optional[Tensor], keep_idx: Optional[Tensor],
                       alpha: float) -> Tensor:
        """x (+) alpha * gamma * attention(x[keep_idx]) with residual, layer scale and drop-path merge fused."""
        qw, qb, kw_, kb, qk_eps = self._qk_norm_inputs()
        opt = _sublayer_opts(head_dim=self._head_dim, eps=self.norm.eps or 1e-5, layernorm=self._use_layer_norm,
                             attention_dropout=self.attention_dropout.p, dropout=self.dropout.p, training=self.training,
                             residual=True, alpha=alpha, qk_eps=qk_eps, qk_layernorm=self._use_layer_norm)
        return ops.AttentionSublayerFn.apply(x, self.norm.weight, get_norm_bias(self.norm), self.qkv_proj.weight,
                                             self.qkv_proj.bias, self.out_proj.weight, self.out_proj.bias, gamma, rope,
                                             keep_idx, qw, qb, kw_, kb, opt)

    def _qk_norm_inputs(self):
        """(q weight, q bias, k weight, k bias, eps), all None / 1e-5 without qk_normalization (vit/attention.py:61-68)."""
        if self.q_norm is None or self.k_norm is None:
            return None, None, None, None, 1e-5
        return (self.q_norm.weight, get_norm_bias(self.q_norm), self.k_norm.weight, get_norm_bias(self.k_norm),
                self.q_norm.eps or 1e-5)

    def forward_weights(self, x: Tensor, rope: Optional[Tensor] = None) -> Tensor:
        """Attention probabilities [B, H, S, S] (vit/attention.py:455-477)."""
        qw, qb, kw_, kb, qk_eps = self._qk_norm_inputs()
        opt = _sublayer_opts(head_dim=self._head_dim, eps=self.norm.eps or 1e-5, layernorm=self._use_layer_norm,
                             attention_dropout=0.0, dropout=0.0, training=False, residual=False, qk_eps=qk_eps,
                             qk_layernorm=self._use_layer_norm)
        return ops.self_attention_weights(x, self.norm.weight, get_norm_bias(self.norm), self.qkv_proj.weight,
                                          self.qkv_proj.bias, qw, qb, kw_, kb, rope, opt)

    def forward(self, x: Tensor, attn_mask: Optional[Tensor] = None, rope: Optional[Tensor] = None) -> Tensor:
        return attention_qkv_packed(
            x, self.qkv_proj.weight, self.qkv_proj.bias, self.norm.weight, get_norm_bias(self.norm), self._use_layer_norm,
            self._head_dim, self.out_proj.weight, self.out_proj.bias, attn_mask, self.norm.eps or 1e-5,
            *self._qk_norm_inputs()[:4], self._use_layer_norm, self._qk_norm_inputs()[4], self._qk_normalization,
            self.attention_dropout.p, self.dropout.p, self.training, rope)


def attention_q_kv_packed(q, kv, w_q, b_q, w_kv, b_kv, w_norm, b_norm, use_layer_norm, head_dim, w_out, b_out, attn_mask,
                          eps, q_norm_weight, q_norm_bias, k_norm_weight, k_norm_bias, qk_use_layer_norm, qk_eps,
                          qk_normalization, attention_dropout, dropout, training, rope_q=None, rope_k=None) -> Tensor:
    """Same positional signature as the reference function (vit/attention.py:218-280)."""
    if attn_mask is not None:
        raise NotImplementedError("vit_amd: attention masks are not built")
    if qk_normalization:
        assert q_norm_weight is not None and k_norm_weight is not None
    else:
        q_norm_weight = q_norm_bias = k_norm_weight = k_norm_bias = None
    opt = _sublayer_opts(head_dim=head_dim, eps=eps, layernorm=use_layer_norm, attention_dropout=attention_dropout,
                         dropout=dropout, training=training, residual=False, qk_eps=qk_eps, qk_layernorm=qk_use_layer_norm)
    return ops.CrossAttentionFn.apply(q, kv, w_norm, b_norm, w_q, b_q, w_kv, b_kv, w_out, b_out, q_norm_weight, q_norm_bias,
                                      k_norm_weight, k_norm_bias, None, None, rope_q, rope_k, opt)


class CrossAttention(nn.Module):
    """Queries from one sequence, keys and values from another (vit/attention.py:480-592); pre-norm on the query side."""

    def __init__(self, hidden_size: int, num_attention_heads: int, hidden_dropout: float = 0.1,
                 attention_dropout: float = 0.1, bias: bool = True, eps: float = 1e-5,
                 qkv_quantization_config: Any = None, out_quantization_config: Any = None, device=None, dtype=None,
                 norm_type: str = "rmsnorm", qk_normalization: bool = False):
        super().__init__()
        if qkv_quantization_config is not None or out_quantization_config is not None:
            raise NotImplementedError("vit_amd: torchao quantization is outside the encoder hot path")
        kw = {"device": device, "dtype": dtype}
        self.norm = make_norm(hidden_size, norm_type, eps=eps, **kw)
        self._use_layer_norm = is_layer_norm(norm_type)
        self.q_proj = nn.Linear(hidden_size, hidden_size, bias=bias, **kw)
        self.kv_proj = nn.Linear(hidden_size, 2 * hidden_size, bias=bias, **kw)
        self.out_proj = nn.Linear(hidden_size, hidden_size, bias=bias, **kw)
        self.dropout = nn.Dropout(hidden_dropout)
        self.attention_dropout = nn.Dropout(attention_dropout)
        self._head_dim = hidden_size // num_attention_heads
        self._qk_normalization = qk_normalization
        if qk_normalization:
            self.q_norm = make_norm(self._head_dim, norm_type, eps=eps, **kw)
            self.k_norm = make_norm(self._head_dim, norm_type, eps=eps, **kw)
        else:
            self.q_norm = self.k_norm = None
        self.reset_parameters()

    def reset_parameters(self) -> None:
        init_linear(self.q_proj)
        init_linear(self.kv_proj)
        init_linear(self.out_proj)

    def apply_quantization(self, qkv_quantization_config: Any = None, out_quantization_config: Any = None) -> None:
        if qkv_quantization_config is not None or out_quantization_config is not None:
            raise NotImplementedError("vit_amd: torchao quantization is outside the encoder hot path")

    def _qk_norm_inputs(self):
        if self.q_norm is None or self.k_norm is None:
            return None, None, None, None, 1e-5
        return (self.q_norm.weight, get_norm_bias(self.q_norm), self.k_norm.weight, get_norm_bias(self.k_norm),
                self.q_norm.eps or 1e-5)

    def fused_residual(self, x: Tensor, kv: Tensor, gamma: Optional[Tensor], keep_idx: Optional[Tensor], alpha: float,
                       rope_q: Optional[Tensor] = None, rope_k: Optional[Tensor] = None) -> Tensor:
        """x (+) alpha * gamma * cross_attention(x[keep_idx], kv[keep_idx]): residual, layer scale and the drop-path merge
        in the out_proj epilogue (the cross sub-layer of vit/transformer.py:262-284)."""
        qw, qb, kw_, kb, qk_eps = self._qk_norm_inputs()
        opt = _sublayer_opts(head_dim=self._head_dim, eps=self.norm.eps or 1e-5, layernorm=self._use_layer_norm,
                             attention_dropout=self.attention_dropout.p, dropout=self.dropout.p, training=self.training,
                             residual=True, alpha=alpha, qk_eps=qk_eps, qk_layernorm=self._use_layer_norm)
        return ops.CrossAttentionFn.apply(x, kv, self.norm.weight, get_norm_bias(self.norm), self.q_proj.weight,
                                          self.q_proj.bias, self.kv_proj.weight, self.kv_proj.bias, self.out_proj.weight,
                                          self.out_proj.bias, qw, qb, kw_, kb, gamma, keep_idx, rope_q, rope_k, opt)

    def forward(self, q: Tensor, kv: Tensor, attn_mask: Optional[Tensor] = None, rope_q: Optional[Tensor] = None,
                rope_k: Optional[Tensor] = None) -> Tensor:
        qw, qb, kw_, kb, qk_eps = self._qk_norm_inputs()
        return attention_q_kv_packed(
            q, kv, self.q_proj.weight, self.q_proj.bias, self.kv_proj.weight, self.kv_proj.bias, self.norm.weight,
            get_norm_bias(self.norm), self._use_layer_norm, self._head_dim, self.out_proj.weight, self.out_proj.bias,
            attn_mask, self.norm.eps or 1e-5, qw, qb, kw_, kb, self._use_layer_norm, qk_eps, self._qk_normalization,
            self.attention_dropout.p, self.dropout.p, self.training, rope_q, rope_k)

    def forward_weights(self, q: Tensor, kv: Tensor, rope_q: Optional[Tensor] = None,
                        rope_k: Optional[Tensor] = None) -> Tensor:
        """Attention probabilities [B, H, Lq, Lk] (vit/attention.py:568-592)."""
        qw, qb, kw_, kb, qk_eps = self._qk_norm_inputs()
        opt = _sublayer_opts(head_dim=self._head_dim, eps=self.norm.eps or 1e-5, layernorm=self._use_layer_norm,
                             attention_dropout=0.0, dropout=0.0, training=False, residual=False, qk_eps=qk_eps,
                             qk_layernorm=self._use_layer_norm)
        return ops.cross_attention_weights(q, kv, self.norm.weight, get_norm_bias(self.norm), self.q_proj.weight,
                                           self.q_proj.bias, self.kv_proj.weight, self.kv_proj.bias, qw, qb, kw_, kb,
                                           rope_q, rope_k, opt)


class AttentivePool(nn.Module):
    """``num_queries`` learned queries attend over the tokens (vit/attention.py:594-667): [B, L, D] -> [B, D], or
    [B, num_queries, D] when there are several."""

    def __init__(self, hidden_size: int, num_attention_heads: int, num_queries: int = 1, hidden_dropout: float = 0.0,
                 attention_dropout: float = 0.0, bias: bool = True, device=None, dtype=None, norm_type: str = "rmsnorm",
                 qk_normalization: bool = False):
        super().__init__()
        if hidden_size % num_attention_heads != 0:
            raise ValueError(
                f"hidden_size ({hidden_size}) must be divisible by num_attention_heads ({num_attention_heads})")
        if num_queries < 1:
            raise ValueError(f"num_queries must be positive, got {num_queries}")
        self.query = nn.Parameter(torch.empty(1, num_queries, hidden_size, device=device, dtype=dtype))
        self.cross_attention = CrossAttention(hidden_size, num_attention_heads, hidden_dropout=hidden_dropout,
                                              attention_dropout=attention_dropout, bias=bias, device=device, dtype=dtype,
                                              norm_type=norm_type, qk_normalization=qk_normalization)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        trunc_normal_(self.query)
        self.cross_attention.reset_parameters()

    def forward(self, x: Tensor) -> Tensor:
        y = self.cross_attention(self.query.expand(x.shape[0], -1, -1), x)
        return y.squeeze(1) if y.shape[1] == 1 else y

    def forward_weights(self, x: Tensor) -> Tensor:
        """[B, H, num_queries, L] (vit/attention.py:658-659)."""
        return self.cross_attention.forward_weights(self.query.expand(x.shape[0], -1, -1), x)
