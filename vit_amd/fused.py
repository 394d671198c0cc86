"""Pre-norm MLP of the encoder block (reference: vit/fused.py:87-258): norm -> fc1 -> activation (plain or
GLU) -> dropout -> fc2 -> dropout, as one autograd function over libvitk kernels."""
from __future__ import annotations

from typing import Any, Optional

import torch.nn as nn
from torch import Tensor

from . import ops
from .helpers import activation_code, get_activation
from .initialization import init_linear
from .norm import get_norm_bias, is_layer_norm, make_norm


def _mlp_opts(*, glu, activation, eps, layernorm, dropout, training, limit, extra_bias, residual, alpha=1.0):
    return dict(glu=bool(glu), act=activation_code(activation), eps=float(eps), layernorm=bool(layernorm),
                dropout=float(dropout), training=bool(training), limit=limit, extra_bias=extra_bias, residual=residual,
                alpha=float(alpha))


def norm_mlp(x, fc1_weight, fc1_bias, fc2_weight, fc2_bias, norm_weight, norm_bias, use_layer_norm, activation, eps,
             dropout, training) -> Tensor:
    """Reference signature vit/fused.py:96-111."""
    opt = _mlp_opts(glu=False, activation=activation, eps=eps, layernorm=use_layer_norm, dropout=dropout,
                    training=training, limit=None, extra_bias=None, residual=False)
    return ops.MlpSublayerFn.apply(x, norm_weight, norm_bias, fc1_weight, fc1_bias, fc2_weight, fc2_bias, None, None, opt)


def norm_mlp_glu(x, fc1_weight, fc1_bias, fc2_weight, fc2_bias, norm_weight, norm_bias, use_layer_norm, activation, eps,
                 dropout, training, limit=None, extra_bias=None) -> Tensor:
    """Reference signature vit/fused.py:131-148: fc1 output = [x_linear | x_glu], h = act(x_glu) * x_linear."""
    opt = _mlp_opts(glu=True, activation=activation, eps=eps, layernorm=use_layer_norm, dropout=dropout,
                    training=training, limit=limit, extra_bias=extra_bias, residual=False)
    return ops.MlpSublayerFn.apply(x, norm_weight, norm_bias, fc1_weight, fc1_bias, fc2_weight, fc2_bias, None, None, opt)


class NormMLP(nn.Module):
    def __init__(self, hidden_size: int, ffn_hidden_size: int, bias: bool = True, activation: str = "gelu",
                 eps: float = 1e-5, dropout: float = 0.1, limit: Optional[float] = None,
                 extra_bias: Optional[float] = None, quantization_config: Any = None, device=None, dtype=None,
                 norm_type: str = "rmsnorm"):
        super().__init__()
        if quantization_config is not None:
            raise NotImplementedError("vit_amd: torchao quantization is outside the encoder hot path")
        kw = {"device": device, "dtype": dtype}
        self.norm = make_norm(hidden_size, norm_type, eps=eps, **kw)
        self._use_layer_norm = is_layer_norm(norm_type)
        self._is_glu = activation.endswith("glu")
        self.fc1 = nn.Linear(hidden_size, (2 if self._is_glu else 1) * ffn_hidden_size, bias=bias, **kw)
        self.fc2 = nn.Linear(ffn_hidden_size, hidden_size, bias=bias, **kw)
        self.dropout = nn.Dropout(dropout)
        self.activation = get_activation(activation)
        self.limit = limit
        self.extra_bias = extra_bias
        self.reset_parameters()

    def reset_parameters(self) -> None:
        init_linear(self.fc1)
        init_linear(self.fc2)

    def apply_quantization(self, quantization_config: Any) -> None:
        if quantization_config is not None:
            raise NotImplementedError("vit_amd: torchao quantization is outside the encoder hot path")

    def _opts(self, residual: bool, alpha: float = 1.0):
        return _mlp_opts(glu=self._is_glu, activation=self.activation, eps=self.norm.eps or 1e-5,
                         layernorm=self._use_layer_norm, dropout=self.dropout.p, training=self.training,
                         limit=self.limit if self._is_glu else None, extra_bias=self.extra_bias if self._is_glu else None,
                         residual=residual, alpha=alpha)

    def fused_residual(self, x: Tensor, gamma: Optional[Tensor], keep_idx: Optional[Tensor], alpha: float) -> Tensor:
        return ops.MlpSublayerFn.apply(x, self.norm.weight, get_norm_bias(self.norm), self.fc1.weight, self.fc1.bias,
                                       self.fc2.weight, self.fc2.bias, gamma, keep_idx, self._opts(True, alpha))

    def forward(self, x: Tensor) -> Tensor:
        return ops.MlpSublayerFn.apply(x, self.norm.weight, get_norm_bias(self.norm), self.fc1.weight, self.fc1.bias,
                                       self.fc2.weight, self.fc2.bias, None, None, self._opts(False))
