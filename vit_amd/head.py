"""Heads applied to the encoder's features (vit/head.py): the linear ``Head`` and the ``AttentivePoolHead`` of the
reference's README, with their config dataclasses and YAML tags.  The transposed-convolution and upsampling heads
(dense prediction) are not built."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Optional

import torch
import torch.nn as nn
import yaml
from torch import Tensor

from . import ops
from .attention import AttentivePool
from .initialization import init_linear

__all__ = ["Head", "HeadConfig", "AttentivePoolHead", "AttentivePoolHeadConfig"]


class Head(nn.Module):
    """dropout -> linear (vit/head.py:183-207)."""

    def __init__(self, in_features: int, out_features: int, dropout: float = 0.0, device=None, dtype=None):
        super().__init__()
        self.dropout = nn.Dropout(dropout)
        self.proj = nn.Linear(in_features, out_features, device=device, dtype=dtype)
        self.reset_parameters()

    def reset_parameters(self, bias: float = 0.0) -> None:
        init_linear(self.proj)
        if self.proj.bias is not None and bias != 0.0:
            nn.init.constant_(self.proj.bias, bias)

    def forward(self, x: Tensor) -> Tensor:
        x = ops.dropout(x, self.dropout.p, self.training)
        return ops.linear(x, self.proj.weight, self.proj.bias)


class AttentivePoolHead(nn.Module):
    """AttentivePool over the tokens, then ``Head`` (vit/head.py:213-257)."""

    def __init__(self, in_features: int, out_features: int, num_attention_heads: int, num_queries: int = 1,
                 hidden_dropout: float = 0.0, attention_dropout: float = 0.0, dropout: float = 0.0, bias: bool = True,
                 device=None, dtype=None, norm_type: str = "rmsnorm", qk_normalization: bool = False):
        super().__init__()
        self.pool = AttentivePool(in_features, num_attention_heads, num_queries=num_queries, hidden_dropout=hidden_dropout,
                                  attention_dropout=attention_dropout, bias=bias, device=device, dtype=dtype,
                                  norm_type=norm_type, qk_normalization=qk_normalization)
        self.proj = Head(in_features, out_features, dropout=dropout, device=device, dtype=dtype)

    def reset_parameters(self) -> None:
        self.pool.reset_parameters()
        self.proj.reset_parameters()

    def forward(self, x: Tensor) -> Tensor:
        return self.proj(self.pool(x))

    def forward_weights(self, x: Tensor) -> Tensor:
        return self.pool.forward_weights(x)


@dataclass
class HeadConfig:
    in_features: Optional[int] = None
    out_features: Optional[int] = None
    dropout: float = 0.0

    def instantiate(self, backbone_config: Any, device=None, dtype=None) -> Head:
        return Head(self.in_features or backbone_config.hidden_size, self.out_features or backbone_config.hidden_size,
                    self.dropout, device=device, dtype=dtype)


@dataclass
class AttentivePoolHeadConfig:
    in_features: Optional[int] = None
    out_features: Optional[int] = None
    num_attention_heads: Optional[int] = None
    num_queries: int = 1
    hidden_dropout: float = 0.0
    attention_dropout: float = 0.0
    dropout: float = 0.0

    def instantiate(self, backbone_config: Any, device=None, dtype=None) -> AttentivePoolHead:
        return AttentivePoolHead(
            in_features=self.in_features or backbone_config.hidden_size,
            out_features=self.out_features or backbone_config.hidden_size,
            num_attention_heads=self.num_attention_heads or backbone_config.num_attention_heads,
            num_queries=self.num_queries, hidden_dropout=self.hidden_dropout, attention_dropout=self.attention_dropout,
            dropout=self.dropout, bias=backbone_config.attention_bias, device=device, dtype=dtype,
            norm_type=backbone_config.norm_type, qk_normalization=backbone_config.qk_normalization)


def _yaml_tags(config_type, names) -> None:
    def construct(loader, node):
        return config_type(**loader.construct_mapping(node, deep=True))

    for name in names:
        for loader in (yaml.SafeLoader, yaml.FullLoader, yaml.UnsafeLoader):
            loader.add_constructor(f"tag:yaml.org,2002:python/object:{name}", construct)


# the reference's tags load as these classes, so its config files keep working (vit/head.py:46-76)
_yaml_tags(HeadConfig, ["vit.head.HeadConfig", "vit.HeadConfig", "vit_amd.head.HeadConfig"])
_yaml_tags(AttentivePoolHeadConfig, ["vit.head.AttentivePoolHeadConfig", "vit.AttentivePoolHeadConfig",
                                     "vit_amd.head.AttentivePoolHeadConfig"])
