"""RMSNorm / LayerNorm (reference: vit/norm.py).  Modules are stock nn.RMSNorm / nn.LayerNorm parameter
holders (same state_dict keys); the arithmetic is libvitk's norm kernel via ops.apply_norm."""
from __future__ import annotations

from typing import Literal, Optional

import torch
import torch.nn as nn
from torch import Tensor

from . import ops

NormType = Literal["rmsnorm", "layernorm"]
NORM_TYPE_CHOICES = ("rmsnorm", "layernorm")
NormModule = (nn.RMSNorm, nn.LayerNorm)


def is_layer_norm(norm_type: str) -> bool:
    return norm_type == "layernorm"


class _KernelRMSNorm(nn.RMSNorm):
    def forward(self, x: Tensor) -> Tensor:
        return ops.apply_norm(x, self.weight, None, self.eps, use_layer_norm=False)


class _KernelLayerNorm(nn.LayerNorm):
    def forward(self, x: Tensor) -> Tensor:
        return ops.apply_norm(x, self.weight, self.bias, self.eps, use_layer_norm=True)


def make_norm(hidden_size: int, norm_type: str, eps: Optional[float] = None, *, device=None, dtype=None):
    if norm_type == "rmsnorm":
        return _KernelRMSNorm(hidden_size, eps=eps, device=device, dtype=dtype)
    if norm_type == "layernorm":
        return _KernelLayerNorm(hidden_size, eps=1e-5 if eps is None else eps, device=device, dtype=dtype)
    raise ValueError(f"Unsupported norm_type: {norm_type}")


def get_norm_bias(norm: nn.Module) -> Optional[Tensor]:
    return norm.bias if isinstance(norm, nn.LayerNorm) else None


apply_norm = ops.apply_norm
