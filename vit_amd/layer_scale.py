"""LayerScale parameter holder (reference: vit/layer_scale.py).  Inside the encoder block the multiply is
fused into the out_proj / fc2 GEMM epilogue; the module's own forward is the plain product."""
import torch
import torch.nn as nn
from torch import Tensor


def layer_scale(x: Tensor, gamma: Tensor, inplace: bool = False) -> Tensor:
    return x.mul_(gamma) if (inplace and not x.requires_grad) else x * gamma


class LayerScale(nn.Module):
    def __init__(self, dim: int, init_value: float = 1e-5, inplace: bool = False, device=None, dtype=None):
        super().__init__()
        self.inplace = inplace
        self.gamma = nn.Parameter(torch.empty(dim, device=device, dtype=dtype))
        self.reset_parameters(init_value)

    def reset_parameters(self, value: float = 1e-5) -> None:
        nn.init.constant_(self.gamma, value)

    def forward(self, x: Tensor) -> Tensor:
        return layer_scale(x, self.gamma, self.inplace)
