"""Parameter initialisers (reference: vit/initialization.py:6-36): N(0, 0.02) truncated at two sigma, zero biases."""
import torch
import torch.nn as nn
from torch import Tensor

INIT_STD = 0.02
TRUNC_STD_SCALE = 2.0


@torch.no_grad()
def trunc_normal_(tensor: Tensor, std: float = INIT_STD) -> Tensor:
    lim = TRUNC_STD_SCALE * std
    return nn.init.trunc_normal_(tensor, mean=0.0, std=std, a=-lim, b=lim)


@torch.no_grad()
def zero_bias_if_present(module: nn.Module) -> None:
    b = getattr(module, "bias", None)
    if isinstance(b, Tensor):
        b.zero_()


@torch.no_grad()
def init_linear(module: nn.Linear, std: float = INIT_STD, *, zero_bias: bool = True) -> None:
    trunc_normal_(module.weight, std=std)
    if zero_bias:
        zero_bias_if_present(module)


init_conv = init_linear  # same rule for the patch convolution (vit/initialization.py:30-36)
