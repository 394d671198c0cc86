"""How fresh parameters are drawn.  The rule the reference applies everywhere (vit/initialization.py:6-36) is one
distribution -- a normal of standard deviation 0.02 cut off at two standard deviations -- for every weight, learned
token and position table, and zeros for every bias.  It is stated once here, as ``_fill``; the names the modules import
are thin bindings of it."""
from functools import partial

import torch
from torch import Tensor, nn

INIT_STD = 0.02        # the reference's default standard deviation
_CUTOFF_SIGMAS = 2.0   # draws are truncated at +- this many standard deviations


def _fill(weight: Tensor, std: float, bias=None) -> Tensor:
    """Redraw ``weight`` from N(0, std) truncated at two sigma; clear ``bias`` when one is given."""
    with torch.no_grad():
        edge = _CUTOFF_SIGMAS * std
        nn.init.trunc_normal_(weight, mean=0.0, std=std, a=-edge, b=edge)
        if isinstance(bias, Tensor):
            bias.zero_()
    return weight


def trunc_normal_(tensor: Tensor, std: float = INIT_STD) -> Tensor:
    return _fill(tensor, std)


def zero_bias_if_present(module: nn.Module) -> None:
    bias = getattr(module, "bias", None)
    if isinstance(bias, Tensor):
        with torch.no_grad():
            bias.zero_()


def _init_layer(module: nn.Module, std: float = INIT_STD, *, zero_bias: bool = True) -> None:
    _fill(module.weight, std, getattr(module, "bias", None) if zero_bias else None)


# nn.Linear and the patch convolution follow the same rule (vit/initialization.py:22-36)
init_linear = partial(_init_layer)
init_conv = partial(_init_layer)
