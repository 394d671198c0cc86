"""Activation registry of the MLP (reference: vit/helpers.py:9-48).

The arithmetic lives in libvitk's activation kernel; the callables returned here only carry the
activation code so that ``norm_mlp(..., activation=get_activation("swiglu"))`` keeps the reference's
call shape.  Calling one on a tensor runs the kernel (forward and backward)."""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor

from . import _lib as L
from . import kernels as K


class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, code):
        L.require_gpu(x)
        shape = x.shape
        p = x.detach().reshape(-1, shape[-1]).contiguous()
        ctx.save_for_backward(p)
        ctx.code, ctx.shape = code, shape
        return K.act_fwd(p, shape[-1], False, code, None, None, 0.0, 0).view(shape)

    @staticmethod
    def backward(ctx, dy):
        (p,) = ctx.saved_tensors
        dp = K.act_bwd(dy.reshape(p.shape).to(p.dtype), p, p.shape[1], False, ctx.code, None, None, 0.0, 0)
        return dp.view(ctx.shape), None


class Activation:
    """Callable activation that remembers which kernel code it stands for."""

    def __init__(self, name: str):
        self.name = name
        self.code = L.ACT_CODES[name]

    def __call__(self, x: Tensor) -> Tensor:
        return _ActFn.apply(x, self.code)

    def __repr__(self) -> str:
        return f"Activation({self.name!r})"


_BUILTIN = {F.relu: "relu", F.silu: "silu", F.gelu: "gelu", torch.relu: "relu"}


def activation_code(activation) -> int:
    if isinstance(activation, Activation):
        return activation.code
    if isinstance(activation, str):
        return L.ACT_CODES[activation]
    if activation in _BUILTIN:
        return L.ACT_CODES[_BUILTIN[activation]]
    raise ValueError(f"vit_amd: unsupported activation callable {activation!r}; use get_activation(name)")


srelu = Activation("srelu")
openswiglu = Activation("openswiglu")


def get_activation(activation: str) -> Activation:
    if activation not in L.ACT_CODES:
        raise ValueError(f"Activation {activation} not supported")
    return Activation(activation)
