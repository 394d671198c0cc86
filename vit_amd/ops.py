"""Autograd functions of the encoder hot path.  Each chains libvitk kernels in forward and backward.

Dtype contract (what the reference's eager path produces, measured against it):
  * pure mode      -- parameters and activations in one dtype T (fp32 or bf16); everything is T.
  * autocast mode  -- fp32 master parameters under torch.autocast(bf16): linears / attention run in bf16
                      (compute dtype TC = bf16), the residual stream, norm statistics and layer scale stay
                      fp32 (residual dtype TR = fp32); weight gradients are produced directly in fp32.
"""
from __future__ import annotations

import math
import os
import weakref
from typing import Optional

import torch
from torch import Tensor

from . import _lib as L
from . import ddp as _ddp
from . import kernels as K


# ------------------------------------------------------------------------------------------------
# mode / rng helpers
# ------------------------------------------------------------------------------------------------
def compute_dtype(param_dtype: torch.dtype) -> torch.dtype:
    """dtype the linears and attention run in for parameters of ``param_dtype`` right now."""
    if torch.is_autocast_enabled("cuda"):
        ac = torch.get_autocast_dtype("cuda")
        if ac != torch.bfloat16:
            raise NotImplementedError(f"vit_amd supports torch.autocast only with bfloat16, got {ac}")
        return torch.bfloat16
    if param_dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"vit_amd supports float32 and bfloat16 parameters, got {param_dtype}")
    return param_dtype


def default_rms_eps(dtype: torch.dtype) -> float:
    """nn.RMSNorm(eps=None): torch 2.10 resolves the default in the fp32 op-math type for fp32 and bf16 inputs."""
    return torch.finfo(torch.float32).eps


def next_seed(device: torch.device) -> int:
    """A fresh 64-bit dropout stream id taken from torch's CUDA generator (no device sync).  Because it
    advances the generator offset, torch.utils.checkpoint replays the same ids during recomputation."""
    gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
    off = gen.get_offset()
    gen.set_offset(off + 4)
    x = (gen.initial_seed() * 0x9E3779B97F4A7C15 + off * 0xD1B54A32D192ED03 + 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 29
    return (x * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF


_ARANGE: dict = {}   # (S, device) -> arange(S) as a [1, S] int64 row


def token_rows(keep_idx: Optional[Tensor], S: int) -> Optional[Tensor]:
    """Row numbers in the flattened [B*S, D] stream of the samples selected by ``keep_idx``."""
    if keep_idx is None:
        return None
    key = (S, keep_idx.device)
    ar = _ARANGE.get(key)
    if ar is None:
        ar = torch.arange(S, device=keep_idx.device, dtype=torch.int64)[None, :]
        if len(_ARANGE) < 64:   # a handful of sequence lengths per process; never an unbounded cache
            _ARANGE[key] = ar
    return torch.add(ar, keep_idx.to(torch.int64)[:, None], alpha=S).reshape(-1)   # one launch: keep_idx * S + arange(S)


def _cast(t: Optional[Tensor], dtype: torch.dtype) -> Optional[Tensor]:
    if t is None:
        return None
    t = t.detach()
    return (t if t.dtype == dtype else t.to(dtype)).contiguous()


# Compute-dtype copies of parameters (fp32 master weights under autocast), kept until the parameter changes: an
# in-place update (optimizer step, load_state_dict) bumps ``_version``, a replaced ``.data`` changes the address.  One
# entry per parameter object; the entry dies with the parameter.
_PARAM_COPIES: dict = {}


_PARAM_CACHE_ON = os.environ.get("VITK_NO_PARAM_CACHE", "") == ""


def clear_param_copies() -> None:
    """Forget the kept compute-dtype copies: for code that rewrites parameters through ``.data`` (no version bump)."""
    _PARAM_COPIES.clear()
    _QK_PARAMS.clear()


# A write through ``param.data`` (``p.data.mul_()``, EMA weight swaps, some third-party optimizers) changes neither the
# version counter nor the address, so the kept copies would go stale without an error.  Two guards:
#   * every optimizer step (torch's global post-step hook) and every train() / eval() switch of a ViT drops the copies,
#     so the usual places such writes happen in are covered at the cost of one round of casts;
#   * VITK_CHECK_PARAM_CACHE=1 compares a kept copy with a fresh cast on every use and raises on a difference
#     (a debugging mode: one extra cast and a device-to-host sync per parameter use).
_PARAM_CACHE_CHECK = os.environ.get("VITK_CHECK_PARAM_CACHE", "") not in ("", "0")


def _drop_copies_after_step(*_a, **_k) -> None:
    clear_param_copies()


try:   # one registration for every optimizer of the process
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post_step

    _reg_post_step(_drop_copies_after_step)
except ImportError:   # pragma: no cover - torch without global optimizer hooks
    pass


def _cast_param(t: Optional[Tensor], dtype: torch.dtype, make=None) -> Optional[Tensor]:
    """``_cast`` for parameters, cached per parameter.  ``make(t)`` builds the copy when it is not a plain cast (the
    zero-padded patch-embedding weight)."""
    if t is None:
        return None
    if make is None and t.dtype == dtype:
        return t.detach().contiguous()
    key = (t._version, t.data_ptr(), dtype, tuple(t.shape))
    ent = _PARAM_COPIES.get(id(t))
    if ent is not None and ent[0]() is t and ent[1] == key:
        if _PARAM_CACHE_CHECK:
            fresh = make(t.detach()) if make is not None else t.detach().to(dtype)
            if not torch.equal(fresh, ent[2]):
                raise RuntimeError("vit_amd: a cached compute-dtype copy of a parameter is stale: the parameter was changed "
                                   "without a version bump (a write through .data?); call vit_amd.ops.clear_param_copies() "
                                   "after such writes or set VITK_NO_PARAM_CACHE=1")
        return ent[2]
    c = make(t.detach()) if make is not None else t.detach().to(dtype).contiguous()
    if not _PARAM_CACHE_ON:
        return c
    ident = id(t)
    _PARAM_COPIES[ident] = (weakref.ref(t, lambda _r, ident=ident: _PARAM_COPIES.pop(ident, None)), key, c)
    return c


def _ptrs(*params: Optional[Tensor]) -> tuple:
    """Storage addresses of master parameters, kept by a forward pass for its backward's gradient sinks."""
    return tuple(None if p is None else p.data_ptr() for p in params)


def _grad_buffer(param_ptr: Optional[int], shape, dtype: torch.dtype, device) -> Tensor:
    """Where a parameter's gradient is produced: the slice of the data-parallel reducer's bucket when one is attached
    to the parameter (autograd adopts it as ``.grad`` without a copy), else fresh memory."""
    if param_ptr is not None and _ddp._SINKS:
        sink = _ddp.grad_sink(param_ptr, shape, dtype)
        if sink is not None:
            return sink
    return torch.empty(shape, dtype=dtype, device=device)


def _linear_bwd(dy: Tensor, inp: Tensor, w_c: Tensor, need_dx: bool, need_dw: bool, need_db: bool, wdtype: torch.dtype,
                dx_dtype: Optional[torch.dtype] = None, tag: str = "", w_ptr: Optional[int] = None,
                b_ptr: Optional[int] = None):
    """Backward of y = inp @ w^T + b.  dy [M,N], inp [M,Kin], w_c [N,Kin] (compute dtype).  ``w_ptr`` / ``b_ptr``: storage
    addresses of the master weight / bias (keys of the reducer's gradient sinks)."""
    M, N = dy.shape
    Kin = inp.shape[1]
    dx = dw = db = None
    if need_dx:
        dx = torch.empty((M, Kin), dtype=dx_dtype or dy.dtype, device=dy.device)
        K.gemm(dy, w_c, trans_a=False, trans_b=True, M=M, N=Kin, K=N, out=dx, tag=tag + ".dgrad")
    if need_dw:
        dw = _grad_buffer(w_ptr, (N, Kin), wdtype, dy.device)
        if need_db:  # the bias gradient dy.sum(0) comes out of the weight-gradient pass over dy
            db = _grad_buffer(b_ptr, (N,), wdtype, dy.device)
        K.gemm(dy, inp, trans_a=True, trans_b=True, M=N, N=Kin, K=M, out=dw, a_colsum=db, tag=tag + ".wgrad")
    elif need_db:
        db = K.colsum(dy, M, N, N, wdtype)
    return dx, dw, db


# ------------------------------------------------------------------------------------------------
# apply_norm  (vit/norm.py:40-50)
# ------------------------------------------------------------------------------------------------
class NormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps, layernorm, out_dtype):
        L.require_gpu(x, w)
        shape = x.shape
        x2 = x.detach().reshape(-1, shape[-1])
        y, rstd, mean = K.norm_fwd(x2, w.detach(), None if b is None else b.detach(), eps, layernorm, out_dtype)
        ctx.save_for_backward(x2, w, rstd, mean)
        ctx.layernorm, ctx.shape, ctx.has_b = layernorm, shape, b is not None
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w, rstd, mean = ctx.saved_tensors
        dy2 = dy.reshape(-1, ctx.shape[-1])
        dx = torch.empty_like(x2)
        dw, db = K.norm_bwd(dy2, x2, w.detach(), rstd, mean, ctx.layernorm, None, dx, ctx.needs_input_grad[1],
                            ctx.has_b and ctx.needs_input_grad[2])
        return dx.view(ctx.shape) if ctx.needs_input_grad[0] else None, dw, db, None, None, None


def apply_norm(x: Tensor, weight: Tensor, bias: Optional[Tensor], eps: Optional[float], *, use_layer_norm: bool) -> Tensor:
    """F.rms_norm / F.layer_norm over the last dim; output dtype = input dtype (as the eager ops)."""
    if eps is None:
        eps = 1e-5 if use_layer_norm else default_rms_eps(x.dtype)
    out_dtype = x.dtype
    if use_layer_norm and torch.is_autocast_enabled("cuda"):
        out_dtype = torch.float32  # layer_norm is on autocast's fp32 list (rms_norm is not), measured on ROCm torch 2.10
    return NormFn.apply(x, weight, bias, float(eps), bool(use_layer_norm), out_dtype)


# ------------------------------------------------------------------------------------------------
# generic linear  (F.linear) -- used for the Fourier position projection
# ------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, tc):
        L.require_gpu(x, w)
        shape = x.shape
        x2 = _cast(x.reshape(-1, shape[-1]), tc)
        wc, bc = _cast_param(w, tc), _cast_param(b, tc)
        M, Kin, N = x2.shape[0], x2.shape[1], w.shape[0]
        y = torch.empty((M, N), dtype=tc, device=x.device)
        K.gemm(x2, wc, trans_a=False, trans_b=False, M=M, N=N, K=Kin, out=y, bias=bc)
        ctx.save_for_backward(x2, wc)
        ctx.shape, ctx.xdtype, ctx.wdtype, ctx.has_b = shape, x.dtype, w.dtype, b is not None
        return y.view(*shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, wc = ctx.saved_tensors
        dy2 = _cast(dy.reshape(-1, dy.shape[-1]), wc.dtype)
        dx, dw, db = _linear_bwd(dy2, x2, wc, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                 ctx.has_b and ctx.needs_input_grad[2], ctx.wdtype)
        if dx is not None:
            dx = dx.view(ctx.shape).to(ctx.xdtype)
        return dx, dw, db, None


def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    return LinearFn.apply(x, w, b, compute_dtype(w.dtype))


class FourierFeaturesFn(torch.autograd.Function):
    """cat([sin(grid @ w), cos(grid @ w)], -1) / sqrt(J) in one launch, in the dtype the projection behind it computes in
    (vit/pos_enc.py:137-149; the reference forms the features in fp32 with autocast off and the linear casts them)."""

    @staticmethod
    def forward(ctx, grid, w, out_dtype):
        L.require_gpu(grid, w)
        g2 = grid.detach().reshape(-1, grid.shape[-1]).float()
        w2 = w.detach().float()
        ctx.save_for_backward(g2, w2)
        ctx.wdtype = w.dtype
        return K.fourier_features(g2, w2, out_dtype).view(*grid.shape[:-1], 2 * w.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        g2, w2 = ctx.saved_tensors
        d2 = dy.reshape(-1, dy.shape[-1])
        if d2.dtype not in (torch.float32, torch.bfloat16):
            d2 = d2.float()
        return None, K.fourier_features_bwd(g2, w2, d2).to(ctx.wdtype), None


# ------------------------------------------------------------------------------------------------
# F.dropout on a [..., N] tensor (N % 4 == 0): mask from (seed, element index), regenerated in backward
# ------------------------------------------------------------------------------------------------
class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        L.require_gpu(x)
        N = x.shape[-1]
        x2 = x.detach().reshape(-1, N)
        ctx.p, ctx.seed, ctx.shape = p, seed, x.shape
        return K.scale_cast(x2, x2.shape[0], N, x.dtype, drop_p=p, seed=seed).view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        N = ctx.shape[-1]
        dy2 = dy.reshape(-1, N)
        return K.scale_cast(dy2, dy2.shape[0], N, dy.dtype, drop_p=ctx.p, seed=ctx.seed).view(ctx.shape), None, None


def dropout(x: Tensor, p: float, training: bool) -> Tensor:
    if not training or p <= 0.0:
        return x
    return DropoutFn.apply(x, float(p), next_seed(x.device))


# ------------------------------------------------------------------------------------------------
# residual sub-layers of the encoder block (vit/transformer.py:146-157)
# ------------------------------------------------------------------------------------------------
def _residual_dtype(x: Tensor, gamma: Optional[Tensor], tc: torch.dtype) -> torch.dtype:
    t = torch.promote_types(x.dtype, tc)
    return torch.promote_types(t, gamma.dtype) if gamma is not None else t


def _dropped_samples(keep_idx: Optional[Tensor]) -> Optional[Tensor]:
    """The complement of a drop-path subset, when the code that drew the subset attached it (vit_amd/transformer.py)."""
    return getattr(keep_idx, "_vitk_dropped", None) if keep_idx is not None else None


def _passthrough(src2: Tensor, dropped: Optional[Tensor], rows_per_sample: int) -> tuple:
    """(out, res) for a kernel that writes out[r] = res[r] + ... on the rows of the kept samples only: ``out`` already
    holds ``src2``'s rows of the dropped samples.  With the list of dropped samples only their rows are copied (the
    kernel reads the kept rows from src2 and writes every one of them); without it the whole stream is cloned and
    updated in place."""
    if dropped is None or src2.shape[0] % rows_per_sample:
        out = src2.clone()
        return out, out
    width = rows_per_sample * src2.shape[1]
    if not src2.is_contiguous() or (width * src2.element_size()) % 16:
        out = src2.clone()
        return out, out
    out = torch.empty_like(src2)
    if dropped.numel():
        K.copy_samples_(out, src2, dropped, width)
    return out, src2


def _residual_gemm(inp: Tensor, w_c: Tensor, b_c: Optional[Tensor], x2: Tensor, gamma: Optional[Tensor], rows: Optional[Tensor],
                   alpha: float, drop_p: float, seed: int, want_aux: bool, tag: str = "", dropped: Optional[Tensor] = None,
                   rows_per_sample: int = 1):
    """x_new = x (+ scatter) alpha * gamma * dropout(inp @ w^T + b): the out_proj / fc2 GEMM with its fused epilogue."""
    M, Kin = inp.shape
    N = w_c.shape[0]
    aux = torch.empty((M, N), dtype=inp.dtype, device=inp.device) if want_aux else None
    if rows is None:
        out = torch.empty_like(x2)
        res = x2
    else:
        out, res = _passthrough(x2, dropped, rows_per_sample)  # rows of dropped samples pass through
    K.gemm(inp, w_c, trans_a=False, trans_b=False, M=M, N=N, K=Kin, out=out, bias=b_c, epilogue=L.EPI_RESIDUAL, res=res,
           gamma=gamma, row_idx=rows, aux=aux, alpha=alpha, drop_p=drop_p, seed=seed, tag=tag)
    return out, aux


# The gradient of the residual stream leaves one sub-layer's norm backward in fp32 and is the first thing the previous
# sub-layer casts to bf16 (when there is no layer scale, drop-path or dropout to apply on the way).  The norm backward
# kernel can write that cast next to the fp32 rows, which saves a pass over them.  The copy travels with the gradient
# itself: the producing backward attaches it to the tensor it returns, and the consuming backward reads it from the very
# tensor object autograd hands it.  If autograd had to build another tensor on the way (a gradient accumulated from two
# consumers, a dtype conversion), the attribute is simply not there and the consumer casts as usual; an accumulation
# done in place into the carrying tensor changes its version counter, which the consumer checks.
#
# Why the tensor object and not a table keyed by address: a freed gradient's address is reused by the caching allocator
# for the next tensor of that size, so only the identity of the Python object proves the values are the ones the copy was
# made from.  Why not one autograd node per block: the sub-layers are modules whose hooks must keep firing, and the
# hand-off from one block's attention sub-layer to the previous block's MLP crosses nodes in any case.
# A lost hand-off costs one cast kernel and no accuracy; carry_stats() counts offers and takes so that a training loop
# (or a test) can see when autograd stopped handing the tensors through unchanged.
_CARRY_BF16 = os.environ.get("VITK_NO_SHADOW", "") == ""
_CARRY_STATS = {"offered": 0, "taken": 0}


def carry_stats(reset: bool = False) -> dict:
    """{"offered": bf16 gradient copies attached by a sub-layer's backward, "taken": copies a consumer used in place of a
    cast}.  In a plain stack of encoder blocks every offer but the first block's is taken."""
    out = dict(_CARRY_STATS)
    if reset:
        _CARRY_STATS["offered"] = _CARRY_STATS["taken"] = 0
    return out


def _shadow_buffer(dx: Tensor, tc: torch.dtype, rows: Optional[Tensor]) -> Optional[Tensor]:
    if not _CARRY_BF16 or rows is not None or dx.dtype != torch.float32 or tc != torch.bfloat16:
        return None
    return torch.empty(dx.shape, dtype=torch.bfloat16, device=dx.device)


def _attach_bf16(dx: Tensor, shadow: Optional[Tensor]) -> Tensor:
    """``dx`` is the gradient about to be returned to autograd; ``shadow`` holds the same values in bf16 (flattened rows)."""
    if shadow is not None:
        dx._vitk_bf16 = (shadow, dx._version)   # the version tells an in-place accumulation into dx apart
        _CARRY_STATS["offered"] += 1
    return dx


def _carried_bf16(dnew: Tensor, D: int) -> Optional[Tensor]:
    entry = getattr(dnew, "_vitk_bf16", None)
    if entry is None:
        return None
    shadow, version = entry
    if (dnew._version != version or dnew.dtype != torch.float32 or not dnew.is_contiguous()
            or shadow.numel() != dnew.numel() or shadow.device != dnew.device):
        return None
    _CARRY_STATS["taken"] += 1
    return shadow.view(-1, D)


_FUSE_DGAMMA = os.environ.get("VITK_NO_FUSED_DGAMMA", "") == ""   # A/B switch: layer-scale gradient as a pass of its own


def _residual_grad(dnew2: Tensor, tc: torch.dtype, gamma: Optional[Tensor], rows: Optional[Tensor], nrows: int, alpha: float,
                   drop_p: float, seed: int, aux: Optional[Tensor], need_dgamma: bool, shadow: Optional[Tensor] = None):
    """Gradient w.r.t. the linear's output, and w.r.t. gamma.  ``shadow``: dnew2 in bf16 if the producer supplied it."""
    N = dnew2.shape[1]
    if shadow is not None and gamma is None and rows is None and alpha == 1.0 and drop_p == 0.0 and tc == torch.bfloat16:
        return shadow, None
    if gamma is None and rows is None and alpha == 1.0 and drop_p == 0.0 and dnew2.dtype == tc:
        dy = dnew2   # nothing to scale, gather or cast (pure-bf16 parameters without layer scale or drop-path)
    elif _FUSE_DGAMMA and gamma is not None and need_dgamma and aux is not None and N <= 4096 and N % 4 == 0:
        # layer scale: the gradient of gamma comes out of the pass that scales and casts the branch gradient
        return K.scale_cast_dgamma(dnew2, aux, nrows, N, tc, gamma, row_idx=rows, alpha=alpha, drop_p=drop_p, seed=seed)
    else:
        dy = K.scale_cast(dnew2, nrows, N, tc, row_idx=rows, gamma=gamma, alpha=alpha, drop_p=drop_p, seed=seed)
    dgamma = None
    if gamma is not None and need_dgamma:
        dgamma = K.colsum(dnew2, nrows, N, N, gamma.dtype, a_idx=rows, b=aux, ldb=N, scale=alpha)
    return dy, dgamma


def _rope_table(rope: Optional[Tensor], keep_idx: Optional[Tensor], B: int, Bk: int, tc: torch.dtype):
    """(fp32 table for the rope kernels, emulate-bf16 flag).  [2, N, Dh] is shared by all samples; [2, B, 1, N', Dh] holds
    one table per sample (masked tokens, vit/vit.py:522-527) and follows a drop-path subset (vit/transformer.py:41-46)."""
    if rope is None:
        return None, False
    emu = rope.dtype == torch.bfloat16 and tc == torch.bfloat16
    if rope.dim() == 5:
        table = rope.detach()[:, :, 0].to(torch.float32)
        if keep_idx is not None and table.shape[1] == B:
            table = table.index_select(1, keep_idx)
        table = table.contiguous()
        if table.shape[1] != Bk:
            raise ValueError(f"vit_amd: RoPE table for {table.shape[1]} samples, batch is {Bk}")
    elif rope.dim() == 3:
        table = rope.detach().to(torch.float32).contiguous()
    else:
        raise ValueError(f"vit_amd: RoPE table must be [2, N, Dh] or [2, B, 1, N, Dh], got {tuple(rope.shape)}")
    return table, emu


_QK_PARAMS: dict = {}   # id(q_norm.weight) -> (key of all four parameters, weakref, stacked fp32 weights, stacked fp32 biases)


def _qk_norm_params(qw: Tensor, qb: Optional[Tensor], kw: Tensor, kb: Optional[Tensor]):
    """[2, Dh] fp32 weights (and biases) of q_norm and k_norm for the projection epilogue, rebuilt when any of them changes."""
    key = tuple(None if t is None else (id(t), t._version, t.data_ptr()) for t in (qw, qb, kw, kb))
    ent = _QK_PARAMS.get(id(qw))
    if ent is not None and ent[0] == key and ent[1]() is qw:
        return ent[2], ent[3]
    w2 = torch.stack([qw.detach().float(), kw.detach().float()]).contiguous()
    b2 = None if qb is None or kb is None else torch.stack([qb.detach().float(), kb.detach().float()]).contiguous()
    ident = id(qw)
    _QK_PARAMS[ident] = (key, weakref.ref(qw, lambda _r, ident=ident: _QK_PARAMS.pop(ident, None)), w2, b2)
    return w2, b2


class AttentionSublayerFn(torch.autograd.Function):
    """x -> x (+) alpha * gamma * dropout(out_proj(SDPA(rope(qkv_proj(norm(x))))))   (vit/attention.py:163-215)."""

    @staticmethod
    def forward(ctx, x, norm_w, norm_b, w_qkv, b_qkv, w_out, b_out, gamma, rope, keep_idx, qn_w, qn_b, kn_w, kn_b, opt):
        L.require_gpu(x, norm_w, w_qkv, w_out)
        B, S, D = x.shape
        Dh = opt["head_dim"]
        H = D // Dh
        tc = compute_dtype(w_qkv.dtype)
        residual = opt["residual"]
        tr = _residual_dtype(x, gamma, tc) if residual else tc
        x2 = _cast(x.reshape(B * S, D), tr if residual else x.dtype)
        rows = token_rows(keep_idx, S)
        Bk = B if keep_idx is None else keep_idx.numel()
        M = Bk * S
        training = opt["training"]
        p_attn = opt["attention_dropout"] if training else 0.0
        p_hid = opt["dropout"] if training else 0.0

        xn, rstd, mean = K.norm_fwd(x2, norm_w.detach(), None if norm_b is None else norm_b.detach(), opt["eps"],
                                    opt["layernorm"], tc, rows)
        wq, bq, wo, bo = _cast_param(w_qkv, tc), _cast_param(b_qkv, tc), _cast_param(w_out, tc), _cast_param(b_out, tc)
        qkv = torch.empty((M, 3 * D), dtype=tc, device=x.device)
        table, emu = _rope_table(rope, keep_idx, B, Bk, tc)
        # head_dim 64, one table for all samples: q and k are rotated in the projection's epilogue and their gradients
        # un-rotated in the attention backward's stores; otherwise the rotation is a pass of its own over q and k
        fuse_rope = K.can_fuse_rope(tc, Dh, table)
        fuse_norm = qn_w is not None and K.can_fuse_qk_norm(tc, Dh)
        qk_saved = (None,) * 6
        if fuse_norm:
            # per-head q/k normalisation (vit/attention.py:25-41, :95-105) in the projection's epilogue, before the rotation;
            # the un-normalised heads and the row statistics are written beside qkv for the backward pass
            qk_ln = opt["qk_layernorm"]
            w2, b2 = _qk_norm_params(qn_w, qn_b if qk_ln else None, kn_w, kn_b if qk_ln else None)
            pre = torch.empty((2, M, D), dtype=tc, device=x.device)
            stat_r = torch.empty((2, M * H), dtype=torch.float32, device=x.device)
            stat_m = torch.empty((2, M * H), dtype=torch.float32, device=x.device) if qk_ln else None
            K.gemm(xn, wq, trans_a=False, trans_b=False, M=M, N=3 * D, K=D, out=qkv, bias=bq, tag="qkv", epilogue=L.EPI_ROPE,
                   rope=table if fuse_rope else None, rope_width=D, rope_emulate_bf16=emu, seq=S,
                   prefix=S - table.shape[-2] if fuse_rope else 0, aux=pre,
                   qk_norm=dict(weight=w2, bias=b2, rstd=stat_r, mean=stat_m, eps=opt["qk_eps"], layernorm=qk_ln))
            qk_saved = [pre[0].view(M * H, Dh), stat_r[0], None if stat_m is None else stat_m[0],
                        pre[1].view(M * H, Dh), stat_r[1], None if stat_m is None else stat_m[1]]
        elif fuse_rope and qn_w is None:
            K.gemm(xn, wq, trans_a=False, trans_b=False, M=M, N=3 * D, K=D, out=qkv, bias=bq, tag="qkv", epilogue=L.EPI_ROPE,
                   rope=table, rope_width=D, rope_emulate_bf16=emu, seq=S, prefix=S - table.shape[-2])
        else:
            fuse_rope = False
            K.gemm(xn, wq, trans_a=False, trans_b=False, M=M, N=3 * D, K=D, out=qkv, bias=bq, tag="qkv")
        # otherwise the heads of q (and of k) are copied out as a contiguous [M*H, Dh] matrix, which is what the row-norm
        # kernels take and what their backward needs; the normalised rows go back into the packed buffer.
        if qn_w is not None and not fuse_norm:
            qk_saved = []
            for which, (w_, b_) in enumerate(((qn_w, qn_b), (kn_w, kn_b))):
                pre = qkv[:, which * D:(which + 1) * D].contiguous().view(M * H, Dh)
                nrm, r_, m_ = K.norm_fwd(pre, w_.detach(), None if b_ is None else b_.detach(), opt["qk_eps"],
                                         opt["qk_layernorm"], tc)
                qkv[:, which * D:(which + 1) * D].copy_(nrm.view(M, D))
                qk_saved += [pre, r_, m_]
        if table is not None and not fuse_rope:
            K.rope_qk_(qkv, table, Bk, S, H, Dh, S - table.shape[-2], False, emu)
        seed_a = next_seed(x.device) if p_attn > 0 else 0
        o, lse = K.attn_fwd(qkv, Bk, S, H, Dh, Dh ** -0.5, p_attn, seed_a)
        seed_h = next_seed(x.device) if p_hid > 0 else 0
        g = _cast(gamma, tr)
        if residual:
            out, aux = _residual_gemm(o, wo, bo, x2, g, rows, opt["alpha"], p_hid, seed_h, want_aux=g is not None, tag="out",
                                      dropped=_dropped_samples(keep_idx), rows_per_sample=S)
            y = out.view(B, S, D)
        else:
            zero = torch.zeros((M, D), dtype=tc, device=x.device)
            out, aux = _residual_gemm(o, wo, bo, zero, None, None, 1.0, p_hid, seed_h, want_aux=False)
            y = out.view(Bk, S, D)
        ctx.save_for_backward(x2, norm_w, rstd, mean, xn, qkv, o, lse, wq, wo, g, aux, table, rows, qn_w, kn_w, *qk_saved)
        ctx.meta = (B, S, D, H, Dh, Bk, tc, tr, p_attn, p_hid, seed_a, seed_h, emu, opt, x.dtype, w_qkv.dtype,
                    norm_b is not None, b_qkv is not None, b_out is not None, qn_b is not None, kn_b is not None, fuse_rope)
        ctx.ptrs = _ptrs(w_qkv, b_qkv, w_out, b_out)
        ctx.dropped = _dropped_samples(keep_idx)
        return y

    @staticmethod
    def backward(ctx, dnew):
        (x2, norm_w, rstd, mean, xn, qkv, o, lse, wq, wo, g, aux, table, rows, qn_w, kn_w, pre_q, rstd_q, mean_q, pre_k,
         rstd_k, mean_k) = ctx.saved_tensors
        (B, S, D, H, Dh, Bk, tc, tr, p_attn, p_hid, seed_a, seed_h, emu, opt, xdtype, wdtype, has_nb, has_bq,
         has_bo, has_qnb, has_knb, fuse_rope) = ctx.meta
        need = ctx.needs_input_grad
        ptrs = ctx.ptrs
        M = Bk * S
        residual = opt["residual"]
        dnew2 = _cast(dnew.reshape(-1, D), tr)
        dy, dgamma = _residual_grad(dnew2, tc, g if residual else None, rows if residual else None, M,
                                    opt["alpha"] if residual else 1.0, p_hid, seed_h, aux, need[7], _carried_bf16(dnew, D))
        do, dwo, dbo = _linear_bwd(dy, o, wo, True, need[5], has_bo and need[6], wdtype, tag="out", w_ptr=ptrs[2], b_ptr=ptrs[3])
        dqkv = K.attn_bwd(qkv, o, lse, do, Bk, S, H, Dh, Dh ** -0.5, p_attn, seed_a, rope=table if fuse_rope else None,
                          rope_emulate_bf16=emu)
        if table is not None and not fuse_rope:
            K.rope_qk_(dqkv, table, Bk, S, H, Dh, S - table.shape[-2], True, emu)
        dqn = [None, None, None, None]   # d q_norm.weight, d q_norm.bias, d k_norm.weight, d k_norm.bias
        if qn_w is not None and K.can_qk_norm_bwd_inplace(dqkv, Dh) and pre_q.dtype == dqkv.dtype and qn_w.dtype == kn_w.dtype:
            # per-head normalisation backward in place on the packed gradient: no head is copied out or back
            dqn = list(K.qk_norm_bwd_(dqkv, pre_q, pre_k, rstd_q, rstd_k, mean_q, mean_k, qn_w.detach(), kn_w.detach(), H, Dh,
                                      opt["qk_layernorm"], (need[10], has_qnb and need[11], need[12], has_knb and need[13])))
        elif qn_w is not None:
            for which, (w_, pre, r_, m_, has_b) in enumerate(((qn_w, pre_q, rstd_q, mean_q, has_qnb),
                                                              (kn_w, pre_k, rstd_k, mean_k, has_knb))):
                dpost = dqkv[:, which * D:(which + 1) * D].contiguous().view(M * H, Dh)
                dpre = torch.empty_like(dpost)
                dw_, db_ = K.norm_bwd(dpost, pre, w_.detach(), r_, m_, opt["qk_layernorm"], None, dpre, need[10 + 2 * which],
                                      has_b and need[11 + 2 * which])
                dqkv[:, which * D:(which + 1) * D].copy_(dpre.view(M, D))
                dqn[2 * which], dqn[2 * which + 1] = dw_, db_
        dxn, dwq, dbq = _linear_bwd(dqkv, xn, wq, True, need[3], has_bq and need[4], wdtype, tag="qkv", w_ptr=ptrs[0], b_ptr=ptrs[1])
        if residual:
            if rows is None:
                dx = torch.empty_like(dnew2)
                add = dnew2
            else:
                dx, add = _passthrough(dnew2, ctx.dropped, S)   # dropped samples: the gradient passes through
        else:
            dx = torch.zeros_like(x2) if rows is not None else torch.empty_like(x2)
            add = None
        shadow = _shadow_buffer(dx, tc, rows) if residual else None
        dnw, dnb = K.norm_bwd(dxn, x2, norm_w.detach(), rstd, mean, opt["layernorm"], add, dx, need[1], has_nb and need[2], rows,
                              dx_bf16=shadow)
        dx = dx.view(B, S, D)
        if dx.dtype != xdtype:
            dx = dx.to(xdtype)
        else:
            _attach_bf16(dx, shadow)
        if dgamma is not None and dgamma.dtype != g.dtype:
            dgamma = dgamma.to(g.dtype)
        return (dx if need[0] else None, dnw, dnb, dwq, dbq, dwo, dbo, dgamma, None, None, *dqn, None)


class CrossAttentionFn(torch.autograd.Function):
    """q -> q (+) alpha * gamma * dropout(out_proj(SDPA(q_proj(norm(q)), kv_proj(kv))))   (vit/attention.py:112-162,
    :218-280; the residual form is the cross sub-layer of vit/transformer.py:262-284 and :367-384).  Lq queries attend
    to Lk keys; the norm is on the query side only.  ``keep_idx`` (drop-path) selects samples of q and of kv alike."""

    @staticmethod
    def forward(ctx, q, kv, norm_w, norm_b, w_q, b_q, w_kv, b_kv, w_out, b_out, qn_w, qn_b, kn_w, kn_b, gamma, keep_idx,
                rope_q, rope_k, opt):
        L.require_gpu(q, kv, norm_w, w_q, w_kv, w_out)
        B, Lq, D = q.shape
        Lk = kv.shape[1]
        if kv.shape[0] != B or kv.shape[2] != D:
            raise ValueError(f"vit_amd: q {tuple(q.shape)} and kv {tuple(kv.shape)} do not go together")
        Dh = opt["head_dim"]
        H = D // Dh
        tc = compute_dtype(w_q.dtype)
        residual = opt["residual"]
        tr = _residual_dtype(q, gamma, tc) if residual else tc
        training = opt["training"]
        p_attn = opt["attention_dropout"] if training else 0.0
        p_hid = opt["dropout"] if training else 0.0
        rows = token_rows(keep_idx, Lq)
        Bk = B if keep_idx is None else keep_idx.numel()
        Mq, Mk = Bk * Lq, Bk * Lk

        q2 = _cast(q.reshape(B * Lq, D), tr if residual else q.dtype)
        qn, rstd, mean = K.norm_fwd(q2, norm_w.detach(), None if norm_b is None else norm_b.detach(), opt["eps"],
                                    opt["layernorm"], tc, rows)
        kv_kept = kv.detach() if keep_idx is None else kv.detach().index_select(0, keep_idx)   # vit/transformer.py:49-54
        kv2 = _cast(kv_kept.reshape(Mk, D), tc)
        wq, bq, wkv, bkv, wo, bo = (_cast(t, tc) for t in (w_q, b_q, w_kv, b_kv, w_out, b_out))
        qp = torch.empty((Mq, D), dtype=tc, device=q.device)
        K.gemm(qn, wq, trans_a=False, trans_b=False, M=Mq, N=D, K=D, out=qp, bias=bq)
        kvp = torch.empty((Mk, 2 * D), dtype=tc, device=q.device)
        K.gemm(kv2, wkv, trans_a=False, trans_b=False, M=Mk, N=2 * D, K=D, out=kvp, bias=bkv)
        qk_saved = (None,) * 6
        if qn_w is not None:   # per-head normalisation of q and k (vit/attention.py:25-41)
            pre_q = qp.view(Mq * H, Dh)
            nq, rq, mq = K.norm_fwd(pre_q, qn_w.detach(), None if qn_b is None else qn_b.detach(), opt["qk_eps"],
                                    opt["qk_layernorm"], tc)
            qp = nq.view(Mq, D)
            pre_k = kvp[:, :D].contiguous().view(Mk * H, Dh)
            nk, rk, mk = K.norm_fwd(pre_k, kn_w.detach(), None if kn_b is None else kn_b.detach(), opt["qk_eps"],
                                    opt["qk_layernorm"], tc)
            kvp[:, :D].copy_(nk.view(Mk, D))
            qk_saved = (pre_q, rq, mq, pre_k, rk, mk)
        tq, emu_q = _rope_table(rope_q, keep_idx, B, Bk, tc)   # vit/attention.py:156-159
        tk, emu_k = _rope_table(rope_k, keep_idx, B, Bk, tc)
        if tq is not None:
            if not qp.is_contiguous():
                qp = qp.contiguous()
            K.rope_rows_(qp, tq, Bk, Lq, H, Dh, False, emu_q)
        if tk is not None:
            K.rope_rows_(kvp, tk, Bk, Lk, H, Dh, False, emu_k)
        seed_a = next_seed(q.device) if p_attn > 0 else 0
        o, lse = K.attn_cross_fwd(qp, kvp, Bk, Lq, Lk, H, Dh, Dh ** -0.5, p_attn, seed_a)
        seed_h = next_seed(q.device) if p_hid > 0 else 0
        g = _cast(gamma, tr)
        if residual:
            out, aux = _residual_gemm(o, wo, bo, q2, g, rows, opt["alpha"], p_hid, seed_h, want_aux=g is not None)
            y = out.view(B, Lq, D)
        else:
            zero = torch.zeros((Mq, D), dtype=tc, device=q.device)
            out, aux = _residual_gemm(o, wo, bo, zero, None, None, 1.0, p_hid, seed_h, want_aux=False)
            y = out.view(Bk, Lq, D)
        ctx.save_for_backward(q2, norm_w, rstd, mean, qn, kv2, qp, kvp, o, lse, wq, wkv, wo, g, aux, rows, keep_idx, qn_w, kn_w,
                              tq, tk, *qk_saved)
        ctx.emu = (emu_q, emu_k)
        ctx.meta = (B, Bk, Lq, Lk, D, H, Dh, tc, tr, p_attn, p_hid, seed_a, seed_h, opt, q.dtype, kv.dtype, w_q.dtype,
                    norm_b is not None, b_q is not None, b_kv is not None, b_out is not None, qn_b is not None,
                    kn_b is not None)
        return y

    @staticmethod
    def backward(ctx, dnew):
        (q2, norm_w, rstd, mean, qn, kv2, qp, kvp, o, lse, wq, wkv, wo, g, aux, rows, keep_idx, qn_w, kn_w, tq, tk, pre_q,
         rstd_q, mean_q, pre_k, rstd_k, mean_k) = ctx.saved_tensors
        (B, Bk, Lq, Lk, D, H, Dh, tc, tr, p_attn, p_hid, seed_a, seed_h, opt, qdtype, kvdtype, wdtype, has_nb, has_bq,
         has_bkv, has_bo, has_qnb, has_knb) = ctx.meta
        need = ctx.needs_input_grad
        residual = opt["residual"]
        Mq, Mk = Bk * Lq, Bk * Lk
        dnew2 = _cast(dnew.reshape(-1, D), tr)
        dy, dgamma = _residual_grad(dnew2, tc, g if residual else None, rows if residual else None, Mq,
                                    opt["alpha"] if residual else 1.0, p_hid, seed_h, aux, need[14])
        do, dwo, dbo = _linear_bwd(dy, o, wo, True, need[8], has_bo and need[9], wdtype)
        dqp, dkvp = K.attn_cross_bwd(qp, kvp, o, lse, do, Bk, Lq, Lk, H, Dh, Dh ** -0.5, p_attn, seed_a)
        if tq is not None:
            K.rope_rows_(dqp, tq, Bk, Lq, H, Dh, True, ctx.emu[0])
        if tk is not None:
            K.rope_rows_(dkvp, tk, Bk, Lk, H, Dh, True, ctx.emu[1])
        dqn = [None, None, None, None]
        if qn_w is not None:
            dpre = torch.empty_like(dqp)
            dqn[0], dqn[1] = K.norm_bwd(dqp.view(Mq * H, Dh), pre_q, qn_w.detach(), rstd_q, mean_q, opt["qk_layernorm"], None,
                                        dpre.view(Mq * H, Dh), need[10], has_qnb and need[11])
            dqp = dpre
            dk_post = dkvp[:, :D].contiguous().view(Mk * H, Dh)
            dk_pre = torch.empty_like(dk_post)
            dqn[2], dqn[3] = K.norm_bwd(dk_post, pre_k, kn_w.detach(), rstd_k, mean_k, opt["qk_layernorm"], None, dk_pre,
                                        need[12], has_knb and need[13])
            dkvp[:, :D].copy_(dk_pre.view(Mk, D))
        dqn_in, dwq, dbq = _linear_bwd(dqp, qn, wq, True, need[4], has_bq and need[5], wdtype)
        dkv_in, dwkv, dbkv = _linear_bwd(dkvp, kv2, wkv, need[1], need[6], has_bkv and need[7], wdtype)
        if residual:
            if rows is None:
                dq_in = torch.empty_like(dnew2)
                add = dnew2
            else:
                dq_in = dnew2.clone()
                add = dq_in
        else:
            dq_in = torch.zeros_like(q2) if rows is not None else torch.empty_like(q2)
            add = None
        dnw, dnb = K.norm_bwd(dqn_in, q2, norm_w.detach(), rstd, mean, opt["layernorm"], add, dq_in, need[2],
                              has_nb and need[3], rows)
        dq_in = dq_in.view(B, Lq, D)
        if dq_in.dtype != qdtype:
            dq_in = dq_in.to(qdtype)
        if dkv_in is not None:
            dkv_in = dkv_in.view(Bk, Lk, D).to(kvdtype)
            if keep_idx is not None:   # samples outside the drop-path subset did not take part
                dkv_in = dkv_in.new_zeros((B, Lk, D)).index_copy_(0, keep_idx, dkv_in)
        if dgamma is not None and dgamma.dtype != g.dtype:
            dgamma = dgamma.to(g.dtype)
        return (dq_in if need[0] else None, dkv_in, dnw, dnb, dwq, dbq, dwkv, dbkv, dwo, dbo, *dqn, dgamma, None, None, None,
                None)


def _weights_dtype(tc: torch.dtype) -> torch.dtype:
    """dtype of (q @ k.mT * s).softmax(-1): softmax is on autocast's fp32 list, otherwise the operands' dtype."""
    return torch.float32 if torch.is_autocast_enabled("cuda") else tc


@torch.no_grad()
def self_attention_weights(x, norm_w, norm_b, w_qkv, b_qkv, qn_w, qn_b, kn_w, kn_b, rope, opt) -> Tensor:
    """Attention probabilities [B, H, S, S] of a self-attention layer (vit/attention.py:281-321)."""
    L.require_gpu(x, norm_w, w_qkv)
    B, S, D = x.shape
    Dh = opt["head_dim"]
    H = D // Dh
    tc = compute_dtype(w_qkv.dtype)
    M = B * S
    xn, _, _ = K.norm_fwd(x.reshape(M, D).contiguous(), norm_w, norm_b, opt["eps"], opt["layernorm"], tc)
    qkv = torch.empty((M, 3 * D), dtype=tc, device=x.device)
    K.gemm(xn, _cast_param(w_qkv, tc), trans_a=False, trans_b=False, M=M, N=3 * D, K=D, out=qkv, bias=_cast_param(b_qkv, tc))
    if qn_w is not None:
        for which, (w_, b_) in enumerate(((qn_w, qn_b), (kn_w, kn_b))):
            pre = qkv[:, which * D:(which + 1) * D].contiguous().view(M * H, Dh)
            nrm, _, _ = K.norm_fwd(pre, w_, b_, opt["qk_eps"], opt["qk_layernorm"], tc)
            qkv[:, which * D:(which + 1) * D].copy_(nrm.view(M, D))
    table, emu = _rope_table(rope, None, B, B, tc)
    if table is not None:
        K.rope_qk_(qkv, table, B, S, H, Dh, S - table.shape[-2], False, emu)
    return K.attn_weights(qkv[:, :D], qkv[:, D:2 * D], B, S, S, H, Dh, Dh ** -0.5).to(_weights_dtype(tc))


@torch.no_grad()
def cross_attention_weights(q, kv, norm_w, norm_b, w_q, b_q, w_kv, b_kv, qn_w, qn_b, kn_w, kn_b, rope_q, rope_k, opt) -> Tensor:
    """Attention probabilities [B, H, Lq, Lk] of a cross-attention layer (vit/attention.py:324-373)."""
    L.require_gpu(q, kv, norm_w, w_q, w_kv)
    B, Lq, D = q.shape
    Lk = kv.shape[1]
    Dh = opt["head_dim"]
    H = D // Dh
    tc = compute_dtype(w_q.dtype)
    Mq, Mk = B * Lq, B * Lk
    qn, _, _ = K.norm_fwd(q.reshape(Mq, D).contiguous(), norm_w, norm_b, opt["eps"], opt["layernorm"], tc)
    qp = torch.empty((Mq, D), dtype=tc, device=q.device)
    K.gemm(qn, _cast_param(w_q, tc), trans_a=False, trans_b=False, M=Mq, N=D, K=D, out=qp, bias=_cast_param(b_q, tc))
    kvp = torch.empty((Mk, 2 * D), dtype=tc, device=q.device)
    K.gemm(_cast(kv.reshape(Mk, D), tc), _cast_param(w_kv, tc), trans_a=False, trans_b=False, M=Mk, N=2 * D, K=D, out=kvp,
           bias=_cast_param(b_kv, tc))
    if qn_w is not None:
        qp = K.norm_fwd(qp.view(Mq * H, Dh), qn_w, qn_b, opt["qk_eps"], opt["qk_layernorm"], tc)[0].view(Mq, D)
        nk = K.norm_fwd(kvp[:, :D].contiguous().view(Mk * H, Dh), kn_w, kn_b, opt["qk_eps"], opt["qk_layernorm"], tc)[0]
        kvp[:, :D].copy_(nk.view(Mk, D))
    tq, emu_q = _rope_table(rope_q, None, B, B, tc)
    tk, emu_k = _rope_table(rope_k, None, B, B, tc)
    if tq is not None:
        K.rope_rows_(qp, tq, B, Lq, H, Dh, False, emu_q)
    if tk is not None:
        K.rope_rows_(kvp, tk, B, Lk, H, Dh, False, emu_k)
    return K.attn_weights(qp, kvp[:, :D], B, Lq, Lk, H, Dh, Dh ** -0.5).to(_weights_dtype(tc))


class RopeFn(torch.autograd.Function):
    """apply_rope on a [B, H, S, Dh] tensor (vit/rope.py:139-163): the last N rows of every head are rotated, computed in
    the table's dtype and cast back; the backward is the transposed rotation."""

    @staticmethod
    def forward(ctx, x, rope):
        L.require_gpu(x, rope)
        B, H, S, Dh = x.shape
        work = rope.dtype   # x.type_as(rope): the rotation is computed in the table's dtype
        emu = rope.dtype == torch.bfloat16
        if rope.dim() == 5:   # [2, B, 1, N, Dh] -> one table per (sample, head) row block
            table = rope.detach()[:, :, 0].to(torch.float32)[:, :, None].expand(2, B, H, -1, Dh).reshape(2, B * H, -1, Dh)
        else:
            table = rope.detach().to(torch.float32)
        table = table.contiguous()
        y = x.detach().to(work).contiguous().view(B * H * S, Dh).clone()
        K.rope_rows_(y, table, B * H, S, 1, Dh, False, emu)
        ctx.save_for_backward(table)
        ctx.meta = (B, H, S, Dh, emu, work, x.dtype)
        return y.view(B, H, S, Dh).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        (table,) = ctx.saved_tensors
        B, H, S, Dh, emu, work, xdtype = ctx.meta
        g = dy.to(work).contiguous().view(B * H * S, Dh).clone()
        K.rope_rows_(g, table, B * H, S, 1, Dh, True, emu)
        return g.view(B, H, S, Dh).to(xdtype), None


def apply_rope(x: Tensor, rope: Tensor) -> Tensor:
    return RopeFn.apply(x, rope)


class MlpSublayerFn(torch.autograd.Function):
    """x -> x (+) alpha * gamma * dropout(fc2(dropout(act(fc1(norm(x))))))   (vit/fused.py:96-166)."""

    @staticmethod
    def forward(ctx, x, norm_w, norm_b, w1, b1, w2, b2, gamma, keep_idx, opt):
        L.require_gpu(x, w1, w2)
        shape = x.shape
        D = shape[-1]
        S = shape[-2] if x.dim() >= 2 else 1
        tc = compute_dtype(w1.dtype)
        residual = opt["residual"]
        tr = _residual_dtype(x, gamma, tc) if residual else tc
        x2 = _cast(x.reshape(-1, D), tr if residual else x.dtype)
        rows = token_rows(keep_idx, S)
        M = x2.shape[0] if rows is None else rows.numel()
        training = opt["training"]
        p_hid = opt["dropout"] if training else 0.0
        F1, F = w1.shape[0], w2.shape[1]
        glu = opt["glu"]

        if norm_w is not None:
            xn, rstd, mean = K.norm_fwd(x2, norm_w.detach(), None if norm_b is None else norm_b.detach(), opt["eps"],
                                        opt["layernorm"], tc, rows)
        else:
            xn, rstd, mean = _cast(x2 if rows is None else x2.index_select(0, rows), tc), None, None
        w1c, b1c, w2c, b2c = _cast_param(w1, tc), _cast_param(b1, tc), _cast_param(w2, tc), _cast_param(b2, tc)
        p = torch.empty((M, F1), dtype=tc, device=x.device)
        seed_1 = next_seed(x.device) if p_hid > 0 else 0
        fused = K.can_fuse_activation(tc, D, F)
        if fused:  # activation (and GLU product) applied in the fc1 GEMM epilogue; p keeps the pre-activations
            h = torch.empty((M, F), dtype=tc, device=x.device)
            K.gemm(xn, w1c, trans_a=False, trans_b=False, M=M, N=F1, K=D, out=h, bias=b1c, aux=p,
                   epilogue=L.EPI_GLU if glu else L.EPI_ACT, act=opt["act"], limit=opt["limit"], extra=opt["extra_bias"],
                   drop_p=p_hid, seed=seed_1, tag="fc1")
        else:
            K.gemm(xn, w1c, trans_a=False, trans_b=False, M=M, N=F1, K=D, out=p, bias=b1c, tag="fc1")
            h = K.act_fwd(p, F, glu, opt["act"], opt["limit"], opt["extra_bias"], p_hid, seed_1)
        seed_2 = next_seed(x.device) if p_hid > 0 else 0
        g = _cast(gamma, tr)
        if residual:
            out, aux = _residual_gemm(h, w2c, b2c, x2, g, rows, opt["alpha"], p_hid, seed_2, want_aux=g is not None, tag="fc2",
                                      dropped=_dropped_samples(keep_idx), rows_per_sample=S)
            y = out.view(shape)
        else:
            zero = torch.zeros((M, D), dtype=tc, device=x.device)
            out, aux = _residual_gemm(h, w2c, b2c, zero, None, None, 1.0, p_hid, seed_2, want_aux=False)
            y = out.view(*shape[:-1], D) if rows is None else out.view(-1, S, D)
        ctx.save_for_backward(x2, norm_w, rstd, mean, xn, p, h, w1c, w2c, g, aux, rows)
        ctx.meta = (shape, D, F, tc, tr, p_hid, seed_1, seed_2, opt, x.dtype, w1.dtype, norm_b is not None, b1 is not None,
                    b2 is not None, fused)
        ctx.ptrs = _ptrs(w1, b1, w2, b2)
        ctx.dropped, ctx.rows_per_sample = _dropped_samples(keep_idx), S
        return y

    @staticmethod
    def backward(ctx, dnew):
        x2, norm_w, rstd, mean, xn, p, h, w1c, w2c, g, aux, rows = ctx.saved_tensors
        shape, D, F, tc, tr, p_hid, seed_1, seed_2, opt, xdtype, wdtype, has_nb, has_b1, has_b2, fused = ctx.meta
        need = ctx.needs_input_grad
        ptrs = ctx.ptrs
        M = xn.shape[0]
        residual = opt["residual"]
        dnew2 = _cast(dnew.reshape(-1, D), tr)
        dy, dgamma = _residual_grad(dnew2, tc, g if residual else None, rows if residual else None, M,
                                    opt["alpha"] if residual else 1.0, p_hid, seed_2, aux, need[7], _carried_bf16(dnew, D))
        if fused:  # fc2 dgrad with the activation backward in its epilogue: dh never reaches memory
            _, dw2, db2 = _linear_bwd(dy, h, w2c, False, need[5], has_b2 and need[6], wdtype, tag="fc2", w_ptr=ptrs[2],
                                      b_ptr=ptrs[3])
            dp = torch.empty_like(p)
            K.gemm(dy, w2c, trans_a=False, trans_b=True, M=M, N=F, K=D, out=dp, aux=p,
                   epilogue=L.EPI_DGLU if opt["glu"] else L.EPI_DACT, act=opt["act"], limit=opt["limit"],
                   extra=opt["extra_bias"], drop_p=p_hid, seed=seed_1, ldo=p.shape[1], tag="fc2.dgrad")
        else:
            dh, dw2, db2 = _linear_bwd(dy, h, w2c, True, need[5], has_b2 and need[6], wdtype, tag="fc2", w_ptr=ptrs[2],
                                       b_ptr=ptrs[3])
            dp = K.act_bwd(dh, p, F, opt["glu"], opt["act"], opt["limit"], opt["extra_bias"], p_hid, seed_1)
        dxn, dw1, db1 = _linear_bwd(dp, xn, w1c, True, need[3], has_b1 and need[4], wdtype, tag="fc1", w_ptr=ptrs[0], b_ptr=ptrs[1])
        dnw = dnb = None
        if norm_w is not None:
            if residual:
                if rows is None:
                    dx, add = torch.empty_like(dnew2), dnew2
                else:
                    dx, add = _passthrough(dnew2, ctx.dropped, ctx.rows_per_sample)
            else:
                dx = torch.zeros_like(x2) if rows is not None else torch.empty_like(x2)
                add = None
            shadow = _shadow_buffer(dx, tc, rows) if residual else None
            dnw, dnb = K.norm_bwd(dxn, x2, norm_w.detach(), rstd, mean, opt["layernorm"], add, dx, need[1],
                                  has_nb and need[2], rows, dx_bf16=shadow)
        else:
            shadow = None
            dx = dxn.to(x2.dtype)
            if residual:
                dx = dx + dnew2 if rows is None else dnew2.index_add(0, rows, dx)
        dx = dx.view(shape)
        if dx.dtype != xdtype:
            dx = dx.to(xdtype)
        else:
            _attach_bf16(dx, shadow)
        if dgamma is not None and dgamma.dtype != g.dtype:
            dgamma = dgamma.to(g.dtype)
        return (dx if need[0] else None, dnw, dnb, dw1, db1, dw2, db2, dgamma, None, None)


# ------------------------------------------------------------------------------------------------
# patch embedding: Conv2d(k = stride = patch) as im2col + GEMM, "+ pos" in the epilogue (vit/patch_embed.py:52-59)
# ------------------------------------------------------------------------------------------------
class PatchEmbedFn(torch.autograd.Function):
    """``fold`` > 1: x holds ``fold`` depth slabs per sample as consecutive batch entries (3-D patches with the depth
    offsets folded into the channels); the position table then covers fold * gh * gw tokens per sample.

    With ``cls_tokens`` / ``register_tokens`` ([1, n, D] each) the result is the whole input sequence of the blocks,
    cat([cls, register, patches], dim=1) in the promoted dtype (vit/vit.py:500-505): the GEMM epilogue writes the patch
    rows behind the prefix rows, one small kernel broadcasts the learned tokens, and no copy of the sequence is made."""

    @staticmethod
    def forward(ctx, x, w, b, pos, fold, cls_tokens, register_tokens):
        L.require_gpu(x, w)
        tc = compute_dtype(w.dtype)
        B, Cc, H, W = x.shape
        D, _, ph, pw = w.shape
        gh, gw = H // ph, W // pw
        N = gh * gw
        Kc = Cc * ph * pw
        Kp = (Kc + 63) // 64 * 64  # pad the contraction to the MFMA kernel's K-tile; padding columns are zero
        cols = K.im2col(x.detach(), ph, pw, Kp, tc)

        def padded(wd: Tensor) -> Tensor:
            wp = torch.zeros((D, Kp), dtype=tc, device=wd.device)
            wp[:, :Kc] = wd.reshape(D, Kc)
            return wp

        wc = _cast_param(w, tc, make=padded)
        bc = _cast_param(b, tc)
        # "y + pos" (vit/patch_embed.py:57): a table computed under autocast arrives in the compute dtype and the sum is
        # rounded to it; a wider table (the learned fp32 parameter under autocast) promotes the sum to its dtype
        pos_wide = pos is not None and torch.promote_types(pos.dtype, tc) != tc
        posc = None if pos is None else _cast(pos.reshape(N * fold, D), pos.dtype if pos_wide else tc)
        Bs, T = B // fold, N * fold          # samples, patch tokens per sample
        with_prefix = cls_tokens is not None
        ncls = cls_tokens.shape[1] if with_prefix else 0
        nreg = register_tokens.shape[1] if with_prefix else 0
        npre = ncls + nreg
        odt = posc.dtype if pos_wide else tc
        if with_prefix:
            odt = torch.promote_types(torch.promote_types(cls_tokens.dtype, register_tokens.dtype), odt)
        y = torch.empty((Bs, npre + T, D), dtype=odt, device=x.device)
        K.gemm(cols, wc, trans_a=False, trans_b=False, M=B * N, N=D, K=Kp, out=y, bias=bc, epilogue=L.EPI_PATCH, pos=posc,
               tokens=T, seq=npre + T, prefix=npre)
        if npre:
            pdt = torch.promote_types(cls_tokens.dtype, register_tokens.dtype)
            K.prefix_fill_(y, cls_tokens.detach().to(pdt), register_tokens.detach().to(pdt))
        ctx.save_for_backward(cols, wc)
        ctx.meta = (x.shape, w.shape, x.dtype, w.dtype, tc, N, Kc, Kp, b is not None, None if pos is None else (pos.shape, pos.dtype),
                    fold, ncls, nreg, None if not with_prefix else (cls_tokens.dtype, register_tokens.dtype), pos_wide)
        return y

    @staticmethod
    def backward(ctx, dy):
        cols, wc = ctx.saved_tensors
        xshape, wshape, xdtype, wdtype, tc, N, Kc, Kp, has_b, posinfo, fold, ncls, nreg, pdts, pos_wide = ctx.meta
        need = ctx.needs_input_grad
        B = xshape[0]
        D = wshape[0]
        Bs, T, npre = B // fold, N * fold, ncls + nreg
        S = npre + T
        dy = dy.contiguous()
        if npre:   # the patch rows of every sample, gathered and cast in one pass
            dy2 = K.scale_cast(dy.view(Bs, S * D)[:, npre * D:], Bs, T * D, tc, ldg=S * D).view(B * N, D)
        else:
            dy2 = _cast(dy.reshape(B * N, D), tc)
        dx = dw = db = dpos = dcls = dreg = None
        dcols, dwp, db = _linear_bwd(dy2, cols, wc, need[0], need[1], has_b and need[2], wdtype)
        if dwp is not None:
            dw = dwp[:, :Kc].reshape(wshape).contiguous()
        if dcols is not None:
            dx = K.col2im(dcols, xshape, wshape[2], wshape[3], xdtype)
        if posinfo is not None and need[3]:
            if pos_wide:   # the table entered the sum in the output's dtype: its gradient is the unrounded batch sum
                dpos = K.colsum(dy.view(Bs, S * D)[:, npre * D:], Bs, T * D, S * D, posinfo[1]).view(posinfo[0])
            else:
                dpos = K.colsum(dy2, Bs, T * D, T * D, posinfo[1]).view(posinfo[0])
        if npre:
            flat = dy.view(Bs, S * D)
            if need[5] and ncls:
                dcls = K.colsum(flat, Bs, ncls * D, S * D, pdts[0]).view(1, ncls, D)
            if need[6] and nreg:
                dreg = K.colsum(flat[:, ncls * D:], Bs, nreg * D, S * D, pdts[1]).view(1, nreg, D)
        return dx, dw, db, dpos, None, dcls, dreg


def patch_embed(x: Tensor, weight: Tensor, bias: Optional[Tensor], pos: Optional[Tensor], fold: int = 1,
                cls_tokens: Optional[Tensor] = None, register_tokens: Optional[Tensor] = None) -> Tensor:
    return PatchEmbedFn.apply(x, weight, bias, pos, fold, cls_tokens, register_tokens)


# ------------------------------------------------------------------------------------------------
# prefix tokens: cat([cls, reg, patches], dim=1)   (vit/vit.py:500-505)
# ------------------------------------------------------------------------------------------------
class PrefixConcatFn(torch.autograd.Function):
    """Copies are plain device memcpys; the only arithmetic, the batch-sum of the prefix-token gradients, is a
    libvitk column reduction."""

    @staticmethod
    def forward(ctx, cls_tokens, register_tokens, x):
        L.require_gpu(x, cls_tokens, register_tokens)
        B, N, D = x.shape
        ncls, nreg = cls_tokens.shape[1], register_tokens.shape[1]
        dtype = torch.promote_types(torch.promote_types(cls_tokens.dtype, register_tokens.dtype), x.dtype)
        out = torch.empty((B, ncls + nreg + N, D), dtype=dtype, device=x.device)
        if ncls:
            out[:, :ncls].copy_(cls_tokens.detach().expand(B, -1, -1))
        if nreg:
            out[:, ncls:ncls + nreg].copy_(register_tokens.detach().expand(B, -1, -1))
        out[:, ncls + nreg:].copy_(x.detach())
        ctx.meta = (B, N, D, ncls, nreg, cls_tokens.dtype, register_tokens.dtype, x.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, N, D, ncls, nreg, cdt, rdt, xdt = ctx.meta
        S = ncls + nreg + N
        dout = dout.contiguous()
        flat = dout.view(B, S * D)
        dcls = dreg = dx = None
        if ctx.needs_input_grad[0] and ncls:
            dcls = K.colsum(flat, B, ncls * D, S * D, cdt).view(1, ncls, D)
        if ctx.needs_input_grad[1] and nreg:
            dreg = K.colsum(flat[:, ncls * D:], B, nreg * D, S * D, rdt).view(1, nreg, D)
        if ctx.needs_input_grad[2]:
            dx = dout[:, ncls + nreg:].to(xdt)
        return dcls, dreg, dx


def concat_prefix(cls_tokens: Tensor, register_tokens: Tensor, x: Tensor) -> Tensor:
    return PrefixConcatFn.apply(cls_tokens, register_tokens, x)
