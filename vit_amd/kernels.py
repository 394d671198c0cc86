"""Tensor-level wrappers over the C ABI: one Python function per libvitk entry point.

These check devices/contiguity, allocate outputs and workspaces with torch (PyTorch owns all device
memory), and pass raw pointers + sizes + the current HIP stream.  No arithmetic happens here.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch
from torch import Tensor

from . import _lib as L

_ALGO = 0  # 0 auto, 1 portable kernels, 2 MFMA kernels (tests flip this)


def set_algo(algo: int) -> int:
    global _ALGO
    old, _ALGO = _ALGO, int(algo)
    return old


_DYNAMIC_TILES = False  # large-tile GEMM blocks draw tiles from counters (set by the data-parallel reducer)


def set_dynamic_tiles(on: bool) -> bool:
    """Ask the large-tile GEMM launches of this process to hand out tiles dynamically: for processes whose collectives
    run on a side stream beside the GEMMs and hold CUs meanwhile.  Returns the previous setting."""
    global _DYNAMIC_TILES
    old, _DYNAMIC_TILES = _DYNAMIC_TILES, bool(on)
    return old


# Tuning knobs of the MFMA GEMM (VitkGemm.store_policy / panel_m): 0 = the library's choice.  Experiments set them per
# process through the environment or per call through gemm()'s arguments.
_STORE_POLICY = int(os.environ.get("VITK_STORE_POLICY", "0"))
_PANEL_M = int(os.environ.get("VITK_PANEL_M", "0"))


def _c(t: Tensor) -> Tensor:
    return t if t.is_contiguous() else t.contiguous()


_WS: dict = {}

# Optional per-launch timing of tagged GEMMs with HIP events on the launch stream (used by bench.py for the
# roofline of the dominant kernel).  _EVENTS = {"tags": set[str], "pairs": {tag: [(start, end), ...]}} or None.
_EVENTS: Optional[dict] = None


def profile_gemms(tags) -> dict:
    global _EVENTS
    _EVENTS = None if tags is None else {"tags": set(tags), "pairs": {t: [] for t in tags}}
    return _EVENTS


def _workspace(nbytes: int, device) -> Optional[Tensor]:
    """Per-(device, stream) scratch, grown on demand; contents never outlive one kernel sequence."""
    if nbytes <= 0:
        return None
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


# ------------------------------------------------------------------------------------------------
def norm_fwd(x2: Tensor, w: Tensor, b: Optional[Tensor], eps: float, layernorm: bool, out_dtype: torch.dtype,
             row_idx: Optional[Tensor] = None):
    """x2 [R_in, D] -> y [rows, D], rstd [rows], mean [rows] | None.  rows = len(row_idx) or R_in."""
    lib = L.load()
    L.require_gpu(x2, w, b, row_idx)
    x2 = _c(x2)
    D = x2.shape[-1]
    rows = x2.shape[0] if row_idx is None else row_idx.numel()
    y = torch.empty((rows, D), dtype=out_dtype, device=x2.device)
    rstd = torch.empty((rows,), dtype=torch.float32, device=x2.device)
    mean = torch.empty((rows,), dtype=torch.float32, device=x2.device) if layernorm else None
    if b is not None and b.dtype != w.dtype:
        b = b.to(w.dtype)
    L.check(lib.vitk_norm_fwd(L.ptr(x2), L.dt(x2), L.ptr(_c(w)), L.ptr(None if b is None else _c(b)), L.dt(w), L.ptr(y),
                              L.dt(y), L.ptr(rstd), L.ptr(mean), L.ptr(row_idx), rows, D, float(eps), int(layernorm),
                              L.stream()))
    return y, rstd, mean


def norm_bwd(dy: Tensor, x2: Tensor, w: Tensor, rstd: Tensor, mean: Optional[Tensor], layernorm: bool,
             add: Optional[Tensor], dx: Tensor, want_dw: bool, want_db: bool, row_idx: Optional[Tensor] = None,
             dx_bf16: Optional[Tensor] = None):
    """dx[row(r)] = (add[row(r)] if add is not None else 0) + dnorm/dx; returns (dw, db) in w.dtype.
    ``dx_bf16`` (same shape as dx, bfloat16) also receives the written rows cast to bf16."""
    lib = L.load()
    L.require_gpu(dy, x2, w, dx)
    dy, x2 = _c(dy), _c(x2)
    rows, D = dy.shape
    assert dx.is_contiguous() and (add is None or (add.is_contiguous() and add.dtype == dx.dtype))
    assert dx_bf16 is None or (dx_bf16.dtype == torch.bfloat16 and dx_bf16.shape == dx.shape and dx_bf16.is_contiguous())
    dw = torch.empty_like(w) if want_dw else None
    db = torch.empty_like(w) if (want_db and layernorm) else None
    ws = _workspace(lib.vitk_norm_bwd_workspace_bytes(D), dy.device) if (dw is not None or db is not None) else None
    # the kernel computes dw and db together; ask for dw storage whenever either is wanted
    dw_buf = dw if dw is not None else (torch.empty_like(w) if db is not None else None)
    L.check(lib.vitk_norm_bwd(L.ptr(dy), L.dt(dy), L.ptr(x2), L.dt(x2), L.ptr(_c(w)), L.dt(w), L.ptr(rstd), L.ptr(mean),
                              L.ptr(add), L.ptr(dx), L.dt(dx), L.ptr(dw_buf), L.ptr(db), L.ptr(row_idx), rows, D,
                              int(layernorm), L.ptr(ws), L.ptr(dx_bf16), L.stream()))
    return dw, db


def can_qk_norm_bwd_inplace(dqkv: Tensor, Dh: int) -> bool:
    return dqkv.is_contiguous() and Dh % 8 == 0 and Dh <= 128 and dqkv.dtype in (torch.float32, torch.bfloat16)


def qk_norm_bwd_(dqkv: Tensor, pre_q: Tensor, pre_k: Tensor, rstd_q: Tensor, rstd_k: Tensor, mean_q: Optional[Tensor],
                 mean_k: Optional[Tensor], w_q: Tensor, w_k: Tensor, H: int, Dh: int, layernorm: bool, want: tuple):
    """In place on the packed gradient [M, 3 * H * Dh]: sections q and k go from the gradient of the normalised heads to
    the gradient of the heads themselves.  ``want`` = (dw_q, db_q, dw_k, db_k) flags; returns those four (or None)."""
    lib = L.load()
    L.require_gpu(dqkv, pre_q, pre_k, w_q, w_k)
    M = dqkv.shape[0]
    assert dqkv.is_contiguous() and pre_q.is_contiguous() and pre_k.is_contiguous() and pre_q.dtype == dqkv.dtype == pre_k.dtype
    assert pre_q.numel() == M * H * Dh == pre_k.numel() and w_q.dtype == w_k.dtype
    outs = [torch.empty_like(w_q) if (f and (i % 2 == 0 or layernorm)) else None for i, f in enumerate(want)]
    # the kernel sums weight and bias gradients together: a wanted bias gradient needs its weight slot too
    slots = list(outs)
    for i in (0, 2):
        if slots[i] is None and slots[i + 1] is not None:
            slots[i] = torch.empty_like(w_q)
    ws = _workspace(lib.vitk_qk_norm_bwd_workspace_bytes(Dh), dqkv.device) if any(t is not None for t in slots) else None
    L.check(lib.vitk_qk_norm_bwd(L.ptr(dqkv), L.dt(dqkv), L.ptr(pre_q), L.ptr(pre_k), L.ptr(rstd_q), L.ptr(rstd_k), L.ptr(mean_q),
                                 L.ptr(mean_k), L.ptr(_c(w_q)), L.ptr(_c(w_k)), L.dt(w_q), L.ptr(slots[0]), L.ptr(slots[1]),
                                 L.ptr(slots[2]), L.ptr(slots[3]), M, H, Dh, dqkv.shape[1], int(layernorm), L.ptr(ws), L.stream()))
    return tuple(outs)


def gemm(a: Tensor, b: Tensor, *, trans_a: bool, trans_b: bool, M: int, N: int, K: int, out: Tensor,
         bias: Optional[Tensor] = None, epilogue: int = L.EPI_STORE, res: Optional[Tensor] = None,
         gamma: Optional[Tensor] = None, row_idx: Optional[Tensor] = None, aux: Optional[Tensor] = None,
         pos: Optional[Tensor] = None, alpha: float = 1.0, drop_p: float = 0.0, seed: int = 0,
         tokens: int = 0, seq: int = 0, prefix: int = 0, ldo: Optional[int] = None, algo: Optional[int] = None,
         tag: str = "", act: int = 0, limit: Optional[float] = None, extra: Optional[float] = None,
         a_colsum: Optional[Tensor] = None, rope: Optional[Tensor] = None, rope_width: int = 0,
         rope_emulate_bf16: bool = False, qk_norm: Optional[dict] = None, store_policy: Optional[int] = None,
         panel_m: Optional[int] = None) -> Tensor:
    """``a_colsum`` ([M], dtype of out; trans_a products only) receives sum_k A[k, m] from the same pass.
    ``rope`` (EPI_ROPE): fp32 table [2, seq - prefix, 64]; sections of ``rope_width`` columns, the first two rotated."""
    lib = L.load()
    L.require_gpu(a, b, out, bias, res, gamma, aux, pos, a_colsum)
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1, "gemm operands must be row-major 2-D"
    assert a.dtype == b.dtype
    for t in (bias, aux):
        assert t is None or (t.dtype == a.dtype and t.is_contiguous())
    assert rope is None or seq > prefix >= 0
    # a position table wider than the operands (a learned fp32 table under autocast) is added unrounded to an fp32 output
    pos_f32 = pos is not None and pos.dtype == torch.float32 and a.dtype != torch.float32
    assert pos is None or (pos.is_contiguous() and (pos.dtype == a.dtype or (pos_f32 and out.dtype == torch.float32)))
    for t in (res, gamma):
        assert t is None or (t.dtype == out.dtype and t.is_contiguous())
    assert out.is_contiguous()
    g = L.VitkGemm()
    g.a, g.b, g.out = L.ptr(a), L.ptr(b), L.ptr(out)
    g.bias, g.res, g.gamma, g.row_idx, g.aux, g.pos = (L.ptr(bias), L.ptr(res), L.ptr(gamma), L.ptr(row_idx), L.ptr(aux),
                                                       L.ptr(pos))
    assert a_colsum is None or (a_colsum.dtype == out.dtype and a_colsum.is_contiguous() and a_colsum.numel() == M)
    g.a_colsum = L.ptr(a_colsum)
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb = a.stride(0), b.stride(0)
    g.ldo = out.shape[-1] if ldo is None else ldo
    g.act, g.use_limit, g.use_extra = act, int(limit is not None), int(extra is not None)
    g.limit, g.extra_bias = float(limit or 0.0), float(extra or 0.0)
    g.tokens, g.seq, g.prefix = tokens, seq, prefix
    g.seed, g.alpha, g.drop_p = seed & 0xFFFFFFFFFFFFFFFF, alpha, drop_p
    g.trans_a, g.trans_b = int(trans_a), int(trans_b)
    g.in_dtype, g.out_dtype = L.dt(a), L.dt(out)
    g.epilogue = epilogue
    g.algo = _ALGO if algo is None else algo
    g.dynamic_tiles = int(_DYNAMIC_TILES)
    g.store_policy = _STORE_POLICY if store_policy is None else store_policy
    g.panel_m = _PANEL_M if panel_m is None else panel_m
    g.pos_f32 = int(pos_f32)
    if rope is not None:
        L.require_gpu(rope)
        assert rope.dtype == torch.float32 and rope.is_contiguous() and tuple(rope.shape) == (2, seq - prefix, 64)
    g.rope, g.rope_width, g.rope_emulate_bf16 = L.ptr(rope), rope_width, int(rope_emulate_bf16)
    if qk_norm is not None:   # {"weight": [2, 64] f32, "bias": [2, 64] f32 | None, "rstd", "mean" | None, "eps", "layernorm"}
        for t in (qk_norm["weight"], qk_norm["bias"], qk_norm["rstd"], qk_norm["mean"]):
            assert t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous())
        assert aux is not None and aux.numel() == 2 * M * rope_width
        g.qk_norm_weight, g.qk_norm_bias = L.ptr(qk_norm["weight"]), L.ptr(qk_norm["bias"])
        g.qk_rstd, g.qk_mean = L.ptr(qk_norm["rstd"]), L.ptr(qk_norm["mean"])
        g.qk_eps, g.qk_layernorm = float(qk_norm["eps"]), int(qk_norm["layernorm"])
    g.workspace, g.workspace_bytes = None, 0
    need = lib.vitk_gemm_workspace_bytes(C.byref(g))
    if need:
        ws = _workspace(need, a.device)
        g.workspace, g.workspace_bytes = L.ptr(ws), ws.numel()
    if _EVENTS is not None and tag in _EVENTS["tags"]:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.vitk_gemm(C.byref(g), L.stream()))
        e1.record()
        _EVENTS["pairs"][tag].append((e0, e1))
        return out
    L.check(lib.vitk_gemm(C.byref(g), L.stream()))
    return out


def can_fuse_rope(dtype: torch.dtype, Dh: int, table: Optional[Tensor]) -> bool:
    """The QKV GEMM can rotate q and k in its epilogue, and the attention backward un-rotate dq and dk in its stores, for
    bf16 operands, head_dim 64 and one table shared by all samples."""
    if os.environ.get("VITK_NO_ROPE_FUSE"):   # measurements: the rotation as a pass of its own
        return False
    return table is not None and table.dim() == 3 and dtype == torch.bfloat16 and _ALGO != 1 and Dh == 64


def can_fuse_qk_norm(dtype: torch.dtype, Dh: int) -> bool:
    """The QKV GEMM can normalise the heads of q and k in its epilogue (bf16 operands, head_dim 64)."""
    if os.environ.get("VITK_NO_ROPE_FUSE"):
        return False
    return dtype == torch.bfloat16 and _ALGO != 1 and Dh == 64


def can_fuse_activation(dtype: torch.dtype, D: int, F: int) -> bool:
    """The MFMA GEMM can apply the MLP activation (and its backward) in its epilogue for bf16 operands."""
    return dtype == torch.bfloat16 and _ALGO != 1 and F % 32 == 0 and D % 8 == 0


def _attn_desc(qkv: Tensor, o: Tensor, lse: Tensor, B: int, S: int, H: int, Dh: int, scale: float, drop_p: float,
               seed: int, algo: Optional[int]) -> L.VitkAttn:
    """qkv is the packed projection [B*S, 3*H*Dh]; o is [B*S, H*Dh]."""
    D = H * Dh
    es = qkv.element_size()
    a = L.VitkAttn()
    base = qkv.data_ptr()
    a.q, a.k, a.v = base, base + D * es, base + 2 * D * es
    a.o, a.lse = L.ptr(o), L.ptr(lse)
    a.B, a.H, a.S, a.Dh = B, H, S, Dh
    a.q_sb, a.q_sh, a.q_ss = S * 3 * D, Dh, 3 * D
    a.o_sb, a.o_sh, a.o_ss = S * D, Dh, D
    a.seed, a.scale, a.drop_p = seed & 0xFFFFFFFFFFFFFFFF, scale, drop_p
    a.dtype = L.dt(qkv)
    a.algo = _ALGO if algo is None else algo
    return a


def attn_fwd(qkv: Tensor, B: int, S: int, H: int, Dh: int, scale: float, drop_p: float = 0.0, seed: int = 0,
             algo: Optional[int] = None):
    lib = L.load()
    L.require_gpu(qkv)
    assert qkv.is_contiguous() and qkv.shape == (B * S, 3 * H * Dh)
    o = torch.empty((B * S, H * Dh), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=qkv.device)
    a = _attn_desc(qkv, o, lse, B, S, H, Dh, scale, drop_p, seed, algo)
    L.check(lib.vitk_attn_fwd(C.byref(a), L.stream()))
    return o, lse


def attn_bwd(qkv: Tensor, o: Tensor, lse: Tensor, dout: Tensor, B: int, S: int, H: int, Dh: int, scale: float,
             drop_p: float = 0.0, seed: int = 0, algo: Optional[int] = None, rope: Optional[Tensor] = None,
             rope_emulate_bf16: bool = False, qk_norm: Optional[dict] = None) -> Tensor:
    """Returns dqkv in the packed layout of qkv.  ``rope`` (fp32 [2, S - prefix, 64]): q and k were rotated with this
    table before the attention; dq and dk come back through the transposed rotation."""
    lib = L.load()
    L.require_gpu(qkv, o, lse, dout)
    dout = _c(dout)
    D = H * Dh
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, H, S), dtype=torch.float32, device=qkv.device)
    a = _attn_desc(qkv, o, lse, B, S, H, Dh, scale, drop_p, seed, algo)
    es = qkv.element_size()
    a.dout, a.delta = L.ptr(dout), L.ptr(delta)
    base = dqkv.data_ptr()
    a.dq, a.dk, a.dv = base, base + D * es, base + 2 * D * es
    a.dq_sb, a.dq_sh, a.dq_ss = S * 3 * D, Dh, 3 * D
    if rope is not None:
        L.require_gpu(rope)
        assert rope.dtype == torch.float32 and rope.is_contiguous() and rope.dim() == 3 and rope.shape[2] == Dh == 64
        a.rope, a.rope_prefix, a.rope_emulate_bf16 = L.ptr(rope), S - rope.shape[1], int(rope_emulate_bf16)
    L.check(lib.vitk_attn_bwd(C.byref(a), L.stream()))
    return dqkv


def _cross_desc(q: Tensor, kv: Tensor, o: Tensor, lse: Tensor, B: int, Sq: int, Sk: int, H: int, Dh: int, scale: float,
                drop_p: float, seed: int, algo: Optional[int] = None) -> L.VitkAttn:
    """q [B*Sq, H*Dh]; kv [B*Sk, 2*H*Dh] = [k | v]; o [B*Sq, H*Dh]."""
    D = H * Dh
    a = L.VitkAttn()
    a.q, a.k, a.v = q.data_ptr(), kv.data_ptr(), kv.data_ptr() + D * kv.element_size()
    a.o, a.lse = L.ptr(o), L.ptr(lse)
    a.B, a.H, a.S, a.Dh, a.Sk = B, H, Sq, Dh, Sk
    a.q_sb, a.q_sh, a.q_ss = Sq * D, Dh, D
    a.kv_sb, a.kv_sh, a.kv_ss = Sk * 2 * D, Dh, 2 * D
    a.o_sb, a.o_sh, a.o_ss = Sq * D, Dh, D
    a.seed, a.scale, a.drop_p = seed & 0xFFFFFFFFFFFFFFFF, scale, drop_p
    a.dtype = L.dt(q)
    a.algo = _ALGO if algo is None else algo
    return a


def attn_cross_fwd(q: Tensor, kv: Tensor, B: int, Sq: int, Sk: int, H: int, Dh: int, scale: float, drop_p: float = 0.0,
                   seed: int = 0, algo: Optional[int] = None):
    """Attention of Sq queries over Sk keys per sample (vit/attention.py:269-273 with Lq != Lk)."""
    lib = L.load()
    L.require_gpu(q, kv)
    D = H * Dh
    assert q.is_contiguous() and kv.is_contiguous() and q.shape == (B * Sq, D) and kv.shape == (B * Sk, 2 * D)
    assert q.dtype == kv.dtype
    o = torch.empty_like(q)
    lse = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
    a = _cross_desc(q, kv, o, lse, B, Sq, Sk, H, Dh, scale, drop_p, seed, algo)
    L.check(lib.vitk_attn_fwd(C.byref(a), L.stream()))
    return o, lse


def attn_cross_bwd(q: Tensor, kv: Tensor, o: Tensor, lse: Tensor, dout: Tensor, B: int, Sq: int, Sk: int, H: int, Dh: int,
                   scale: float, drop_p: float = 0.0, seed: int = 0, algo: Optional[int] = None):
    """Returns (dq [B*Sq, D], dkv [B*Sk, 2D])."""
    lib = L.load()
    L.require_gpu(q, kv, o, lse, dout)
    dout = _c(dout)
    D = H * Dh
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    delta = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
    a = _cross_desc(q, kv, o, lse, B, Sq, Sk, H, Dh, scale, drop_p, seed, algo)
    a.dout, a.delta = L.ptr(dout), L.ptr(delta)
    a.dq, a.dk, a.dv = dq.data_ptr(), dkv.data_ptr(), dkv.data_ptr() + D * dkv.element_size()
    a.dq_sb, a.dq_sh, a.dq_ss = Sq * D, Dh, D
    a.dkv_sb, a.dkv_sh, a.dkv_ss = Sk * 2 * D, Dh, 2 * D
    L.check(lib.vitk_attn_bwd(C.byref(a), L.stream()))
    return dq, dkv


def attn_weights(q: Tensor, k: Tensor, B: int, Sq: int, Sk: int, H: int, Dh: int, scale: float) -> Tensor:
    """softmax(scale * q k^T) as [B, H, Sq, Sk] fp32.  q, k: 2-D views [B*Sq, >= H*Dh] / [B*Sk, >= H*Dh] whose rows hold
    the heads side by side from column 0 of the view (row strides may exceed H*Dh: packed projections)."""
    lib = L.load()
    L.require_gpu(q, k)
    assert q.dtype == k.dtype and q.stride(1) == 1 and k.stride(1) == 1
    w = torch.empty((B, H, Sq, Sk), dtype=torch.float32, device=q.device)
    a = L.VitkAttn()
    a.q, a.k, a.v, a.o, a.lse = q.data_ptr(), k.data_ptr(), k.data_ptr(), None, None
    a.B, a.H, a.S, a.Dh, a.Sk = B, H, Sq, Dh, Sk
    a.q_sb, a.q_sh, a.q_ss = Sq * q.stride(0), Dh, q.stride(0)
    a.kv_sb, a.kv_sh, a.kv_ss = Sk * k.stride(0), Dh, k.stride(0)
    a.scale, a.dtype = scale, L.dt(q)
    L.check(lib.vitk_attn_weights(C.byref(a), L.ptr(w), L.stream()))
    return w


def rope_qk_(qkv: Tensor, table_f32: Tensor, B: int, S: int, H: int, Dh: int, prefix: int, inverse: bool,
             emulate_bf16: bool) -> Tensor:
    lib = L.load()
    L.require_gpu(qkv, table_f32)
    assert qkv.is_contiguous() and table_f32.is_contiguous() and table_f32.dtype == torch.float32
    per_sample = table_f32.dim() == 4   # [2, B, S - prefix, Dh]: one table per sample
    want = (2, B, S - prefix, Dh) if per_sample else (2, S - prefix, Dh)
    assert tuple(table_f32.shape) == want, f"rope table {tuple(table_f32.shape)} vs {want}"
    L.check(lib.vitk_rope_qk(L.ptr(qkv), L.dt(qkv), L.ptr(table_f32), B, S, H, Dh, prefix, int(inverse), int(emulate_bf16),
                             int(per_sample), L.stream()))
    return qkv


def rope_rows_(x: Tensor, table_f32: Tensor, B: int, S: int, H: int, Dh: int, inverse: bool, emulate_bf16: bool) -> Tensor:
    """Rotate, in place, the first H*Dh columns of x [B*S, W] (W = H*Dh for q, 2*H*Dh for a packed [k | v])."""
    lib = L.load()
    L.require_gpu(x, table_f32)
    assert x.is_contiguous() and x.shape[0] == B * S and x.shape[1] >= H * Dh
    assert table_f32.is_contiguous() and table_f32.dtype == torch.float32
    per_sample = table_f32.dim() == 4
    n_rot = table_f32.shape[-2]
    want = (2, B, n_rot, Dh) if per_sample else (2, n_rot, Dh)
    assert tuple(table_f32.shape) == want and n_rot <= S, f"rope table {tuple(table_f32.shape)} vs {want}"
    L.check(lib.vitk_rope_rows(L.ptr(x), L.dt(x), L.ptr(table_f32), B, S, H, Dh, S - n_rot, x.shape[1], 1, int(inverse),
                               int(emulate_bf16), int(per_sample), L.stream()))
    return x


def act_fwd(p: Tensor, F: int, glu: bool, act: int, limit: Optional[float], extra: Optional[float], drop_p: float,
            seed: int) -> Tensor:
    lib = L.load()
    L.require_gpu(p)
    rows = p.shape[0]
    h = torch.empty((rows, F), dtype=p.dtype, device=p.device)
    L.check(lib.vitk_act_fwd(L.ptr(p), L.ptr(h), L.dt(p), rows, F, int(glu), act, int(limit is not None), float(limit or 0.0),
                             int(extra is not None), float(extra or 0.0), drop_p, seed & 0xFFFFFFFFFFFFFFFF, L.stream()))
    return h


def act_bwd(dh: Tensor, p: Tensor, F: int, glu: bool, act: int, limit: Optional[float], extra: Optional[float],
            drop_p: float, seed: int) -> Tensor:
    lib = L.load()
    L.require_gpu(dh, p)
    dh = _c(dh)
    dp = torch.empty_like(p)
    L.check(lib.vitk_act_bwd(L.ptr(dh), L.ptr(p), L.ptr(dp), L.dt(p), p.shape[0], F, int(glu), act, int(limit is not None),
                             float(limit or 0.0), int(extra is not None), float(extra or 0.0), drop_p,
                             seed & 0xFFFFFFFFFFFFFFFF, L.stream()))
    return dp


def im2col(x: Tensor, ph: int, pw: int, ldc: int, out_dtype: torch.dtype) -> Tensor:
    lib = L.load()
    L.require_gpu(x)
    x = _c(x)
    B, Cc, H, W = x.shape
    cols = torch.empty((B * (H // ph) * (W // pw), ldc), dtype=out_dtype, device=x.device)
    L.check(lib.vitk_im2col(L.ptr(x), L.dt(x), L.ptr(cols), L.dt(cols), B, Cc, H, W, ph, pw, ldc, L.stream()))
    return cols


def col2im(cols: Tensor, shape, ph: int, pw: int, out_dtype: torch.dtype) -> Tensor:
    lib = L.load()
    L.require_gpu(cols)
    B, Cc, H, W = shape
    dx = torch.empty(shape, dtype=out_dtype, device=cols.device)
    L.check(lib.vitk_col2im(L.ptr(_c(cols)), L.dt(cols), L.ptr(dx), L.dt(dx), B, Cc, H, W, ph, pw, cols.shape[1], L.stream()))
    return dx


def colsum(a: Tensor, rows: int, N: int, lda: int, out_dtype: torch.dtype, *, a_idx: Optional[Tensor] = None,
           b: Optional[Tensor] = None, ldb: int = 0, scale: float = 1.0) -> Tensor:
    """out[n] = scale * sum_r a[ra(r)*lda + n] * (b[r*ldb + n] if b is not None else 1)."""
    lib = L.load()
    L.require_gpu(a, b, a_idx)
    out = torch.empty((N,), dtype=out_dtype, device=a.device)
    ws = _workspace(lib.vitk_colsum_workspace_bytes(N), a.device)
    L.check(lib.vitk_colsum(L.ptr(a), L.dt(a), lda, L.ptr(a_idx), L.ptr(b), L.dt(b) if b is not None else 0, ldb, L.ptr(out),
                            L.dt(out), rows, N, scale, L.ptr(ws), L.stream()))
    return out


def scale_cast(g: Tensor, rows: int, N: int, out_dtype: torch.dtype, *, row_idx: Optional[Tensor] = None,
               gamma: Optional[Tensor] = None, alpha: float = 1.0, drop_p: float = 0.0, seed: int = 0,
               ldg: Optional[int] = None) -> Tensor:
    """``ldg``: elements between consecutive rows of ``g`` (default N); with it ``g`` may be a view whose rows are
    N contiguous elements ldg apart."""
    lib = L.load()
    L.require_gpu(g, gamma, row_idx)
    if ldg is None:
        g, ldg = _c(g), N
    assert gamma is None or gamma.dtype == g.dtype
    out = torch.empty((rows, N), dtype=out_dtype, device=g.device)
    L.check(lib.vitk_scale_cast(L.ptr(g), L.dt(g), ldg, L.ptr(row_idx), L.ptr(gamma), L.ptr(out), L.dt(out), rows, N, alpha,
                                drop_p, seed & 0xFFFFFFFFFFFFFFFF, L.stream()))
    return out


def scale_cast_dgamma(g: Tensor, aux: Tensor, rows: int, N: int, out_dtype: torch.dtype, gamma: Tensor, *,
                      row_idx: Optional[Tensor] = None, alpha: float = 1.0, drop_p: float = 0.0, seed: int = 0):
    """scale_cast with the layer-scale gradient from the same pass: returns (out, dgamma), dgamma[n] = alpha * sum_r
    g[row(r), n] * aux[r, n] in the dtype of gamma."""
    lib = L.load()
    L.require_gpu(g, aux, gamma, row_idx)
    g, aux = _c(g), _c(aux)
    assert gamma.dtype == g.dtype and aux.shape[-1] == N and N <= 4096
    out = torch.empty((rows, N), dtype=out_dtype, device=g.device)
    dgamma = torch.empty_like(gamma)
    ws = _workspace(lib.vitk_scale_cast_dgamma_workspace_bytes(N), g.device)
    L.check(lib.vitk_scale_cast_dgamma(L.ptr(g), L.dt(g), N, L.ptr(row_idx), L.ptr(_c(gamma)), L.ptr(aux), L.dt(aux), L.ptr(out),
                                       L.dt(out), L.ptr(dgamma), rows, N, alpha, drop_p, seed & 0xFFFFFFFFFFFFFFFF, L.ptr(ws),
                                       L.stream()))
    return out, dgamma


def fourier_features(grid: Tensor, w: Tensor, out_dtype: torch.dtype) -> Tensor:
    """grid [N, Kc] fp32, w [Kc, J] fp32 -> [N, 2 J] = cat(sin(grid @ w), cos(grid @ w)) / sqrt(J) in out_dtype."""
    lib = L.load()
    L.require_gpu(grid, w)
    assert grid.dtype == torch.float32 and w.dtype == torch.float32 and grid.shape[1] == w.shape[0]
    grid, w = _c(grid), _c(w)
    N, Kc, J = grid.shape[0], grid.shape[1], w.shape[1]
    out = torch.empty((N, 2 * J), dtype=out_dtype, device=grid.device)
    L.check(lib.vitk_fourier_features(L.ptr(grid), L.ptr(w), L.ptr(out), L.dt(out), N, Kc, J, L.stream()))
    return out


def fourier_features_bwd(grid: Tensor, w: Tensor, g: Tensor) -> Tensor:
    """d w [Kc, J] fp32 from g = d features [N, 2 J]."""
    lib = L.load()
    L.require_gpu(grid, w, g)
    grid, w, g = _c(grid), _c(w), _c(g)
    N, Kc, J = grid.shape[0], grid.shape[1], w.shape[1]
    assert g.shape == (N, 2 * J)
    dw = torch.empty((Kc, J), dtype=torch.float32, device=grid.device)
    L.check(lib.vitk_fourier_features_bwd(L.ptr(grid), L.ptr(w), L.ptr(g), L.dt(g), L.ptr(dw), N, Kc, J, L.stream()))
    return dw


def copy_samples_(dst: Tensor, src: Tensor, idx: Tensor, sample_elems: int) -> Tensor:
    """dst[idx[i]] = src[idx[i]] with dst / src viewed as [-1, sample_elems] (same dtype, contiguous; idx int64)."""
    lib = L.load()
    L.require_gpu(dst, src, idx)
    assert dst.is_contiguous() and src.is_contiguous() and dst.dtype == src.dtype and dst.numel() == src.numel()
    assert idx.dtype == torch.int64 and dst.numel() % sample_elems == 0
    L.check(lib.vitk_copy_samples(L.ptr(src), L.ptr(dst), L.ptr(_c(idx)), idx.numel(), sample_elems * dst.element_size(),
                                  L.stream()))
    return dst


def prefix_fill_(out: Tensor, cls_tokens: Tensor, register_tokens: Tensor) -> Tensor:
    """Rows 0 .. ncls + nreg - 1 of every sample of out [B, S, D] <- cls [1, ncls, D], register tokens [1, nreg, D]."""
    lib = L.load()
    L.require_gpu(out, cls_tokens, register_tokens)
    B, S, D = out.shape
    ncls, nreg = cls_tokens.shape[1], register_tokens.shape[1]
    assert out.is_contiguous() and cls_tokens.dtype == register_tokens.dtype
    cls_tokens, register_tokens = _c(cls_tokens), _c(register_tokens)
    L.check(lib.vitk_prefix_fill(L.ptr(cls_tokens) if ncls else None, ncls, L.ptr(register_tokens) if nreg else None, nreg,
                                 L.dt(cls_tokens), L.ptr(out), L.dt(out), B, S, D, L.stream()))
    return out
