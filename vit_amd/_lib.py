"""ctypes binding of libvitk.so (declared in include/vitk.h).  Fails loudly: there is no CPU path."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch
from torch import Tensor

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VITK_LIB") or os.path.join(_HERE, "libvitk.so")   # VITK_LIB: an experimental build

F32, BF16 = 0, 1
EPI_STORE, EPI_RESIDUAL, EPI_PATCH, EPI_ACT, EPI_GLU, EPI_DACT, EPI_DGLU, EPI_ROPE = 0, 1, 2, 3, 4, 5, 6, 7
ACT_CODES = {"relu": 0, "reglu": 0, "silu": 1, "swiglu": 1, "gelu": 2, "geglu": 2, "srelu": 3, "openswiglu": 4}

EXPORTS = [
    "vitk_version", "vitk_last_error", "vitk_norm_fwd", "vitk_norm_bwd", "vitk_norm_bwd_workspace_bytes",
    "vitk_qk_norm_bwd", "vitk_qk_norm_bwd_workspace_bytes",
    "vitk_gemm", "vitk_gemm_workspace_bytes", "vitk_attn_fwd", "vitk_attn_bwd", "vitk_attn_weights", "vitk_rope_qk", "vitk_rope_rows", "vitk_debug_hold_cus", "vitk_act_fwd",
    "vitk_act_bwd", "vitk_im2col", "vitk_col2im", "vitk_colsum", "vitk_colsum_static", "vitk_colsum_workspace_bytes", "vitk_scale_cast",
    "vitk_scale_cast_dgamma", "vitk_scale_cast_dgamma_workspace_bytes",
    "vitk_prefix_fill", "vitk_copy_samples", "vitk_fourier_features", "vitk_fourier_features_bwd",
]


class VitkGemm(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("b", C.c_void_p), ("out", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p),
        ("gamma", C.c_void_p), ("row_idx", C.c_void_p), ("aux", C.c_void_p), ("pos", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64),
        ("lda", C.c_int64), ("ldb", C.c_int64), ("ldo", C.c_int64),
        ("tokens", C.c_int64), ("seq", C.c_int64), ("prefix", C.c_int64),
        ("seed", C.c_uint64), ("alpha", C.c_float), ("drop_p", C.c_float),
        ("trans_a", C.c_int32), ("trans_b", C.c_int32), ("in_dtype", C.c_int32), ("out_dtype", C.c_int32),
        ("epilogue", C.c_int32), ("algo", C.c_int32),
        ("act", C.c_int32), ("use_limit", C.c_int32), ("use_extra", C.c_int32), ("limit", C.c_float), ("extra_bias", C.c_float),
        ("a_colsum", C.c_void_p), ("dynamic_tiles", C.c_int32), ("pos_f32", C.c_int32),
        ("rope", C.c_void_p), ("rope_width", C.c_int64), ("rope_emulate_bf16", C.c_int32),
        ("qk_norm_weight", C.c_void_p), ("qk_norm_bias", C.c_void_p), ("qk_rstd", C.c_void_p), ("qk_mean", C.c_void_p),
        ("qk_eps", C.c_float), ("qk_layernorm", C.c_int32),
        ("store_policy", C.c_int32), ("panel_m", C.c_int32),
    ]


class VitkAttn(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p),
        ("dout", C.c_void_p), ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p),
        ("lse", C.c_void_p), ("delta", C.c_void_p),
        ("B", C.c_int64), ("H", C.c_int64), ("S", C.c_int64), ("Dh", C.c_int64),
        ("q_sb", C.c_int64), ("q_sh", C.c_int64), ("q_ss", C.c_int64),
        ("o_sb", C.c_int64), ("o_sh", C.c_int64), ("o_ss", C.c_int64),
        ("dq_sb", C.c_int64), ("dq_sh", C.c_int64), ("dq_ss", C.c_int64),
        ("seed", C.c_uint64), ("scale", C.c_float), ("drop_p", C.c_float),
        ("dtype", C.c_int32), ("algo", C.c_int32),
        ("Sk", C.c_int64), ("kv_sb", C.c_int64), ("kv_sh", C.c_int64), ("kv_ss", C.c_int64),
        ("dkv_sb", C.c_int64), ("dkv_sh", C.c_int64), ("dkv_ss", C.c_int64),
        ("rope", C.c_void_p), ("rope_prefix", C.c_int64), ("rope_emulate_bf16", C.c_int32),
    ]


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libvitk.so once.  Raises if the extension has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"vit_amd: {LIB_PATH} is missing. Build it with `make -C vit_amd/csrc` (needs hipcc, gfx950). "
            "There is no CPU or PyTorch fallback for the encoder kernels."
        )
    lib = C.CDLL(LIB_PATH)
    lib.vitk_last_error.restype = C.c_char_p
    lib.vitk_version.restype = C.c_int
    for name in ("vitk_norm_bwd_workspace_bytes", "vitk_colsum_workspace_bytes", "vitk_qk_norm_bwd_workspace_bytes",
                 "vitk_scale_cast_dgamma_workspace_bytes"):
        getattr(lib, name).restype = C.c_size_t
        getattr(lib, name).argtypes = [C.c_int64]
    lib.vitk_gemm_workspace_bytes.restype = C.c_size_t
    lib.vitk_gemm_workspace_bytes.argtypes = [C.POINTER(VitkGemm)]
    vp, i32, i64, f32, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64
    sigs = {
        "vitk_norm_fwd": [vp, i32, vp, vp, i32, vp, i32, vp, vp, vp, i64, i64, f32, i32, vp],
        "vitk_norm_bwd": [vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, i32, vp, vp, vp, i64, i64, i32, vp, vp, vp],
        "vitk_qk_norm_bwd": [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i64, i64, i64, i64, i32, vp, vp],
        "vitk_gemm": [C.POINTER(VitkGemm), vp],
        "vitk_attn_fwd": [C.POINTER(VitkAttn), vp],
        "vitk_attn_bwd": [C.POINTER(VitkAttn), vp],
        "vitk_attn_weights": [C.POINTER(VitkAttn), vp, vp],
        "vitk_debug_hold_cus": [i32, C.c_longlong, vp],
        "vitk_rope_qk": [vp, i32, vp, i64, i64, i64, i64, i64, i32, i32, i32, vp],
        "vitk_rope_rows": [vp, i32, vp, i64, i64, i64, i64, i64, i64, i64, i32, i32, i32, vp],
        "vitk_act_fwd": [vp, vp, i32, i64, i64, i32, i32, i32, f32, i32, f32, f32, u64, vp],
        "vitk_act_bwd": [vp, vp, vp, i32, i64, i64, i32, i32, i32, f32, i32, f32, f32, u64, vp],
        "vitk_im2col": [vp, i32, vp, i32, i64, i64, i64, i64, i64, i64, i64, vp],
        "vitk_col2im": [vp, i32, vp, i32, i64, i64, i64, i64, i64, i64, i64, vp],
        "vitk_colsum": [vp, i32, i64, vp, vp, i32, i64, vp, i32, i64, i64, f32, vp, vp],
        "vitk_colsum_static": [vp, i32, i64, vp, vp, i32, i64, vp, i32, i64, i64, f32, vp],
        "vitk_scale_cast": [vp, i32, i64, vp, vp, vp, i32, i64, i64, f32, f32, u64, vp],
        "vitk_scale_cast_dgamma": [vp, i32, i64, vp, vp, vp, i32, vp, i32, vp, i64, i64, f32, f32, u64, vp, vp],
        "vitk_prefix_fill": [vp, i64, vp, i64, i32, vp, i32, i64, i64, i64, vp],
        "vitk_copy_samples": [vp, vp, vp, i64, i64, vp],
        "vitk_fourier_features": [vp, vp, vp, i32, i64, i64, i64, vp],
        "vitk_fourier_features_bwd": [vp, vp, vp, i32, vp, i64, i64, i64, vp],
    }
    for name, args in sigs.items():
        if not hasattr(lib, name) and os.environ.get("VITK_LIB"):
            continue   # an experimental build named by VITK_LIB may lack newer entry points
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError("libvitk: " + load().vitk_last_error().decode())


def dt(t_or_dtype) -> int:
    d = t_or_dtype.dtype if isinstance(t_or_dtype, Tensor) else t_or_dtype
    if d == torch.float32:
        return F32
    if d == torch.bfloat16:
        return BF16
    raise TypeError(f"vit_amd kernels take float32 or bfloat16, got {d}")


def ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def require_gpu(*tensors: Optional[Tensor]) -> None:
    """Every tensor on the GPU, and on the CURRENT one: launches go to the current device's stream and its workspace, so
    a model on another device has to run under ``torch.cuda.device(...)`` (one process per GPU does this once)."""
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "vit_amd: the encoder kernels run on an MI355X (ROCm) device only; got a tensor on "
                f"'{t.device}'. There is no CPU fallback."
            )
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise RuntimeError(
                f"vit_amd: tensor on '{t.device}' but the current device is cuda:{cur}; kernels are launched on the "
                "current device's stream. Call torch.cuda.set_device(...) (or run under torch.cuda.device(...)) first."
            )
