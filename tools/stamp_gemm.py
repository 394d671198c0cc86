"""Reads the per-segment cycle sums an EXP_STAMP build of gemm_mfma256d_k keeps (diagnostics only):
VITK_LIB=exp/libvitk_stamp.so python tools/stamp_gemm.py [store|glu|dglu|res|out|fc2|wgrad1|wgrad2|wgradq|nn1|nnq]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import _lib as L
from vit_amd import kernels as K
M, D, F = 256 * 272, 768, 3072
def rnd(*s): return (torch.rand(*s, device="cuda") * 2 - 1).to(torch.bfloat16)
mode = sys.argv[1] if len(sys.argv) > 1 else "store"
a = rnd(M, D)
if mode == "glu":
    w, aux, out = rnd(2 * F, D), torch.empty(M, 2 * F, device="cuda", dtype=torch.bfloat16), torch.empty(M, F, device="cuda", dtype=torch.bfloat16)
    run = lambda: K.gemm(a, w, trans_a=False, trans_b=False, M=M, N=2 * F, K=D, out=out, aux=aux, bias=rnd(2 * F), epilogue=L.EPI_GLU, act=1, algo=5)
elif mode == "store":
    w, out = rnd(2 * F, D), torch.empty(M, 2 * F, device="cuda", dtype=torch.bfloat16)
    run = lambda: K.gemm(a, w, trans_a=False, trans_b=False, M=M, N=2 * F, K=D, out=out, algo=5)
elif mode in ("out", "fc2"):   # narrow outputs: N = D, K = D or F, plain bf16 store
    kk = D if mode == "out" else F
    a = rnd(M, kk)
    w, out = rnd(D, kk), torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    run = lambda: K.gemm(a, w, trans_a=False, trans_b=False, M=M, N=D, K=kk, out=out, algo=5)
elif mode.startswith("wgrad"):   # weight gradients: dW[n_out, n_in] = dy^T @ x, contraction over the M rows, split-K partials
    n_out, n_in = {"wgrad1": (2 * F, D), "wgrad2": (D, F), "wgradq": (3 * D, D)}[mode]
    dy, x, out = rnd(M, n_out), rnd(M, n_in), torch.empty(n_out, n_in, device="cuda")
    run = lambda: K.gemm(dy, x, trans_a=True, trans_b=True, M=n_out, N=n_in, K=M, out=out, algo=5)
elif mode.startswith("nn"):      # input gradients: dx[M, n_in] = dy[M, n_out] @ W[n_out, n_in]
    n_out, n_in = {"nn1": (2 * F, D), "nnq": (3 * D, D)}[mode]
    dy, w, out = rnd(M, n_out), rnd(n_out, n_in), torch.empty(M, n_in, device="cuda", dtype=torch.bfloat16)
    run = lambda: K.gemm(dy, w, trans_a=False, trans_b=True, M=M, N=n_in, K=n_out, out=out, algo=5)
elif mode == "res":
    a = rnd(M, F)
    w, out, res = rnd(D, F), torch.empty(M, D, device="cuda"), torch.randn(M, D, device="cuda")
    run = lambda: K.gemm(a, w, trans_a=False, trans_b=False, M=M, N=D, K=F, out=out, res=res, bias=rnd(D), epilogue=L.EPI_RESIDUAL, algo=5)
else:
    w, aux, out = rnd(D, F), rnd(M, 2 * F), torch.empty(M, 2 * F, device="cuda", dtype=torch.bfloat16)
    run = lambda: K.gemm(a, w, trans_a=False, trans_b=True, M=M, N=F, K=D, out=out, aux=aux, epilogue=L.EPI_DGLU, act=1, ldo=2 * F, algo=5)
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
buf = np.zeros((512, 8), dtype=np.uint64)
lib = L.load()
lib.vitk_debug_stamps.argtypes = [ctypes.c_void_p]
assert lib.vitk_debug_stamps(buf.ctypes.data) == 0
names = ["setup+first stage", "main loop", "post-loop sync", "next-tile request", "epilogue", "final wait"]
for g in (0, 1):
    d = buf[g::2].astype(np.float64)
    d = d[d[:, 6] > 0]
    tiles = d[:, 6].mean()
    per = d[:, :6].sum(0) / d[:, 6].sum()
    tot = per.sum()
    print(f"{mode} group {g}: {ms if False else ''}tiles/block {tiles:.1f}; cycles per tile {tot:.0f}: " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names, per)))
print(f"launch {e0.elapsed_time(e1):.4f} ms")
