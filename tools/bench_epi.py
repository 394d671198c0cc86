"""Epilogue-only timing: fused-epilogue GEMMs with a tiny contraction (K = 64), so the time is the epilogue's memory pass."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import _lib as L
from vit_amd import kernels as K

M, D, F = 256 * 272, 768, 3072
dev = "cuda"
def rnd(*s): return (torch.rand(*s, device=dev) * 2 - 1).to(torch.bfloat16)
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for Kd in (64, 768):
    a, b = rnd(M, Kd), rnd(Kd, F)
    aux, out = rnd(M, 2 * F), torch.empty(M, 2 * F, device=dev, dtype=torch.bfloat16)
    for algo in (4, 3):
        t = timeit(lambda: K.gemm(a, b, trans_a=False, trans_b=True, M=M, N=F, K=Kd, out=out, aux=aux, epilogue=L.EPI_DGLU, act=1, ldo=2 * F, algo=algo))
        gb = (aux.numel() + out.numel()) * 2 / 1e9
        print(f"DGLU K={Kd} algo={algo}: {t:.3f} ms  epilogue bytes {gb:.2f} GB -> {gb / t:.2f} TB/s")
    o2 = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
    for algo in (4, 3):
        t = timeit(lambda: K.gemm(a, b, trans_a=False, trans_b=True, M=M, N=F, K=Kd, out=o2, algo=algo))
        print(f"STORE K={Kd} algo={algo}: {t:.3f} ms  bytes {o2.numel() * 2 / 1e9:.2f} GB -> {o2.numel() * 2 / 1e9 / t:.2f} TB/s")
    w1 = rnd(2 * F, Kd)
    h = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
    for algo in (4, 3):
        t = timeit(lambda: K.gemm(a, w1, trans_a=False, trans_b=False, M=M, N=2 * F, K=Kd, out=h, aux=aux, bias=rnd(2 * F), epilogue=L.EPI_GLU, act=1, algo=algo))
        gb = (aux.numel() + h.numel()) * 2 / 1e9
        print(f"GLU K={Kd} algo={algo}: {t:.3f} ms  epilogue bytes {gb:.2f} GB -> {gb / t:.2f} TB/s")
    x = torch.randn(M, D, device=dev)
    wo = rnd(D, Kd)
    xo = torch.empty_like(x)
    for algo in (4, 3):
        t = timeit(lambda: K.gemm(a, wo, trans_a=False, trans_b=False, M=M, N=D, K=Kd, out=xo, res=x, bias=rnd(D), epilogue=L.EPI_RESIDUAL, algo=algo))
        gb = (x.numel() + xo.numel()) * 4 / 1e9
        print(f"RESIDUAL K={Kd} algo={algo}: {t:.3f} ms  epilogue bytes {gb:.2f} GB -> {gb / t:.2f} TB/s")
# plain device copy as the bandwidth yardstick
src = torch.empty(M * 2 * F, device=dev, dtype=torch.bfloat16); dst = torch.empty_like(src)
t = timeit(lambda: dst.copy_(src))
print(f"copy {src.numel() * 4 / 1e9:.2f} GB: {t:.3f} ms -> {src.numel() * 4 / 1e9 / t:.2f} TB/s")
