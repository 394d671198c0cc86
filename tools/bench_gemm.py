"""GEMM micro-benchmark on the encoder's shapes (random bf16 data), libvitk vs torch.matmul (hipBLASLt) as a
same-hardware reference point.  Usage: python tools/bench_gemm.py [B]"""
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import _lib as L
from vit_amd import kernels as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ONLY = sys.argv[2] if len(sys.argv) > 2 else None      # run one named case only
NOREF = os.environ.get("NOREF") == "1"                  # skip the torch.matmul comparison (for counter runs)
M, D, F = B * 272, 768, 3072
dev = "cuda"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def rnd(*shape, dtype=torch.bfloat16):
    return (torch.rand(*shape, device=dev) * 2 - 1).to(dtype)


cases = [  # name, layout, M, N, K, epilogue, out dtype
    ("qkv fwd", "nt", M, 3 * D, D, "store", torch.bfloat16),
    ("fc1 fwd", "nt", M, 2 * F, D, "store", torch.bfloat16),
    ("out fwd", "nt", M, D, D, "res", torch.float32),
    ("fc2 fwd", "nt", M, D, F, "res", torch.float32),
    ("fc1 glu", "nt", M, 2 * F, D, "glu", torch.bfloat16),
    ("fc2 dgrad", "nn", M, F, D, "store", torch.bfloat16),
    ("fc2 dglu", "nn", M, F, D, "dglu", torch.bfloat16),
    ("fc2 dglu0", "nn", M, F, D, "dglu0", torch.bfloat16),   # relu gate: the same memory traffic, trivial arithmetic
    ("fc1 dgrad", "nn", M, D, 2 * F, "store", torch.bfloat16),
    ("qkv dgrad", "nn", M, D, 3 * D, "store", torch.bfloat16),
    ("out dgrad", "nn", M, D, D, "store", torch.bfloat16),
    ("fc1 wgrad", "tn", 2 * F, D, M, "store", torch.float32),
    ("fc2 wgrad", "tn", D, F, M, "store", torch.float32),
    ("qkv wgrad", "tn", 3 * D, D, M, "store", torch.float32),
    ("out wgrad", "tn", D, D, M, "store", torch.float32),
]
tot_ms, tot_fl = 0.0, 0.0
for name, layout, m, n, k, epi, odt in cases:
    if ONLY and name != ONLY:
        continue
    if layout == "nt":
        a, b = rnd(m, k), rnd(n, k)
        ref = lambda: torch.matmul(a, b.t())
    elif layout == "nn":
        a, b = rnd(m, k), rnd(k, n)
        ref = lambda: torch.matmul(a, b)
    else:
        a, b = rnd(k, m), rnd(k, n)
        ref = lambda: torch.matmul(a.t(), b)
    out = torch.empty(m, n, device=dev, dtype=odt)
    kw = {}
    if epi == "res":
        res = torch.randn(m, n, device=dev)
        kw = dict(epilogue=L.EPI_RESIDUAL, res=res, bias=rnd(n))
    if epi == "glu":    # out = silu(gate) * value, pre-activations kept in aux
        out = torch.empty(m, n // 2, device=dev, dtype=odt)
        kw = dict(epilogue=L.EPI_GLU, aux=torch.empty(m, n, device=dev, dtype=odt), bias=rnd(n), act=1)
    if epi in ("dglu", "dglu0"):   # out = d(pre-activations) [m, 2n] from dh = a @ b and the saved pre-activations
        out = torch.empty(m, 2 * n, device=dev, dtype=odt)
        kw = dict(epilogue=L.EPI_DGLU, aux=rnd(m, 2 * n), act=int(epi == "dglu"), ldo=2 * n)
    fn = lambda: K.gemm(a, b, trans_a=layout == "tn", trans_b=layout != "nt", M=m, N=n, K=k, out=out, algo=4, **kw)
    fn256 = lambda: K.gemm(a, b, trans_a=layout == "tn", trans_b=layout != "nt", M=m, N=n, K=k, out=out, algo=3, **kw)
    fn256d = lambda: K.gemm(a, b, trans_a=layout == "tn", trans_b=layout != "nt", M=m, N=n, K=k, out=out, algo=5, **kw)
    ms = timeit(fn, 5 if NOREF else 20)
    ms256 = timeit(fn256, 5 if NOREF else 20)
    ms256d = timeit(fn256d, 5 if NOREF else 20)
    fnauto = lambda: K.gemm(a, b, trans_a=layout == "tn", trans_b=layout != "nt", M=m, N=n, K=k, out=out, algo=0, **kw)
    msauto = timeit(fnauto, 5 if NOREF else 20)

    ms_ref = ms if NOREF else timeit(ref)
    fl = 2.0 * m * n * k
    tot_ms += ms
    tot_fl += fl
    print(f"{name:10s} {layout} M={m:6d} N={n:5d} K={k:6d}  t128 {ms:6.3f} ms {fl / ms / 1e9:6.1f} TF  t256 {ms256:6.3f} ms {fl / ms256 / 1e9:6.1f} TF  t256dma {ms256d:6.3f} ms {fl / ms256d / 1e9:6.1f} TF  auto {msauto:6.3f} ms {fl / msauto / 1e9:6.1f} TF  torch.matmul {ms_ref:7.3f} ms {fl / ms_ref / 1e9:7.1f} TF/s", flush=True)
print(f"sum over one layer: {tot_ms:.3f} ms, {tot_fl / tot_ms / 1e9:.1f} TF/s")
