"""A/B of the MFMA GEMM's tuning knobs (VitkGemm.store_policy, panel_m) on the encoder step's products with their real
epilogues: variants are timed in interleaved rounds inside one process, median and minimum per variant.
Usage: python tools/bench_policy.py [rounds] [case-substring]     (PANELS=8,4,16,...: a panel_m sweep at the default store policy)"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import _lib as L
from vit_amd import kernels as K

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 7
ONLY = sys.argv[2] if len(sys.argv) > 2 else None
B, S, D, F = 256, 272, 768, 3072
M = B * S
dev = "cuda"
VARIANTS = [(1, 8), (2, 8), (3, 8), (1, 4), (2, 4), (2, 6), (2, 12), (2, 16), (1, 16)]   # (store_policy, panel_m)
if os.environ.get("PANELS"):
    VARIANTS = [(0, int(v)) for v in os.environ["PANELS"].split(",")]


def rnd(*shape, dtype=torch.bfloat16):
    return (torch.rand(*shape, device=dev) * 2 - 1).to(dtype)


def case(name, layout, m, n, k, epi):
    if layout == "nt":
        a, b = rnd(m, k), rnd(n, k)
    elif layout == "nn":
        a, b = rnd(m, k), rnd(k, n)
    else:
        a, b = rnd(k, m), rnd(k, n)
    odt = torch.float32 if epi in ("res", "wgrad") else torch.bfloat16
    out = torch.empty(m, n, device=dev, dtype=odt)
    kw = {}
    if epi == "res":
        kw = dict(epilogue=L.EPI_RESIDUAL, res=torch.randn(m, n, device=dev), bias=rnd(n))
    if epi == "glu":
        out = torch.empty(m, n // 2, device=dev, dtype=odt)
        kw = dict(epilogue=L.EPI_GLU, aux=torch.empty(m, n, device=dev, dtype=odt), bias=rnd(n), act=1)
    if epi == "dglu":
        out = torch.empty(m, 2 * n, device=dev, dtype=odt)
        kw = dict(epilogue=L.EPI_DGLU, aux=rnd(m, 2 * n), act=1, ldo=2 * n)
    if epi == "bias":
        kw = dict(bias=rnd(n))
    return name, (lambda sp, pm: K.gemm(a, b, trans_a=layout == "tn", trans_b=layout != "nt", M=m, N=n, K=k, out=out,
                                        store_policy=sp, panel_m=pm, **kw)), 2.0 * m * n * k


CASES = [
    ("fc1 + SwiGLU", "nt", M, 2 * F, D, "glu"), ("fc2 dgrad + DGLU", "nn", M, F, D, "dglu"), ("fc2 + residual", "nt", M, D, F, "res"),
    ("out + residual", "nt", M, D, D, "res"), ("qkv fwd", "nt", M, 3 * D, D, "bias"), ("fc1 dgrad", "nn", M, D, 2 * F, "store"),
    ("qkv dgrad", "nn", M, D, 3 * D, "store"), ("fc1 wgrad", "tn", 2 * F, D, M, "wgrad"), ("fc2 wgrad", "tn", D, F, M, "wgrad"),
    ("qkv wgrad", "tn", 3 * D, D, M, "wgrad"), ("out wgrad", "tn", D, D, M, "wgrad"), ("out dgrad", "nn", M, D, D, "store"),
]
for spec in CASES:
    if ONLY and ONLY not in spec[0]:
        continue
    name, fn, fl = case(*spec)
    times = {v: [] for v in VARIANTS}
    for v in VARIANTS:
        fn(*v)
    torch.cuda.synchronize()
    for _ in range(ROUNDS):
        for v in VARIANTS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                fn(*v)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 4)
    base = statistics.median(times[VARIANTS[0]])
    print(f"{name}  ({fl / 1e9:.0f} GFLOP)")
    for v in VARIANTS:
        med, mn = statistics.median(times[v]), min(times[v])
        print(f"   store_policy {v[0]} panel_m {v[1]:2d}: median {med:.4f} ms  min {mn:.4f} ms  {fl / med / 1e9:7.1f} TF/s  {med / base:5.3f} x", flush=True)
