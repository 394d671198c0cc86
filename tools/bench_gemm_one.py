"""One GEMM shape, one kernel family, a few launches (for counter runs): python tools/bench_gemm_one.py LAYOUT M N K ALGO"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import kernels as K
layout, m, n, k, algo = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
def rnd(*s): return (torch.rand(*s, device="cuda") * 2 - 1).to(torch.bfloat16)
if layout == "nt": a, b = rnd(m, k), rnd(n, k)
elif layout == "nn": a, b = rnd(m, k), rnd(k, n)
else: a, b = rnd(k, m), rnd(k, n)
out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
for _ in range(6):
    K.gemm(a, b, trans_a=layout == "tn", trans_b=layout != "nt", M=m, N=n, K=k, out=out, algo=algo)
torch.cuda.synchronize()
