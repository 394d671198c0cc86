"""One GEMM shape, one kernel family: python tools/bench_gemm_one.py LAYOUT M N K ALGO [ITERS]   (prints ms and TFLOP/s;
a few launches only when ITERS is small, for counter runs)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import kernels as K
layout, m, n, k, algo = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 6
def rnd(*s): return (torch.rand(*s, device="cuda") * 2 - 1).to(torch.bfloat16)
if layout == "nt": a, b = rnd(m, k), rnd(n, k)
elif layout == "nn": a, b = rnd(m, k), rnd(k, n)
else: a, b = rnd(k, m), rnd(k, n)
out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
run = lambda: K.gemm(a, b, trans_a=layout == "tn", trans_b=layout != "nt", M=m, N=n, K=k, out=out, algo=algo)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"{os.environ.get('VITK_LIB', 'libvitk')} {layout} M={m} N={n} K={k} algo={algo}: {ms:.4f} ms  {2.0 * m * n * k / ms / 1e9:.1f} TF/s")
