import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import _lib as L
from vit_amd import kernels as K
M, D, F = 256 * 272, 768, 3072
def rnd(*s): return (torch.rand(*s, device="cuda") * 2 - 1).to(torch.bfloat16)
a, w, aux = rnd(M, D), rnd(D, F), rnd(M, 2 * F)
out = torch.empty(M, 2 * F, device="cuda", dtype=torch.bfloat16)
run = lambda: K.gemm(a, w, trans_a=False, trans_b=True, M=M, N=F, K=D, out=out, aux=aux, epilogue=L.EPI_DGLU, act=1, ldo=2 * F)
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print(f"dglu {e0.elapsed_time(e1)/50*1e3:.1f} us")
