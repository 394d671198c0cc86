import os, sys, torch
sys.path.insert(0, '/root/repo')
from vit_amd import kernels as K, _lib as L
M, N = 69632, 768
def rnd(*s): return (torch.rand(*s, device="cuda") * 2 - 1).to(torch.bfloat16)
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for Kd in (768, 3072):
    a, b = rnd(M, Kd), rnd(N, Kd)
    res = torch.randn(M, N, device="cuda")
    o16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    o32 = torch.empty(M, N, device="cuda")
    for algo in (4, 5):
        t1 = timeit(lambda: K.gemm(a, b, trans_a=False, trans_b=False, M=M, N=N, K=Kd, out=o16, algo=algo))
        t2 = timeit(lambda: K.gemm(a, b, trans_a=False, trans_b=False, M=M, N=N, K=Kd, out=o32, algo=algo))
        t3 = timeit(lambda: K.gemm(a, b, trans_a=False, trans_b=False, M=M, N=N, K=Kd, out=o32, algo=algo, epilogue=L.EPI_RESIDUAL, res=res))
        print(f"K={Kd} algo={algo}: store bf16 {t1*1e3:.0f} us, store fp32 {t2*1e3:.0f} us, residual fp32 {t3*1e3:.0f} us", flush=True)
# pure streaming add for reference: out = res + x (fp32 read, bf16 read, fp32 write)
x16 = rnd(M, N); res = torch.randn(M, N, device="cuda"); o32 = torch.empty(M, N, device="cuda")
t = timeit(lambda: torch.add(res, x16, out=o32))
print(f"torch add fp32+bf16->fp32: {t*1e3:.0f} us  ({(M*N*10)/t/1e9:.2f} TB/s)")
