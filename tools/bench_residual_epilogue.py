"""The out_proj / fc2 forward with the residual epilogue, timed alone against plain stores of the same product:
python tools/bench_residual_epilogue.py [M N]   (defaults: the ViT-B/14 headline's rows and width; VITK_LIB picks the library)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import kernels as K, _lib as L
M, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (69632, 768)
def rnd(*s): return (torch.rand(*s, device="cuda") * 2 - 1).to(torch.bfloat16)
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
lib = os.environ.get("VITK_LIB", "libvitk")
for Kd in (N, 4 * N):
    a, b = rnd(M, Kd), rnd(N, Kd)
    res = torch.randn(M, N, device="cuda")
    res16 = res.to(torch.bfloat16)
    o16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    o32 = torch.empty(M, N, device="cuda")
    for algo in (4, 5):
        t1 = timeit(lambda: K.gemm(a, b, trans_a=False, trans_b=False, M=M, N=N, K=Kd, out=o16, algo=algo))
        t2 = timeit(lambda: K.gemm(a, b, trans_a=False, trans_b=False, M=M, N=N, K=Kd, out=o32, algo=algo))
        t3 = timeit(lambda: K.gemm(a, b, trans_a=False, trans_b=False, M=M, N=N, K=Kd, out=o32, algo=algo, epilogue=L.EPI_RESIDUAL, res=res))
        t4 = timeit(lambda: K.gemm(a, b, trans_a=False, trans_b=False, M=M, N=N, K=Kd, out=o16, algo=algo, epilogue=L.EPI_RESIDUAL, res=res16))
        print(f"{lib} K={Kd} algo={algo}: store bf16 {t1*1e3:.0f} us, store fp32 {t2*1e3:.0f} us, residual fp32 {t3*1e3:.0f} us, "
              f"residual bf16 {t4*1e3:.0f} us  checksums {o32.abs().mean().item():.6f} {o16.float().abs().mean().item():.6f}", flush=True)
