"""Norm forward / backward micro-benchmark on the encoder's shapes (fp32 residual stream, bf16 compute) through the C ABI."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import kernels as K
M, D = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (256 * 272, 768)))
x = torch.randn(M, D, device="cuda"); w = torch.ones(D, device="cuda")
dy = torch.randn(M, D, device="cuda").bfloat16(); add = torch.randn(M, D, device="cuda")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
y, rstd, _ = K.norm_fwd(x, w, None, 1e-5, False, torch.bfloat16)
dx = torch.empty_like(x); sh = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
f = timeit(lambda: K.norm_fwd(x, w, None, 1e-5, False, torch.bfloat16))
b = timeit(lambda: K.norm_bwd(dy, x, w, rstd, None, False, add, dx, True, False, None, dx_bf16=sh))
gb_f = M * D * 6 / 1e9; gb_b = M * D * (2 + 4 + 4 + 4 + 2) / 1e9
print(f"M={M} D={D}: fwd {f*1e3:.1f} us ({gb_f/f:.2f} TB/s)  bwd {b*1e3:.1f} us ({gb_b/b:.2f} TB/s)")
