"""What a GEMM on the main stream loses while idle workgroups on a side stream hold some CUs (a stand-in for a
collective that runs beside the backward pass)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import _lib as L
from vit_amd import kernels as K
lib = L.load()
K.set_dynamic_tiles(os.environ.get("DYNAMIC_TILES", "1") != "0")   # DYNAMIC_TILES=0: fixed shares
M, D, F = 256 * 272, 768, 3072
def rnd(*s): return (torch.rand(*s, device="cuda") * 2 - 1).to(torch.bfloat16)
a, w = rnd(M, D), rnd(2 * F, D)
out = torch.empty(M, 2 * F, device="cuda", dtype=torch.bfloat16)
run = lambda: K.gemm(a, w, trans_a=False, trans_b=False, M=M, N=2 * F, K=D, out=out)
side = torch.cuda.Stream()
def timed(hog_blocks):
    for _ in range(3): run()
    torch.cuda.synchronize()
    if hog_blocks:
        with torch.cuda.stream(side):
            L.check(lib.vitk_debug_hold_cus(hog_blocks, 60_000_000, side.cuda_stream))   # ~30 ms
        torch.cuda._sleep(2_000_000)   # let the idle blocks settle on their CUs before the GEMMs arrive
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10
base = timed(0)
print(f"fc1 forward GEMM alone: {base:.3f} ms")
for hb in (16, 32, 64, 128, 192):
    t = timed(hb)
    print(f"  with {hb:3d} CUs held: {t:.3f} ms  ({t / base:.2f}x; ideal {256 / (256 - hb):.2f}x)")
