"""fc2 input gradient with the activation backward in its epilogue, timed alone:
python tools/bench_dglu.py [dglu|dact|glu] [M D F]   (defaults: the ViT-B/14 headline's shape; VITK_LIB picks the library)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import _lib as L
from vit_amd import kernels as K
mode = sys.argv[1] if len(sys.argv) > 1 else "dglu"
M, D, F = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (256 * 272, 768, 3072)
def rnd(*s): return (torch.rand(*s, device="cuda") * 2 - 1).to(torch.bfloat16)
gl = 2 if mode in ("dglu", "glu") else 1
a, w, aux = rnd(M, D), rnd(D, F), rnd(M, gl * F)
out = torch.empty(M, gl * F, device="cuda", dtype=torch.bfloat16)
epi = L.EPI_DGLU if mode == "dglu" else L.EPI_DACT
run = lambda: K.gemm(a, w, trans_a=False, trans_b=True, M=M, N=F, K=D, out=out, aux=aux, epilogue=epi, act=int(os.environ.get("ACT", "1")), ldo=gl * F)
if mode == "glu":   # the forward: fc1 + bias + gated activation, pre-activations kept
    w1, b1, h = rnd(2 * F, D), rnd(2 * F), torch.empty(M, F, device="cuda", dtype=torch.bfloat16)
    out = h
    run = lambda: K.gemm(a, w1, trans_a=False, trans_b=False, M=M, N=2 * F, K=D, out=h, aux=aux, bias=b1, epilogue=L.EPI_GLU, act=int(os.environ.get("ACT", "1")))
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print(f"{os.environ.get('VITK_LIB', 'libvitk')} {mode} M={M} D={D} F={F}: {e0.elapsed_time(e1)/50*1e3:.1f} us  checksum {out.float().abs().mean().item():.6f}")
