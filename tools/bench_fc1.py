"""The fc1 forward GEMM of the headline workload (ViT-B/14, batch 256: [69632, 768] x [6144, 768]^T with the fused
SwiGLU epilogue), launched a few times through the C ABI -- the program rocprofv3 counter passes are taken on."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import _lib as L
from vit_amd import kernels as K
M, D, F = 256 * 272, 768, 3072
def rnd(*s): return (torch.rand(*s, device="cuda") * 2 - 1).to(torch.bfloat16)
a, w, b = rnd(M, D), rnd(2 * F, D), rnd(2 * F)
aux = torch.empty(M, 2 * F, device="cuda", dtype=torch.bfloat16)
out = torch.empty(M, F, device="cuda", dtype=torch.bfloat16)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    K.gemm(a, w, trans_a=False, trans_b=False, M=M, N=2 * F, K=D, out=out, aux=aux, bias=b, epilogue=L.EPI_GLU, act=1)
torch.cuda.synchronize()
