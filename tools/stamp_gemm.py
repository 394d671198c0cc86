"""Reads the cycle stamps an EXP_STAMP build of gemm_mfma256d_k leaves in the aux buffer (experiments only)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import _lib as L
from vit_amd import kernels as K
M, D, F = 256 * 272, 768, 3072
def rnd(*s): return (torch.rand(*s, device="cuda") * 2 - 1).to(torch.bfloat16)
mode = sys.argv[1] if len(sys.argv) > 1 else "glu"
a = rnd(M, D)
if mode == "glu":
    w, aux, out = rnd(2 * F, D), torch.empty(M, 2 * F, device="cuda", dtype=torch.bfloat16), torch.empty(M, F, device="cuda", dtype=torch.bfloat16)
    run = lambda: K.gemm(a, w, trans_a=False, trans_b=False, M=M, N=2 * F, K=D, out=out, aux=aux, bias=rnd(2 * F), epilogue=L.EPI_GLU, act=1, algo=5)
    nb = (M // 256) * (2 * F // 256)
else:
    w, aux, out = rnd(D, F), rnd(M, 2 * F), torch.empty(M, 2 * F, device="cuda", dtype=torch.bfloat16)
    run = lambda: K.gemm(a, w, trans_a=False, trans_b=True, M=M, N=F, K=D, out=out, aux=aux, epilogue=L.EPI_DGLU, act=1, ldo=2 * F, algo=5)
    nb = (M // 256) * (F // 256)
for _ in range(3): run()
torch.cuda.synchronize()
d = aux.view(torch.int32).flatten()[: nb * 8].view(nb, 8).cpu().to(torch.int64) & 0xFFFFFFFF
print(f"{mode}: blocks {nb}: prologue {d[:,2].float().mean():.0f}  loop {d[:,3].float().mean():.0f}  epilogue issue {d[:,4].float().mean():.0f}  store drain {d[:,5].float().mean():.0f}")
t0 = d[:, 0] + (d[:, 1] << 32); dur = d[:, 2] + d[:, 3] + d[:, 4] + d[:, 5]
key = ((d[:, 7] & 15) * 1000 + ((d[:, 6] >> 13) & 7) * 100 + ((d[:, 6] >> 12) & 1) * 10 + ((d[:, 6] >> 8) & 15)).tolist()
g = collections.defaultdict(list)
for i, k in enumerate(key): g[k].append(i)
spans, conc, gaps = [], [], []
for k, idx in g.items():
    idx = torch.tensor(idx); o = torch.argsort(t0[idx]); idx = idx[o]
    s = (t0[idx] + dur[idx]).max() - t0[idx].min(); spans.append(s.item()); conc.append(dur[idx].sum().item() / s.item())
    gaps += (t0[idx][1:] - (t0[idx][:-1] + dur[idx][:-1])).tolist()
import statistics
print(f"CUs {len(g)}: span {statistics.mean(spans):.0f} ticks, resident fraction {statistics.mean(conc):.2f}, gap between blocks mean {statistics.mean(gaps):.0f} ticks")
