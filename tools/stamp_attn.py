"""Reads the cycle stamps an EXP_STAMP build of attn_fwd_k leaves in the lse buffer (experiments only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import kernels as K
B, S, H, Dh = 256, 272, 12, 64
qkv = torch.randn(B * S, 3 * H * Dh, device="cuda").to(torch.bfloat16)
for _ in range(3):
    o, lse = K.attn_fwd(qkv, B, S, H, Dh, Dh ** -0.5)
torch.cuda.synchronize()
nw = int(os.environ.get("VITK_ATTN_NW", "3"))
gx = -(-S // (32 * nw))
nb = gx * H * B
d = lse.view(torch.int32).flatten()[: nb * 8].view(nb, 8).cpu().to(torch.int64) & 0xFFFFFFFF
# only blocks whose wave 0 lane 0 reaches the end record (q-blocks with valid first query: all)
t0 = d[:, 0] + (d[:, 1] << 32)
dur = d[:, 2] + d[:, 3] + d[:, 4] + d[:, 5]
print(f"blocks {nb}: prologue {d[:,2].float().mean():.0f}  loop {d[:,3].float().mean():.0f}  tail {d[:,4].float().mean():.0f}  epilogue {d[:,5].float().mean():.0f}  total {dur.float().mean():.0f}")
xcc = d[:, 7] & 15
hw = d[:, 6]
cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
for x in range(8):
    sel = xcc == x
    if sel.sum() == 0: continue
    span = (t0[sel] + dur[sel]).max() - t0[sel].min()
    ncu = len(set(zip(cu[sel].tolist(), sh[sel].tolist(), se[sel].tolist())))
    print(f"  xcc {x}: blocks {int(sel.sum())}, CUs seen {ncu}, span {int(span)} ticks, resident block-ticks/span = {dur[sel].sum().item() / span.item():.1f} (per CU {dur[sel].sum().item() / span.item() / max(ncu,1):.2f})")
os.makedirs("gpurun_out", exist_ok=True)
torch.save(d.to(torch.int64), "gpurun_out/stamps.pt")
