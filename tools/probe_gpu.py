"""One-off probe of torch semantics on the GPU box that the host code must mirror (run by hand)."""
import torch
import torch.nn.functional as F

print(torch.__version__, torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count)
x = torch.full((1, 8), 1e-3, device="cuda")
w = torch.ones(8, device="cuda")
for dt in (torch.float32, torch.bfloat16):
    xd = x.to(dt)
    y = F.rms_norm(xd, (8,), w.to(dt), None).float()[0, 0].item()
    ms = xd.float().square().mean().item()
    print("rms_norm default eps for", dt, "=", (xd[0, 0].float().item() / y) ** 2 - ms)
with torch.autocast("cuda", dtype=torch.bfloat16):
    xb = torch.randn(4, 8, device="cuda", dtype=torch.bfloat16)
    xf = xb.float()
    print("autocast rms_norm(bf16 in) ->", F.rms_norm(xb, (8,), w, None).dtype, " (fp32 in) ->", F.rms_norm(xf, (8,), w, None).dtype)
    print("autocast layer_norm(bf16 in) ->", F.layer_norm(xb, (8,), w, None).dtype, " (fp32 in) ->", F.layer_norm(xf, (8,), w, None).dtype)
    print("autocast cat(fp32 empty, bf16) ->", torch.cat([torch.empty(4, 0, 8, device="cuda"), xb[:, None].expand(4, 2, 8)], 1).dtype)
    print("autocast linear ->", F.linear(xf, torch.randn(8, 8, device="cuda")).dtype)
    print("autocast bf16*fp32 ->", (xb * w).dtype, " silu(bf16) ->", F.silu(xb).dtype)
