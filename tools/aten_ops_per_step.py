"""Which torch operators (outside libvitk) launch kernels in one benchmark step:
python tools/aten_ops_per_step.py [--workload W]   (torch.profiler over 2 steps after warm-up; counts per step)"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="vit_h14_224_droppath_ls_ckpt", choices=list(bench.WORKLOADS))
args = ap.parse_args()
w = bench.WORKLOADS[args.workload]
device = torch.device("cuda", 0)
cfg, model = bench.make_model(w, device)
x = torch.randn(w["batch"], cfg.in_channels, *cfg.img_size, device=device, dtype=torch.float32 if w["autocast"] else cfg.dtype)


def step():
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=w["autocast"]):
        loss = model(x).dense_features.mean()
    loss.backward()
    model.zero_grad(set_to_none=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
STEPS = 2
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
rows = collections.Counter()
dev_us = collections.Counter()
for e in prof.key_averages():
    dt = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
    if e.key.startswith("aten::") and dt > 0:
        rows[e.key] += e.count
        dev_us[e.key] += dt
for k, c in sorted(rows.items(), key=lambda kv: -dev_us[kv[0]])[:25]:
    print(f"{k:40s} {c / STEPS:8.1f} calls/step {dev_us[k] / STEPS / 1e3:8.3f} ms/step")
