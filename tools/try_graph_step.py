"""Feasibility check: the benchmark's training step captured into one HIP graph (torch.cuda.CUDAGraph) and replayed,
against the eager step.  python tools/try_graph_step.py [--workload W] [--steps K]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="vit_b14_224_swiglu_reg16_fourier", choices=list(bench.WORKLOADS))
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
w = bench.WORKLOADS[args.workload]
B = w["batch"]
cfg, model = bench.make_model(w, device)
torch.manual_seed(1337)
x = torch.randn(B, cfg.in_channels, *cfg.img_size, device=device, dtype=torch.float32 if w["autocast"] else cfg.dtype)


def step():
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=w["autocast"]):
        loss = model(x).dense_features.mean()
    loss.backward()


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(3):
    step()
    model.zero_grad(set_to_none=True)
eager = timed(lambda: (step(), model.zero_grad(set_to_none=True)), args.steps)
step()
want = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
model.zero_grad(set_to_none=True)

# capture: gradients land in static buffers (kept allocated between replays, zeroed inside the graph by set_to_none=False)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
        model.zero_grad(set_to_none=True)
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
model.zero_grad(set_to_none=True)
with torch.cuda.graph(graph):
    step()
graph.replay()
torch.cuda.synchronize()
worst = 0.0
for n, p in model.named_parameters():
    if p.grad is not None:
        worst = max(worst, (p.grad.float() - want[n].float()).abs().max().item() / max(want[n].float().abs().max().item(), 1e-12))
graphed = timed(graph.replay, args.steps)
print(f"{args.workload}: eager {eager:.3f} ms/step, graph replay {graphed:.3f} ms/step, worst |grad diff| / max|grad| = {worst:.2e}")
