"""Which hipBLASLt (Tensile) kernels torch.matmul picks for the encoder step's six product shapes.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/blaslt -o blaslt -- python3 tools/hipblaslt_names.py run
    python3 tools/hipblaslt_names.py report gpurun_out/blaslt > profiles/r03_hipblaslt_kernels.txt

`run` calls torch.matmul REPS times per shape, in a fixed order, with a marker kernel (a fill of a one-element tensor
with the shape's index) in front of each group; `report` cuts the trace at the markers and prints, per shape, the kernel
names and their average durations.  The Tensile names encode the macro tile (MT), the MFMA shape (MI), the wave layout
(WG), the contraction depth per iteration (DepthU via the K in MT AxBxK), prefetch depths (PGR / PLR), direct-to-LDS (DTL)
and the store path, which is what this build's own kernels are compared with."""
import collections
import csv
import os
import sys

REPS = 6
B, S, D, F = 256, 272, 768, 3072
M = B * S
SHAPES = [  # name, layout, M, N, K
    ("qkv fwd", "nt", M, 3 * D, D), ("fc1 fwd", "nt", M, 2 * F, D), ("out fwd", "nt", M, D, D), ("fc2 fwd", "nt", M, D, F),
    ("fc2 dgrad", "nn", M, F, D), ("fc1 dgrad", "nn", M, D, 2 * F), ("qkv dgrad", "nn", M, D, 3 * D), ("out dgrad", "nn", M, D, D),
    ("fc1 wgrad", "tn", 2 * F, D, M), ("fc2 wgrad", "tn", D, F, M), ("qkv wgrad", "tn", 3 * D, D, M), ("out wgrad", "tn", D, D, M),
]


def run():
    import torch

    dev = "cuda"
    rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).to(torch.bfloat16)  # noqa: E731
    for name, layout, m, n, k in SHAPES:
        if layout == "nt":
            a, b = rnd(m, k), rnd(n, k)
            fn = lambda: torch.matmul(a, b.t())  # noqa: E731
        elif layout == "nn":
            a, b = rnd(m, k), rnd(k, n)
            fn = lambda: torch.matmul(a, b)  # noqa: E731
        else:
            a, b = rnd(k, m), rnd(k, n)
            fn = lambda: torch.matmul(a.t(), b)  # noqa: E731
        fn()
        torch.cuda.synchronize()
        marker = torch.empty(1, device=dev, dtype=torch.int64)
        marker.fill_(7)          # marker kernel: one fill per group, right in front of its REPS products
        for _ in range(REPS):
            fn()
        torch.cuda.synchronize()


def report(d):
    path = None
    for root, _, files in os.walk(d):
        for fn in files:
            if fn.endswith("kernel_trace.csv"):
                path = os.path.join(root, fn)
    rows = sorted(csv.DictReader(open(path, newline="")), key=lambda r: int(r["Start_Timestamp"]))
    groups, cur = [], None
    for r in rows:
        n = r["Kernel_Name"]
        if "fill" in n.lower() and int(r["Grid_Size_X"]) <= 512:   # a marker
            cur = collections.defaultdict(list)
            groups.append(cur)
        elif cur is not None and ("Cijk" in n or "gemm" in n.lower()):
            cur[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    groups = [g for g in groups if g]
    print(f"hipBLASLt kernels behind torch.matmul, bf16, random operands in [-1, 1); {REPS} calls per shape (B = {B}, S = {S})")
    for (name, layout, m, n, k), g in zip(SHAPES, groups[-len(SHAPES):]):
        fl = 2.0 * m * n * k
        for kn, ts in g.items():
            us = sum(ts) / len(ts) / 1e3
            print(f"{name:10s} {layout} M={m} N={n} K={k}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s  x{len(ts)}  {kn}")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
