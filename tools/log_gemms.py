"""The GEMM calls of one training step of a bench.py workload: count, tag, M, N, K, epilogue code, operand layouts, whether
rows are gathered, output dtype.  python tools/log_gemms.py <workload>"""
import sys, os, collections, torch
sys.path.insert(0, os.getcwd())
import bench
from vit_amd import kernels as K
w = bench.WORKLOADS[sys.argv[1]]
cfg, model = bench.make_model(w, torch.device("cuda", 0))
x = torch.randn(w["batch"], cfg.in_channels, *cfg.img_size, device="cuda", dtype=torch.float32 if w["autocast"] else cfg.dtype)
seen = collections.Counter()
orig = K.gemm
def logged(a, b, **kw):
    seen[(kw.get("tag", ""), kw["M"], kw["N"], kw["K"], kw.get("epilogue", 0), kw["trans_a"], kw["trans_b"], kw.get("row_idx") is not None, str(kw["out"].dtype))] += 1
    return orig(a, b, **kw)
K.gemm = logged
import vit_amd.ops as ops
with torch.autocast("cuda", dtype=torch.bfloat16, enabled=w["autocast"]):
    loss = model(x).dense_features.mean()
loss.backward()
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(v, k)
