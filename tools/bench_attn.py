"""Attention micro-benchmark (forward, backward) on the encoder's shapes through the C ABI.
python tools/bench_attn.py [B S H Dh]      (DROP=0.1 in the environment: with dropout on the probabilities)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import kernels as K

B, S, H, Dh = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (256, 272, 12, 64)))
qkv = torch.randn(B * S, 3 * H * Dh, device="cuda").to(torch.bfloat16)
dout = torch.randn(B * S, H * Dh, device="cuda").to(torch.bfloat16)


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


DROP = float(os.environ.get("DROP", "0"))
o, lse = K.attn_fwd(qkv, B, S, H, Dh, Dh ** -0.5, DROP, 1234)
f = timeit(lambda: K.attn_fwd(qkv, B, S, H, Dh, Dh ** -0.5, DROP, 1234))
b = timeit(lambda: K.attn_bwd(qkv, o, lse, dout, B, S, H, Dh, Dh ** -0.5, DROP, 1234))
fl = 4.0 * B * H * S * S * Dh
print(f"B={B} S={S} H={H} Dh={Dh} drop={DROP}: fwd {f:.3f} ms ({fl / f / 1e9:.0f} TF/s)  bwd {b:.3f} ms ({2.5 * fl / b / 1e9:.0f} TF/s algorithmic)")
