"""Rehearsal of the multi-rank training step on one GPU: the benchmark's model and step under BucketedGradAllReduce on a
ONE-rank RCCL group with always_reduce=True, so every bucket goes through a real RCCL all-reduce on its side stream while
the backward GEMMs draw their tiles dynamically -- everything of `bench.py --gpus N` except a second GPU.

Checks that the gradients equal those of the plain step (an average over one rank changes nothing) and prints both
step times.  python tools/rehearse_ddp_one_rank.py [--workload W] [--batch B] [--steps K]"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="vit_b14_224_swiglu_reg16_fourier", choices=list(bench.WORKLOADS))
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--bucket-mb", type=int, default=64)
    args = ap.parse_args()

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)

    from vit_amd.ddp import BucketedGradAllReduce

    w = bench.WORKLOADS[args.workload]
    B = args.batch or w["batch"]
    cfg, model = bench.make_model(w, device)
    torch.manual_seed(1337)
    in_dtype = torch.float32 if w["autocast"] else cfg.dtype
    x = torch.randn(B, cfg.in_channels, *cfg.img_size, device=device, dtype=in_dtype)

    def step(reducer, keep_grads=False):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=w["autocast"]):
            loss = model(x).dense_features.mean()
        loss.backward()
        if reducer is not None:
            reducer.finish()
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None} if keep_grads else None
        model.zero_grad(set_to_none=True)
        return grads

    def timed(reducer):
        for _ in range(2):
            step(reducer)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(reducer)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3

    state = torch.cuda.get_rng_state(0)
    want = step(None, keep_grads=True)
    plain_ms = timed(None)

    reducer = BucketedGradAllReduce(model.parameters(), bucket_bytes=args.bucket_mb << 20, always_reduce=True)
    assert reducer.dynamic_tiles, "the RCCL path must switch the GEMMs to dynamic tile shares"
    torch.cuda.set_rng_state(state, 0)
    got = step(reducer, keep_grads=True)
    reduced_ms = timed(reducer)
    reducer.remove()

    assert set(got) == set(want), set(got) ^ set(want)
    worst = 0.0
    for n, g in want.items():
        scale = max(g.float().abs().max().item(), 1e-12)
        worst = max(worst, (got[n].float() - g.float()).abs().max().item() / scale)
    print(f"buckets={len(reducer.buckets)} plain={plain_ms:.2f} ms/step reducer+RCCL(1 rank)={reduced_ms:.2f} ms/step "
          f"worst |grad diff| / max|grad| = {worst:.3e}")
    # the dynamic tile order changes nothing in a tile's arithmetic, and averaging over one rank is the identity
    assert worst == 0.0, worst
    dist.barrier()
    dist.destroy_process_group()
    print("rehearsal ok")


if __name__ == "__main__":
    main()
