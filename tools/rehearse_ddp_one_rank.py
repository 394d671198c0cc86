"""Rehearsal of the multi-rank training step on one GPU: the benchmark's model and step under BucketedGradAllReduce on a
ONE-rank RCCL group with always_reduce=True, so every bucket goes through a real RCCL all-reduce on its side stream while
the backward GEMMs draw their tiles dynamically -- everything of `bench.py --gpus N` except a second GPU.

Checks that the gradients equal those of the plain step (an average over one rank changes nothing) and prints both
step times.  With --json it also writes, per bucket, the point of the backward pass at which the collective was
queued (an event on the compute stream) and the collective's stand-alone duration on this one-rank group, which shows
how much of the backward pass is left to hide each bucket behind.
python tools/rehearse_ddp_one_rank.py [--workload W] [--batch B] [--steps K] [--json OUT]"""
import argparse
import json
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
    ap.add_argument("--json", default="")
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
    record = None
    if args.json:
        record = traced_step(args, w, B, model, x, reducer, plain_ms, reduced_ms)
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
    if record is not None:
        record["worst_grad_diff_over_max"] = worst
        os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
        with open(args.json, "w") as f:
            json.dump(record, f, indent=1)
    dist.barrier()
    dist.destroy_process_group()
    print("rehearsal ok")


def traced_step(args, w, B, model, x, reducer, plain_ms, reduced_ms) -> dict:
    """One more step with the reducer's timeline on, then every bucket's all-reduce alone."""
    torch.cuda.synchronize()
    reducer.timeline = []
    e_fwd, e_bwd0, e_bwd1, e_done = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    e_fwd.record()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=w["autocast"]):
        loss = model(x).dense_features.mean()
    e_bwd0.record()
    h0 = time.perf_counter()
    loss.backward()
    e_bwd1.record()
    reducer.finish()
    e_done.record()
    torch.cuda.synchronize()
    timeline, reducer.timeline = reducer.timeline, None
    model.zero_grad(set_to_none=True)
    bwd_ms = e_bwd0.elapsed_time(e_bwd1)
    buckets = []
    for idx, ev, host_t in timeline:
        b = reducer.buckets[idx]
        # the collective by itself: queue it on an idle GPU and wait for it
        alone = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dist.all_reduce(b.flat, op=dist.ReduceOp.AVG, async_op=True).wait()
            torch.cuda.synchronize()
            alone.append((time.perf_counter() - t0) * 1e3)
        at = e_bwd0.elapsed_time(ev)
        buckets.append({"bucket": idx, "mbytes": round(b.flat.numel() * b.flat.element_size() / 2 ** 20, 2),
                        "params": len(b.params), "queued_at_ms_of_backward": round(at, 3),
                        "queued_at_host_ms": round((host_t - h0) * 1e3, 3),
                        "backward_left_ms": round(max(bwd_ms - at, 0.0), 3),
                        "all_reduce_alone_ms_one_rank": round(min(alone), 3)})
    return {"tool": "tools/rehearse_ddp_one_rank.py", "workload": args.workload, "per_gpu_batch": B,
            "group": "RCCL, world_size 1, always_reduce (the launches, side stream and dynamic tile shares of the N-rank step; "
                     "no second GPU, so no xGMI traffic: the stand-alone durations are launch + local copy cost only)",
            "bucket_mb": args.bucket_mb, "plain_ms_per_step": round(plain_ms, 3), "reducer_ms_per_step": round(reduced_ms, 3),
            "forward_ms": round(e_fwd.elapsed_time(e_bwd0), 3), "backward_ms": round(bwd_ms, 3),
            "finish_ms_after_backward": round(e_bwd1.elapsed_time(e_done), 3), "buckets": buckets}


if __name__ == "__main__":
    main()
