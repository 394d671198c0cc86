#!/usr/bin/env python3
"""Headline benchmark: train images/sec of ViT-B/14 224 px (RMSNorm + SwiGLU, 16 register tokens, Fourier
pos-enc) bf16, forward + backward on MI355X, one process per GPU.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A step = forward + loss (dense_features.mean(), as the reference's benchmark/benchmark.py:200-217) + backward +
bucketed gradient all-reduce (N > 1) + zero_grad.  The reference has no optimiser, so none is timed.  The only
mode in which the reference runs this configuration is fp32 master parameters under torch.autocast(bf16)
(its Fourier encoder cannot multiply an fp32 grid by bf16 parameters); that is the mode measured here.

Prints ONE JSON line: metric/value (whole-job images/s), roofline of the dominant kernel (fc1 forward GEMM,
timed with HIP events inside the timed region), and the CPU baseline (the oracle on the host cores, bounded
sample)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # BASELINE config 2 (the configuration the metric is quoted on)
    "vit_b14_224_swiglu_reg16_fourier": dict(
        cfg=dict(in_channels=3, patch_size=(14, 14), img_size=(224, 224), depth=12, hidden_size=768, ffn_hidden_size=3072,
                 num_attention_heads=12, hidden_dropout=0.0, attention_dropout=0.0, activation="swiglu", pos_enc="fourier",
                 num_register_tokens=16, dtype="float32"),
        autocast=True, batch=256),
    # BASELINE config 1: the reference's own CPU-runnable case, forward only (plumbing; fp32 takes the portable kernels)
    "vit_s16_224_fp32_fwd": dict(
        cfg=dict(in_channels=3, patch_size=(16, 16), img_size=(224, 224), depth=12, hidden_size=384, ffn_hidden_size=1536,
                 num_attention_heads=6, hidden_dropout=0.0, attention_dropout=0.0, activation="srelu", pos_enc="rope",
                 dtype="float32"),   # activation / positions: the reference's defaults (vit/vit.py:83,87)
        autocast=False, batch=8, forward_only=True),
    # the README's training configuration (README.md:28-48) on the headline shape: hidden / attention dropout 0.1, layer
    # scale 1e-5, normalised patch embedding.  Its drop_path_rate is left out: under autocast the reference's own
    # index_add of the bf16 branch into the fp32 stream raises, so that combination does not run there either.
    "vit_b14_224_readme_dropout": dict(
        cfg=dict(in_channels=3, patch_size=(14, 14), img_size=(224, 224), depth=12, hidden_size=768, ffn_hidden_size=3072,
                 num_attention_heads=12, hidden_dropout=0.1, attention_dropout=0.1, activation="swiglu", pos_enc="fourier",
                 num_register_tokens=16, patch_embed_normalization=True, layer_scale=1e-5, dtype="float32"),
        autocast=True, batch=256),
    # BASELINE config 3 / 4 / 5 shapes (secondary; pure bf16 parameters)
    "vit_l16_384_srelu_rope": dict(
        cfg=dict(in_channels=3, patch_size=(16, 16), img_size=(384, 384), depth=24, hidden_size=1024, ffn_hidden_size=4096,
                 num_attention_heads=16, hidden_dropout=0.0, attention_dropout=0.0, activation="srelu", pos_enc="rope",
                 dtype="bfloat16"),
        autocast=False, batch=64),
    "vit_h14_224_droppath_ls_ckpt": dict(
        cfg=dict(in_channels=3, patch_size=(14, 14), img_size=(224, 224), depth=32, hidden_size=1280, ffn_hidden_size=5120,
                 num_attention_heads=16, hidden_dropout=0.0, attention_dropout=0.0, activation="srelu", pos_enc="rope",
                 drop_path_rate=0.1, layer_scale=1e-5, activation_checkpointing=True, dtype="bfloat16"),
        autocast=False, batch=64),
    "vit_g14_1024_swiglu_rope": dict(
        cfg=dict(in_channels=3, patch_size=(14, 14), img_size=(1024, 1024), depth=40, hidden_size=1408, ffn_hidden_size=6144,
                 num_attention_heads=16, hidden_dropout=0.0, attention_dropout=0.0, activation="swiglu", pos_enc="rope",
                 dtype="bfloat16"),
        autocast=False, batch=8),
}


def flops_per_image(c: dict) -> float:
    """Forward FLOPs (2 x MAC) per image: 2*[N*C*p^2*D + depth*(4*S*D^2 + S*D*F1 + S*F*D + 2*S^2*D)]."""
    ph, pw = c["patch_size"]
    N = (c["img_size"][0] // ph) * (c["img_size"][1] // pw)
    S = N + c.get("num_register_tokens", 0) + c.get("num_cls_tokens", 0)
    D, F = c["hidden_size"], c["ffn_hidden_size"]
    F1 = 2 * F if c["activation"].endswith("glu") else F
    return 2.0 * (N * c["in_channels"] * ph * pw * D + c["depth"] * (4 * S * D * D + S * D * F1 + S * F * D + 2 * S * S * D))


def make_model(w: dict, device):
    from vit_amd import ViTConfig

    kw = dict(w["cfg"])
    kw["dtype"] = getattr(torch, kw["dtype"])
    cfg = ViTConfig(**kw)
    torch.manual_seed(0)  # identical replicas on every rank
    model = cfg.instantiate(device=device)
    with torch.no_grad():  # the reference zero-initialises out_proj / fc2; give them weights so no branch multiplies by zero
        for blk in model.blocks:
            blk.self_attention.out_proj.weight.normal_(std=0.02)
            blk.mlp.fc2.weight.normal_(std=0.02)
    return cfg, model.train()


def fc1_kernel_name(c: dict, autocast: bool) -> str:
    """The kernel behind the "fc1" tag: the 256x256-tile LDS-DMA GEMM with the activation fused in its epilogue; fp32
    parameters without autocast run the portable fp32 kernel."""
    if c["dtype"] == "float32" and not autocast:
        return "gemm_simple_k<float> (fc1 forward, portable fp32 kernel)"
    epi = "GLU" if c["activation"].endswith("glu") else "ACT"
    return f"gemm_mfma256d_k<NT,{epi},bf16> (fc1 forward)"


PMC_SUMMARY = os.path.join("profiles", "r03_pmc_summary.json")
FC1_KERNEL_ROW = "gemm_mfma256d_k<false, false, 4, bf16>"   # its name in tools/pmc_summary.py's table


def counter_summary(workload: str, batch: int):
    """Figures that only hardware counters give, from the committed summary of separate rocprofv3 --pmc passes over
    this same command (tools/pmc_collect.sh, tools/pmc_summary.py): memory-side bytes per fc1 launch (FETCH_SIZE with
    the gfx950 correction + WRITE_SIZE) and the matrix-unit busy fractions.  They are not measured by this run -- a
    process cannot read the counters of its own kernels -- so the build they were taken on is reported beside them.
    The third value gives the normalisation kernels (HBM-bound) their memory-side rate over the 8 TB/s peak.
    (None, None, None) for a workload / batch the passes were not taken on."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), PMC_SUMMARY)
    if workload != "vit_b14_224_swiglu_reg16_fourier" or batch != 256 or not os.path.exists(path):
        return None, None, None
    with open(path) as f:
        summ = json.load(f)
    row = next((k for k in summ["kernels"] if k["kernel"] == FC1_KERNEL_ROW), None)
    traffic = None if row is None else row["read_bytes_per_call"] + row["write_bytes_per_call"]
    util = {"whole_step_mfma_busy_frac": summ["whole_run"]["mfma_busy_frac"],
            "fc1_mfma_busy_frac": None if row is None else row["mfma_busy_frac"],
            "definition": summ["formulas"]["mfma_busy_frac"], "source": PMC_SUMMARY,
            "taken_on_build": summ["build"]}
    norm = {"peak_gbps": PEAK_HBM_GBS, "source": PMC_SUMMARY,
            "definition": "(2 * FETCH_SIZE + WRITE_SIZE) KiB of the kernel's launches / their summed duration"}
    for key, prefix in (("norm_fwd", "norm_fwd_k<float, bool"), ("norm_bwd", "norm_bwd_k<bool")):
        r = next((k for k in summ["kernels"] if k["kernel"].startswith(prefix)), None)   # the bf16-output instances
        if r is not None and r.get("hbm_side_gbps"):
            norm[key] = {"kernel": r["kernel"], "ms_per_step": r["ms_per_step"], "achieved_gbps": r["hbm_side_gbps"],
                         "frac": round(r["hbm_side_gbps"] / PEAK_HBM_GBS, 4)}
    return traffic, util, norm


def cpu_baseline(w: dict, seconds: float = 20.0) -> dict:
    """The oracle (torch-op restatement of the reference's eager path) on the host cores: same configuration and
    mode, batch 8 per step, as many steps as fit in ~`seconds`."""
    from oracle import vit_oracle as O

    kw = dict(w["cfg"])
    kw["dtype"] = getattr(torch, kw["dtype"])
    cfg = O.OracleConfig(**kw)
    sd = O.init_state(cfg, seed=0)
    B = 8
    x = torch.randn(B, cfg.in_channels, *cfg.img_size).to(cfg.dtype)
    ac = torch.bfloat16 if w["autocast"] else None
    if w.get("forward_only"):
        def run():
            with torch.no_grad():
                O.vit_forward(cfg, sd, x, training=False)
    else:
        run = lambda: O.run_fwd_bwd(cfg, sd, x, training=True, autocast=ac, loss="mean")  # noqa: E731
    run()  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        run()
        n += 1
        el = time.perf_counter() - t0
        if el > seconds or n >= 16:
            break
    return {"value": round(n * B / el, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} {'forward' if w.get('forward_only') else 'fwd+bwd'} steps of batch {B}, same model/mode, torch {torch.__version__} eager CPU ops "
                      f"({os.cpu_count()} logical CPUs visible)"}


def cpu_baseline_config1() -> dict:
    """BASELINE config 1 exactly, on the host cores: ViT-S/16 224 px fp32 forward, batch 8, eval + inference_mode,
    3 warm-up + 20 timed iterations of the oracle (the reference's benchmark/benchmark.py:51-84 loop)."""
    from oracle import vit_oracle as O

    w = WORKLOADS["vit_s16_224_fp32_fwd"]
    kw = dict(w["cfg"])
    kw["dtype"] = getattr(torch, kw["dtype"])
    cfg = O.OracleConfig(**kw)
    sd = O.init_state(cfg, seed=0)
    x = torch.randn(8, cfg.in_channels, *cfg.img_size)
    with torch.inference_mode():
        for _ in range(3):
            O.vit_forward(cfg, sd, x, training=False)
        t0 = time.perf_counter()
        for _ in range(20):
            O.vit_forward(cfg, sd, x, training=False)
        el = time.perf_counter() - t0
    return {"workload": "vit_s16_224_fp32_fwd", "value": round(20 * 8 / el, 2), "unit": "images/s", "ms_per_batch": round(el / 20 * 1e3, 2),
            "cores": torch.get_num_threads(), "cpu_count": os.cpu_count(), "kind": "port",
            "sample": "20 forward passes of batch 8 after 3 warm-up passes, eval + inference_mode, fp32"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--workload", default="vit_b14_224_swiglu_reg16_fourier", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bucket-mb", type=int, default=64)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: run the N ranks as children under torch.distributed.run (this
        # process has not touched the GPU) and leave with their exit code
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; start it as `python bench.py --gpus N` or under "
                 f"torch.distributed.run with --nproc-per-node equal to --gpus")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from vit_amd import kernels as K
    from vit_amd.ddp import BucketedGradAllReduce

    w = WORKLOADS[args.workload]
    B = args.batch or w["batch"]
    cfg, model = make_model(w, device)
    reducer = BucketedGradAllReduce(model.parameters(), bucket_bytes=args.bucket_mb << 20) if world > 1 else None
    torch.manual_seed(1337 + rank)  # the reference's benchmark seed, offset per rank
    in_dtype = torch.float32 if w["autocast"] else cfg.dtype
    x = torch.randn(B, cfg.in_channels, *cfg.img_size, device=device, dtype=in_dtype)

    fwd_only = bool(w.get("forward_only"))
    if fwd_only:
        model.eval()

    def step():
        if fwd_only:
            with torch.no_grad():
                model(x)
            return
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=w["autocast"]):
            out = model(x)
            loss = out.dense_features.mean()
        loss.backward()
        if reducer is not None:
            reducer.finish()
        model.zero_grad(set_to_none=True)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    prof = K.profile_gemms(["fc1"])
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    K.profile_gemms(None)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    c = w["cfg"]
    ph, pw = c["patch_size"]
    S = (c["img_size"][0] // ph) * (c["img_size"][1] // pw) + c.get("num_register_tokens", 0) + c.get("num_cls_tokens", 0)
    F1 = (2 if c["activation"].endswith("glu") else 1) * c["ffn_hidden_size"]
    keep = B
    if c.get("drop_path_rate", 0.0) > 0:
        keep = max(1, min(B, int(B * (1 - c["drop_path_rate"]))))
    # dominant kernel: the fc1 forward GEMM  [keep*S, D] x [F1, D]^T, 2*M*N*K FLOPs per launch
    pairs = prof["pairs"]["fc1"]
    k_ms = sum(a.elapsed_time(b) for a, b in pairs) / max(len(pairs), 1)
    k_flop = 2.0 * keep * S * F1 * c["hidden_size"]
    achieved = k_flop / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0

    if rank == 0:
        imgs = B * world * args.steps / elapsed
        fpi = flops_per_image(c)
        traffic, mfma_util, norm_hbm = counter_summary(args.workload, B)
        rec = {
            "metric": "train images/sec ViT-B/14 224px bf16" if args.workload == "vit_b14_224_swiglu_reg16_fourier" else
                      f"{'forward' if fwd_only else 'train'} images/sec {args.workload}",
            "value": round(imgs, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if (w["autocast"] or c["dtype"] == "bfloat16") else "f32", "data": "synthetic",
            "config": {"workload": args.workload, "per_gpu_batch": B, "global_batch": B * world, "seq_len": S,
                       "mode": "fp32 master params + torch.autocast(bf16)" if w["autocast"] else f"pure {c['dtype']} params",
                       "parallelism": f"dp{world}",
                       "step": "forward (eval, no_grad)" if fwd_only else "fwd + mean loss + bwd + bucketed grad all-reduce + zero_grad",
                       "fwd_bwd_gflop_per_image": round((1 if fwd_only else 3) * fpi / 1e9, 1)},
            "step_mfma_frac": round(imgs / world * (1 if fwd_only else 3) * fpi / (PEAK_BF16_TFLOPS * 1e12), 4),
            "roofline": {"bound": "mfma", "kernel": fc1_kernel_name(c, w["autocast"]), "achieved": round(achieved, 1),
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                         "traffic": traffic, "traffic_source": PMC_SUMMARY if traffic is not None else None,
                         "launch_ms": round(k_ms, 4), "launches_timed": len(pairs), "flop_per_launch": k_flop},
            "mfma_util": mfma_util,
            "norm_hbm": norm_hbm,
        }
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(w)
            if args.workload == "vit_b14_224_swiglu_reg16_fourier":   # the headline run also times the reference's own CPU case
                rec["cpu_baseline_config1"] = cpu_baseline_config1()
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
