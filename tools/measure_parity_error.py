"""How far apart are two bf16 computations of the same model, and how far is each from fp32 arithmetic?

For BASELINE configs 3 and 4 at depth 2 (pure bf16 parameters) this prints, for the output, the input gradient and the
worst parameter gradient, the max-norm-relative error  max|a - b| / max|b|  of
    hip  vs oracle-bf16     (what tests/test_gpu_model.py bounds)
    hip  vs oracle-fp32     (the same bf16 weights and input, fp32 arithmetic)
    oracle-bf16 vs oracle-fp32
The tolerances of the two tests are justified by the last two columns: both bf16 computations sit at a similar distance
from the fp32 result, and their mutual distance is of the same size.   Usage: python tools/measure_parity_error.py"""
import os
import sys

print("importing torch", flush=True)
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vit_oracle as O  # noqa: E402
from vit_amd import ViTConfig  # noqa: E402

DEV = "cuda"


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def oracle_cfg(cfg, dtype=None):
    c = O.OracleConfig(**{k: v for k, v in cfg.__dict__.items()})
    if dtype is not None:
        c.dtype = dtype
    return c


def run(name, cfg, B, hw, seed_randperm):
    print(f"== {name}", flush=True)
    model = cfg.instantiate(device=DEV)
    sd = O.randomize_state(O.init_state(oracle_cfg(cfg), seed=3), seed=4)
    model.load_state_dict({k: v.to(DEV) for k, v in sd.items()}, strict=True)
    x = torch.randn(B, 3, *hw, generator=torch.Generator().manual_seed(0)).to(cfg.dtype)
    model.train()
    xg = x.to(DEV).clone().requires_grad_(True)
    torch.manual_seed(1234)
    out = model(xg).dense_features
    out.float().square().mean().backward()
    hip = {"out": out.detach(), "x_grad": xg.grad, "grads": {k: p.grad for k, p in model.named_parameters() if p.grad is not None}}
    refs = {}
    print("   hip pass done", flush=True)
    for tag, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        print(f"   oracle {tag} ...", flush=True)
        sdd = {k: (v.to(dt) if v.is_floating_point() else v).to(DEV) for k, v in sd.items()}
        from torch.nn.attention import SDPBackend, sdpa_kernel

        torch.backends.cudnn.enabled = False   # no MIOpen search for the oracle's convolution
        with sdpa_kernel(SDPBackend.MATH):   # plain ATen math, as in tests/test_gpu_model.py
            refs[tag] = O.run_fwd_bwd(oracle_cfg(cfg, dt), sdd, x.to(DEV, dt), training=True, autocast=None, loss="square_mean",
                                      seed=1234)
    for what in ("out", "x_grad"):
        print(f"{what:8s} hip~bf16 {rel(hip[what], refs['bf16'][what]):.2e}   hip~fp32 {rel(hip[what], refs['fp32'][what]):.2e}   "
              f"bf16~fp32 {rel(refs['bf16'][what], refs['fp32'][what]):.2e}")
    worst = [0.0, 0.0, 0.0, ""]
    for k, g in refs["fp32"]["grads"].items():
        e = (rel(hip["grads"][k], refs["bf16"]["grads"][k]), rel(hip["grads"][k], g), rel(refs["bf16"]["grads"][k], g))
        if e[0] > worst[0]:
            worst = [*e, k]
    print(f"worst parameter gradient ({worst[3]}): hip~bf16 {worst[0]:.2e}   hip~fp32 {worst[1]:.2e}   bf16~fp32 {worst[2]:.2e}")


def main():
    run("config 3: ViT-L/16 384 px, RoPE + SReLU, depth 2, bf16",
        ViTConfig(in_channels=3, patch_size=(16, 16), img_size=(384, 384), depth=2, hidden_size=1024, ffn_hidden_size=4096,
                  num_attention_heads=16, hidden_dropout=0.0, attention_dropout=0.0, activation="srelu", pos_enc="rope",
                  dtype=torch.bfloat16), 2, (384, 384), False)
    run("config 4: ViT-H/14 224 px, drop-path + layer scale + checkpointing, depth 2, bf16",
        ViTConfig(in_channels=3, patch_size=(14, 14), img_size=(224, 224), depth=2, hidden_size=1280, ffn_hidden_size=5120,
                  num_attention_heads=16, hidden_dropout=0.0, attention_dropout=0.0, activation="srelu", pos_enc="rope",
                  drop_path_rate=0.25, layer_scale=1.0, activation_checkpointing=True, dtype=torch.bfloat16), 4, (224, 224), True)


if __name__ == "__main__":
    main()
