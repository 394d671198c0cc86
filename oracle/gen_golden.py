"""TEST INFRASTRUCTURE, container-only: write golden vectors from the LIVE reference.

Run in the build container:  python oracle/gen_golden.py
It imports /root/reference through oracle/refshim.py, runs the reference's own modules on seeded
inputs and stores inputs + expected outputs (data only) as tests/golden/*.pt.  It refuses to run
when /root/reference is absent.  The reference has no golden vectors of its own (its tests are
self-consistency checks), so these files are what pins the oracle and the HIP path.

Each fixture: {"cfg": plain dict, "mode": str, "state": {name: tensor}, "x": tensor,
               "eval_out": tensor, "train_out": tensor, "x_grad": tensor, "grads": {name: tensor},
               "train_seed": int}
Loss for the backward pass: dense_features.float().square().mean()  (tests/test_vit.py:841-846).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refshim  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

BASE = dict(in_channels=3, patch_size=(8, 8), img_size=(32, 32), depth=2, hidden_size=64, ffn_hidden_size=128,
            num_attention_heads=2, hidden_dropout=0.0, attention_dropout=0.0)

# name -> (config overrides, batch, input HxW, mode)   mode: fp32 | autocast_bf16 | bf16
CASES = {
    "g01_fourier_swiglu_reg_fp32": (dict(activation="swiglu", pos_enc="fourier", num_register_tokens=4), 2, (32, 32), "fp32"),
    "g01_fourier_swiglu_reg_autocast": (dict(activation="swiglu", pos_enc="fourier", num_register_tokens=4), 2, (32, 32), "autocast_bf16"),
    "g02_rope_srelu_fp32": (dict(activation="srelu", pos_enc="rope", img_size=(32, 48)), 2, (32, 48), "fp32"),
    "g02_rope_srelu_bf16": (dict(activation="srelu", pos_enc="rope", img_size=(32, 48)), 2, (32, 48), "bf16"),
    "g03_layernorm_gelu_learnable_cls_fp32": (dict(activation="gelu", pos_enc="learnable", norm_type="layernorm",
                                                   num_cls_tokens=1, num_register_tokens=2), 2, (32, 32), "fp32"),
    "g04_droppath_layerscale_fp32": (dict(activation="srelu", pos_enc="rope", layer_scale=0.1, drop_path_rate=0.25), 8, (32, 32), "fp32"),
    "g05_rope_srelu_ckpt_fp32": (dict(activation="srelu", pos_enc="rope", img_size=(32, 48), activation_checkpointing=True), 2, (32, 48), "fp32"),
    "g06_dh80_fp32": (dict(activation="srelu", pos_enc="rope", hidden_size=160, ffn_hidden_size=96, depth=1, img_size=(40, 24),
                           num_register_tokens=1), 2, (40, 24), "fp32"),
    "g06_dh88_bf16": (dict(activation="swiglu", pos_enc="rope", hidden_size=176, ffn_hidden_size=96, depth=1, img_size=(40, 24),
                           num_register_tokens=1), 2, (40, 24), "bf16"),
    "g07_glu_limit_bias_fp32": (dict(activation="swiglu", pos_enc="rope", glu_limit=0.05, glu_extra_bias=1.0), 2, (32, 32), "fp32"),
    "g07_qknorm_fp32": (dict(activation="srelu", pos_enc="rope", qk_normalization=True, num_register_tokens=2), 2, (32, 32), "fp32"),
    "g08_patchnorm_fp32": (dict(activation="openswiglu", pos_enc="fourier", patch_embed_normalization=True,
                                attention_bias=False, mlp_bias=False), 2, (32, 32), "fp32"),
    "g09_relu_none_layernorm_fp32": (dict(activation="relu", pos_enc="none", norm_type="layernorm", num_cls_tokens=2), 2, (32, 32), "fp32"),
    "g10_geglu_resize_fp32": (dict(activation="geglu", pos_enc="fourier", num_register_tokens=3), 2, (48, 40), "fp32"),
    # masked-token forward (vit/vit.py:545, :522-527): a fixed non-ragged mask is part of the fixture ("mask" key)
    "g11_masked_rope_fp32": (dict(activation="srelu", pos_enc="rope", num_register_tokens=2, img_size=(32, 48)), 3, (32, 48), "fp32"),
    "g11_masked_rope_autocast": (dict(activation="swiglu", pos_enc="rope", num_register_tokens=2, img_size=(32, 48)), 3, (32, 48),
                                 "autocast_bf16"),
    "g11_masked_fourier_droppath_fp32": (dict(activation="swiglu", pos_enc="fourier", num_register_tokens=1, drop_path_rate=0.25,
                                              layer_scale=0.1), 8, (32, 32), "fp32"),
    "g11_masked_rope_droppath_fp32": (dict(activation="srelu", pos_enc="rope", drop_path_rate=0.25), 8, (32, 32), "fp32"),
}
# volumetric input (PatchEmbed3d, vit/patch_embed.py:62-109); depth 18 leaves two slices no patch covers
CASES["g14_patch3d_fourier_fp32"] = (dict(activation="swiglu", pos_enc="fourier", patch_size=(4, 8, 8), img_size=(16, 32, 32),
                                          num_register_tokens=2), 2, (18, 32, 32), "fp32")
CASES["g14_patch3d_learnable_autocast"] = (dict(activation="srelu", pos_enc="learnable", patch_size=(4, 8, 8), img_size=(16, 32, 32),
                                                num_cls_tokens=1), 2, (16, 32, 32), "autocast_bf16")
# learned 2-D position table under autocast: the fp32 table is added to the bf16 tokens and the sum stays fp32
# (vit/pos_enc.py:79-90, vit/patch_embed.py:57)
CASES["g15_learnable_autocast"] = (dict(activation="swiglu", pos_enc="learnable", num_cls_tokens=1, num_register_tokens=2), 2,
                                   (32, 32), "autocast_bf16")
CASES["g15_learnable_resized_autocast"] = (dict(activation="srelu", pos_enc="learnable", num_register_tokens=1), 2, (48, 40),
                                           "autocast_bf16")
MASKED = {k for k in CASES if k.startswith("g11_masked")}


def plain_cfg(cfg) -> dict:
    d = {k: v for k, v in cfg.__dict__.items() if k != "heads"}
    d["dtype"] = str(d["dtype"]).replace("torch.", "")
    d["patch_size"] = list(d["patch_size"])
    d["img_size"] = list(d["img_size"])
    return d


@torch.no_grad()
def perturb(model, gen):
    """Make parity non-vacuous: the reference zero-initialises out_proj/fc2 (vit/transformer.py:132)."""
    for name, p in model.named_parameters():
        if p.numel() == 0:
            continue
        noise = torch.randn(p.shape, generator=gen).to(p.dtype)
        if name.endswith("out_proj.weight") or name.endswith("fc2.weight"):
            p.copy_(0.02 * noise)
        elif name.endswith("bias"):
            p.add_(0.02 * noise)
        elif name.endswith("norm.weight") or name.endswith("gamma"):
            p.mul_(1 + 0.1 * noise)


def make(ref, name, overrides, B, hw, mode, seed):
    kw = dict(BASE)
    kw.update(overrides)
    kw["dtype"] = torch.bfloat16 if mode == "bf16" else torch.float32
    cfg = ref.ViTConfig(**kw)
    torch.manual_seed(seed)
    model = cfg.instantiate()
    gen = torch.Generator().manual_seed(seed + 17)
    perturb(model, gen)
    x = torch.randn(B, cfg.in_channels, *hw, generator=gen).to(cfg.dtype)
    ac = torch.autocast("cpu", dtype=torch.bfloat16, enabled=(mode == "autocast_bf16"))
    mask = None
    if name in MASKED:   # keep 5/8 of the tokens of every sample (a non-ragged mask, as the reference's create_mask makes)
        L = (hw[0] // cfg.patch_size[0]) * (hw[1] // cfg.patch_size[1])
        keep = (5 * L) // 8
        mask = torch.zeros(B, L, dtype=torch.bool)
        for b in range(B):
            mask[b, torch.randperm(L, generator=gen)[:keep]] = True

    model.eval()
    with torch.no_grad(), ac:
        eval_out = model(x, mask).dense_features.clone()
        eval_out_nonorm = model(x, mask, output_norm=False).dense_features.clone()

    model.train()
    xg = x.clone().requires_grad_(True)
    train_seed = seed + 101
    torch.manual_seed(train_seed)
    with ac:
        out = model(xg, mask).dense_features
    out.float().square().mean().backward()
    fix = {
        "cfg": plain_cfg(cfg),
        "mode": mode,
        "state": {k: v.detach().clone() for k, v in model.state_dict().items()},
        "x": x,
        "eval_out": eval_out,
        "eval_out_nonorm": eval_out_nonorm,
        "train_out": out.detach().clone(),
        "x_grad": xg.grad.clone(),
        "grads": {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None},
        "train_seed": train_seed,
    }
    if mask is not None:
        fix["mask"] = mask
    path = os.path.join(OUT, name + ".pt")
    torch.save(fix, path)
    print(f"{name:45s} out={tuple(eval_out.shape)} {eval_out.dtype} bytes={os.path.getsize(path)}")


def make_op_level(ref):
    """Op-level vectors straight from the reference's L1 functions (apply_norm, apply_rope)."""
    from vit.norm import apply_norm
    from vit.rope import RopePositionEmbedding, apply_rope

    g = torch.Generator().manual_seed(7)
    fix = {}
    x = torch.randn(5, 7, 96, generator=g)
    w = 1 + 0.1 * torch.randn(96, generator=g)
    b = 0.1 * torch.randn(96, generator=g)
    for dt, tag in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        xd, wd, bd = x.to(dt), w.to(dt), b.to(dt)
        fix[f"norm_x_{tag}"], fix[f"norm_w_{tag}"], fix[f"norm_b_{tag}"] = xd, wd, bd
        fix[f"rms_eps1e-5_{tag}"] = apply_norm(xd, wd, None, 1e-5, use_layer_norm=False)
        fix[f"rms_epsNone_{tag}"] = torch.nn.functional.rms_norm(xd, (96,), weight=wd, eps=None)
        fix[f"ln_eps1e-5_{tag}"] = apply_norm(xd, wd, bd, 1e-5, use_layer_norm=True)
    for dt, tag in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        rope_mod = RopePositionEmbedding(128, num_heads=2, base=100.0, dtype=dt)
        table = rope_mod(H=3, W=5)
        q = torch.randn(2, 2, 3 + 15, 64, generator=g).to(dt)
        fix[f"rope_table_{tag}"] = table
        fix[f"rope_periods_{tag}"] = rope_mod.periods.clone()
        fix[f"rope_q_{tag}"] = q
        fix[f"rope_out_{tag}"] = apply_rope(q, table)
    # bf16 activations rotated with an fp32 table: the autocast mode of the model
    fix["rope_out_bf16_x_fp32_table"] = apply_rope(fix["rope_q_bf16"], fix["rope_table_fp32"])
    path = os.path.join(OUT, "g07_ops.pt")
    torch.save(fix, path)
    print(f"{'g07_ops':45s} bytes={os.path.getsize(path)}")


MASK_CASES = [   # (grid, mask_ratio, batch, scale, roll, seed)
    ((16, 16), 0.25, 3, 1, False, 11), ((14, 10), 0.6, 2, 1, False, 12), ((16, 16), 0.5, 4, 2, False, 13),
    ((16, 16), 0.4, 5, 2, True, 14), ((24,), 0.3, 2, 1, False, 15), ((12, 12), 0.75, 3, 4, True, 16),
]


def make_masks(ref):
    """Token masks drawn by the reference from a seeded default CPU generator (vit/tokens.py:107-218): create_mask with and
    without block scale / roll, and a non-overlapping second mask drawn right after the first."""
    from vit.tokens import create_mask, generate_non_overlapping_mask

    fix = {"cases": []}
    for size, ratio, B, scale, roll, seed in MASK_CASES:
        torch.manual_seed(seed)
        mask = create_mask(size, ratio, batch_size=B, scale=scale, roll=roll)
        second = generate_non_overlapping_mask(~mask, ratio, 0.2) if ratio + 0.2 <= 1 else None
        fix["cases"].append({"size": tuple(size), "mask_ratio": ratio, "batch": B, "scale": scale, "roll": roll, "seed": seed,
                             "mask": mask, "second_p2": 0.2, "second": second})
    path = os.path.join(OUT, "g16_masks.pt")
    torch.save(fix, path)
    print(f"{'g16_masks':45s} bytes={os.path.getsize(path)}")


HEAD_CASES = {
    # name -> (backbone overrides, mode)
    "g12_heads_fp32": (dict(activation="swiglu", pos_enc="fourier", num_register_tokens=2), "fp32"),
    "g12_heads_qknorm_layernorm_fp32": (dict(activation="srelu", pos_enc="rope", qk_normalization=True, norm_type="layernorm",
                                             attention_bias=True), "fp32"),
    "g12_heads_autocast": (dict(activation="swiglu", pos_enc="fourier", num_register_tokens=2), "autocast_bf16"),
}


def make_heads(ref, name, overrides, mode, seed):
    """Encoder + the README's heads (vit/head.py): AttentivePoolHead with 1 and with 3 queries on the visual tokens,
    the linear Head on the register/cls-free mean.  Fixture: state, x, the three head outputs, x_grad, grads of the
    loss sum(out.float().square().mean())."""
    from vit.head import AttentivePoolHeadConfig, HeadConfig

    kw = dict(BASE)
    kw.update(overrides)
    kw["dtype"] = torch.float32
    kw["heads"] = {"pool": AttentivePoolHeadConfig(out_features=10), "pool3": AttentivePoolHeadConfig(out_features=6, num_queries=3),
                   "lin": HeadConfig(out_features=5)}
    cfg = ref.ViTConfig(**kw)
    torch.manual_seed(seed)
    model = cfg.instantiate()
    gen = torch.Generator().manual_seed(seed + 17)
    perturb(model, gen)
    x = torch.randn(2, cfg.in_channels, 32, 32, generator=gen)
    ac = torch.autocast("cpu", dtype=torch.bfloat16, enabled=(mode == "autocast_bf16"))
    model.eval()
    xg = x.clone().requires_grad_(True)
    with ac:
        feats = model(xg)
        tokens = feats.visual_tokens
        outs = {"pool": model.heads["pool"](tokens), "pool3": model.heads["pool3"](tokens),
                "lin": model.heads["lin"](tokens.mean(dim=1))}
    sum(o.float().square().mean() for o in outs.values()).backward()
    with torch.no_grad(), ac:   # diagnostics: forward_attention_weights (no RoPE in it) and the pooling weights
        weights = {k: v.clone() for k, v in model.forward_attention_weights(x).items()}
        weights["pool3"] = model.heads["pool3"].forward_weights(tokens.detach()).clone()
    d = plain_cfg(cfg)
    d["heads"] = {"pool": dict(out_features=10, num_queries=1), "pool3": dict(out_features=6, num_queries=3),
                  "lin": dict(out_features=5)}
    fix = {"cfg": d, "mode": mode, "state": {k: v.detach().clone() for k, v in model.state_dict().items()}, "x": x,
           "outs": {k: v.detach().clone() for k, v in outs.items()}, "weights": weights, "x_grad": xg.grad.clone(),
           "grads": {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}}
    path = os.path.join(OUT, name + ".pt")
    torch.save(fix, path)
    print(f"{name:45s} outs={[tuple(v.shape) for v in outs.values()]} bytes={os.path.getsize(path)}")


LAYER_CASES = {
    # name -> (backbone overrides, mode)
    "g13_decoder_layers_fp32": (dict(activation="swiglu", pos_enc="fourier", layer_scale=0.5), "fp32"),
    "g13_decoder_layers_qknorm_ln_fp32": (dict(activation="gelu", pos_enc="fourier", norm_type="layernorm", qk_normalization=True),
                                          "fp32"),
    "g13_decoder_layers_autocast": (dict(activation="swiglu", pos_enc="fourier", layer_scale=0.5), "autocast_bf16"),
}


def make_layers(ref, name, overrides, mode, seed):
    """TransformerDecoderLayer and CrossAttentionTransformer from the backbone's factories (vit/vit.py:373-433),
    applied to x [3, 7, D] with memory kv [3, 11, D].  Fixture: the two state dicts, inputs, outputs, input and
    parameter gradients of out.float().square().mean()."""
    kw = dict(BASE)
    kw.update(overrides)
    kw["dtype"] = torch.float32
    cfg = ref.ViTConfig(**kw)
    torch.manual_seed(seed)
    model = cfg.instantiate()
    gen = torch.Generator().manual_seed(seed + 17)
    layers = {"decoder": model.create_decoder_layer(), "cross": model.create_cross_attention_layer()}
    D = cfg.hidden_size
    x = torch.randn(3, 7, D, generator=gen)
    kv = torch.randn(3, 11, D, generator=gen)
    fix = {"cfg": plain_cfg(cfg), "mode": mode, "x": x, "kv": kv}
    ac = torch.autocast("cpu", dtype=torch.bfloat16, enabled=(mode == "autocast_bf16"))
    for tag, layer in layers.items():
        perturb(layer, gen)
        layer.eval()
        xg, kvg = x.clone().requires_grad_(True), kv.clone().requires_grad_(True)
        with ac:
            out = layer(xg, kvg)
        out.float().square().mean().backward()
        fix[tag] = {"state": {k: v.detach().clone() for k, v in layer.state_dict().items()}, "out": out.detach().clone(),
                    "x_grad": xg.grad.clone(), "kv_grad": kvg.grad.clone(),
                    "grads": {k: p.grad.clone() for k, p in layer.named_parameters() if p.grad is not None}}
    path = os.path.join(OUT, name + ".pt")
    torch.save(fix, path)
    print(f"{name:45s} out={tuple(fix['decoder']['out'].shape)} bytes={os.path.getsize(path)}")


def main():
    ref = refshim.import_reference()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)  # fixed reduction order
    only = sys.argv[1:]   # optional name prefixes: regenerate just those fixtures (seeds do not depend on the selection)
    for i, (name, (ov, B, hw, mode)) in enumerate(CASES.items()):
        if only and not any(name.startswith(p) for p in only):
            continue
        make(ref, name, ov, B, hw, mode, seed=1000 + i)
    if not only or any("g07_ops".startswith(p) for p in only):
        make_op_level(ref)
    if not only or any("g16_masks".startswith(p) for p in only):
        make_masks(ref)
    for i, (name, (ov, mode)) in enumerate(HEAD_CASES.items()):
        if only and not any(name.startswith(p) for p in only):
            continue
        make_heads(ref, name, ov, mode, seed=2000 + i)
    for i, (name, (ov, mode)) in enumerate(LAYER_CASES.items()):
        if only and not any(name.startswith(p) for p in only):
            continue
        make_layers(ref, name, ov, mode, seed=3000 + i)


if __name__ == "__main__":
    main()
