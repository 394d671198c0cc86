"""TEST INFRASTRUCTURE, container-only: assert the oracle is bit-identical to the LIVE reference.

Run in the build container:  python oracle/check_against_reference.py
For each configuration the reference model (vit/vit.py) and oracle.vit_oracle.vit_forward run on the
same state_dict and input with one thread; outputs and every gradient must be torch.equal in fp32
(same ATen kernels, same order).  bf16 / autocast cases must be equal too (identical op sequence).
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import refshim  # noqa: E402
from gen_golden import BASE, CASES, perturb  # noqa: E402
from oracle import vit_oracle as O  # noqa: E402

EXTRA = {
    "vit_s16_224_cfg1": (dict(patch_size=(16, 16), img_size=(224, 224), depth=12, hidden_size=384, ffn_hidden_size=1536,
                              num_attention_heads=6), 2, (224, 224), "fp32"),
    "vit_b14_cfg2_depth2_autocast": (dict(patch_size=(14, 14), img_size=(224, 224), depth=2, hidden_size=768,
                                         ffn_hidden_size=3072, num_attention_heads=12, activation="swiglu",
                                         pos_enc="fourier", num_register_tokens=16), 2, (224, 224), "autocast_bf16"),
}


def main():
    ref = refshim.import_reference()
    torch.set_num_threads(1)
    bad = 0
    for i, (name, (ov, B, hw, mode)) in enumerate({**CASES, **EXTRA}.items()):
        kw = dict(BASE)
        kw.update(ov)
        kw["dtype"] = torch.bfloat16 if mode == "bf16" else torch.float32
        cfg = ref.ViTConfig(**kw)
        torch.manual_seed(i)
        model = cfg.instantiate()
        gen = torch.Generator().manual_seed(i + 5)
        perturb(model, gen)
        x = torch.randn(B, 3, *hw, generator=gen).to(cfg.dtype)
        ac = mode == "autocast_bf16"
        model.train()
        xg = x.clone().requires_grad_(True)
        torch.manual_seed(77)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=ac):
            out = model(xg).dense_features
        out.float().square().mean().backward()
        ocfg = O.OracleConfig(**{k: v for k, v in cfg.__dict__.items() if k != "heads"})
        r = O.run_fwd_bwd(ocfg, model.state_dict(), x, training=True, autocast=torch.bfloat16 if ac else None, seed=77)
        ok = torch.equal(r["out"], out.detach()) and torch.equal(r["x_grad"], xg.grad)
        for k, p in model.named_parameters():
            if p.grad is not None:
                ok = ok and torch.equal(r["grads"][k], p.grad)
        print(f"{name:45s} {mode:14s} bit-identical={ok}")
        bad += not ok
    print("ALL BIT-IDENTICAL" if bad == 0 else f"{bad} MISMATCH")
    return bad


if __name__ == "__main__":
    sys.exit(main())
