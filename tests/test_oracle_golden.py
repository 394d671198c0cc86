"""The CPU oracle reproduces the live reference's outputs stored in tests/golden/ (CPU, no GPU)."""
import pytest
import torch

from conftest import golden_names, head_golden_names, layer_golden_names, load_golden
from oracle import vit_oracle as O

DT = {"float32": torch.float32, "bfloat16": torch.bfloat16}


def cfg_from(fix) -> O.OracleConfig:
    kw = dict(fix["cfg"])
    kw["dtype"] = DT[kw["dtype"]]
    return O.OracleConfig(**kw)


def tol(mode):
    # fp32: same ATen kernels, same order -> equal up to threading; bf16: identical ops, allow 1 ulp-ish
    return dict(rtol=1e-5, atol=1e-6) if mode == "fp32" else dict(rtol=2e-2, atol=2e-3)


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_fixture(name):
    fix = load_golden(name)
    cfg = cfg_from(fix)
    ac = torch.bfloat16 if fix["mode"] == "autocast_bf16" else None
    sd, x = fix["state"], fix["x"]
    mask = fix.get("mask")   # masked-token fixtures carry the mask the reference was run with
    torch.set_num_threads(1)
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16, enabled=ac is not None):
        ev = O.vit_forward(cfg, sd, x, training=False, mask=mask)
        ev2 = O.vit_forward(cfg, sd, x, training=False, output_norm=False, mask=mask)
    torch.testing.assert_close(ev, fix["eval_out"], **tol(fix["mode"]))
    torch.testing.assert_close(ev2, fix["eval_out_nonorm"], **tol(fix["mode"]))
    assert ev.dtype == fix["eval_out"].dtype

    r = O.run_fwd_bwd(cfg, sd, x, training=True, autocast=ac, seed=fix["train_seed"], mask=mask)
    torch.testing.assert_close(r["out"], fix["train_out"], **tol(fix["mode"]))
    torch.testing.assert_close(r["x_grad"], fix["x_grad"], **tol(fix["mode"]))
    assert set(r["grads"]) == set(fix["grads"])
    for k, g in fix["grads"].items():
        torch.testing.assert_close(r["grads"][k], g, **tol(fix["mode"]), msg=lambda m, k=k: f"{k}: {m}")


def oracle_heads(cfg, sd, x, num_heads):
    """Encoder, then the three heads of the g12 fixtures (see oracle/gen_golden.py: make_heads)."""
    feats = O.vit_forward(cfg, sd, x, training=False)
    tokens = feats[:, cfg.num_cls_tokens + cfg.num_register_tokens:]
    return {"pool": O.attentive_pool_head(cfg, sd, "heads.pool.", tokens, num_heads),
            "pool3": O.attentive_pool_head(cfg, sd, "heads.pool3.", tokens, num_heads),
            "lin": O.linear_head(sd, "heads.lin.", tokens.mean(dim=1))}


@pytest.mark.parametrize("name", head_golden_names())
def test_oracle_heads_match_reference_fixture(name):
    fix = load_golden(name)
    kw = dict(fix["cfg"])
    kw.pop("heads")
    kw["dtype"] = DT[kw["dtype"]]
    cfg = O.OracleConfig(**kw)
    torch.set_num_threads(1)
    params = {k: v.detach().clone().requires_grad_(v.is_floating_point() and k != "rope.periods" and v.numel() > 0)
              for k, v in fix["state"].items()}
    x = fix["x"].clone().requires_grad_(True)
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=fix["mode"] == "autocast_bf16"):
        outs = oracle_heads(cfg, params, x, cfg.num_attention_heads)
    sum(o.float().square().mean() for o in outs.values()).backward()
    t = tol(fix["mode"])
    for k, v in fix["outs"].items():
        assert outs[k].dtype == v.dtype and outs[k].shape == v.shape
        torch.testing.assert_close(outs[k].detach(), v, **t, msg=lambda m, k=k: f"{k}: {m}")
    torch.testing.assert_close(x.grad, fix["x_grad"], **t)
    got = {k: p.grad for k, p in params.items() if p.requires_grad and p.grad is not None}
    assert set(got) == set(fix["grads"])
    for k, g in fix["grads"].items():
        torch.testing.assert_close(got[k], g, **t, msg=lambda m, k=k: f"{k}: {m}")
    # attention probabilities (vit/vit.py:575-596, vit/attention.py:658-659)
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16, enabled=fix["mode"] == "autocast_bf16"):
        sd = fix["state"]
        w = O.vit_attention_weights(cfg, sd, fix["x"])
        feats = O.vit_forward(cfg, sd, fix["x"], training=False)
        w["pool3"] = O.attentive_pool_weights(cfg, sd, "heads.pool3.pool.",
                                              feats[:, cfg.num_cls_tokens + cfg.num_register_tokens:], cfg.num_attention_heads)
    assert set(w) == set(fix["weights"])
    for k, v in fix["weights"].items():
        assert w[k].dtype == v.dtype and w[k].shape == v.shape, k
        torch.testing.assert_close(w[k], v, **t, msg=lambda m, k=k: f"{k}: {m}")


@pytest.mark.parametrize("name", layer_golden_names())
@pytest.mark.parametrize("kind", ["decoder", "cross"])
def test_oracle_decoder_layers_match_reference_fixture(name, kind):
    fix = load_golden(name)
    cfg = cfg_from(fix)
    part = fix[kind]
    torch.set_num_threads(1)
    params = {k: v.detach().clone().requires_grad_(True) for k, v in part["state"].items()}
    x, kv = fix["x"].clone().requires_grad_(True), fix["kv"].clone().requires_grad_(True)
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=fix["mode"] == "autocast_bf16"):
        out = O.decoder_layer(cfg, params, "", x, kv, False, self_attn=kind == "decoder")
    out.float().square().mean().backward()
    t = tol(fix["mode"])
    assert out.dtype == part["out"].dtype
    torch.testing.assert_close(out.detach(), part["out"], **t)
    torch.testing.assert_close(x.grad, part["x_grad"], **t)
    torch.testing.assert_close(kv.grad, part["kv_grad"], **t)
    got = {k: p.grad for k, p in params.items() if p.grad is not None}
    assert set(got) == set(part["grads"])
    for k, g in part["grads"].items():
        torch.testing.assert_close(got[k], g, **t, msg=lambda m, k=k: f"{k}: {m}")


def test_op_level_vectors():
    fix = load_golden("g07_ops")
    for tag in ("fp32", "bf16"):
        x, w, b = fix[f"norm_x_{tag}"], fix[f"norm_w_{tag}"], fix[f"norm_b_{tag}"]
        assert torch.equal(O.apply_norm(x, w, None, 1e-5, False), fix[f"rms_eps1e-5_{tag}"])
        assert torch.equal(O.apply_norm(x, w, None, None, False), fix[f"rms_epsNone_{tag}"])
        assert torch.equal(O.apply_norm(x, w, b, 1e-5, True), fix[f"ln_eps1e-5_{tag}"])
        dt = x.dtype
        periods = O.rope_periods(64, 100.0, dt)
        assert torch.equal(periods, fix[f"rope_periods_{tag}"])
        table = O.rope_table(3, 5, periods, dt)
        assert torch.equal(table, fix[f"rope_table_{tag}"])
        assert torch.equal(O.apply_rope(fix[f"rope_q_{tag}"], table), fix[f"rope_out_{tag}"])
    assert torch.equal(O.apply_rope(fix["rope_q_bf16"], fix["rope_table_fp32"]), fix["rope_out_bf16_x_fp32_table"])


def test_rmsnorm_default_eps_is_opmath_epsilon():
    # nn.RMSNorm(eps=None) (vit/vit.py:316): with torch 2.10 the default resolves to finfo(float32).eps
    # for fp32 AND bf16 inputs (the statistic is computed in the fp32 op-math type), measured here:
    x = torch.full((1, 8), 1e-3)
    w = torch.ones(8)
    for dt in (torch.float32, torch.bfloat16):
        xd = x.to(dt)
        y = O.apply_norm(xd, w.to(dt), None, None, False).float()[0, 0].item()
        ms = xd.float().square().mean().item()
        eps = (xd[0, 0].float().item() / y) ** 2 - ms
        assert abs(eps - torch.finfo(torch.float32).eps) < 3e-8 if dt == torch.float32 else abs(eps) < 5e-7
