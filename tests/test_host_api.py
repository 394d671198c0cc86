"""Host-side contract of the drop-in (no GPU, no kernel calls): configuration, module tree, state_dict layout,
initialisation rules, drop-path bookkeeping, feature views, the C ABI surface, and loud failure without a GPU.
Reference behaviours cited per test (paths relative to the reference checkout)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT, load_golden
from vit_amd import ViT, ViTConfig, ViTFeatures
from vit_amd import _lib
from vit_amd.transformer import TransformerEncoderLayer, _select_keep_indices, drop_path_keep_count


def small(**kw):
    base = dict(in_channels=3, patch_size=(8, 8), img_size=(32, 32), depth=2, hidden_size=64, ffn_hidden_size=128,
                num_attention_heads=2, dtype=torch.float32)
    base.update(kw)
    return ViTConfig(**base)


def test_config_validation_errors():  # vit/vit.py:115-125
    with pytest.raises(ValueError, match="divisible"):
        small(hidden_size=65)
    with pytest.raises(ValueError, match="norm_type"):
        small(norm_type="batchnorm")
    with pytest.raises(ValueError, match="even"):
        small(hidden_size=63, num_attention_heads=3, pos_enc="fourier")


def test_config_defaults_match_reference():  # vit/vit.py:79-110
    c = small()
    assert (c.hidden_dropout, c.attention_dropout, c.activation, c.pos_enc, c.norm_type) == (0.1, 0.1, "srelu", "rope", "rmsnorm")
    assert c.rope_base == 100 and c.rope_normalize_coords == "separate" and c.layer_scale is None
    assert ViTConfig(in_channels=1, patch_size=(4, 4), img_size=(8, 8), depth=1, hidden_size=32, ffn_hidden_size=64,
                     num_attention_heads=2).dtype == torch.bfloat16


def test_yaml_round_trip():  # vit/vit.py:130-152
    c = small(activation="swiglu", pos_enc="fourier", num_register_tokens=4, layer_scale=1e-5, dtype=torch.bfloat16)
    assert ViTConfig.from_yaml(c.to_yaml()) == c
    assert "bfloat16" in c.to_yaml()


def test_state_dict_keys_match_the_reference_fixture():
    """Key set and shapes equal the live reference's state_dict for several configurations (tests/golden/*.pt)."""
    for name in ("g01_fourier_swiglu_reg_fp32", "g02_rope_srelu_bf16", "g03_layernorm_gelu_learnable_cls_fp32",
                 "g04_droppath_layerscale_fp32", "g07_qknorm_fp32", "g08_patchnorm_fp32"):
        fix = load_golden(name)
        kw = dict(fix["cfg"])
        kw["dtype"] = getattr(torch, kw["dtype"])
        model = ViTConfig(**kw).instantiate()
        ours = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        ref = {k: tuple(v.shape) for k, v in fix["state"].items()}
        assert ours == ref, name
        model.load_state_dict(fix["state"], strict=True)
        for k, v in model.state_dict().items():
            assert v.dtype == fix["state"][k].dtype, (name, k)


def test_parameters_are_created_in_config_dtype_on_device():  # tests/test_vit.py:210-298 of the reference
    for dt in (torch.float32, torch.bfloat16):
        m = small(dtype=dt, pos_enc="fourier", num_register_tokens=2, layer_scale=0.1).instantiate(device=torch.device("cpu"))
        assert all(p.dtype == dt for p in m.parameters())
        assert all(b.dtype == dt for b in m.buffers())


def test_residual_outputs_zero_initialised_and_norm_eps():  # vit/transformer.py:57-66,132; vit/vit.py:316
    m = small().instantiate()
    for blk in m.blocks:
        assert isinstance(blk, TransformerEncoderLayer)
        assert torch.all(blk.self_attention.out_proj.weight == 0) and torch.all(blk.mlp.fc2.weight == 0)
        assert torch.all(blk.self_attention.out_proj.bias == 0) and torch.all(blk.mlp.fc2.bias == 0)
        assert blk.self_attention.norm.eps == 1e-5 and blk.mlp.norm.eps == 1e-5
        assert blk.self_attention.qkv_proj.weight.abs().max() <= 0.04 + 1e-6  # trunc normal at 2 sigma
    assert m.output_norm.eps is None
    assert small(patch_embed_normalization=True).instantiate().patch_embed_norm.eps is None
    assert small(norm_type="layernorm").instantiate().output_norm.eps == 1e-5


def test_token_parameters_and_frozen_flags():  # vit/vit.py:297-306,598-615
    m = small(num_register_tokens=3, num_cls_tokens=0, mlp_requires_grad=False).instantiate()
    assert m.register_tokens.shape == (1, 3, 64) and m.register_tokens.requires_grad
    assert m.cls_tokens.shape == (1, 0, 64) and not m.cls_tokens.requires_grad
    assert all(not p.requires_grad for blk in m.blocks for p in blk.mlp.parameters())
    assert all(p.requires_grad for blk in m.blocks for p in blk.self_attention.parameters())
    m.backbone_requires_grad_(False)
    assert not any(p.requires_grad for p in m.parameters())


def test_glu_doubles_fc1_and_layer_scale_modules():  # vit/fused.py:189-195, vit/transformer.py:122-131
    m = small(activation="swiglu", layer_scale=0.5).instantiate()
    assert m.blocks[0].mlp.fc1.weight.shape == (256, 64) and m.blocks[0].mlp.fc2.weight.shape == (64, 128)
    assert torch.all(m.blocks[0].layer_scale_attn.gamma == 0.5)
    assert isinstance(small().instantiate().blocks[0].layer_scale_mlp, torch.nn.Identity)


def test_drop_path_keep_count_and_scale():  # vit/transformer.py:14-27, reference tests/test_transformer.py:137-197
    assert drop_path_keep_count(8, 0.25) == 6
    assert drop_path_keep_count(64, 0.1) == 57
    assert drop_path_keep_count(2, 0.99) == 1
    x = torch.zeros(8, 4, 16)
    idx, scale = _select_keep_indices(x, 0.25, True)
    assert idx.numel() == 6 and len(set(idx.tolist())) == 6 and abs(scale - 8 / 6) < 1e-12
    assert _select_keep_indices(x, 0.25, False) == (None, 1.0)      # eval
    assert _select_keep_indices(x, 0.0, True) == (None, 1.0)        # p <= 0
    assert _select_keep_indices(x[:1], 0.5, True) == (None, 1.0)    # B <= 1
    torch.manual_seed(5)
    a = _select_keep_indices(x, 0.5, True)[0]
    torch.manual_seed(5)
    assert torch.equal(a, torch.randperm(8)[:4])                   # same draw as the reference's randperm


def test_features_views():  # vit/vit.py:155-251
    d = torch.arange(2 * 7 * 3, dtype=torch.float32).view(2, 7, 3)
    f = ViTFeatures(d, num_register_tokens=2, num_cls_tokens=1, tokenized_size=(2, 2))
    assert torch.equal(f.cls_tokens, d[:, :1]) and torch.equal(f.register_tokens, d[:, 1:3])
    assert torch.equal(f.visual_tokens, d[:, 3:]) and f.visual_tokens.data_ptr() == d[:, 3:].data_ptr()
    assert f.visual_tokens_as_grid.shape == (2, 2, 2, 3)
    c, r, v = f
    assert v.shape == (2, 4, 3)
    g = ViTFeatures.from_separate_features(c, r, v, (2, 2))
    assert torch.equal(g.dense_features, d)
    assert torch.equal(f.apply(lambda t: t * 2).dense_features, d * 2)
    with pytest.raises(ValueError):
        ViTFeatures(d, 2, 1).visual_tokens_as_grid


def test_rope_table_host_code_matches_reference_vectors():  # vit/rope.py:60-136
    from vit_amd.rope import RopePositionEmbedding

    fix = load_golden("g07_ops")
    for tag, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        rope = RopePositionEmbedding(128, num_heads=2, base=100.0, dtype=dt)
        assert torch.equal(rope.periods, fix[f"rope_periods_{tag}"])
        assert torch.equal(rope(H=3, W=5), fix[f"rope_table_{tag}"])
    with pytest.raises(AssertionError):
        RopePositionEmbedding(100, num_heads=3)


def test_unsupported_features_fail_loudly():
    with pytest.raises(NotImplementedError):
        small(heads={"cls": object()}).instantiate()
    with pytest.raises(ValueError):
        small(patch_size=(4, 4, 4, 4), img_size=(8, 8, 8, 8)).instantiate()
    vol = small(patch_size=(4, 4, 4), img_size=(8, 8, 8)).instantiate()   # volumetric stem: Conv3d parameter holder
    assert vol.stem.patch.weight.dim() == 5 and vol.stem.tokenized_size((8, 8, 8)) == (2, 2, 2)


def test_forward_on_cpu_raises_instead_of_falling_back():
    m = small().instantiate()
    with pytest.raises(RuntimeError, match="no CPU fallback|MI355X"):
        m(torch.randn(1, 3, 32, 32))


def test_c_abi_library_exports_every_declared_symbol():
    """Every function declared in include/vitk.h is exported by vit_amd/libvitk.so (loading needs no GPU)."""
    header = open(os.path.join(ROOT, "include", "vitk.h")).read()
    declared = set(re.findall(r"\b(vitk_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert os.path.exists(_lib.LIB_PATH), "run `python __graft_entry__.py build` first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), sym
    lib.vitk_version.restype = ctypes.c_int
    assert lib.vitk_version() >= 100


def test_ctypes_struct_layout_matches_header_field_order():
    header = open(os.path.join(ROOT, "include", "vitk.h")).read()
    for cname, cls in (("VitkGemm", _lib.VitkGemm), ("VitkAttn", _lib.VitkAttn)):
        body = header[header.index(f"typedef struct {cname} {{"):header.index(f"}} {cname};")]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for stmt in body.split("{", 1)[1].split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            parts = [re.sub(r"[*\s]", "", piece.split()[-1]) for piece in stmt.split(",")]
            names += parts
        assert names == [f[0] for f in cls._fields_], cname


# ---- token masks (vit/tokens.py): the cases of the reference's tests/test_tokens.py:10-57 ----------------------
@pytest.mark.parametrize("mask, exp", [
    (([True, True], [True, True]), False),
    (([True, False], [True, True]), True),
    (([False, True], [True, True]), True),
    (([False, False], [True, True]), True),
    (([False, False], [False, False]), False),
])
def test_mask_is_ragged(mask, exp):
    from vit_amd.tokens import mask_is_ragged

    assert mask_is_ragged(torch.tensor(mask, dtype=torch.bool)) == exp


@pytest.mark.parametrize("mask, fill_value, padding_value, exp", [
    (([True, True], [True, True]), None, 0, [[0, 1], [0, 1]]),
    (([True, False], [False, True]), None, 0, [[0], [1]]),
    (([False, True], [True, False]), None, 0, [[1], [0]]),
    (([False, False], [False, False]), None, 0, [[], []]),
    (([True, False], [False, True]), -1.0, 0, [[0, -1.0], [-1.0, 1]]),
    (([True, False], [False, True]), torch.tensor(-1.0), 0, [[0, -1.0], [-1.0, 1]]),
    (([True, False], [True, True]), None, 0, [[0, 0], [0, 1]]),
    (([False, True], [True, True]), None, 0, [[1, 0], [0, 1]]),
    (([True, False], [True, True]), None, -1.0, [[0, -1.0], [0, 1]]),
    (([False, True], [True, True]), None, -1.0, [[1, -1.0], [0, 1]]),
    (([False, True], [True, True]), -1.0, 0, [[-1.0, 1], [0, 1]]),
])
def test_apply_mask_cases(mask, fill_value, padding_value, exp):
    from vit_amd.tokens import apply_mask

    mask = torch.tensor(mask, dtype=torch.bool)
    N, L = mask.shape
    x = torch.arange(L).view(1, L, 1).expand(N, L, 1)
    o = apply_mask(mask, x, fill_value, padding_value)
    assert torch.equal(o.float().reshape(N, -1), torch.tensor(exp).float().reshape(N, -1))


@pytest.mark.parametrize("size, ratio, scale, roll", [((8, 8), 0.5, 1, False), ((8, 12), 0.25, 2, False),
                                                      ((16, 16), 0.75, 4, True), ((10,), 0.3, 1, False)])
def test_create_mask_drops_the_requested_share(size, ratio, scale, roll):
    from vit_amd.tokens import apply_mask, create_mask, mask_is_ragged, unapply_mask

    torch.manual_seed(0)
    B = 6
    L = 1
    for s in size:
        L *= s
    mask = create_mask(size, ratio, batch_size=B, scale=scale, roll=roll)
    assert mask.shape == (B, L) and mask.dtype == torch.bool and not mask_is_ragged(mask)
    coarse = L // scale ** len(size)
    assert int((~mask[0]).sum()) == round(coarse * ratio) * scale ** len(size)
    assert not torch.equal(mask[0], mask[1])   # drawn per sample
    x = torch.randn(B, L, 5)
    back = unapply_mask(mask, apply_mask(mask, x))
    assert torch.equal(back[mask], x[mask]) and bool((back[~mask] == 0).all())
    with pytest.raises(ValueError):
        create_mask(size, 1.5)


def test_masks_follow_the_references_random_stream():
    """tests/golden/g16_masks.pt: masks the live reference drew from a seeded CPU generator (oracle/gen_golden.py:make_masks;
    vit/tokens.py:107-218).  The same seed here draws the same masks, block scale, roll and the non-overlapping second
    mask included."""
    import os

    from vit_amd.tokens import create_mask, generate_non_overlapping_mask

    fix = torch.load(os.path.join(os.path.dirname(__file__), "golden", "g16_masks.pt"), weights_only=True)
    assert len(fix["cases"]) >= 6
    for c in fix["cases"]:
        torch.manual_seed(c["seed"])
        mask = create_mask(c["size"], c["mask_ratio"], batch_size=c["batch"], scale=c["scale"], roll=c["roll"])
        assert torch.equal(mask, c["mask"]), (c["size"], c["scale"], c["roll"])
        if c["second"] is not None:
            second = generate_non_overlapping_mask(~mask, c["mask_ratio"], c["second_p2"])
            assert torch.equal(second, c["second"]), c["size"]


def test_generate_non_overlapping_mask():   # tests/test_tokens.py:147-170
    from vit_amd.tokens import create_mask, generate_non_overlapping_mask

    torch.manual_seed(1)
    m1 = ~create_mask((100,), 0.4, batch_size=4)   # True = selected by the first mask
    m2 = generate_non_overlapping_mask(m1, 0.4, 0.3)
    assert not bool((m1 & m2).any()) and bool((m2.sum(1) == 30).all())
    with pytest.raises(ValueError):
        generate_non_overlapping_mask(m1, 0.6, 0.6)


def test_masked_rope_table_is_per_sample():   # vit/vit.py:507-529
    cfg = small(pos_enc="rope")
    model = cfg.instantiate()
    x = torch.zeros(3, cfg.in_channels, *cfg.img_size)
    mask = model.create_mask(x, unmasked_ratio=0.5, scale=1)
    grid = model.stem.tokenized_size(tuple(cfg.img_size))
    full = model.prepare_rope(grid)
    per = model.prepare_rope(grid, mask)
    kept = int(mask[0].sum())
    assert per.shape == (2, 3, 1, kept, full.shape[-1])
    for b in range(3):
        assert torch.equal(per[0, b, 0], full[0][mask[b]]) and torch.equal(per[1, b, 0], full[1][mask[b]])
