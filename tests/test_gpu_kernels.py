"""GPU parity of each libvitk kernel (called through the C ABI) against fp32/fp64 torch math on the same inputs.
Tolerances: fp32 kernels rtol 1e-4 (north-star bar); bf16 kernels rtol 2e-2; integer-valued MFMA checks exact."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from vit_amd import _lib as L
    from vit_amd import kernels as K

DEV = "cuda"


def close(a, b, rtol, atol_rel=None, msg=""):
    a, b = a.float(), b.float()
    atol = (atol_rel if atol_rel is not None else rtol) * max(b.abs().max().item(), 1e-6)
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol, msg=lambda m: f"{msg}: {m}")


def test_library_loads_and_reports_errors():
    lib = L.load()
    assert lib.vitk_version() >= 100
    x = torch.zeros(4, 6, device=DEV)
    with pytest.raises(RuntimeError, match="multiple of 4"):
        K.norm_fwd(x, torch.ones(6, device=DEV), None, 1e-5, False, torch.float32)


# ---------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
@pytest.mark.parametrize("shape", [(128, 128, 64), (200, 136, 72), (384, 264, 200), (1000, 72, 328), (768, 512, 448), (520, 392, 1280)])
def test_mfma_gemm_exact_on_integers(layout, shape):
    """A = small integers, B = asymmetric small integers: every product and sum is exact in bf16/fp32, so the
    MFMA kernel (operand lane maps, transposed LDS reads, C layout, tails) must match torch bit for bit."""
    M, N, Kd = shape
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
    A = torch.randint(-3, 4, (M, Kd), generator=g).float()
    B = torch.randint(-2, 3, (Kd, N), generator=g).float() + (torch.arange(N)[None, :] % 3 == 0).float()
    ref = (A @ B).to(DEV)
    a = (A if layout != "tn" else A.t()).contiguous().to(DEV, torch.bfloat16)
    b = (B.t() if layout == "nt" else B).contiguous().to(DEV, torch.bfloat16)
    out = torch.empty(M, N, device=DEV, dtype=torch.float32)
    K.gemm(a, b, trans_a=layout == "tn", trans_b=layout != "nt", M=M, N=N, K=Kd, out=out, algo=2)
    assert torch.equal(out, ref), f"max err {(out - ref).abs().max().item()}"
    for algo in (1, 3, 4, 5):  # portable, 256x256-tile MFMA (register staged), 128x128-tile MFMA, 256x256 tile with LDS-DMA staging
        out1 = torch.empty_like(out)
        K.gemm(a, b, trans_a=layout == "tn", trans_b=layout != "nt", M=M, N=N, K=Kd, out=out1, algo=algo)
        assert torch.equal(out1, ref), f"algo {algo}: max err {(out1 - ref).abs().max().item()}"


@pytest.mark.parametrize("shape", [(256, 128, 4096), (200, 136, 328), (768, 512, 8192), (2304, 768, 2176)])
def test_gemm_fused_bias_gradient_row_sums(shape):
    """a_colsum[m] = sum_k A[k, m] from the weight-gradient pass (exact on small integers): every kernel family,
    with and without split-K, fp32 and bf16 outputs."""
    M, N, Kd = shape
    g = torch.Generator().manual_seed(M + Kd)
    A = torch.randint(-2, 3, (Kd, M), generator=g).float()
    B = torch.randint(-2, 3, (Kd, N), generator=g).float()
    ref, ref_sum = (A.t() @ B).to(DEV), A.sum(0).to(DEV)
    a, b = A.to(DEV, torch.bfloat16), B.to(DEV, torch.bfloat16)
    for od in (torch.float32, torch.bfloat16):
        for algo in (0, 1, 3, 4, 5):
            out = torch.empty(M, N, device=DEV, dtype=od)
            sums = torch.full((M,), float("nan"), device=DEV, dtype=od)
            K.gemm(a, b, trans_a=True, trans_b=True, M=M, N=N, K=Kd, out=out, a_colsum=sums, algo=algo)
            assert torch.equal(out.float(), ref.to(od).float()), f"algo {algo} {od}: product"
            assert torch.equal(sums.float(), ref_sum.to(od).float()), f"algo {algo} {od}: sums {(sums.float() - ref_sum).abs().max().item()}"
    with pytest.raises(RuntimeError, match="a_colsum"):
        K.gemm(a.t().contiguous(), b, trans_a=False, trans_b=True, M=M, N=N, K=Kd, out=out, a_colsum=sums)


@pytest.mark.parametrize("Kd", [192, 2304])
def test_gemm_ragged_last_round_is_cut_between_tile_sizes(Kd):
    """270 tiles of 256x256 on 256 CUs: with the long contraction the automatic path runs whole rounds on the 256-tile
    kernel and the remaining rows on the 128-tile kernel (those few tiles are split along K and the residual epilogue
    is applied by the pass that sums the partial products); with the short one a second round costs less than a
    second launch and the problem stays whole.  Integer operands make every kernel exact, so the cut must
    be invisible; the residual epilogue is checked with a row gather and dropout (same mask as the uncut launch)."""
    M, N = 256 * 90, 768
    g = torch.Generator().manual_seed(11)
    A = torch.randint(-2, 3, (M, Kd), generator=g).float()
    B = torch.randint(-2, 3, (Kd, N), generator=g).float()
    ref = (A @ B).to(DEV)
    a = A.to(DEV, torch.bfloat16)
    for layout, b in (("nt", B.t().contiguous().to(DEV, torch.bfloat16)), ("nn", B.to(DEV, torch.bfloat16))):
        out = torch.empty(M, N, device=DEV, dtype=torch.float32)
        K.gemm(a, b, trans_a=False, trans_b=layout == "nn", M=M, N=N, K=Kd, out=out, algo=0)
        assert torch.equal(out, ref), layout
    bt = B.t().contiguous().to(DEV, torch.bfloat16)
    rows = torch.randperm(M + 512, generator=g)[:M].to(DEV)
    x = torch.randn(M + 512, N, generator=g).to(DEV)
    outs = []
    for algo in (0, 4):
        o = x.clone()
        aux = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        K.gemm(a, bt, trans_a=False, trans_b=False, M=M, N=N, K=Kd, out=o, res=o, epilogue=L.EPI_RESIDUAL, row_idx=rows,
               aux=aux, alpha=0.5, drop_p=0.25, seed=321, algo=algo)
        outs.append((o, aux))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    kept = (outs[0][1] != 0).float().mean().item()
    assert 0.6 < kept < 0.8   # ~75% kept (zeros of the product itself excluded by the loose band)


def test_mfma_gemm_splitk_wgrad_shape():
    M, N, Kd = 256, 128, 4096  # few output tiles, long contraction -> split-K path
    g = torch.Generator().manual_seed(3)
    A = torch.randint(-2, 3, (Kd, M), generator=g).float()
    B = torch.randint(-2, 3, (Kd, N), generator=g).float()
    ref = (A.t() @ B).to(DEV)
    for od in (torch.float32, torch.bfloat16):
        out = torch.empty(M, N, device=DEV, dtype=od)
        K.gemm(A.to(DEV, torch.bfloat16), B.to(DEV, torch.bfloat16), trans_a=True, trans_b=True, M=M, N=N, K=Kd, out=out, algo=2)
        close(out, ref, 1e-2 if od == torch.bfloat16 else 0, 1e-2 if od == torch.bfloat16 else 0)


def test_gemm_weight_gradient_with_a_ragged_second_round_splits_k():
    """288 tiles of 256x256 (ViT-g/14's fc1 weight gradient has 48 x 6) on 256 CUs: the second round would be 1/8 full,
    so the automatic path splits the long contraction instead.  Integer operands keep every partial sum exact; the
    fused bias-gradient row sums travel through the same partial buffers."""
    M, N, Kd = 256 * 48, 256 * 6 - 128, 8192
    g = torch.Generator().manual_seed(48)
    A = torch.randint(-2, 3, (Kd, M), generator=g).to(torch.bfloat16)
    B = torch.randint(-2, 3, (Kd, N), generator=g).to(torch.bfloat16)
    a, b = A.to(DEV), B.to(DEV)
    ref = a.float().t() @ b.float()
    ref_sum = a.float().sum(0)
    for od in (torch.float32, torch.bfloat16):
        out = torch.empty(M, N, device=DEV, dtype=od)
        sums = torch.full((M,), float("nan"), device=DEV, dtype=od)
        K.gemm(a, b, trans_a=True, trans_b=True, M=M, N=N, K=Kd, out=out, a_colsum=sums, algo=0)
        assert torch.equal(out.float(), ref.to(od).float()), od
        assert torch.equal(sums.float(), ref_sum.to(od).float()), od


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_random_bias_and_tolerance(dtype):
    M, N, Kd = 520, 384, 256
    a = torch.randn(M, Kd, device=DEV)
    w = torch.randn(N, Kd, device=DEV) * 0.05
    bias = torch.randn(N, device=DEV)
    ref = a.to(dtype).double() @ w.to(dtype).double().t() + bias.to(dtype).double()
    out = torch.empty(M, N, device=DEV, dtype=dtype)
    K.gemm(a.to(dtype), w.to(dtype), trans_a=False, trans_b=False, M=M, N=N, K=Kd, out=out, bias=bias.to(dtype))
    close(out, ref, 1e-4 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("algo", [1, 2])
def test_gemm_residual_epilogue_scatter_gamma(algo):
    M, N, Kd, R = 256, 128, 64, 400
    g = torch.Generator().manual_seed(5)
    a = torch.randn(M, Kd, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(N, Kd, generator=g) * 0.1).to(DEV, torch.bfloat16)
    bias = torch.randn(N, generator=g).to(DEV, torch.bfloat16)
    res = torch.randn(R, N, generator=g).to(DEV)
    gamma = torch.randn(N, generator=g).to(DEV)
    rows = torch.randperm(R, generator=g)[:M].to(DEV)
    out = res.clone()
    aux = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    K.gemm(a, w, trans_a=False, trans_b=False, M=M, N=N, K=Kd, out=out, bias=bias, epilogue=L.EPI_RESIDUAL, res=out,
           gamma=gamma, row_idx=rows, aux=aux, alpha=1.5, algo=algo)
    lin = (a.double() @ w.double().t() + bias.double()).to(torch.bfloat16)
    ref = res.clone().double()
    ref[rows] += 1.5 * gamma.double() * lin.double()
    close(out, ref, 1e-2)
    close(aux, lin, 1e-2)
    untouched = torch.ones(R, dtype=torch.bool, device=DEV)
    untouched[rows] = False
    assert torch.equal(out[untouched], res[untouched])


def test_gemm_patch_epilogue_places_rows():
    B, T, N, Kd, S, P = 3, 10, 64, 64, 14, 4
    g = torch.Generator().manual_seed(6)
    a = torch.randn(B * T, Kd, generator=g).to(DEV)
    w = torch.randn(N, Kd, generator=g).to(DEV)
    pos = torch.randn(T, N, generator=g).to(DEV)
    out = torch.full((B, S, N), 7.0, device=DEV)
    K.gemm(a, w, trans_a=False, trans_b=False, M=B * T, N=N, K=Kd, out=out, epilogue=L.EPI_PATCH, pos=pos, tokens=T, seq=S,
           prefix=P)
    ref = (a.double() @ w.double().t()).view(B, T, N) + pos.double()
    close(out[:, P:P + T], ref, 1e-4)
    assert torch.all(out[:, :P] == 7.0)


# ---------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("ln", [False, True])
@pytest.mark.parametrize("xdt,ydt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("D", [64, 520, 768, 1408, 2048])
def test_norm_forward_backward(ln, xdt, ydt, D):
    R = 37
    g = torch.Generator().manual_seed(D + ln)
    x = (torch.randn(50, D, generator=g) * 2 + 0.3).to(DEV, xdt)
    w = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV, xdt)
    b = (0.1 * torch.randn(D, generator=g)).to(DEV, xdt) if ln else None
    rows = torch.randperm(50, generator=g)[:R].to(DEV)
    y, rstd, mean = K.norm_fwd(x, w, b, 1e-5, ln, ydt, rows)
    xd = x.double()[rows].requires_grad_(True)
    wd = w.double().requires_grad_(True)
    bd = b.double().requires_grad_(True) if ln else None
    ref = torch.nn.functional.layer_norm(xd, (D,), wd, bd, 1e-5) if ln else torch.nn.functional.rms_norm(xd, (D,), wd, 1e-5)
    tol = 1e-4 if ydt == torch.float32 and xdt == torch.float32 else 1e-2
    close(y, ref, tol)
    dy = torch.randn(R, D, generator=g).to(DEV, ydt)
    ref.backward(dy.double())
    add = torch.randn(50, D, generator=g).to(DEV, xdt)
    dx = add.clone()
    dw, db = K.norm_bwd(dy, x, w, rstd, mean, ln, dx, dx, True, ln, rows)
    dx_ref = add.double().clone()
    dx_ref[rows] += xd.grad
    close(dx, dx_ref, tol)
    close(dw, wd.grad, tol, 2 * tol)
    if ln:
        close(db, bd.grad, tol, 2 * tol)


# ---------------------------------------------------------------------------------------------- attention
def attn_reference(qkv, B, S, H, Dh, dout=None):
    D = H * Dh
    q, k, v = [t.reshape(B, S, H, Dh).transpose(1, 2).double() for t in qkv.double().reshape(B * S, 3, D).unbind(1)]
    q.requires_grad_(True), k.requires_grad_(True), v.requires_grad_(True)
    s = (q @ k.transpose(-1, -2)) * Dh ** -0.5
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(B * S, D)
    lse = torch.logsumexp(s, -1)
    if dout is None:
        return o, lse, None
    o.backward(dout.double())
    dqkv = torch.stack([t.grad.transpose(1, 2).reshape(B * S, D) for t in (q, k, v)], dim=1).reshape(B * S, 3 * D)
    return o.detach(), lse.detach(), dqkv


@pytest.mark.parametrize("S,Dh", [(64, 64), (272, 64), (100, 32), (197, 80), (130, 88), (96, 128), (577, 64)])
@pytest.mark.parametrize("mode", ["portable_fp32", "portable_bf16", "mfma"])
def test_attention_forward_backward(S, Dh, mode):
    B, H = 2, 3
    dtype = torch.float32 if mode == "portable_fp32" else torch.bfloat16
    algo = 2 if mode == "mfma" else 1
    g = torch.Generator().manual_seed(S * 131 + Dh)
    qkv = torch.randn(B * S, 3 * H * Dh, generator=g).to(DEV, dtype)
    dout = torch.randn(B * S, H * Dh, generator=g).to(DEV, dtype)
    o, lse = K.attn_fwd(qkv, B, S, H, Dh, Dh ** -0.5, algo=algo)
    dqkv = K.attn_bwd(qkv, o, lse, dout, B, S, H, Dh, Dh ** -0.5, algo=algo)
    o_ref, lse_ref, dqkv_ref = attn_reference(qkv, B, S, H, Dh, dout)
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    close(o, o_ref, tol, msg="o")
    close(lse, lse_ref, 1e-4 if dtype == torch.float32 else 1e-3, msg="lse")
    D = H * Dh
    for i, name in enumerate("qkv"):
        close(dqkv[:, i * D:(i + 1) * D], dqkv_ref[:, i * D:(i + 1) * D], tol, msg="d" + name)


def cross_attn_reference(q, kv, B, Sq, Sk, H, Dh, dout):
    D = H * Dh
    qd = q.double().reshape(B, Sq, H, Dh).transpose(1, 2).requires_grad_(True)
    k, v = [t.reshape(B, Sk, H, Dh).transpose(1, 2).requires_grad_(True) for t in kv.double().reshape(B * Sk, 2, D).unbind(1)]
    s = (qd @ k.transpose(-1, -2)) * Dh ** -0.5
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(B * Sq, D)
    o.backward(dout.double())
    dq = qd.grad.transpose(1, 2).reshape(B * Sq, D)
    dkv = torch.stack([t.grad.transpose(1, 2).reshape(B * Sk, D) for t in (k, v)], dim=1).reshape(B * Sk, 2 * D)
    return o.detach(), torch.logsumexp(s, -1).detach(), dq, dkv


@pytest.mark.parametrize("Sq,Sk,Dh", [(1, 272, 64), (3, 100, 32), (70, 197, 80), (300, 45, 88), (130, 577, 128), (64, 16, 64)])
@pytest.mark.parametrize("mode", ["portable_fp32", "portable_bf16", "mfma"])
def test_cross_attention_forward_backward(Sq, Sk, Dh, mode):
    """Sq queries over Sk keys per sample (vit/attention.py:269-273 with Lq != Lk): pooling-head sized and decoder sized
    query counts, keys shorter and longer than a 64-key tile, on the portable and on the MFMA kernels against fp64 math."""
    B, H = 2, 3
    D = H * Dh
    dtype = torch.float32 if mode == "portable_fp32" else torch.bfloat16
    algo = 2 if mode == "mfma" else 1
    g = torch.Generator().manual_seed(Sq * 977 + Sk * 31 + Dh)
    q = torch.randn(B * Sq, D, generator=g).to(DEV, dtype)
    kv = torch.randn(B * Sk, 2 * D, generator=g).to(DEV, dtype)
    dout = torch.randn(B * Sq, D, generator=g).to(DEV, dtype)
    o, lse = K.attn_cross_fwd(q, kv, B, Sq, Sk, H, Dh, Dh ** -0.5, algo=algo)
    dq, dkv = K.attn_cross_bwd(q, kv, o, lse, dout, B, Sq, Sk, H, Dh, Dh ** -0.5, algo=algo)
    o_ref, lse_ref, dq_ref, dkv_ref = cross_attn_reference(q, kv, B, Sq, Sk, H, Dh, dout)
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    close(o, o_ref, tol, msg="o")
    close(lse, lse_ref, 1e-4 if dtype == torch.float32 else 1e-3, msg="lse")
    close(dq, dq_ref, tol, msg="dq")
    close(dkv[:, :D], dkv_ref[:, :D], tol, msg="dk")
    close(dkv[:, D:], dkv_ref[:, D:], tol, msg="dv")


def test_cross_attention_dropout_masks_agree_between_kernel_families():
    """Both families draw the keep decision of (query, key) from the same counter, query * (Sk rounded up to even) + key, so with dropout on they
    agree with each other to rounding, forward and backward."""
    B, H, Sq, Sk, Dh, p = 2, 2, 70, 150, 64, 0.3
    D = H * Dh
    g = torch.Generator().manual_seed(77)
    q = torch.randn(B * Sq, D, generator=g).to(DEV, torch.bfloat16)
    kv = torch.randn(B * Sk, 2 * D, generator=g).to(DEV, torch.bfloat16)
    dout = torch.randn(B * Sq, D, generator=g).to(DEV, torch.bfloat16)
    res = []
    for algo in (1, 2):
        o, lse = K.attn_cross_fwd(q, kv, B, Sq, Sk, H, Dh, Dh ** -0.5, drop_p=p, seed=4321, algo=algo)
        dq, dkv = K.attn_cross_bwd(q, kv, o, lse, dout, B, Sq, Sk, H, Dh, Dh ** -0.5, drop_p=p, seed=4321, algo=algo)
        res.append((o, dq, dkv))
    for a, b, name in zip(res[0], res[1], ("o", "dq", "dkv")):
        close(b, a.double(), 2e-2, msg=name)


def test_attention_long_sequence_head_dim_88():
    """The long-sequence shape of ViT-g/14 at 1024 px: S = 5329 (84 key tiles, ragged last tile), head_dim 88 (the 96-wide
    kernel instances, 4 waves per block for S >= 2048), forward and both backward kernels against fp64 math."""
    B, H, S, Dh = 1, 2, 5329, 88
    g = torch.Generator().manual_seed(5329)
    qkv = torch.randn(B * S, 3 * H * Dh, generator=g).to(DEV, torch.bfloat16)
    dout = torch.randn(B * S, H * Dh, generator=g).to(DEV, torch.bfloat16)
    o, lse = K.attn_fwd(qkv, B, S, H, Dh, Dh ** -0.5, algo=2)
    dqkv = K.attn_bwd(qkv, o, lse, dout, B, S, H, Dh, Dh ** -0.5, algo=2)
    o_ref, lse_ref, dqkv_ref = attn_reference(qkv, B, S, H, Dh, dout)
    close(o, o_ref, 2e-2, msg="o")
    close(lse, lse_ref, 1e-3, msg="lse")
    D = H * Dh
    for i, name in enumerate("qkv"):
        close(dqkv[:, i * D:(i + 1) * D], dqkv_ref[:, i * D:(i + 1) * D], 2e-2, msg="d" + name)


def test_attention_online_softmax_rescale_spike():
    """Force the running max to jump at a late key tile (a rare data-dependent branch of online softmax)."""
    B, H, S, Dh = 1, 1, 256, 64
    qkv = torch.randn(B * S, 3 * Dh, device=DEV) * 0.1
    qkv[200, Dh:2 * Dh] = qkv[5, :Dh] * 400.0  # key 200 aligned with query 5: huge score in the 4th tile
    qkv = qkv.to(torch.bfloat16)
    o, lse = K.attn_fwd(qkv, B, S, H, Dh, Dh ** -0.5, algo=2)
    o_ref, lse_ref, _ = attn_reference(qkv, B, S, H, Dh)
    close(o, o_ref, 2e-2)
    close(lse, lse_ref, 1e-3)


@pytest.mark.parametrize("algo", [1, 2])
def test_attention_dropout_statistics_and_fwd_bwd_mask_agreement(algo):
    B, H, S, Dh, p = 2, 2, 128, 64, 0.25
    qkv = torch.zeros(B * S, 3 * H * Dh, device=DEV, dtype=torch.bfloat16)
    qkv[:, 2 * H * Dh:] = 1.0  # v = 1, uniform attention: o = (#kept / S) / (1 - p)
    o, lse = K.attn_fwd(qkv, B, S, H, Dh, Dh ** -0.5, drop_p=p, seed=1234, algo=algo)
    kept = o.float()[:, ::Dh] * (1 - p)  # fraction of kept keys per (row, head)
    assert abs(kept.mean().item() - (1 - p)) < 0.01
    assert kept.std().item() > 0.01
    o2, _ = K.attn_fwd(qkv, B, S, H, Dh, Dh ** -0.5, drop_p=p, seed=1234, algo=algo)
    assert torch.equal(o, o2)
    o3, _ = K.attn_fwd(qkv, B, S, H, Dh, Dh ** -0.5, drop_p=p, seed=99, algo=algo)
    assert not torch.equal(o, o3)
    # dV = P_dropped^T dO: with dO = 1 and uniform P, dv[key] = (#queries keeping key)/S/(1-p); its mean must equal mean(o)
    dout = torch.ones(B * S, H * Dh, device=DEV, dtype=torch.bfloat16)
    dqkv = K.attn_bwd(qkv, o, lse, dout, B, S, H, Dh, Dh ** -0.5, drop_p=p, seed=1234, algo=algo)
    dv = dqkv[:, 2 * H * Dh:].float()
    assert abs(dv.mean().item() - o.float().mean().item()) < 2e-3


def test_attention_dropout_mask_is_the_same_in_both_kernel_families():
    B, H, S, Dh, p = 1, 2, 96, 64, 0.5
    qkv = torch.zeros(B * S, 3 * H * Dh, device=DEV, dtype=torch.bfloat16)
    qkv[:, 2 * H * Dh:] = 1.0
    o1, _ = K.attn_fwd(qkv, B, S, H, Dh, Dh ** -0.5, drop_p=p, seed=7, algo=1)
    o2, _ = K.attn_fwd(qkv, B, S, H, Dh, Dh ** -0.5, drop_p=p, seed=7, algo=2)
    close(o1, o2, 1e-2)


@pytest.mark.parametrize("dims", [(16, 16), (5, 7), (3, 4, 5)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_fourier_features_forward_backward(dims, out_dtype):
    """vitk_fourier_features / _bwd against the reference's op sequence (vit/pos_enc.py:137-149) in torch."""
    import math

    from vit_amd import ops
    from vit_amd.pos_enc import create_grid
    J = 24
    g = torch.Generator().manual_seed(len(dims) * 100 + dims[0])
    grid = create_grid(dims, device=DEV)
    w = torch.randn(len(dims), J, generator=g).to(DEV).requires_grad_(True)
    wr = w.detach().clone().requires_grad_(True)
    f = grid @ wr
    ref = torch.cat([f.sin(), f.cos()], dim=-1) / math.sqrt(J)
    got = ops.FourierFeaturesFn.apply(grid, w, out_dtype)
    assert got.shape == ref.shape and got.dtype == out_dtype
    close(got, ref.to(out_dtype), 1e-5 if out_dtype == torch.float32 else 8e-3)
    dy = torch.randn(ref.shape, generator=g).to(DEV)
    ref.backward(dy.to(out_dtype).float())
    got.backward(dy.to(out_dtype))
    close(w.grad, wr.grad, 1e-4)


def test_copy_samples_moves_only_the_listed_samples():
    """vitk_copy_samples: the drop-path pass-through (samples outside the subset keep their rows, vit/transformer.py:41-46)."""
    B, S, D = 9, 5, 24
    g = torch.Generator().manual_seed(9)
    for dtype in (torch.float32, torch.bfloat16):
        src = torch.randn(B * S, D, generator=g).to(DEV, dtype)
        dst = torch.full_like(src, -7.0)
        idx = torch.randperm(B, generator=g).to(DEV)[6:]          # a slice with a storage offset, like the caller's
        K.copy_samples_(dst, src, idx, S * D)
        want = torch.full_like(src, -7.0).view(B, S * D)
        want[idx] = src.view(B, S * D)[idx]
        assert torch.equal(dst.view(B, S * D), want)
    K.copy_samples_(dst, src, idx[:0], S * D)   # an empty list is a no-op
    with pytest.raises(RuntimeError, match="16-byte"):
        K.copy_samples_(dst.view(-1, 6), src.view(-1, 6), idx, 6)


# ---------------------------------------------------------------------------------------------- elementwise
@pytest.mark.parametrize("Dh", [32, 80, 88, 12])   # 16-byte, 16-byte, 8-byte and scalar accesses in bf16
@pytest.mark.parametrize("dtype,emu", [(torch.float32, False), (torch.bfloat16, False), (torch.bfloat16, True)])
def test_rope_matches_half_split_rotation_and_inverse_is_transpose(dtype, emu, Dh):
    B, S, H, prefix = 2, 19, 3, 4
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(B * S, 3 * H * Dh, generator=g).to(DEV, dtype)
    ang = torch.rand(S - prefix, Dh, generator=g) * 6.28
    table = torch.stack([ang.sin(), ang.cos()]).to(DEV)
    if emu:
        table = table.to(torch.bfloat16).float()
    out = K.rope_qk_(qkv.clone(), table, B, S, H, Dh, prefix, False, emu)
    x = qkv.double().view(B, S, 3, H, Dh)
    ref = x.clone()
    body = x[:, prefix:, :2]
    x1, x2 = body.chunk(2, -1)
    sin, cos = table.double()[0][None, :, None, None, :], table.double()[1][None, :, None, None, :]
    ref[:, prefix:, :2] = body * cos + torch.cat([-x2, x1], -1) * sin
    close(out.view(B, S, 3, H, Dh), ref, 1e-5 if dtype == torch.float32 else 2e-2)
    assert torch.equal(out.view(B, S, 3, H, Dh)[:, :prefix], qkv.view(B, S, 3, H, Dh)[:, :prefix])
    assert torch.equal(out.view(B, S, 3, H, Dh)[:, :, 2], qkv.view(B, S, 3, H, Dh)[:, :, 2])
    # <R x, y> == <x, R^T y>
    y = torch.randn(B * S, 3 * H * Dh, generator=g).to(DEV, torch.float32)
    xf = qkv.float()
    Rx = K.rope_qk_(xf.clone(), table, B, S, H, Dh, prefix, False, False)
    RTy = K.rope_qk_(y.clone(), table, B, S, H, Dh, prefix, True, False)
    assert abs((Rx * y).sum().item() - (xf * RTy).sum().item()) < 1e-2 * abs((Rx * y).sum().item()) + 1e-2


@pytest.mark.parametrize("emu", [False, True])
@pytest.mark.parametrize("algo", [0, 3, 4, 5])
@pytest.mark.parametrize("shape", [(3, 37, 2, 5), (9, 197, 6, 1), (256, 272, 12, 16)])
def test_gemm_rope_epilogue_equals_projection_then_rotation(shape, algo, emu):
    """VITK_EPI_ROPE (head_dim 64): the QKV projection with q and k rotated in its epilogue is bit-identical to the plain
    projection followed by vitk_rope_qk; prefix rows and the v section pass through.  The largest shape is the ViT-B/14
    step's own (a ragged last round: its rows are cut between the 256x256 and the 128x128 kernel)."""
    B, S, H, prefix = shape
    if B * S > 10000 and (algo != 0 or emu):
        pytest.skip("the full-size shape runs once, on the automatic path")
    Dh, D = 64, 64 * H
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B * S, D, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(3 * D, D, generator=g) * D ** -0.5).to(DEV, torch.bfloat16)
    bias = torch.randn(3 * D, generator=g).to(DEV, torch.bfloat16)
    ang = torch.rand(S - prefix, Dh, generator=g) * 6.28
    table = torch.stack([ang.sin(), ang.cos()]).to(DEV)
    if emu:
        table = table.to(torch.bfloat16).float()
    plain = torch.empty(B * S, 3 * D, device=DEV, dtype=torch.bfloat16)
    K.gemm(x, w, trans_a=False, trans_b=False, M=B * S, N=3 * D, K=D, out=plain, bias=bias, algo=algo)
    want = K.rope_qk_(plain.clone(), table, B, S, H, Dh, prefix, False, emu)
    got = torch.empty_like(plain)
    K.gemm(x, w, trans_a=False, trans_b=False, M=B * S, N=3 * D, K=D, out=got, bias=bias, algo=algo, epilogue=L.EPI_ROPE,
           rope=table, rope_width=D, rope_emulate_bf16=emu, seq=S, prefix=prefix)
    assert torch.equal(got, want)
    assert not torch.equal(got, plain)


@pytest.mark.parametrize("gather", [False, True])
@pytest.mark.parametrize("case", [(3000, 768, "float32", 0.0), (517, 1280, "bfloat16", 0.0), (64, 4096, "float32", 0.1), (1, 8, "float32", 0.25),
                                  (2300, 1408, "float32", 0.1)])
def test_scale_cast_with_layer_scale_gradient_equals_the_two_passes(case, gather):
    """vitk_scale_cast_dgamma: the scaled / masked / cast branch gradient is bit-equal to vitk_scale_cast's, and the layer-scale
    gradient equals vitk_colsum of g * aux (fp64 reference for the sum), with and without a row gather."""
    rows, N, dtype, p = case
    dt = getattr(torch, dtype)
    gen = torch.Generator().manual_seed(rows + N)
    src_rows = rows + 37 if gather else rows
    g = torch.randn(src_rows, N, generator=gen).to(dt).to(DEV)
    aux = torch.randn(rows, N, generator=gen).to(torch.bfloat16).to(DEV)
    gamma = (0.1 + torch.rand(N, generator=gen)).to(dt).to(DEV)
    idx = torch.randperm(src_rows, generator=gen)[:rows].to(DEV) if gather else None
    want = K.scale_cast(g, rows, N, torch.bfloat16, row_idx=idx, gamma=gamma, alpha=0.5, drop_p=p, seed=77)
    got, dgamma = K.scale_cast_dgamma(g, aux, rows, N, torch.bfloat16, gamma, row_idx=idx, alpha=0.5, drop_p=p, seed=77)
    assert torch.equal(got, want)
    picked = g if idx is None else g[idx]
    ref = 0.5 * (picked.double() * aux.double()).sum(0)
    assert dgamma.dtype == dt
    close(dgamma.double(), ref, 1e-5 if dt == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
@pytest.mark.parametrize("layernorm", [False, True])
@pytest.mark.parametrize("shape", [(37, 3, 64), (5, 2, 80), (130, 4, 88), (1, 1, 128), (600, 12, 64)])
def test_per_head_qk_norm_backward_in_place(shape, layernorm, dtype):
    """vitk_qk_norm_bwd rewrites the q and k sections of a packed gradient [M, 3, H, Dh] in place, from the gradient of the
    normalised heads to the gradient of the heads, and sums the weight / bias gradients (vit/attention.py:25-41 through
    autograd, in fp64); the v section is left alone."""
    M, H, Dh = shape
    dt = getattr(torch, dtype)
    D = H * Dh
    g = torch.Generator(device="cpu").manual_seed(M * 131 + Dh)
    dqkv = torch.randn(M, 3 * D, generator=g).to(dt).cuda()
    pre = [(torch.randn(M * H, Dh, generator=g) * 1.5 + 0.3).to(dt).cuda() for _ in range(2)]
    w = [(1 + 0.2 * torch.randn(Dh, generator=g)).cuda() for _ in range(2)]
    eps = 1e-5
    want_sections, want_dw, want_db, stats = [], [], [], []
    for which in range(2):
        x = pre[which].double().requires_grad_(True)
        wd = w[which].double().requires_grad_(True)
        bd = torch.zeros(Dh, dtype=torch.float64, device="cuda", requires_grad=True)
        if layernorm:
            y = torch.nn.functional.layer_norm(x, (Dh,), wd, bd, eps)
            mean = x.detach().mean(-1)
            rstd = (x.detach().var(-1, unbiased=False) + eps).rsqrt()
        else:
            y = torch.nn.functional.rms_norm(x, (Dh,), wd, eps)
            mean = None
            rstd = (x.detach().square().mean(-1) + eps).rsqrt()
        dpost = dqkv[:, which * D:(which + 1) * D].reshape(M * H, Dh).double()
        y.backward(dpost)
        want_sections.append(x.grad.view(M, D))
        want_dw.append(wd.grad)
        want_db.append(bd.grad if layernorm else None)
        stats.append((rstd.float(), None if mean is None else mean.float()))
    got = dqkv.clone()
    outs = K.qk_norm_bwd_(got, pre[0], pre[1], stats[0][0], stats[1][0], stats[0][1], stats[1][1], w[0], w[1], H, Dh, layernorm,
                          (True, layernorm, True, layernorm))
    rtol = 1e-4 if dt == torch.float32 else 2e-2
    for which in range(2):
        close(got[:, which * D:(which + 1) * D].double(), want_sections[which], rtol, msg=f"section {which}")
        close(outs[2 * which].double(), want_dw[which], rtol if dt == torch.float32 else 1e-2, msg=f"dw {which}")
        if layernorm:
            close(outs[2 * which + 1].double(), want_db[which], rtol if dt == torch.float32 else 1e-2, msg=f"db {which}")
        else:
            assert outs[2 * which + 1] is None
    assert torch.equal(got[:, 2 * D:], dqkv[:, 2 * D:])
    # frozen norm parameters: no workspace, same rows
    again = dqkv.clone()
    none = K.qk_norm_bwd_(again, pre[0], pre[1], stats[0][0], stats[1][0], stats[0][1], stats[1][1], w[0], w[1], H, Dh, layernorm,
                          (False, False, False, False))
    assert none == (None, None, None, None) and torch.equal(again, got)


@pytest.mark.parametrize("with_rope", [False, True])
@pytest.mark.parametrize("layernorm", [False, True])
@pytest.mark.parametrize("shape", [(3, 37, 2, 5), (40, 197, 6, 1), (256, 272, 12, 16)])
def test_gemm_epilogue_with_per_head_qk_normalisation(shape, layernorm, with_rope):
    """The QKV projection with q and k normalised per head (and then rotated) in its epilogue is bit-identical to the
    plain projection, vitk_norm_fwd on the heads copied out as [M * H, 64] rows, and vitk_rope_qk; the un-normalised heads
    and the row statistics it leaves for the backward pass equal what that path saves."""
    B, S, H, prefix = shape
    if B * S > 10000 and not (layernorm and with_rope):
        pytest.skip("the full-size shape runs once")
    Dh, D, M = 64, 64 * H, B * S
    g = torch.Generator().manual_seed(21)
    x = torch.randn(M, D, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(3 * D, D, generator=g) * D ** -0.5).to(DEV, torch.bfloat16)
    bias = torch.randn(3 * D, generator=g).to(DEV, torch.bfloat16)
    nw = (1 + 0.2 * torch.randn(2, Dh, generator=g)).to(DEV)
    nb = (0.1 * torch.randn(2, Dh, generator=g)).to(DEV) if layernorm else None
    ang = torch.rand(S - prefix, Dh, generator=g) * 6.28
    table = torch.stack([ang.sin(), ang.cos()]).to(DEV)
    eps = 1e-5
    want = torch.empty(M, 3 * D, device=DEV, dtype=torch.bfloat16)
    K.gemm(x, w, trans_a=False, trans_b=False, M=M, N=3 * D, K=D, out=want, bias=bias)
    saved = []
    for which in range(2):
        pre = want[:, which * D:(which + 1) * D].contiguous().view(M * H, Dh)
        nrm, r_, m_ = K.norm_fwd(pre, nw[which].contiguous(), None if nb is None else nb[which].contiguous(), eps, layernorm,
                                 torch.bfloat16)
        want[:, which * D:(which + 1) * D].copy_(nrm.view(M, D))
        saved.append((pre, r_, m_))
    if with_rope:
        K.rope_qk_(want, table, B, S, H, Dh, prefix, False, False)
    got = torch.empty_like(want)
    pre = torch.empty(2, M, D, device=DEV, dtype=torch.bfloat16)
    rstd = torch.empty(2, M * H, device=DEV)
    mean = torch.empty(2, M * H, device=DEV) if layernorm else None
    K.gemm(x, w, trans_a=False, trans_b=False, M=M, N=3 * D, K=D, out=got, bias=bias, epilogue=L.EPI_ROPE,
           rope=table if with_rope else None, rope_width=D, seq=S, prefix=prefix if with_rope else 0, aux=pre,
           qk_norm=dict(weight=nw.contiguous(), bias=nb, rstd=rstd, mean=mean, eps=eps, layernorm=layernorm))
    for which in range(2):
        assert torch.equal(pre[which].view(M * H, Dh), saved[which][0]), which
        assert torch.equal(rstd[which], saved[which][1]), which
        if layernorm:
            assert torch.equal(mean[which], saved[which][2]), which
    assert torch.equal(got, want)


@pytest.mark.parametrize("emu", [False, True])
@pytest.mark.parametrize("S,prefix", [(64, 0), (197, 1), (272, 16), (600, 24)])
def test_attention_backward_with_fused_inverse_rotation(S, prefix, emu):
    """vitk_attn_bwd with a rope table (head_dim 64) is bit-identical to the plain backward followed by
    vitk_rope_qk(inverse) on dq and dk; dv is untouched."""
    B, H, Dh = 2, 3, 64
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(B * S, 3 * H * Dh, generator=g).to(DEV, torch.bfloat16)
    dout = torch.randn(B * S, H * Dh, generator=g).to(DEV, torch.bfloat16)
    ang = torch.rand(S - prefix, Dh, generator=g) * 6.28
    table = torch.stack([ang.sin(), ang.cos()]).to(DEV)
    if emu:
        table = table.to(torch.bfloat16).float()
    o, lse = K.attn_fwd(qkv, B, S, H, Dh, Dh ** -0.5, algo=2)
    plain = K.attn_bwd(qkv, o, lse, dout, B, S, H, Dh, Dh ** -0.5, algo=2)
    want = K.rope_qk_(plain.clone(), table, B, S, H, Dh, prefix, True, emu)
    got = K.attn_bwd(qkv, o, lse, dout, B, S, H, Dh, Dh ** -0.5, algo=2, rope=table, rope_emulate_bf16=emu)
    assert torch.equal(got, want)
    assert not torch.equal(got, plain)


ACTS = {"relu": torch.relu, "silu": torch.nn.functional.silu, "gelu": torch.nn.functional.gelu,
        "srelu": lambda t: t.relu().pow(2), "openswiglu": lambda t: t * torch.sigmoid(1.702 * t)}


@pytest.mark.parametrize("name", list(ACTS))
@pytest.mark.parametrize("glu", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_activation_forward_backward(name, glu, dtype):
    rows, F = 33, 64
    g = torch.Generator().manual_seed(len(name) + glu)
    p = torch.randn(rows, (2 if glu else 1) * F, generator=g).to(DEV, dtype)
    dh = torch.randn(rows, F, generator=g).to(DEV, dtype)
    limit, extra = (1.0, 0.5) if glu else (None, None)
    code = L.ACT_CODES[name]
    h = K.act_fwd(p, F, glu, code, limit, extra, 0.0, 0)
    dp = K.act_bwd(dh, p, F, glu, code, limit, extra, 0.0, 0)
    pd = p.double().requires_grad_(True)
    if glu:
        lin, gl = pd.chunk(2, -1)
        ref = ACTS[name](gl.clamp(max=limit)) * (lin.clamp(-limit, limit) + extra)
    else:
        ref = ACTS[name](pd)
    ref.backward(dh.double())
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    close(h, ref, tol)
    close(dp, pd.grad, tol)


def test_activation_dropout_is_inverted_and_replayed_in_backward():
    rows, F, p = 64, 256, 0.3
    x = torch.ones(rows, F, device=DEV)
    h = K.act_fwd(x, F, False, L.ACT_CODES["relu"], None, None, p, 42)
    keep = h != 0
    assert abs(keep.float().mean().item() - (1 - p)) < 0.02
    close(h[keep], torch.full_like(h[keep], 1 / (1 - p)), 1e-6)
    dp = K.act_bwd(torch.ones_like(x), x, F, False, L.ACT_CODES["relu"], None, None, p, 42)
    assert torch.equal(dp != 0, keep)


def test_im2col_col2im_roundtrip_and_conv_equivalence():
    B, C, H, W, ph, pw, D = 2, 3, 30, 45, 7, 14, 16  # H, W not multiples of the patch: trailing pixels are dropped
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, C, H, W, generator=g).to(DEV)
    w = torch.randn(D, C, ph, pw, generator=g).to(DEV)
    Kc = C * ph * pw
    Kp = (Kc + 63) // 64 * 64
    cols = K.im2col(x, ph, pw, Kp, torch.float32)
    assert torch.all(cols[:, Kc:] == 0)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=(ph, pw)).flatten(2).transpose(1, 2)
    got = cols[:, :Kc].double() @ w.double().reshape(D, Kc).t()
    close(got.view(B, -1, D), ref, 1e-5)
    back = K.col2im(cols, (B, C, H, W), ph, pw, torch.float32)
    Hc, Wc = H // ph * ph, W // pw * pw
    assert torch.equal(back[:, :, :Hc, :Wc], x[:, :, :Hc, :Wc])
    assert torch.all(back[:, :, Hc:] == 0) and torch.all(back[:, :, :, Wc:] == 0)


def test_colsum_and_scale_cast():
    R, N = 3000, 200
    g = torch.Generator().manual_seed(9)
    a = torch.randn(R + 50, N, generator=g).to(DEV, torch.bfloat16)
    b = torch.randn(R, N, generator=g).to(DEV, torch.bfloat16)
    idx = torch.randperm(R + 50, generator=g)[:R].to(DEV)
    out = K.colsum(a, R, N, N, torch.float32, a_idx=idx, b=b, ldb=N, scale=0.5)
    close(out, 0.5 * (a.double()[idx] * b.double()).sum(0), 1e-4)
    out2 = K.colsum(a, R + 50, N, N, torch.float32)
    close(out2, a.double().sum(0), 1e-4)
    gam = torch.randn(N, generator=g).to(DEV)
    src = torch.randn(R + 50, N, generator=g).to(DEV)
    sc = K.scale_cast(src, R, N, torch.bfloat16, row_idx=idx, gamma=gam, alpha=1.25)
    close(sc, src[idx] * gam * 1.25, 1e-2)


@pytest.mark.parametrize("glu", [False, True])
@pytest.mark.parametrize("name", ["silu", "srelu", "gelu"])
@pytest.mark.parametrize("drop", [0.0, 0.2])
def test_fused_activation_epilogues_equal_the_unfused_kernels(glu, name, drop):
    """fc1 GEMM + activation (GLU pairs columns f and F+f inside one wave) and fc2 dgrad + activation backward,
    fused in the MFMA epilogue, must reproduce GEMM -> vitk_act_fwd / vitk_act_bwd (same pre-activations bit for bit, same dropout mask)."""
    M, D, F = 300, 192, 160
    F1 = 2 * F if glu else F
    g = torch.Generator().manual_seed(17)
    x = torch.randn(M, D, generator=g).to(DEV, torch.bfloat16)
    w1 = (torch.randn(F1, D, generator=g) * 0.1).to(DEV, torch.bfloat16)
    b1 = torch.randn(F1, generator=g).to(DEV, torch.bfloat16)
    w2 = (torch.randn(D, F, generator=g) * 0.1).to(DEV, torch.bfloat16)
    dy = torch.randn(M, D, generator=g).to(DEV, torch.bfloat16)
    code = L.ACT_CODES[name]
    limit, extra = (1.5, 0.25) if glu else (None, None)
    # unfused
    p_ref = torch.empty(M, F1, device=DEV, dtype=torch.bfloat16)
    K.gemm(x, w1, trans_a=False, trans_b=False, M=M, N=F1, K=D, out=p_ref, bias=b1, algo=2)
    h_ref = K.act_fwd(p_ref, F, glu, code, limit, extra, drop, 77)
    dh_ref = torch.empty(M, F, device=DEV, dtype=torch.bfloat16)
    K.gemm(dy, w2, trans_a=False, trans_b=True, M=M, N=F, K=D, out=dh_ref, algo=2)
    dp_ref = K.act_bwd(dh_ref, p_ref, F, glu, code, limit, extra, drop, 77)
    # fused
    p = torch.empty_like(p_ref)
    h = torch.empty(M, F, device=DEV, dtype=torch.bfloat16)
    K.gemm(x, w1, trans_a=False, trans_b=False, M=M, N=F1, K=D, out=h, bias=b1, aux=p,
           epilogue=L.EPI_GLU if glu else L.EPI_ACT, act=code, limit=limit, extra=extra, drop_p=drop, seed=77, algo=2)
    dp = torch.empty_like(p_ref)
    K.gemm(dy, w2, trans_a=False, trans_b=True, M=M, N=F, K=D, out=dp, aux=p_ref,
           epilogue=L.EPI_DGLU if glu else L.EPI_DACT, act=code, limit=limit, extra=extra, drop_p=drop, seed=77, ldo=F1,
           algo=2)
    assert torch.equal(p, p_ref)
    # same formulas and the same dropout mask; the two compilations may contract FMAs differently: allow 1 bf16 ulp
    assert torch.equal(h == 0, h_ref == 0) or drop == 0.0
    close(h, h_ref, 8e-3, 1e-3)
    close(dp, dp_ref, 8e-3, 1e-3)


@pytest.mark.parametrize("tag", ["fp32", "bf16"])
def test_standalone_apply_rope_matches_reference_vectors(tag):
    """vit_amd.rope.apply_rope on [B, H, S, Dh] against the reference's own apply_rope outputs (g07_ops), and its
    backward against autograd through the oracle's restatement."""
    from conftest import load_golden
    from oracle import vit_oracle as O
    from vit_amd.rope import apply_rope

    fix = load_golden("g07_ops")
    q, table = fix[f"rope_q_{tag}"].to(DEV), fix[f"rope_table_{tag}"].to(DEV)
    out = apply_rope(q, table)
    assert out.dtype == q.dtype
    if tag == "bf16":   # every product and the sum rounded like eager bf16 arithmetic: the same bits
        assert torch.equal(out.cpu(), fix[f"rope_out_{tag}"])
        mixed = apply_rope(q, fix["rope_table_fp32"].to(DEV))
        torch.testing.assert_close(mixed.cpu().float(), fix["rope_out_bf16_x_fp32_table"].float(), rtol=2e-2, atol=2e-2)
    else:               # fp32: the kernel contracts x1*c - x2*s into an fma
        torch.testing.assert_close(out.cpu(), fix[f"rope_out_{tag}"], rtol=1e-5, atol=1e-6)
    x = q.float().clone().requires_grad_(True)
    t32 = fix["rope_table_fp32"].to(DEV)
    g = torch.randn_like(x)
    (apply_rope(x, t32) * g).sum().backward()
    xr = q.float().clone().requires_grad_(True)
    (O.apply_rope(xr, t32) * g).sum().backward()
    torch.testing.assert_close(x.grad, xr.grad, rtol=1e-5, atol=1e-6)
    per = torch.stack([t32, t32.flip(1)], 1)[:, :, None]   # [2, B=2, 1, N, Dh]
    torch.testing.assert_close(apply_rope(q.float(), per), O.apply_rope(q.float(), per), rtol=1e-5, atol=1e-6)


def test_gemm_dynamic_tile_shares_match_fixed_ones():
    """VitkGemm.dynamic_tiles = 1: the blocks of the 256x256-tile kernels draw tiles from per-XCD counters kept in the
    workspace.  Same exact products as with fixed shares, on launches with fewer, as many and more tiles than CUs, several
    launches in a row (each call zeroes its counters), with split-K, and with idle workgroups on a side stream holding
    CUs meanwhile."""
    from vit_amd import _lib as L

    lib = L.load()
    g = torch.Generator().manual_seed(0)
    cases = [(512, 512, 256, "nt"), (4096, 4096, 128, "nn"), (8192, 4608, 192, "nt"), (768, 768, 16384, "tn")]
    old = K.set_dynamic_tiles(True)
    try:
        side = torch.cuda.Stream()
        for rep in range(3):
            if rep == 2:
                with torch.cuda.stream(side):
                    L.check(lib.vitk_debug_hold_cus(48, 20_000_000, side.cuda_stream))
            for M, N, Kd, layout in cases:
                A = torch.randint(-2, 3, (M, Kd), generator=g).float()
                B = torch.randint(-2, 3, (Kd, N), generator=g).float()
                ref = (A @ B).to(DEV)
                a = (A if layout != "tn" else A.t()).contiguous().to(DEV, torch.bfloat16)
                b = (B.t() if layout == "nt" else B).contiguous().to(DEV, torch.bfloat16)
                out = torch.empty(M, N, device=DEV, dtype=torch.float32)
                K.gemm(a, b, trans_a=layout == "tn", trans_b=layout != "nt", M=M, N=N, K=Kd, out=out, algo=5)
                assert torch.equal(out, ref), f"rep {rep} {M}x{N}x{Kd} {layout}: max err {(out - ref).abs().max().item()}"
        torch.cuda.synchronize()
    finally:
        K.set_dynamic_tiles(old)


def test_tensor_on_another_device_than_the_current_one_is_refused():
    """Launches use the current device's stream: a tensor that lives elsewhere must be refused, not launched on."""
    if torch.cuda.device_count() < 2:
        # one visible GPU: the same rule is exercised through a CPU tensor, which takes the same check
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            L.require_gpu(torch.zeros(4))
        return
    x = torch.zeros(8, 64, device="cuda:1")
    with pytest.raises(RuntimeError, match="current device"):
        L.require_gpu(x)
    with torch.cuda.device(1):
        L.require_gpu(x)
