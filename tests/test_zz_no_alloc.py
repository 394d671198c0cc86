"""The allocation-free column reduction (vitk_colsum_static: partial sums in a device array built into the library)
against vitk_colsum with a caller workspace, through the C ABI.  Same kernels, same order of additions: bit-equal.
Two calls in a row, the second smaller, so stale partials of the first call would show."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from vit_amd import _lib as L
    from vit_amd import kernels as K

DEV = "cuda"


def _both(a, rows, N, lda, out_dtype, a_idx=None, b=None, scale=1.0):
    lib = L.load()
    ref = K.colsum(a, rows, N, lda, out_dtype, a_idx=a_idx, b=b, ldb=0 if b is None else b.stride(0), scale=scale)
    out = torch.full((N,), float("nan"), dtype=out_dtype, device=a.device)
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    L.check(lib.vitk_colsum_static(L.ptr(a), L.dt(a), lda, L.ptr(a_idx), L.ptr(b), L.dt(b) if b is not None else 0,
                                   0 if b is None else b.stride(0), L.ptr(out), L.dt(out), rows, N, scale, L.stream()))
    assert torch.cuda.memory_allocated() == before
    return out, ref


def test_colsum_static_matches_workspace_form_twice():
    g = torch.Generator().manual_seed(7)
    a1 = torch.randn(40000, 3072, generator=g).to(DEV, torch.bfloat16)
    b1 = torch.randn(40000, 3072, generator=g).to(DEV, torch.bfloat16)
    out, ref = _both(a1, 40000, 3072, 3072, torch.float32, b=b1, scale=0.5)
    assert torch.equal(out, ref)
    close_to = (a1.double() * b1.double()).sum(0) * 0.5
    torch.testing.assert_close(out.double(), close_to, rtol=1e-4, atol=1e-4 * close_to.abs().max().item())
    # second call, smaller in rows and columns, with a row gather: must not see the first call's partial sums
    a2 = torch.randn(900, 768, generator=g).to(DEV)
    idx = torch.randperm(900, generator=g)[:517].to(DEV)
    out2, ref2 = _both(a2, 517, 768, 768, torch.bfloat16, a_idx=idx)
    assert torch.equal(out2, ref2)
    torch.testing.assert_close(out2.double(), a2[idx].double().sum(0), rtol=1e-2, atol=1e-2 * ref2.float().abs().max().item())


def test_colsum_static_rejects_wide_inputs():
    a = torch.zeros(8, 8196, device=DEV)
    out = torch.empty(8196, device=DEV)
    rc = L.load().vitk_colsum_static(L.ptr(a), L.dt(a), 8196, None, None, 0, 0, L.ptr(out), L.dt(out), 8, 8196, 1.0, L.stream())
    assert rc != 0 and b"8192" in L.load().vitk_last_error()
