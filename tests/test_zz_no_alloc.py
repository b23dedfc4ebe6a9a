"""cg_inner_product_noalloc (no counterpart in the reference): the dot product of
inner_product_gpu (conjugate-gradient_gpu.cu:239-255) for callers that cannot allocate.  Checked
through the C-ABI against inner_product_gpu itself -- same kernels and launch shape, so the float
results must be the same bits -- on two calls in a row, the second on a smaller input (the static
scratch still holds the first call's partial sums)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _as_bits(v):
    return np.float32(v).view(np.uint32)


def test_noalloc_inner_product_matches_inner_product_gpu(lib):
    assert lib.c.cg_noalloc_prepare() == 0
    g = torch.Generator(device="cuda").manual_seed(7)
    n1, n2 = (1 << 22) + 13, 70001   # 2048 (the launch cap) and 70 workgroups: the second call leaves most old partials behind
    v = torch.rand(n1, device="cuda", generator=g) - 0.5
    w = torch.rand(n1, device="cuda", generator=g) - 0.5
    out = torch.full((2,), float("nan"), device="cuda")
    torch.cuda.synchronize()
    for k, n in enumerate((n1, n2)):
        want = lib.inner_product_gpu(v.data_ptr(), w.data_ptr(), n)
        rc = lib.inner_product_noalloc(v.data_ptr(), w.data_ptr(), n, out.data_ptr() + 4 * k)
        assert rc == 0
        torch.cuda.synchronize()
        got = float(out[k].item())
        assert _as_bits(got) == _as_bits(want), (n, got, want)
        ref = float((v[:n].double() * w[:n].double()).sum().item())
        scale = float((v[:n].double() * w[:n].double()).abs().sum().item())
        assert abs(got - ref) <= 1e-5 * scale
    # v.v through the single-vector path, on a side stream, after the larger call
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    want = lib.inner_product_gpu(v.data_ptr(), v.data_ptr(), n2)
    rc = lib.inner_product_noalloc(v.data_ptr(), v.data_ptr(), n2, out.data_ptr(), s.cuda_stream)
    assert rc == 0
    s.synchronize()
    assert _as_bits(float(out[0].item())) == _as_bits(want)


def test_noalloc_rejects_bad_arguments(lib):
    out = torch.zeros(1, device="cuda")
    v = torch.ones(16, device="cuda")
    assert lib.inner_product_noalloc(v.data_ptr(), v.data_ptr(), 0, out.data_ptr()) != 0
    assert lib.inner_product_noalloc(v.data_ptr(), v.data_ptr(), 1 << 32, out.data_ptr()) != 0
    assert lib.inner_product_noalloc(0, v.data_ptr(), 16, out.data_ptr()) != 0
    assert lib.inner_product_noalloc(v.data_ptr(), v.data_ptr(), 16, out.data_ptr()) == 0
    torch.cuda.synchronize()
    assert float(out.item()) == 16.0
