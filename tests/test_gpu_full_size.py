"""Size-independent properties at the sizes the benchmark configurations name (2-D 8192^2,
3-D 512^3, one 1024 x 1024 x 128 slab, the whole 1024^3 grid on one GPU, 4-D 128^4 in Morton
order), where the CPU oracle is out of reach: exact boundary counts, termination of CG on a
right-hand side with m eigenmodes after m iterations, agreement of the recurrence residual with
b - A x, and agreement of the two layouts with each other."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def stencil(lib, u, ext_L, d, mode=0):
    N = u.numel()
    out = torch.empty_like(u)
    lib.laplace_gpu_jl(out.data_ptr(), u.data_ptr(), 0.0, d, ext_L, N, mode, (N + 255) // 256, 256)
    return out


@pytest.mark.parametrize("L,d,mode", [(8192, 2, 0), (512, 3, 0), (1024, 3, 0), (128, 4, 0), (128, 4, 2)])
def test_stencil_of_ones_counts_boundary_faces(lib, L, d, mode):
    """(A 1)_i = number of neighbours of i outside the grid, so the sum is 2 d L^(d-1) exactly
    (every addend and the total are small integers: exact in float and double)."""
    N = L ** d
    u = torch.ones(N, dtype=torch.float32, device="cuda")
    out = stencil(lib, u, L, d, mode)
    assert out.sum(dtype=torch.float64).item() == 2 * d * L ** (d - 1)
    assert out.max().item() == d and out.min().item() == 0
    # corners: the first and the last point have all d lower / upper neighbours missing in either layout
    assert out[0].item() == d and out[N - 1].item() == d


@pytest.mark.parametrize("L,d", [(8192, 2), (512, 3)])
def test_eigenmode_rhs_full_size(lib, L, d):
    """b = three mid-spectrum eigenmodes of the Dirichlet Laplacian: three distinct eigenvalues,
    so CG is done after three iterations (a fourth for the lagged stop test)."""
    N = L ** d
    ax = torch.arange(1, L + 1, device="cuda", dtype=torch.float64) * (np.pi / (L + 1))
    b = torch.zeros([L] * d, device="cuda", dtype=torch.float64)
    for j in (L // 4, L // 2, 3 * L // 4):
        s = torch.sin(j * ax)
        m = s
        for _ in range(d - 1):
            m = torch.tensordot(m.unsqueeze(-1), s.unsqueeze(0), dims=1)
        b += m
    bf = b.reshape(-1).float().contiguous()
    del b
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    res, iters, hist = lib.conjugate_gradient_gpu_ex(bf.data_ptr(), x.data_ptr(), L, d, hist_len=200)
    assert iters == 4, (iters, hist)
    Ax = stencil(lib, x, L, d)
    assert (Ax - bf).norm().item() <= 1e-5 * bf.norm().item()


@pytest.mark.parametrize("ext", [[8192, 8192], [512, 512, 512], [1024, 1024, 128], [1024, 1024, 1024]])
def test_recurrence_residual_equals_true_residual_full_size(lib, ext):
    """After K iterations r (updated by recurrence inside kernel B) equals b - A x recomputed from
    the solver's x with torch in double precision."""
    d = len(ext)
    N = int(np.prod(ext))
    b = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(b.data_ptr(), N, 99)
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    s = lib.make_solver(ext)
    s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
    s.iterate(25)
    iters, res, hist = s.finish(hist_len=32)
    assert iters == 25 and all(np.isfinite(hist))
    r_ptr = lib.c.cg_solver_r(s.h)
    rec = float(np.sqrt(lib.inner_product_gpu(r_ptr, r_ptr, N)))
    # true residual, plane by plane along the last axis to bound the temporaries
    shape = ext[::-1]                                  # axis 0 of the grid is contiguous
    xg = x.reshape(shape)
    bg = b.reshape(shape)
    acc = 0.0
    nlast = shape[0]
    step = max(1, (1 << 26) // (N // nlast))
    for z0 in range(0, nlast, step):
        z1 = min(nlast, z0 + step)
        lo, hi = max(0, z0 - 1), min(nlast, z1 + 1)
        t = xg[lo:hi].double()
        a = 2 * d * t
        for axis in range(d):
            n = t.shape[axis]
            up = torch.zeros_like(t)
            dn = torch.zeros_like(t)
            sl_a = [slice(None)] * d
            sl_b = [slice(None)] * d
            sl_a[axis], sl_b[axis] = slice(0, n - 1), slice(1, n)
            up[tuple(sl_a)] = t[tuple(sl_b)]
            dn[tuple(sl_b)] = t[tuple(sl_a)]
            a -= up + dn
        a = a[z0 - lo:z0 - lo + (z1 - z0)]
        acc += float(((bg[z0:z1].double() - a) ** 2).sum().item())
    true_res = float(np.sqrt(acc))
    assert abs(true_res - rec) <= 1e-3 * rec, (true_res, rec)
    s.close()


@pytest.mark.parametrize("relayout", ["0", "1"])
def test_morton_128_4_equals_lexicographic(lib, monkeypatch, relayout):
    """The 4-D 128^4 configuration in both layouts on the same function of the coordinates:
    identical stencil values point by point, and the same CG history for 10 iterations, with the
    Z-curve kernels (relayout 0) and with the solver's internal lexicographic order (the default)."""
    monkeypatch.setenv("CG_MORTON_RELAYOUT", relayout)
    L, d = 128, 4
    N = L ** d
    idx = torch.arange(N, dtype=torch.int64, device="cuda")
    # coordinates of Z-curve index i (interleave.cuh:110-118: coordinate k owns bits k, k+d, ...)
    coords = []
    for k in range(d):
        c = torch.zeros(N, dtype=torch.int64, device="cuda")
        for bit in range(7):
            c |= ((idx >> (bit * d + k)) & 1) << bit
        coords.append(c)
    lex_of_morton = coords[0] + L * (coords[1] + L * (coords[2] + L * coords[3]))
    del coords, idx
    u_lex = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(u_lex.data_ptr(), N, 5)
    u_mor = u_lex[lex_of_morton].contiguous()
    a_lex = stencil(lib, u_lex, L, d, 0)
    a_mor = stencil(lib, u_mor, L, d, 2)
    assert torch.equal(a_mor, a_lex[lex_of_morton])
    del a_lex, a_mor
    xl = torch.zeros(N, dtype=torch.float32, device="cuda")
    xm = torch.zeros(N, dtype=torch.float32, device="cuda")
    _, itl, hl = lib.conjugate_gradient_gpu_ex(u_lex.data_ptr(), xl.data_ptr(), L, d, index_mode=0, maxit=10, hist_len=10)
    _, itm, hm = lib.conjugate_gradient_gpu_ex(u_mor.data_ptr(), xm.data_ptr(), L, d, index_mode=2, maxit=10, hist_len=10)
    assert itl == itm == 10
    assert np.allclose(hl, hm, rtol=1e-4)
    diff = (xm - xl[lex_of_morton]).norm().item()
    assert diff <= 1e-4 * xl.norm().item()


def test_mixed_precision_at_its_configuration_size(lib):
    """Config "mixed precision, 3-D 512^3" (conjugate-gradient_cpu.cu:334-392 with the float GPU CG as the
    inner solve): b = A x_true in double, from x0 = 0 the outer loop must reach its 1e-8 |b| tolerance in a
    handful of outer iterations (bounded here at 12), every inner solve must take some float iterations,
    and the fp64 residual recomputed with torch must be below 1e-8 -- a level the float solver alone
    cannot reach on this grid."""
    L, d = 512, 3

    def lap(t):
        out = 2 * d * t.clone()
        for ax in range(d):
            lo = [slice(None)] * d
            hi = [slice(None)] * d
            lo[ax], hi[ax] = slice(0, L - 1), slice(1, L)
            out[tuple(lo)] -= t[tuple(hi)]
            out[tuple(hi)] -= t[tuple(lo)]
        return out

    gen = torch.Generator(device="cuda").manual_seed(7)
    xt = torch.rand([L] * d, dtype=torch.float64, device="cuda", generator=gen)
    b = lap(xt).reshape(-1).contiguous()
    x = torch.zeros_like(b)
    outer, inner = lib.preconditioned_cg_gpu(b.data_ptr(), x.data_ptr(), L, d)
    assert 1 <= outer <= 12, (outer, inner)
    assert len(inner) == outer and all(v > 0 for v in inner), (outer, inner)
    res = ((b - lap(x.reshape([L] * d)).reshape(-1)).norm() / b.norm()).item()
    assert res < 1e-8, res
    assert (x - xt.reshape(-1)).abs().max().item() < 1e-5
