"""Edges of the entry points' domains: empty and tiny vectors, unaligned pointers, a zero right-hand side,
one-point grids, and the largest sizes the reference's types allow (`unsigned N` of inner_product_gpu,
conjugate-gradient_gpu.cu:239; `int N = pow(L, d)` of the solver, :330)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def stencil(lib, u, L, d, mode=0):
    N = u.numel()
    out = torch.empty_like(u)
    lib.laplace_gpu_jl(out.data_ptr(), u.data_ptr(), 0.0, d, L, N, mode, (N + 255) // 256, 256)
    return out


# ---- inner product ---------------------------------------------------------------------------------

def test_inner_product_of_nothing_is_zero(lib):
    v = torch.ones(4, dtype=torch.float32, device="cuda")
    assert lib.inner_product_gpu(v.data_ptr(), v.data_ptr(), 0) == 0.0


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 63, 65, 255, 257])
@pytest.mark.parametrize("off_v,off_w", [(0, 0), (1, 0), (0, 3), (1, 1), (2, 3)])
def test_inner_product_tiny_and_unaligned(lib, n, off_v, off_w):
    """Small integers: every product and partial sum is exact in float, whatever the order.  The offsets
    move the vectors off the 16-byte boundary the vector loads want."""
    rng = np.random.default_rng(n * 16 + off_v * 4 + off_w)
    v = rng.integers(-8, 9, n + 4).astype(np.float32)
    w = rng.integers(-8, 9, n + 4).astype(np.float32)
    vd, wd = dev(v), dev(w)
    got = lib.inner_product_gpu(vd.data_ptr() + 4 * off_v, wd.data_ptr() + 4 * off_w, n)
    assert got == float(np.dot(v[off_v:off_v + n].astype(np.float64), w[off_w:off_w + n].astype(np.float64)))


def test_inner_product_of_the_longest_vector_an_unsigned_length_allows(lib):
    n = (1 << 32) - 1
    v = torch.ones(n, dtype=torch.float32, device="cuda")
    got = lib.inner_product_gpu(v.data_ptr(), v.data_ptr(), n)
    assert got == float(np.float32(n))          # 4294967295 rounds to 2^32 in float
    v[n - 1] = 3.0                               # the very last element is part of the sum
    v[0] = 2.0
    got = lib.inner_product_gpu(v.data_ptr(), v.data_ptr(), n)
    # double partials: (n - 2) + 4 + 9 = n + 11 exactly, then one rounding to float
    assert got == float(np.float32(n + 11))


# ---- solver ----------------------------------------------------------------------------------------

@pytest.mark.parametrize("L,d", [(2, 1), (3, 2), (40, 2), (1024, 2), (128, 3), (256, 3), (64, 4)])
def test_zero_right_hand_side_stops_before_the_first_iteration(lib, L, d):
    """reltol = 1e-6 |b| = 0 and residue = |r| = 0: `residue > reltol` is false at once
    (conjugate-gradient_gpu.cu:347-349); x keeps its values."""
    N = L ** d
    b = torch.zeros(N, dtype=torch.float32, device="cuda")
    x = torch.full((N,), 0.25, dtype=torch.float32, device="cuda")
    res, iters, _ = lib.conjugate_gradient_gpu_ex(b.data_ptr(), x.data_ptr(), L, d)
    assert iters == 0 and res == 0.0
    assert torch.equal(x, torch.full_like(x, 0.25))
    assert lib.conjugate_gradient_gpu(b.data_ptr(), x.data_ptr(), L, d) == 0.0


@pytest.mark.parametrize("d", [1, 2, 3, 5])
def test_grid_of_a_single_point(lib, d):
    """L = 1: every neighbour is outside the grid, A = 2d."""
    u = torch.tensor([3.0], dtype=torch.float32, device="cuda")
    assert stencil(lib, u, 1, d).item() == 6.0 * d


def test_solving_a_single_point_grid_trips_the_reference_assertion():
    """The reference asserts N > nthreads = 1 before it solves (conjugate-gradient_gpu.cu:331): so does the
    library -- an abort of the process, checked in a child."""
    import os
    import signal
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import torch; from __graft_entry__ import load_package; "
            "lib = load_package().load_library(); b = torch.ones(1, device='cuda'); x = torch.zeros(1, device='cuda'); "
            "lib.conjugate_gradient_gpu_ex(b.data_ptr(), x.data_ptr(), 1, 3); print('survived')" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == -signal.SIGABRT, (r.returncode, r.stderr[-300:])
    assert "N > 1" in r.stderr and "survived" not in r.stdout


@pytest.mark.parametrize("L,d", [(2, 1), (2, 2), (2, 5), (3, 1)])
def test_smallest_grids_the_solver_accepts(lib, L, d):
    """Two or three points per axis: every point touches the boundary; checked against a dense solve."""
    N = L ** d
    A = np.zeros((N, N))
    for i in range(N):
        A[i, i] = 2 * d
        c, stride = i, 1
        for k in range(d):
            ck = (i // stride) % L
            if ck > 0:
                A[i, i - stride] = -1
            if ck + 1 < L:
                A[i, i + stride] = -1
            stride *= L
    rng = np.random.default_rng(L * 10 + d)
    xt = rng.random(N)
    b = dev((A @ xt).astype(np.float32))
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    res, iters, _ = lib.conjugate_gradient_gpu_ex(b.data_ptr(), x.data_ptr(), L, d)
    assert 1 <= iters <= N + 2
    assert np.abs(x.cpu().numpy() - xt).max() <= 1e-4


def test_largest_grid_an_int_index_allows(lib):
    """1-D, N = 2^31 - 4 points (the reference computes N into an int; the rows engine marches such a grid in
    rows of 932 points).  Stencil of ones: only the two end points see the boundary.  One eigenmode as
    right-hand side: CG is done after one iteration (a second for the lagged stop test) with x = b / lambda."""
    L = (1 << 31) - 4
    u = torch.ones(L, dtype=torch.float32, device="cuda")
    out = stencil(lib, u, L, 1)
    assert out[0].item() == 1.0 and out[L - 1].item() == 1.0
    assert out.sum(dtype=torch.float64).item() == 2.0
    assert out[1:L - 1].abs().max().item() == 0.0
    del u, out
    j = L // 2 + 1
    theta = np.pi * j / (L + 1)
    b = torch.arange(1, L + 1, device="cuda", dtype=torch.float64)
    b.mul_(theta).sin_()
    bf = b.float()
    del b
    x = torch.zeros(L, dtype=torch.float32, device="cuda")
    res, iters, hist = lib.conjugate_gradient_gpu_ex(bf.data_ptr(), x.data_ptr(), L, 1, hist_len=8)
    assert iters == 2, (iters, hist)
    lam = 2.0 - 2.0 * np.cos(theta)
    x.mul_(lam).sub_(bf)
    assert x.abs().max().item() <= 1e-5


def test_the_drop_in_solve_gives_its_device_memory_back(lib, monkeypatch):
    """The reference frees r, Ap, p before conjugate_gradient_gpu returns (conjugate-gradient_gpu.cu:371-373).
    Here a solver above CG_CACHE_KEEP_MB is freed on return, a smaller one is kept for the next call on the
    same grid and freed by cg_release_cache()."""
    monkeypatch.setenv("CG_GPU_QUIET", "1")
    lib.release_cache()
    L, d = 512, 3
    N = L ** d
    b = torch.rand(N, dtype=torch.float32, device="cuda")
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    lib.conjugate_gradient_gpu_ex(b.data_ptr(), x.data_ptr(), L, d, maxit=4)      # 3 x 513 MiB: above the limit
    free1 = torch.cuda.mem_get_info()[0]
    assert abs(free0 - free1) <= 64 << 20, (free0, free1)
    assert float(x.abs().max()) > 0
    del b, x
    # a small grid stays cached (the second call reuses it) until it is released
    L, d = 256, 2
    N = L ** d
    b = torch.rand(N, dtype=torch.float32, device="cuda")
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    r1 = lib.conjugate_gradient_gpu_ex(b.data_ptr(), x.data_ptr(), L, d, maxit=10)
    x.zero_()
    r2 = lib.conjugate_gradient_gpu_ex(b.data_ptr(), x.data_ptr(), L, d, maxit=10)
    assert r1 == r2
    monkeypatch.setenv("CG_CACHE_KEEP_MB", "0")
    x.zero_()
    r3 = lib.conjugate_gradient_gpu_ex(b.data_ptr(), x.data_ptr(), L, d, maxit=10)  # limit 0: freed on return
    assert r3 == r1
    lib.release_cache()                                                             # nothing left: a no-op
    lib.release_cache()
