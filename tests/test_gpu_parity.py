"""GPU parity tests: the HIP library, called through its C-ABI, against the CPU oracle.

Bars: index mapping and the stencil bit-exact; dot products within 1 ulp-scale relative
error of the oracle's double-accumulated value; float CG solutions within the reference's
own acceptance (per-element abs 1e-3, tests.cu:14-41) and rel-L2 <= 1e-4 of the fp64
solution; iteration counts within max(2, 5 %) of the fp32 CPU emulation of
conjugate-gradient_gpu.cu:327-376.
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def iters_close(a, b):
    return abs(a - b) <= max(2, int(np.ceil(0.05 * max(a, b))))


# ---- index mapping ---------------------------------------------------------------------

def test_neighbour_index_known_answers(lib):
    # main.jl:61-65
    assert lib.neighbour_index_gpu(2, 1, 1, 3, 2, 9, 0) == 5
    assert lib.neighbour_index_gpu(2, 0, 1, 3, 2, 9, 0) == 9
    assert lib.neighbour_index_gpu(3, 0, -1, 3, 2, 9, 0) == 9


# ---- dot product -----------------------------------------------------------------------

def test_inner_product_ones_exact(lib):
    # tests.cu:350-359
    x = torch.ones(1000, dtype=torch.float32, device="cuda")
    assert lib.inner_product_gpu(x.data_ptr(), x.data_ptr(), 1000) == 1000.0


@pytest.mark.parametrize("n", [1, 3, 64, 1001, 4096, 1 << 20, (1 << 22) + 5])
def test_inner_product_random(lib, oracle, n):
    rng = np.random.default_rng(n)
    v = rng.standard_normal(n).astype(np.float32)
    w = rng.standard_normal(n).astype(np.float32)
    vd, wd = dev(v), dev(w)  # keep the tensors alive across the call
    got = lib.inner_product_gpu(vd.data_ptr(), wd.data_ptr(), n)
    want = oracle.inner_product_f32(v, w)
    scale = float(np.sum(np.abs(v.astype(np.float64) * w.astype(np.float64))))
    # fp32 per-lane accumulation over a strided subset, then double: error << 1e-5 * sum|v w|
    assert abs(got - want) <= 1e-5 * scale
    # deterministic: same bits on a second call
    assert lib.inner_product_gpu(vd.data_ptr(), wd.data_ptr(), n) == got


def test_inner_product_host_pointer(lib):
    v = np.arange(100, dtype=np.float32)
    got = lib.inner_product_gpu(v.ctypes.data, v.ctypes.data, 100)
    assert got == float(np.sum(v.astype(np.float64) ** 2))


# ---- stencil ---------------------------------------------------------------------------

STENCIL_CASES = [(1000, 1), (1024, 1), (4, 1), (1000 * 1100, 1), (312 * 4000, 1), (2, 20), (2, 13), (2, 2), (10, 2), (5, 3), (64, 2), (256, 2), (12, 3), (16, 3), (32, 3),
                 (8, 4), (6, 4), (4, 5), (2, 10), (260, 2), (100, 3), (16, 4), (64, 3), (128, 3), (16, 5),
                 (1 << 20, 1), (3 << 20, 1)]


@pytest.mark.parametrize("L,d", STENCIL_CASES)
def test_stencil_bit_exact(lib, oracle, engine_env, L, d):
    N = L ** d
    rng = np.random.default_rng(L * 31 + d)
    u = rng.standard_normal(N).astype(np.float32)
    want = oracle.laplace_f32(u, d, L)
    ud = dev(u)
    for mode in (0, 1):  # identical, as in the reference (tests.cu:477 vs gpu.cu:352)
        out = torch.full((N,), np.nan, dtype=torch.float32, device="cuda")
        lib.laplace_gpu_jl(out.data_ptr(), ud.data_ptr(), 0.5, d, L, N, mode, (N + 255) // 256, 256)
        assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize("L,d", [(10, 2), (64, 2), (8, 3)])
def test_stencil_partial_launch_keeps_reference_meaning(lib, oracle, L, d):
    """blocks*threads < N: only the first blocks*threads points are written (julia.cu:55-60)."""
    N = L ** d
    u = np.random.default_rng(5).standard_normal(N).astype(np.float32)
    want = oracle.laplace_f32(u, d, L)
    out = torch.full((N,), -7.0, dtype=torch.float32, device="cuda")
    blocks, threads = 3, 8
    ud = dev(u)
    lib.laplace_gpu_jl(out.data_ptr(), ud.data_ptr(), 0.0, d, L, N, 0, blocks, threads)
    got = out.cpu().numpy()
    assert np.array_equal(got[:24], want[:24]) and np.all(got[24:] == -7.0)


@pytest.mark.parametrize("L,d", [(2, 2), (4, 2), (16, 2), (8, 3), (16, 3), (8, 4), (16, 4), (4, 5), (2, 7), (16, 1)])
def test_stencil_morton_bit_exact(lib, oracle, L, d):
    N = L ** d
    u = np.random.default_rng(11).standard_normal(N).astype(np.float32)
    want = oracle.laplace_morton_f32(u, d)
    out = torch.zeros(N, dtype=torch.float32, device="cuda")
    ud = dev(u)
    lib.laplace_gpu_jl(out.data_ptr(), ud.data_ptr(), 0.0, d, L, N, 2, (N + 255) // 256, 256)
    assert np.array_equal(out.cpu().numpy(), want)


def test_stencil_sin_and_square_reference_properties(lib):
    """tests.cu:378-434: on 1-D sin(x) and x^2 the ratio / value is constant over the interior."""
    N = 1000
    step = np.float32(2 * np.pi / (N - 1))
    xs = (-np.float32(np.pi) + np.arange(N, dtype=np.float32) * step).astype(np.float32)
    for f, ratio in ((np.sin, True), (np.square, False)):
        u = f(xs).astype(np.float32)
        out = torch.zeros(N, dtype=torch.float32, device="cuda")
        ud = dev(u)
        lib.laplace_gpu_jl(out.data_ptr(), ud.data_ptr(), 0.0, 1, N, N, 0, 4, 256)
        r = out.cpu().numpy()
        if ratio:
            with np.errstate(divide="ignore", invalid="ignore"):
                r = r / u
        assert np.all(np.abs(r[1:N - 1] - r[N // 2]) < 1e-3)


# ---- CG --------------------------------------------------------------------------------

@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "cg_cpu_*.npz"))))
def test_cg_recovers_golden_solution(lib, oracle, engine_env, path):
    """b from the reference's fixtures; the GPU float CG must recover x (tests.cu:466-501)."""
    g = np.load(path)
    L, d = int(g["L"]), int(g["d"])
    N = L ** d
    b = g["b"].astype(np.float32)
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    bd = dev(b)
    res, iters, hist = lib.conjugate_gradient_gpu_ex(bd.data_ptr(), x.data_ptr(), L, d, hist_len=200)
    got = x.cpu().numpy()
    assert np.all(np.abs(got - g["x"]) < 1e-3)
    assert rel_l2(got, g["x_solved"]) <= 1e-4
    # iteration count and printed history vs the fp32 CPU emulation of the reference loop
    xe, rese, iterse, histe = oracle.cg_f32(b, L, d, hist_len=200)
    assert iters_close(iters, iterse), (iters, iterse)
    assert rel_l2(got, xe) <= 1e-4
    n = min(iters, iterse, 10)
    assert np.allclose(hist[:n], histe[:n], rtol=1e-3)
    assert res == pytest.approx(hist[-1]) if iters else True


@pytest.mark.parametrize("L,d,engine", [(64, 2, 1), (64, 2, 2), (16, 3, 1), (16, 3, 2), (8, 4, 1), (8, 4, 2),
                                        (1024, 1, 1), (1024, 1, 2), (128, 2, 1), (16, 3, 3), (24, 3, 3),
                                        (260, 3, 3)])
def test_engines_agree_with_emulation(lib, oracle, L, d, engine):
    N = L ** d
    rng = np.random.default_rng(N)
    xt = rng.random(N).astype(np.float32)
    b = oracle.laplace_f32(xt, d, L)
    s = lib.make_solver([L] * d, engine=engine)
    assert s.engine == engine
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    bd = dev(b)
    s.begin(bd.data_ptr(), x.data_ptr(), 1e-6, 1)
    s.iterate(200 if N < 10 ** 7 else 60, use_graph=(engine != 2))
    iters, res, hist = s.finish(hist_len=200)
    xe, rese, iterse, histe = oracle.cg_f32(b, L, d, maxit=200 if N < 10 ** 7 else 60, hist_len=200)
    assert iters_close(iters, iterse), (iters, iterse)
    got = x.cpu().numpy()
    if iterse < 60:
        assert np.all(np.abs(got - xt) < 1e-3)
    assert rel_l2(got, xe) <= 2e-4
    s.close()


@pytest.mark.parametrize("N", [3 << 20, 1000 * 1100, 1_048_580, 312 * 4000, 516 * 2100])
def test_long_1d_grid_marches_in_rows(lib, oracle, N):
    """1-D grids of 2^20 points and more are cut into contiguous rows by the rows engine (the largest
    multiple of 4 from 1024 down to 512 that divides N: 1024, 1000, none for 4 x 262145, 960, 1008; the
    reference's own driver solves 100000 x 312 points, main.cu:40-50): same stencil bits as the oracle
    (in the stencil cases above) and the same CG as the one-point-per-lane engine."""
    b = np.random.default_rng(4).random(N).astype(np.float32)
    bd = dev(b)
    outs = []
    for engine in (1, 2):
        s = lib.make_solver([N], engine=engine)
        x = torch.zeros(N, dtype=torch.float32, device="cuda")
        s.begin(bd.data_ptr(), x.data_ptr(), 0.0, 1)
        s.iterate(50, use_graph=True)
        iters, res, hist = s.finish(hist_len=64)
        assert iters == 50
        outs.append((x.cpu().numpy(), np.array(hist)))
        s.close()
    assert np.allclose(outs[0][1], outs[1][1], rtol=1e-4)
    assert rel_l2(outs[0][0], outs[1][0]) <= 1e-4
    xe, _, _, histe = oracle.cg_f32(b, N, 1, maxit=50, hist_len=64, rel=0.0)
    assert np.allclose(outs[0][1], histe[:50], rtol=1e-3)


def test_graph_and_plain_launches_are_bit_identical(lib, oracle):
    L, d = 128, 2
    N = L * L
    b = np.random.default_rng(2).random(N).astype(np.float32)
    outs = []
    for use_graph in (True, False):
        s = lib.make_solver([L, L])
        x = torch.zeros(N, dtype=torch.float32, device="cuda")
        bd = dev(b)
        s.begin(bd.data_ptr(), x.data_ptr(), 0.0, 1)
        s.iterate(37, use_graph=use_graph)
        s.iterate(13, use_graph=use_graph)
        iters, res, hist = s.finish(hist_len=64)
        assert iters == 50
        outs.append((x.cpu().numpy(), np.array(hist)))
        s.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_fixed_iteration_history_tracks_emulation(lib, oracle):
    """A problem that cannot converge in 200 float iterations runs all 200 (gpu.cu:348)."""
    L, d = 256, 2
    N = L * L
    b = np.random.default_rng(9).random(N).astype(np.float32)
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    bd = dev(b)
    res, iters, hist = lib.conjugate_gradient_gpu_ex(bd.data_ptr(), x.data_ptr(), L, d, hist_len=200)
    xe, rese, iterse, histe = oracle.cg_f32(b, L, d, hist_len=200)
    assert iters == iterse == 200
    assert np.allclose(hist[:50], histe[:50], rtol=2e-3)
    assert rel_l2(x.cpu().numpy(), xe) <= 1e-3


def test_cg_accumulates_into_x_and_assumes_r_equals_b(lib, oracle):
    """x is added to, not restarted from (gpu.cu:336,356): x_out = x_in + sum alpha p."""
    L, d = 10, 2
    g = np.load(os.path.join(GOLDEN, "cg_cpu_L10_d2.npz"))
    b = g["b"].astype(np.float32)
    x0 = np.full(100, 3.0, np.float32)
    xa = dev(x0.copy())
    xb = torch.zeros(100, dtype=torch.float32, device="cuda")
    bd = dev(b)
    lib.conjugate_gradient_gpu_ex(bd.data_ptr(), xa.data_ptr(), L, d)
    lib.conjugate_gradient_gpu_ex(bd.data_ptr(), xb.data_ptr(), L, d)
    assert np.allclose(xa.cpu().numpy() - 3.0, xb.cpu().numpy(), atol=2e-6)


def test_reference_entry_point_on_managed_fields(lib, capfd):
    """main.cu:46-50 / tests.cu:472-488 usage: managed N+1 fields touched by the host."""
    L, d, N = 10, 2, 100
    g = np.load(os.path.join(GOLDEN, "cg_cpu_L10_d2.npz"))
    pb = lib.c.cg_allocate_field(N)
    px = lib.c.cg_allocate_field(N)
    fb = np.ctypeslib.as_array(C.cast(pb, C.POINTER(C.c_float)), shape=(N + 1,))
    fx = np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_float)), shape=(N + 1,))
    assert np.all(fb == 0) and np.all(fx == 0)
    fb[:N] = g["b"].astype(np.float32)
    res = lib.conjugate_gradient_gpu(pb, px, L, d)
    C.CDLL(None).fflush(None)  # the library prints through C stdio
    out = capfd.readouterr().out.strip().splitlines()
    assert np.all(np.abs(fx[:N] - g["x"]) < 1e-3)
    # the reference prints "i,residue" once per iteration (gpu.cu:369)
    assert len(out) >= 1 and out[0].startswith("1,")
    assert int(out[-1].split(",")[0]) == len(out)
    assert float(out[-1].split(",")[1]) == pytest.approx(res, abs=1e-6)
    lib.c.cg_free_field(pb)
    lib.c.cg_free_field(px)


@pytest.mark.parametrize("L,d", [(16, 2), (8, 3), (8, 4)])
def test_morton_cg_equals_lexicographic_cg(lib, oracle, L, d):
    N = L ** d
    rng = np.random.default_rng(4)
    xt = rng.random(N).astype(np.float32)
    b = oracle.laplace_f32(xt, d, L)
    perm = oracle.lex_to_morton_perm(L, d)
    bm = np.zeros(N, np.float32)
    bm[perm] = b
    xl = torch.zeros(N, dtype=torch.float32, device="cuda")
    xm = torch.zeros(N, dtype=torch.float32, device="cuda")
    bd, bmd = dev(b), dev(bm)
    _, it_l, _ = lib.conjugate_gradient_gpu_ex(bd.data_ptr(), xl.data_ptr(), L, d, index_mode=0)
    _, it_m, _ = lib.conjugate_gradient_gpu_ex(bmd.data_ptr(), xm.data_ptr(), L, d, index_mode=2)
    assert iters_close(it_l, it_m)
    assert np.all(np.abs(xm.cpu().numpy()[perm] - xt) < 1e-3)
    assert rel_l2(xm.cpu().numpy()[perm], xl.cpu().numpy()) <= 1e-4


def test_preconditioner_rule_stops_on_fresh_residual(lib, oracle):
    """lagged = 0 reproduces the CPU preconditioner's stop rule (cpu.cu:299,317) in float."""
    g = np.load(os.path.join(GOLDEN, "pcg_cpu_L50_d2.npz"))
    b = g["b"].astype(np.float32)
    x = torch.zeros(2500, dtype=torch.float32, device="cuda")
    bd = dev(b)
    res, iters, _ = lib.conjugate_gradient_gpu_ex(bd.data_ptr(), x.data_ptr(), 50, 2, maxit=-1, rel=1e-3, lagged=0)
    xe, rese, iterse, _ = oracle.cg_f32(b, 50, 2, maxit=-1, rel=1e-3, lagged=False)
    assert iters_close(iters, iterse), (iters, iterse)
    assert rel_l2(x.cpu().numpy(), xe) <= 1e-3


def test_mixed_precision_matches_fp64_reference(lib, oracle):
    """Double outer CG + float GPU inner CG recovers the reference's fp64 solution."""
    g = np.load(os.path.join(GOLDEN, "pcg_cpu_L50_d2.npz"))
    b = np.ascontiguousarray(g["b"], np.float64)
    x = np.zeros(2500, np.float64)
    outer, inner = lib.preconditioned_cg_gpu(b.ctypes.data, x.ctypes.data, 50, 2)
    assert 1 <= outer <= 8 and len(inner) >= outer
    assert np.all(np.abs(x - g["x"]) < 1e-3)          # tests.cu:114-146 acceptance
    assert rel_l2(x, g["x_pcg"]) <= 1e-7              # outer tolerance is 1e-8 |b|
    # a plain float solve cannot reach that
    xf = torch.zeros(2500, dtype=torch.float32, device="cuda")
    bf = dev(b.astype(np.float32))
    lib.conjugate_gradient_gpu_ex(bf.data_ptr(), xf.data_ptr(), 50, 2)
    assert rel_l2(xf.cpu().numpy(), g["x_pcg"]) > rel_l2(x, g["x_pcg"])


@pytest.mark.parametrize("L,d", [(50, 2), (64, 3), (32, 3), (16, 4), (12, 5), (256, 2)])
def test_outer_loop_stencil_as_a_march_gives_the_same_solve(lib, monkeypatch, L, d):
    """The fp64 stencil + dot of the outer loop (conjugate-gradient_cpu.cu:363-365) has two kernels: one point
    per lane, and a march along the last axis (two points of axis 0 per lane, grids with an even L).  Both add
    the stencil's terms in the reference's order, so forcing either one changes only the order in which the
    partial sums of p.Ap are added: same outer and inner iteration counts, solutions equal to 1e-12."""
    N = L ** d
    gen = torch.Generator(device="cuda").manual_seed(8)
    b = torch.rand(N, dtype=torch.float64, device="cuda", generator=gen)
    out = {}
    for want in ("0", "2"):
        monkeypatch.setenv("CG_MIXED_MARCH", want)
        x = torch.zeros(N, dtype=torch.float64, device="cuda")
        out[want] = (lib.preconditioned_cg_gpu(b.data_ptr(), x.data_ptr(), L, d), x)
    assert out["0"][0] == out["2"][0]
    assert out["0"][0][0] >= 1
    err = (out["0"][1] - out["2"][1]).abs().max().item() / out["0"][1].abs().max().item()
    assert err <= 1e-12, err


# ---- size-independent properties at larger sizes ---------------------------------------

@pytest.mark.parametrize("L,d", [(2048, 2), (128, 3), (32, 4)])
def test_stencil_linearity_and_symmetry_large(lib, engine_env, L, d):
    """<A u, v> == <u, A v> and A(u+v) == A u + A v up to rounding, at sizes the oracle skips."""
    N = L ** d
    gen = torch.Generator(device="cuda").manual_seed(1)
    u = torch.rand(N, device="cuda", generator=gen)
    v = torch.rand(N, device="cuda", generator=gen)
    Au, Av, Auv = torch.empty_like(u), torch.empty_like(u), torch.empty_like(u)
    uv = u + v
    for src, dst in ((u, Au), (v, Av), (uv, Auv)):
        lib.laplace_gpu_jl(dst.data_ptr(), src.data_ptr(), 0.0, d, L, N, 0, (N + 255) // 256, 256)
    a = lib.inner_product_gpu(Au.data_ptr(), v.data_ptr(), N)
    b = lib.inner_product_gpu(u.data_ptr(), Av.data_ptr(), N)
    assert abs(a - b) <= 1e-4 * abs(a)
    assert torch.allclose(Auv, Au + Av, rtol=0, atol=1e-4)
    # against torch on the same data (fp32, different operation order)
    t = u.reshape([L] * d)
    ref = 2 * d * t.clone()
    for ax in range(d):
        ref -= torch.roll(t, 1, ax) * (torch.arange(L, device="cuda") > 0).reshape([-1 if i == ax else 1 for i in range(d)])
        ref -= torch.roll(t, -1, ax) * (torch.arange(L, device="cuda") < L - 1).reshape([-1 if i == ax else 1 for i in range(d)])
    assert torch.allclose(Au.reshape([L] * d), ref, rtol=0, atol=1e-4)


def test_eigenmode_rhs_converges_in_few_iterations(lib):
    """b = sum of m mid-spectrum Laplacian eigenmodes: CG terminates after m iterations (+1 for
    the lagged test) at any grid size.  (Low modes do not show this in fp32: their eigenvalues
    are ~1e-3 and rounding in A p swamps them; the fp32 CPU emulation behaves the same.)"""
    L, d = 1024, 2
    N = L * L
    ax = torch.arange(1, L + 1, device="cuda", dtype=torch.float64) * (np.pi / (L + 1))
    b = torch.zeros(L, L, device="cuda", dtype=torch.float64)
    for j in (L // 4, L // 2, 3 * L // 4):
        b += torch.outer(torch.sin(j * ax), torch.sin(j * ax))
    bf = b.reshape(-1).float().contiguous()
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    res, iters, hist = lib.conjugate_gradient_gpu_ex(bf.data_ptr(), x.data_ptr(), L, d, hist_len=200)
    assert iters == 4, (iters, hist)
    Ax = torch.empty_like(x)
    lib.laplace_gpu_jl(Ax.data_ptr(), x.data_ptr(), 0.0, d, L, N, 0, (N + 255) // 256, 256)
    assert (Ax - bf).norm().item() <= 1e-5 * bf.norm().item()


@pytest.mark.parametrize("L,d", [(2048, 2), (128, 3)])
def test_recurrence_residual_equals_true_residual_large(lib, L, d):
    """After K iterations the solver's r (updated by recurrence) must equal b - A x recomputed from x."""
    N = L ** d
    gen = torch.Generator(device="cuda").manual_seed(3)
    b = torch.rand(N, device="cuda", generator=gen)
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    s = lib.make_solver([L] * d)
    s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
    s.iterate(41)
    iters, res, hist = s.finish(hist_len=64)
    assert iters == 41 and len(hist) == 41
    assert all(np.isfinite(hist)) and hist[0] == pytest.approx(b.double().norm().item(), rel=1e-6)
    r_ptr = lib.c.cg_solver_r(s.h)
    rec = np.sqrt(lib.inner_product_gpu(r_ptr, r_ptr, N))
    Ax = torch.empty_like(x)
    lib.laplace_gpu_jl(Ax.data_ptr(), x.data_ptr(), 0.0, d, L, N, 0, (N + 255) // 256, 256)
    true_res = (b.double() - Ax.double()).norm().item()
    assert abs(true_res - rec) <= 1e-3 * rec, (true_res, rec)
    s.close()


def morton_perm(L, d):
    """perm[lex] = Z-curve index, vectorised (interleave.cuh:110-118: coordinate k owns bits k, k+d, ...)."""
    N = L ** d
    lex = np.arange(N, dtype=np.int64)
    bits = L.bit_length() - 1
    out = np.zeros(N, dtype=np.int64)
    rem = lex.copy()
    for k in range(d):
        c = rem % L
        rem //= L
        for b in range(bits):
            out |= ((c >> b) & 1) << (b * d + k)
    return out


@pytest.mark.parametrize("relayout", ["0", "1"])
@pytest.mark.parametrize("L,d", [(32, 4), (128, 3), (1024, 2), (16, 5), (8, 7)])
def test_morton_tiled_engine_equals_lexicographic_large(lib, oracle, monkeypatch, L, d, relayout):
    """Z-curve data against the lexicographic engines on the same data, at sizes with hundreds of
    tiles: stencil bit-exact, 30 CG iterations to rounding.  The solve runs either on the LDS-tiled
    Z-curve kernels (relayout 0) or, the default, in lexicographic order inside the solver with b and
    x permuted on the way in and out."""
    monkeypatch.delenv("CG_ENGINE", raising=False)   # a forced engine keeps the Z-curve kernels
    monkeypatch.setenv("CG_MORTON_RELAYOUT", relayout)
    N = L ** d
    perm = morton_perm(L, d)
    if L <= 32:
        assert np.array_equal(perm[:4096], oracle.lex_to_morton_perm(L, d)[:4096])
    rng = np.random.default_rng(17)
    u = rng.random(N).astype(np.float32)
    um = np.zeros(N, np.float32)
    um[perm] = u
    ud, umd = dev(u), dev(um)
    a_lex = torch.empty(N, dtype=torch.float32, device="cuda")
    a_mor = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.laplace_gpu_jl(a_lex.data_ptr(), ud.data_ptr(), 0.0, d, L, N, 0, (N + 255) // 256, 256)
    lib.laplace_gpu_jl(a_mor.data_ptr(), umd.data_ptr(), 0.0, d, L, N, 2, (N + 255) // 256, 256)
    assert np.array_equal(a_mor.cpu().numpy()[perm], a_lex.cpu().numpy())
    xl = torch.zeros(N, dtype=torch.float32, device="cuda")
    xm = torch.zeros(N, dtype=torch.float32, device="cuda")
    _, itl, hl = lib.conjugate_gradient_gpu_ex(ud.data_ptr(), xl.data_ptr(), L, d, index_mode=0, maxit=30, hist_len=30)
    _, itm, hm = lib.conjugate_gradient_gpu_ex(umd.data_ptr(), xm.data_ptr(), L, d, index_mode=2, maxit=30, hist_len=30)
    assert itl == itm == 30
    assert np.allclose(hl, hm, rtol=1e-4)
    assert rel_l2(xm.cpu().numpy()[perm], xl.cpu().numpy()) <= 1e-4
    # x is accumulated into (conjugate-gradient_gpu.cu:327-376 never resets it): a second solve adds to the first
    lib.conjugate_gradient_gpu_ex(ud.data_ptr(), xl.data_ptr(), L, d, index_mode=0, maxit=5)
    lib.conjugate_gradient_gpu_ex(umd.data_ptr(), xm.data_ptr(), L, d, index_mode=2, maxit=5)
    assert rel_l2(xm.cpu().numpy()[perm], xl.cpu().numpy()) <= 1e-4
    s = lib.make_solver([L] * d, index_mode=2)
    assert s.kernel_name.startswith("k_morton") == (relayout == "0"), s.kernel_name
    s.close()


def test_mixed_precision_3d(lib):
    """Outer fp64 CG + inner float GPU CG on 64^3: residual far below what float alone reaches."""
    L, d = 64, 3
    N = L ** d
    gen = torch.Generator(device="cuda").manual_seed(5)
    xt = torch.rand([L] * d, dtype=torch.float64, device="cuda", generator=gen)

    def lap(t):
        out = 2 * d * t.clone()
        for ax in range(d):
            lo = [slice(None)] * d
            hi = [slice(None)] * d
            lo[ax], hi[ax] = slice(0, L - 1), slice(1, L)
            out[tuple(lo)] -= t[tuple(hi)]
            out[tuple(hi)] -= t[tuple(lo)]
        return out

    b = lap(xt).reshape(-1).contiguous()
    x = torch.zeros_like(b)
    outer, inner = lib.preconditioned_cg_gpu(b.data_ptr(), x.data_ptr(), L, d)   # device pointers
    assert 1 <= outer <= 10 and len(inner) == outer
    res = (b - lap(x.reshape([L] * d)).reshape(-1)).norm().item() / b.norm().item()
    assert res < 1e-8
    assert (x - xt.reshape(-1)).abs().max().item() < 1e-6


# ---- the remaining exports of the Julia shim (julia.cu:16-60) --------------------------

def test_add_jl(lib):
    a = torch.arange(1024, dtype=torch.float32, device="cuda")
    b = torch.ones(1024, dtype=torch.float32, device="cuda")
    lib.add_jl(a.data_ptr(), b.data_ptr(), 4, 256)
    torch.cuda.synchronize()
    assert torch.equal(a, torch.arange(1024, dtype=torch.float32, device="cuda") + 1)


def test_strong_scaling_and_cpu_timer_return_microseconds(lib):
    # main.jl:139-163 divides one by the other; both are positive microsecond counts
    t_gpu = lib.strong_scaling(1000, 512, 512000, 512000, 1)
    t_gpu1 = lib.strong_scaling(512000, 1, 512000, 512000, 1)       # the reference's <<<N,1>>> shape
    t_cpu = lib.time_cpu_laplace(512000, 512000, 1)
    assert 0 < t_gpu < 1e6 and 0 < t_gpu1 < 1e7 and 0 < t_cpu < 1e7
    t2 = lib.strong_scaling(4096, 256, 1 << 20, 1024, 2)
    assert 0 < t2 < 1e6


def test_event_timers_of_the_stencil(lib):
    """strong_scaling (julia.cu:16-30) re-based on the device clock: cg_strong_scaling_events times the
    caller's launch shape, cg_laplace_time_events the tiled whole-grid stencil.  Both return mean
    microseconds per launch: positive, growing with the grid, and never above the wall-clock figure of
    strong_scaling for the same launch (which adds the launch and the synchronisation)."""
    small = lib.strong_scaling_events(4096, 256, 1 << 20, 1024, 2, reps=10)
    large = lib.strong_scaling_events(4 * 4096, 256, 1 << 22, 2048, 2, reps=10)
    assert 0 < small < large < 1e6
    wall = min(lib.strong_scaling(4 * 4096, 256, 1 << 22, 2048, 2) for _ in range(3))
    assert large <= wall
    # the reference's own shape, one thread per block, is far slower than 256 per block on the same grid
    one = lib.strong_scaling_events(1 << 20, 1, 1 << 20, 1024, 2, reps=3)
    assert one > 2 * small
    t1 = lib.laplace_time_events(1024, 2, 0, reps=10)
    t2 = lib.laplace_time_events(4096, 2, 0, reps=10)
    assert 0 < t1 < t2 < 1e6
    assert t2 < 4 * large             # four times the points of `large`, tiled: not slower per point
    tm = lib.laplace_time_events(64, 3, 2, reps=10)      # Morton layout
    assert 0 < tm < 1e6


def test_reference_kernel_names_via_compat_launch(lib, oracle):
    """laplace_gpu_jl with a launch that does not cover the grid takes the reference-named
    one-point-per-thread kernel; index_mode 2 there is the Morton stencil."""
    L, d = 8, 3
    N = L ** d
    u = np.random.default_rng(8).standard_normal(N).astype(np.float32)
    ud = dev(u)
    for mode, want in ((0, oracle.laplace_f32(u, d, L)), (2, oracle.laplace_morton_f32(u, d))):
        out = torch.full((N,), 9.0, dtype=torch.float32, device="cuda")
        lib.laplace_gpu_jl(out.data_ptr(), ud.data_ptr(), 0.0, d, L, N, mode, 2, 128)   # covers 256 of 512 points
        got = out.cpu().numpy()
        assert np.array_equal(got[:256], want[:256]) and np.all(got[256:] == 9.0)


def test_cpp_drivers_built_against_the_library(pkg):
    """lib/tests: the assert-based C++ executable that launches the reference-named kernels with
    its own shapes on managed fields (in the manner of tests.cu:327-525)."""
    import subprocess
    exe = os.path.join(os.path.dirname(pkg.LIB_PATH), "tests")
    if not os.path.exists(exe):
        pytest.skip("drivers not built (make -C conjugate-gradient_amd drivers)")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "tests passed!" in out.stdout


SHAPES = [[128, 3, 5], [256, 1, 7], [64, 64, 2, 3], [32, 5, 4, 3, 2], [16, 7, 9], [512, 2], [260, 4, 3], [1028, 5],
          [128, 128, 1], [8, 8, 8, 8], [2048, 3, 1, 2], [4, 6, 5]]


@pytest.mark.parametrize("ext", SHAPES, ids=lambda e: "x".join(map(str, e)))
def test_engines_agree_on_non_cubic_grids(lib, ext):
    """Grids with unequal extents (slabs are such grids) are only reachable through the solver object:
    every engine that accepts the shape must reproduce the one-point-per-lane engine."""
    N = int(np.prod(ext))
    b = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(b.data_ptr(), N, 21)
    results = {}
    for engine in (2, 1, 3):
        if engine != 2 and ext[0] % 4:
            continue
        if engine == 3 and len(ext) != 3:
            continue
        s = lib.make_solver(ext, engine=engine)
        assert s.engine == engine
        x = torch.zeros(N, dtype=torch.float32, device="cuda")
        s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
        s.iterate(12, use_graph=False)
        iters, res, hist = s.finish(hist_len=12)
        assert iters == 12
        results[engine] = (x.cpu().numpy(), np.array(hist))
        s.close()
    ref_x, ref_h = results[2]
    for engine, (xe, he) in results.items():
        assert np.allclose(he, ref_h, rtol=1e-5), (engine, he, ref_h)
        assert rel_l2(xe, ref_x) <= 1e-5, engine


@pytest.mark.parametrize("exact", ["0", "1"])
@pytest.mark.parametrize("R,stream", [(2, 0), (2, 1), (1, 1), (0, 0)])
@pytest.mark.parametrize("ext", [[64, 6, 4, 5], [32, 3, 6, 7], [16, 5, 3, 4], [128, 9, 2, 3], [32, 4, 3, 2, 6], [16, 3, 2, 4, 5],
                                 [16, 2, 2, 2, 2], [64, 40, 34, 9], [64, 90, 7], [128, 33, 5], [48, 200, 30], [200, 70, 3],
                                 [96, 10, 6, 5], [20, 9, 8, 7], [80, 12, 6], [24, 6, 4, 4, 3], [100, 7, 6, 5]],
                         ids=lambda e: "x".join(map(str, e)))
def test_blocked_columns_match_the_scalar_engine(lib, monkeypatch, ext, R, stream, exact):
    """3-D to 5-D grids with short rows: R columns per lane along axis d-2, neighbours fetched two
    steps ahead, with and without the streaming hint (the automatic choice depends on the grid's
    size), and the fall-back to one column per lane when axis d-2 has an odd extent.  The stencil must
    have the scalar engine's bits: twelve iterations then agree to the rounding of the dot products."""
    monkeypatch.delenv("CG_ENGINE", raising=False)
    monkeypatch.setenv("CG_COLS_R", str(R))
    monkeypatch.setenv("CG_COLS_STREAM", str(stream))
    monkeypatch.setenv("CG_COLS_EXACT", exact)   # rows of exactly n0/4 lanes (x neighbours through LDS)
    N = int(np.prod(ext))
    b = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(b.data_ptr(), N, 33)
    out = {}
    for engine in (2, 1):
        s = lib.make_solver(ext, engine=engine)
        x = torch.zeros(N, dtype=torch.float32, device="cuda")
        s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
        s.iterate(1, use_graph=False)
        s.sync()
        x1 = x.cpu().numpy().copy()
        s.iterate(11, use_graph=True)
        iters, res, hist = s.finish(hist_len=12)
        assert iters == 12
        out[engine] = (x1, x.cpu().numpy(), np.array(hist))
        s.close()
    # after one iteration x = alpha b: the same bits unless alpha differs in its last place
    assert np.allclose(out[1][0], out[2][0], rtol=3e-7, atol=0)
    assert np.allclose(out[1][2], out[2][2], rtol=1e-5)
    assert rel_l2(out[1][1], out[2][1]) <= 1e-5


@pytest.mark.parametrize("t2", ["2", "4", "8"])
@pytest.mark.parametrize("exact", ["0", "1"])
@pytest.mark.parametrize("ext", [[64, 6, 4, 5], [32, 3, 6, 7], [16, 5, 3, 4], [128, 9, 2, 3], [64, 40, 34, 9], [96, 10, 6, 5],
                                 [20, 9, 8, 7], [100, 7, 6, 5], [32, 32, 32, 8], [64, 17, 13, 6], [128, 16, 2, 40]],
                         ids=lambda e: "x".join(map(str, e)))
def test_tiled_slots_of_blocked_columns_match_the_scalar_engine(lib, monkeypatch, ext, t2, exact):
    """4-D: the column slots of a workgroup as a patch of axis 1 x blocks of axis 2 (CG_COLS_T2 slots along axis 2;
    large grids take 2 by themselves), every block-edge neighbour inside the patch through LDS.  Same check as for
    the strip: the one-point-per-lane engine's bits, patches that hang over the grid's faces included."""
    monkeypatch.delenv("CG_ENGINE", raising=False)
    monkeypatch.setenv("CG_COLS_R", "2")
    monkeypatch.setenv("CG_COLS_LDS_A", "1")
    monkeypatch.setenv("CG_COLS_T2", t2)
    monkeypatch.setenv("CG_COLS_EXACT", exact)
    N = int(np.prod(ext))
    b = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(b.data_ptr(), N, 34)
    out = {}
    for engine in (2, 1):
        s = lib.make_solver(ext, engine=engine)
        if engine == 1:
            assert s.kernel_name == "k_cols"
        x = torch.zeros(N, dtype=torch.float32, device="cuda")
        s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
        s.iterate(1, use_graph=False)
        s.sync()
        x1 = x.cpu().numpy().copy()
        s.iterate(11, use_graph=True)
        iters, res, hist = s.finish(hist_len=12)
        assert iters == 12
        out[engine] = (x1, x.cpu().numpy(), np.array(hist))
        s.close()
    assert np.allclose(out[1][0], out[2][0], rtol=3e-7, atol=0)
    assert np.allclose(out[1][2], out[2][2], rtol=1e-5)
    assert rel_l2(out[1][1], out[2][1]) <= 1e-5


@pytest.mark.parametrize("ext", [[2048, 1024], [1 << 21], [1000, 700], [512, 64, 32], [64, 64, 64, 8]], ids=lambda e: "x".join(map(str, e)))
@pytest.mark.parametrize("env", [{"CG_CACHE_POLICY": "0"}, {"CG_CACHE_POLICY": "1"}, {"CG_CACHE_POLICY": "3"},
                                 {"CG_CACHE_POLICY_A": "3"}, {"CG_CACHE_POLICY": "2", "CG_CACHE_POLICY_A": "1"},
                                 {"CG_CACHE_POLICY": "4", "CG_KEEP_MB": "3"}, {"CG_CACHE_POLICY": "4", "CG_KEEP_MB": "0"},
                                 {"CG_CACHE_POLICY": "1", "CG_CACHE_POLICY_A": "4", "CG_KEEP_MB": "5"}],
                         ids=lambda e: ",".join(f"{k[9:]}={v}" for k, v in e.items()))
def test_cache_policies_change_no_bit(lib, monkeypatch, ext, env):
    """The cache policies (plain accesses, r kept in the Infinity Cache, r and p kept, everything streaming, a
    cache-sized part of r kept -- here 3 or 5 MiB of the 8 MiB vectors; one for kernel A and one for kernel B) only
    choose between temporal and non-temporal loads and stores: a solve under
    any of them has the automatic choice's bits, whichever engine the grid takes (a policy that does not exist
    for that engine falls back)."""
    N = int(np.prod(ext))
    b = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(b.data_ptr(), N, 35)
    out = []
    for use_env in (False, True):
        for k, v in env.items():
            if use_env:
                monkeypatch.setenv(k, v)
        s = lib.make_solver(ext)
        x = torch.zeros(N, dtype=torch.float32, device="cuda")
        s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
        s.iterate(13)
        out.append((s.finish(hist_len=13), x))
        s.close()
    assert out[0][0] == out[1][0] and out[0][0][0] == 13
    assert torch.equal(out[0][1], out[1][1])


@pytest.mark.parametrize("ext", [[512, 256, 64], [1024, 64, 40], [256, 128, 128], [260, 36, 24], [768, 48, 16]],
                         ids=lambda e: "x".join(map(str, e)))
def test_tiles_dealt_to_the_xcds_in_runs_give_the_same_solve(lib, monkeypatch, ext):
    """The 3-D engine hands its tiles to the eight XCDs in contiguous runs when all of them are resident at once
    (CG_PLANES_XCD, csrc/cg_planes.hpp); tile order only regroups the dot products' addends, so histories and x
    agree with the plain order to rounding, and with the one-point-per-lane engine."""
    N = int(np.prod(ext))
    b = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(b.data_ptr(), N, 77)
    out = []
    for xcd, engine in (("0", 3), ("1", 3), ("1", 2)):
        monkeypatch.setenv("CG_PLANES_XCD", xcd)
        s = lib.make_solver(ext, engine=engine)
        x = torch.zeros(N, dtype=torch.float32, device="cuda")
        s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
        s.iterate(15)
        iters, res, hist = s.finish(hist_len=15)
        assert iters == 15
        out.append((np.array(hist), x.cpu().numpy()))
        s.close()
    for h, xv in out[1:]:
        assert np.allclose(h, out[0][0], rtol=1e-5)
        assert rel_l2(xv, out[0][1]) <= 1e-5


# ---- grids that fit one workgroup: whole iterations in a single launch -------------------------

def _solve(lib, ext, b, plan, mode=0):
    """plan: list of ('it', n) -> iterate(n) or ('ab', n) -> n explicit kernel A / kernel B rounds."""
    N = b.numel()
    s = lib.make_solver(ext, index_mode=mode)
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    s.begin(b.data_ptr(), x.data_ptr(), 1e-6, 1)
    for what, n in plan:
        if what == "it":
            s.iterate(n, use_graph=True)
        else:
            s.add_iterations(n)
            for _ in range(n):
                s.kernel_A()
                s.kernel_B()
    iters, res, hist = s.finish(hist_len=300)
    s.close()
    return x.cpu().numpy(), iters, res, np.array(hist)


@pytest.mark.parametrize("ext,mode", [([64, 64], 0), ([50, 50], 0), ([16, 16, 16], 0), ([8, 8, 8, 8], 0), ([5, 5, 5], 0),
                                      ([4, 4, 4, 4, 4], 0), ([8191], 0), ([16, 16, 16], 2), ([64, 64], 2), ([90, 91], 0),
                                      ([8193], 0), ([16384], 0), ([128, 128], 0), ([127, 129], 0), ([25, 25, 25], 0),
                                      ([11, 11, 11, 11], 0), ([7, 7, 7, 7, 6], 0), ([128, 128], 2), ([16384], 2),
                                      ([4, 4, 4, 4, 4, 4, 4], 2)])
def test_single_workgroup_solver_matches_the_two_kernel_iteration(lib, monkeypatch, ext, mode):
    """Grids of at most 16384 points run whole iterations inside one workgroup (csrc/cg_small.hpp: 8 points
    per lane up to 8192 points, 16 beyond);
    CG_SMALL=0 sends the same solve through kernels A and B.  Same iteration count, histories and
    solutions to rounding; and the two kinds of launch can alternate on one solver."""
    N = int(np.prod(ext))
    rng = np.random.default_rng(N)
    b = torch.from_numpy(rng.random(N).astype(np.float32)).cuda()
    monkeypatch.setenv("CG_SMALL", "0")
    x0, it0, res0, h0 = _solve(lib, ext, b, [("it", 200)], mode)
    monkeypatch.setenv("CG_SMALL", "1")
    x1, it1, res1, h1 = _solve(lib, ext, b, [("it", 200)], mode)
    x2, it2, res2, h2 = _solve(lib, ext, b, [("it", 7), ("ab", 3), ("it", 5), ("ab", 2), ("it", 183)], mode)
    for x, it, res, h in ((x1, it1, res1, h1), (x2, it2, res2, h2)):
        assert iters_close(it, it0), (it, it0)
        n = min(len(h), len(h0), 40)
        assert np.allclose(h[:n], h0[:n], rtol=1e-3)
        assert rel_l2(x, x0) <= 1e-4
    # alternating launch kinds changes nothing but the association of the dot products
    assert it2 == it1 or iters_close(it2, it1)


def test_single_workgroup_solver_pauses_and_resumes(lib):
    """iterate(n) on a small grid stops after exactly n iterations and can be continued."""
    ext = [32, 32]
    N = 1024
    b = torch.from_numpy(np.random.default_rng(5).random(N).astype(np.float32)).cuda()
    xa, ita, _, ha = _solve(lib, ext, b, [("it", 30)])
    xb, itb, _, hb = _solve(lib, ext, b, [("it", 10), ("it", 10), ("it", 10)])
    assert ita == itb == 30
    assert np.array_equal(ha, hb) and np.array_equal(xa, xb)


# ---- randomised shapes, error convention ------------------------------------------------------------

def _random_cubic_cases(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        d = int(rng.integers(1, 6))
        L = int(rng.integers(2, 41))
        if 2 <= L ** d <= 300000:
            out.append((L, d))
    return out


@pytest.mark.parametrize("L,d", _random_cubic_cases(24, 2024))
def test_stencil_bit_exact_on_random_shapes(lib, oracle, engine_env, L, d):
    N = L ** d
    u = np.random.default_rng(L * 100 + d).standard_normal(N).astype(np.float32)
    want = oracle.laplace_f32(u, d, L)
    ud = dev(u)
    out = torch.full((N,), np.nan, dtype=torch.float32, device="cuda")
    lib.laplace_gpu_jl(out.data_ptr(), ud.data_ptr(), 0.0, d, L, N, 0, (N + 255) // 256, 256)
    assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize("call,needle", [
    ("lib.conjugate_gradient_gpu_ex(b.data_ptr(), x.data_ptr(), 6, 2, index_mode=2)", "Morton"),
    ("lib.make_solver([16, 16], 0, 1, 0).bind_links(0, [1])", "rank"),
    ("lib.make_solver([6, 6], engine=1)", "rows engine"),
])
def test_precondition_failures_follow_the_reference_error_convention(call, needle):
    """Runtime and precondition failures print 'Error: file:line ... reason: ...' to stderr and exit(1)
    (common.h:7-17); checked in a child process because the library terminates its process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import torch; from __graft_entry__ import load_package; "
            "lib = load_package().load_library(); b = torch.ones(36, device='cuda'); x = torch.zeros(36, device='cuda'); "
            "%s; print('survived')" % (root, call))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1, (r.returncode, r.stdout[-300:], r.stderr[-300:])
    assert "Error:" in r.stderr and needle in r.stderr and "survived" not in r.stdout


# ---- rows that are not a multiple of four points: padded internally ------------------------------

@pytest.mark.parametrize("ext", [[1022, 700], [1001, 1001], [130, 90, 70], [258, 66, 40], [301, 33, 10, 7], [4097, 130],
                                 [500, 40, 30], [1000, 24, 30], [1001, 30, 20], [50, 30, 20, 18]],
                         ids=lambda e: "x".join(map(str, e)))
def test_padded_rows_match_the_scalar_engine(lib, monkeypatch, ext):
    """Large grids whose axis 0 is not a multiple of 4 are kept internally with rows padded to 4 floats
    so that the 16-byte engines run them; CG_PAD_ROWS=0 sends them through the one-point-per-lane
    engine.  (3-D grids with long rows are padded to 32 floats, so that rows start on 128-byte lines.)  Same iterations; x (copied back by finish) agrees to rounding and nothing outside the
    caller's N floats is touched."""
    monkeypatch.delenv("CG_ENGINE", raising=False)   # a forced engine switches the padding off
    N = int(np.prod(ext))
    b = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(b.data_ptr(), N, 77)
    outs = []
    for pad in ("0", "1"):
        monkeypatch.setenv("CG_PAD_ROWS", pad)
        s = lib.make_solver(ext)
        if ext[0] % 4:
            assert (s.engine == 2) == (pad == "0")
        else:  # rows padded to 128-byte lines for the planes engine: the same engine either way
            assert s.engine == 3
        guard = torch.full((N + 64,), 123.0, dtype=torch.float32, device="cuda")
        x = guard[:N]
        x.zero_()
        s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
        s.iterate(25, use_graph=True)
        iters, res, hist = s.finish(hist_len=25)
        assert iters == 25
        assert torch.all(guard[N:] == 123.0)
        outs.append((x.cpu().numpy().copy(), np.array(hist)))
        # a second solve on the same solver, other x buffer, accumulating into a non-zero x
        x2 = torch.ones(N, dtype=torch.float32, device="cuda")
        s.begin(b.data_ptr(), x2.data_ptr(), 0.0, 1)
        s.iterate(25, use_graph=True)
        s.finish()
        assert rel_l2(x2.cpu().numpy() - 1.0, outs[-1][0]) <= 1e-5
        s.close()
    assert np.allclose(outs[0][1], outs[1][1], rtol=1e-5)
    assert rel_l2(outs[1][0], outs[0][0]) <= 1e-5


def test_reference_entry_point_on_an_odd_large_grid(lib, oracle):
    """conjugate_gradient_gpu_ex end to end on 1001^2 (padded rows) against the fp32 emulation."""
    L, d = 1001, 2
    N = L * L
    b = np.random.default_rng(3).random(N).astype(np.float32)
    bd = dev(b)
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    res, iters, hist = lib.conjugate_gradient_gpu_ex(bd.data_ptr(), x.data_ptr(), L, d, maxit=40, hist_len=40)
    xe, rese, iterse, histe = oracle.cg_f32(b, L, d, maxit=40, hist_len=40)
    assert iters == iterse == 40
    assert np.allclose(hist, histe, rtol=1e-3)
    assert rel_l2(x.cpu().numpy(), xe) <= 1e-4


# ---- power iteration together with CG (tasks.md:10-24; not in the reference: parity unpinned, --------
# ---- checked against the operator's analytic spectrum) ------------------------------------------------

@pytest.mark.parametrize("L,d", [(16, 2), (8, 3), (24, 1), (64, 2)])
def test_extreme_eigenvalues_match_the_analytic_spectrum(lib, L, d):
    """The eigenvalues of the 1-D operator are 2 - 2 cos(pi j / (L + 1)), j = 1..L; in d dimensions
    they add up per axis.  Power iteration finds d (2 + 2 cos(pi/(L+1))), inverse iteration through the
    float CG d (2 - 2 cos(pi/(L+1))); the condition number is their ratio."""
    lam1 = 2.0 - 2.0 * np.cos(np.pi / (L + 1))
    lamL = 2.0 + 2.0 * np.cos(np.pi / (L + 1))
    r = lib.eigen_extremes_gpu(L, d, max_steps=20000, tol=1e-9)
    assert abs(r["lambda_max"] - d * lamL) <= 2e-4 * d * lamL, r
    assert abs(r["lambda_min"] - d * lam1) <= 1e-4 * d * lam1, r
    assert r["steps_min"] < 60 and r["cg_iterations"] > 0
    only_max = lib.eigen_extremes_gpu(L, d, max_steps=5, want_min=False)
    assert only_max["steps_max"] == 5 and only_max["steps_min"] == 0 and only_max["lambda_max"] > 0


# ---- two-grid correction scheme (tasks.md:32-36; not in the reference: parity unpinned, checked --------
# ---- through its textbook properties and against the CG solution) --------------------------------------

def _prolong_numpy(ec, Lc, d):
    """d-linear interpolation from Lc^d to (2 Lc + 1)^d, zero boundary, axis 0 fastest."""
    L = 2 * Lc + 1
    e = ec.reshape((Lc,) * d)          # numpy axis order is reversed, the operator is symmetric in the axes
    for ax in range(d):
        shape = list(e.shape)
        shape[ax] = L
        out = np.zeros(shape, dtype=np.float64)
        idx = [slice(None)] * d
        idx[ax] = slice(1, L, 2)
        out[tuple(idx)] = e
        lo, hi = [slice(None)] * d, [slice(None)] * d
        lo[ax], hi[ax] = slice(0, L - 1, 2), slice(2, L + 1, 2)
        out[tuple(lo)] += 0.5 * e
        out[tuple(hi)] += 0.5 * e
        e = out
    return e.reshape(-1)


@pytest.mark.parametrize("L,d", [(63, 2), (31, 3), (255, 1), (15, 4), (127, 2)])
def test_two_grid_cycles_contract_the_residual(lib, oracle, L, d):
    """Damped Jacobi (2+2 steps) with an accurate coarse solve contracts the residual by a factor that
    does not depend on the grid size (textbook: about 0.1 to 0.2 in 2-D); the converged solution is the
    CG solution."""
    N = L ** d
    rng = np.random.default_rng(L * d)
    xt = rng.random(N).astype(np.float32)
    f = oracle.laplace_f32(xt, d, L)
    fd = dev(f)
    u = torch.zeros(N, dtype=torch.float32, device="cuda")
    cycles, hist, inner = lib.two_grid_gpu(fd.data_ptr(), u.data_ptr(), L, d, max_cycles=25, rel=1e-5)
    assert 2 <= cycles < 25 and inner > 0
    factors = [hist[k + 1] / hist[k] for k in range(min(cycles, 5))]
    assert max(factors) < 0.35, factors
    assert hist[cycles] <= 1e-5 * np.linalg.norm(f) * 1.01
    assert np.all(np.abs(u.cpu().numpy() - xt) < 1e-3)
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    lib.conjugate_gradient_gpu_ex(fd.data_ptr(), x.data_ptr(), L, d, maxit=5000, rel=1e-6, lagged=0)
    assert rel_l2(u.cpu().numpy(), x.cpu().numpy()) <= 1e-4


@pytest.mark.parametrize("Lc,d", [(7, 1), (5, 2), (3, 3)])
def test_two_grid_transfer_operators(lib, Lc, d):
    """With no smoothing and the coarse solve switched off by a huge tolerance (e_c = 0) a cycle changes
    nothing; with f = A (P e_c) for a random coarse vector and an exact coarse solve, one cycle of pure
    coarse correction must return P e_c: this pins prolongation (against numpy), restriction (its
    transpose, through the Galerkin-like identity below) and the factor 4."""
    L = 2 * Lc + 1
    N, Nc = L ** d, Lc ** d
    rng = np.random.default_rng(Lc + d)
    ec = rng.random(Nc)
    pe = _prolong_numpy(ec, Lc, d).astype(np.float32)
    ped = dev(pe)
    f = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.laplace_gpu_jl(f.data_ptr(), ped.data_ptr(), 0.0, d, L, N, 0, (N + 255) // 256, 256)
    u = torch.zeros(N, dtype=torch.float32, device="cuda")
    cycles, hist, inner = lib.two_grid_gpu(f.data_ptr(), u.data_ptr(), L, d, nu1=0, nu2=0, max_cycles=1, rel=0.0, coarse_rel=1e9)
    assert cycles == 1 and torch.all(u == 0)          # coarse solve returned 0: nothing prolongated
    # 1-D: R A P = A_c / 4 exactly (Galerkin), so the coarse problem A_c e = 4 R A P e_c has the solution e_c
    if d == 1:
        cycles, hist, inner = lib.two_grid_gpu(f.data_ptr(), u.data_ptr(), L, d, nu1=0, nu2=0, max_cycles=1, rel=0.0, coarse_rel=1e-7)
        assert rel_l2(u.cpu().numpy(), pe) <= 1e-4
        assert hist[1] <= 1e-3 * hist[0]


@pytest.mark.parametrize("L,d", [(63, 2), (31, 3), (127, 2)])
def test_cg_preconditioned_by_a_two_grid_cycle(lib, oracle, L, d):
    """One symmetric two-grid cycle (2+2 Jacobi steps) as M^-1: preconditioned CG needs fewer iterations
    than stand-alone cycles and far fewer than plain CG, and ends at the same solution."""
    N = L ** d
    rng = np.random.default_rng(3 * L + d)
    xt = rng.random(N).astype(np.float32)
    f = oracle.laplace_f32(xt, d, L)
    fd = dev(f)
    u = torch.zeros(N, dtype=torch.float32, device="cuda")
    its, hist, inner = lib.two_grid_pcg_gpu(fd.data_ptr(), u.data_ptr(), L, d, max_iters=30, rel=1e-5)
    cycles, _, _ = lib.two_grid_gpu(fd.data_ptr(), torch.zeros_like(u).data_ptr(), L, d, max_cycles=30, rel=1e-5)
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    _, cg_iters, _ = lib.conjugate_gradient_gpu_ex(fd.data_ptr(), x.data_ptr(), L, d, maxit=5000, rel=1e-5, lagged=0)
    assert 1 <= its <= cycles < cg_iters, (its, cycles, cg_iters)
    assert hist[its] <= 1e-5 * np.linalg.norm(f) * 1.01
    assert np.all(np.abs(u.cpu().numpy() - xt) < 1e-3)   # the reference's recover-x bar (tests.cu:14-41)


@pytest.mark.parametrize("L,d,levels", [(127, 2, 0), (63, 3, 0), (1023, 1, 0), (255, 2, 3), (45, 2, 0)])
def test_multigrid_v_cycles(lib, oracle, L, d, levels):
    """V-cycles over all the levels the grid allows (L = 2^k - 1 coarsens down to 3 points per axis; 45 ->
    22 stops after one coarsening) contract the residual at a rate independent of the size, and with two
    levels the scheme is the two-grid scheme bit for bit."""
    N = L ** d
    rng = np.random.default_rng(L + 7 * d)
    xt = rng.random(N).astype(np.float32)
    f = oracle.laplace_f32(xt, d, L)
    fd = dev(f)
    u = torch.zeros(N, dtype=torch.float32, device="cuda")
    cycles, hist, used, inner = lib.multigrid_gpu(fd.data_ptr(), u.data_ptr(), L, d, levels=levels, max_cycles=30, rel=1e-5)
    want = {(127, 2, 0): 6, (63, 3, 0): 5, (1023, 1, 0): 9, (255, 2, 3): 3, (45, 2, 0): 2}[(L, d, levels)]
    assert used == want, used
    assert 2 <= cycles < 30
    assert max(hist[k + 1] / hist[k] for k in range(min(cycles, 5))) < 0.4
    assert hist[cycles] <= 1e-5 * np.linalg.norm(f) * 1.01
    assert np.all(np.abs(u.cpu().numpy() - xt) < 1e-3)
    if L == 45:
        v = torch.zeros(N, dtype=torch.float32, device="cuda")
        c2, h2, _ = lib.two_grid_gpu(fd.data_ptr(), v.data_ptr(), L, d, max_cycles=30, rel=1e-5)
        assert c2 == cycles and h2 == hist and torch.equal(u, v)



def test_hypercube_of_two_points_per_axis_runs_on_the_quad_kernels(lib, monkeypatch):
    """L = 2: bit k of the lexicographic index is coordinate k, i.e. the index is its own Z-curve index, and
    the solver hands such grids to the quad-per-lane Morton kernels.  Same CG as the one-point-per-lane
    engine (the end point of the reference's dimension sweep, main.jl:85-116, is 2^24)."""
    monkeypatch.delenv("CG_ENGINE", raising=False)
    d = 20
    N = 2 ** d
    b = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(b.data_ptr(), N, 9)
    outs = []
    for engine in (0, 2):
        s = lib.make_solver([2] * d, engine=engine)
        assert s.kernel_name.startswith("k_morton") == (engine == 0), s.kernel_name
        x = torch.zeros(N, dtype=torch.float32, device="cuda")
        s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
        s.iterate(12, use_graph=True)
        iters, res, hist = s.finish(hist_len=12)
        assert iters == 12
        outs.append((x.cpu().numpy(), np.array(hist)))
        s.close()
    assert np.allclose(outs[0][1], outs[1][1], rtol=1e-5)
    assert rel_l2(outs[0][0], outs[1][0]) <= 1e-5


def test_solve_is_ordered_after_the_default_stream(lib):
    """The reference runs everything on the default stream (conjugate-gradient_gpu.cu:327-376), so a
    right-hand side still being produced there when the solver is called is implicitly waited for.  Here
    the solver has a private stream: the entry points must make it wait.  A long chain of default-stream
    work ends with the copy that puts the real right-hand side into b; the call follows without any
    synchronisation and must see that b, not the zeros before it."""
    L, d = 512, 2
    N = L ** d
    g = torch.Generator(device="cuda").manual_seed(3)
    want = torch.rand(N, device="cuda", generator=g)
    x_ref = torch.zeros(N, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    res_ref, it_ref, _ = lib.conjugate_gradient_gpu_ex(want.data_ptr(), x_ref.data_ptr(), L, d, maxit=30)
    b = torch.zeros(N, dtype=torch.float32, device="cuda")
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    big = torch.rand(4096, 4096, device="cuda", generator=g)
    torch.cuda.synchronize()
    acc = big
    for _ in range(40):                     # tens of milliseconds of queued default-stream work
        acc = (acc @ big) * 1e-3
    b.copy_(want + 0.0 * acc[0, 0])         # depends on the whole chain
    res, it, _ = lib.conjugate_gradient_gpu_ex(b.data_ptr(), x.data_ptr(), L, d, maxit=30)
    assert it == it_ref == 30 and res == res_ref
    assert torch.equal(x, x_ref)


def test_history_of_a_solve_longer_than_the_device_ring(lib):
    """The device keeps the last 1024 values of r.r; a longer solve is polled by the host, which keeps
    every value, so the replayed per-iteration print-out (conjugate-gradient_gpu.cu:369) has no gaps."""
    L, d = 1024, 2
    N = L ** d
    b = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(b.data_ptr(), N, 21)
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    res, iters, hist = lib.conjugate_gradient_gpu_ex(b.data_ptr(), x.data_ptr(), L, d, maxit=1500, rel=1e-9, hist_len=1500)
    assert iters == 1500 and len(hist) == 1500
    h = np.array(hist)
    assert np.all(np.isfinite(h)) and np.all(h > 0)
    assert h[0] == np.float32(np.sqrt(np.float64(np.float32((b.double() @ b.double()).item()))))
    assert h[-1] < 1e-2 * h[0]
