"""The CPU oracle against the reference's golden vectors and known answers (no GPU)."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


# ---- known answers held by the reference's own tests --------------------------------

def test_index_known_answers_cpu(oracle):
    # tests.cu:235-307
    assert oracle.get_index([1, 2], 5, 2, 25) == 11
    assert oracle.get_index([5, 5], 5, 2, 25) == 25
    assert oracle.get_index([-1, 0], 5, 2, 25) == 25
    assert oracle.neighbour_index([0, 2], 1, -1, 5, 2, 25) == 5
    assert oracle.neighbour_index([1, 2], 1, -1, 5, 2, 25) == 6


def test_index_known_answers_gpu(oracle):
    # main.jl:61-65 and conjugate-gradient_gpu.cu:287-289
    assert oracle.neighbour_index_gpu(2, 1, 1, 3, 2, 9, 0) == 5
    assert oracle.neighbour_index_gpu(2, 0, 1, 3, 2, 9, 0) == 9
    assert oracle.neighbour_index_gpu(3, 0, -1, 3, 2, 9, 0) == 9
    assert oracle.index_to_cords_gpu(4, 3, 1) == 1
    assert oracle.index_to_cords_gpu(2, 3, 1) == 0
    assert oracle.index_to_cords_gpu(2, 3, 0) == 2


@pytest.mark.parametrize("L,d", [(3, 1), (4, 2), (5, 2), (3, 3), (4, 3), (3, 4), (2, 5)])
def test_gpu_and_cpu_index_flavours_agree(oracle, L, d):
    """neighbour_index_gpu (gpu.cu:128-148) must be consistent with neighbour_index (cpu.cu:54-61)."""
    N = L ** d
    for ind in range(N):
        c = oracle.index_to_cords(ind, L, d)
        for k in range(d):
            assert oracle.index_to_cords_gpu(ind, L, k) == c[k]
            for a in (-1, 0, 1):
                assert oracle.neighbour_index_gpu(ind, k, a, L, d, N, 0) == oracle.neighbour_index(c, k, a, L, d, N)
                assert oracle.neighbour_index_gpu(ind, k, a, L, d, N, 1) == oracle.neighbour_index(c, k, a, L, d, N)


def test_dot_and_norm_known_answers(oracle):
    # tests.cu:212-233
    assert oracle.inner_product([1, 2, 3], [4, 5, 6]) == 32.0
    assert oracle.norm([1, 2, 3]) == np.sqrt(14.0)
    # tests.cu:350-359 (fp32, exact)
    ones = np.ones(1000, np.float32)
    assert oracle.inner_product_f32(ones, ones) == 1000.0


def test_laplace_quadratic_known_answer(oracle):
    # tests.cu:160-202: u = (x^2+y^2)/4 on 5x5, dx = 0.5 -> the unscaled operator gives
    # -laplace(u)*dx^2 = -0.25 at interior points (all inputs exactly representable).
    L, d = 5, 2
    dx = 2.0 / (L - 1)
    u = np.zeros(L * L)
    for i in range(L * L):
        cx, cy = i % L, i // L
        x, y = -1 + cx * dx, -1 + cy * dx
        u[i] = np.float32((x * x + y * y) / 4)
    ddf = oracle.minus_laplace(u, d, L)
    for i in range(L * L):
        cx, cy = i % L, i // L
        if 0 < cx < L - 1 and 0 < cy < L - 1:
            assert ddf[i] == -0.25


# ---- golden vectors produced by the compiled reference ------------------------------

@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "cg_cpu_*.npz"))))
def test_cg_matches_reference_bitwise(oracle, path):
    g = np.load(path)
    L, d = int(g["L"]), int(g["d"])
    b = oracle.minus_laplace(g["x"], d, L)
    assert np.array_equal(b, g["b"]), "stencil differs from the reference"
    xs, ret, iters, _ = oracle.conjugate_gradient(g["b"], L, d)
    assert ret == float(g["ret"])
    assert np.array_equal(xs, g["x_solved"])
    # the reference's own acceptance test (tests.cu:14-26, 44-78)
    assert np.all(np.abs(xs - g["x"]) < 1e-3)
    assert iters > 0


def test_survey_golden_scalars():
    """Values captured from the compiled reference when the build was planned."""
    want = {(64, 1): 4.7294622873200783e-14, (5, 3): 1.5623351468560875e-05,
            (10, 2): 1.3259678524648185e-05, (50, 2): 5.4687667092311695e-05,
            (16, 3): 1.1979987766088884e-04, (8, 4): 1.3911284144635472e-04}
    for (L, d), ret in want.items():
        g = np.load(os.path.join(GOLDEN, f"cg_cpu_L{L}_d{d}.npz"))
        assert float(g["ret"]) == ret
    g = np.load(os.path.join(GOLDEN, "cg_cpu_L5_d3.npz"))
    assert g["x"][0] == 0.84018771715470952
    assert g["b"][0] == 4.0482476125695968


def test_preconditioner_and_pcg_match_reference_bitwise(oracle):
    g = np.load(os.path.join(GOLDEN, "pcg_cpu_L50_d2.npz"))
    L, d = int(g["L"]), int(g["d"])
    xp, res, steps = oracle.preconditioner(g["b"], L, d, 1e-6)
    assert np.array_equal(xp, g["x_precond"]) and res == float(g["res_precond"])
    # step counts for THIS fixture (srand(1) input); bit-equal x and residual pin them,
    # since the reference does not return its counts
    assert steps == 105
    xo, outer, inner = oracle.preconditioned_cg(g["b"], L, d)
    assert np.array_equal(xo, g["x_pcg"])
    assert outer == 4 and inner == [52, 82, 67, 82]
    assert np.all(np.abs(xo - g["x"]) < 1e-3)


def test_live_reference_agrees_when_built(oracle, reference):
    """Where oracle/_ref exists, cross-check on fresh inputs, bit for bit."""
    if not reference.available:
        pytest.skip("oracle/_ref not built on this machine")
    rng = np.random.default_rng(7)
    for L, d in [(7, 2), (6, 3), (33, 1), (4, 4)]:
        N = L ** d
        x = rng.random(N)
        b_ref = reference.minus_laplace(x, d, L)
        assert np.array_equal(oracle.minus_laplace(x, d, L), b_ref)
        xr, rr = reference.conjugate_gradient(b_ref, L, d)
        xo, ro, _, _ = oracle.conjugate_gradient(b_ref, L, d)
        assert rr == ro and np.array_equal(xr, xo)


# ---- Morton -------------------------------------------------------------------------

def test_morton_goldens(oracle):
    with open(os.path.join(GOLDEN, "morton_index.json")) as f:
        g = json.load(f)
    for d, masks in g["masks"].items():
        for k, m in enumerate(masks):
            assert oracle.morton_mask(int(d), k) == m
    # interleave.cuh:14-23 base masks
    base = {1: 0xffffffff, 2: 0x55555555, 3: 0x49249249, 4: 0x11111111, 5: 0x42108421, 6: 0x41041041}
    for d, m in base.items():
        assert oracle.lib.orc_morton_base_mask(d) == m
    for c in g["cases"]:
        d = c["d"]
        assert oracle.morton_from_coords(c["coords"]) == c["index"]
        assert oracle.morton_to_coords(c["index"], d).tolist() == c["decoded"]
        for k in range(d):
            assert oracle.morton_neighbour(c["index"], d, k, +1) == c["neighbours"][k][0]
            assert oracle.morton_neighbour(c["index"], d, k, -1) == c["neighbours"][k][1]
    for x, m, pd, pe in g["pdep_pext"]:
        assert oracle.pdep(x, m) == pd and oracle.pext(x, m) == pe


def test_morton_selftest_printout(oracle):
    """The values the reference's self-test prints (interleave.cuh:122-150) and the 4-D case."""
    i = oracle.morton_from_coords([2, 3])
    assert i == 0xe
    assert oracle.morton_neighbour(i, 2, 1, +1) == 0x24 and oracle.morton_to_coords(0x24, 2).tolist() == [2, 4]
    assert oracle.morton_neighbour(i, 2, 0, +1) == 0xf
    assert oracle.morton_neighbour(i, 2, 1, -1) == 0xc
    assert oracle.morton_neighbour(i, 2, 0, -1) == 0xb
    assert oracle.morton_neighbour(0, 2, 0, -1) == 0x55555555  # outside any grid
    j = oracle.morton_from_coords([127, 0, 5, 127])
    assert j == 0x9999d9d
    plus = [0x18888c8c, 0x9999d9f, 0x9999dd9, 0x81111515]
    minus = [0x9999d9c, 0x2bbbbfbf, 0x9999d99, 0x9999d95]
    for k in range(4):
        assert oracle.morton_neighbour(j, 4, k, +1) == plus[k]
        assert oracle.morton_neighbour(j, 4, k, -1) == minus[k]
    N = 128 ** 4
    assert [p >= N for p in plus] == [True, False, False, True]
    assert [m >= N for m in minus] == [False, True, False, False]


@pytest.mark.parametrize("L,d", [(4, 2), (8, 2), (4, 3), (4, 4)])
def test_morton_stencil_equals_lexicographic_stencil(oracle, L, d):
    """A u in Morton storage is the permutation of A u in lexicographic storage."""
    N = L ** d
    rng = np.random.default_rng(3)
    u = rng.random(N).astype(np.float32)
    perm = oracle.lex_to_morton_perm(L, d)
    assert sorted(perm.tolist()) == list(range(N))
    um = np.zeros(N, np.float32)
    um[perm] = u
    lex = oracle.laplace_f32(u, d, L)
    mor = oracle.laplace_morton_f32(um, d)
    assert np.array_equal(mor[perm], lex)


def test_f32_emulation_tracks_f64(oracle):
    g = np.load(os.path.join(GOLDEN, "cg_cpu_L10_d2.npz"))
    x32, res, iters, hist = oracle.cg_f32(g["b"].astype(np.float32), 10, 2, hist_len=200)
    assert 0 < iters <= 200 and len(hist) == iters
    assert np.all(np.abs(x32 - g["x"]) < 1e-3)  # tests.cu:466-501 acceptance


def test_all_cores_variant_agrees_with_the_serial_oracle(oracle):
    """oracle/cg_oracle_omp.c (the benchmark's separately labelled all-cores CPU figure) runs the same
    loops with OpenMP reductions: same iteration count, solution equal to re-association error."""
    import ctypes as C
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "libcg_oracle_omp.so")
    if not os.path.exists(path):
        pytest.skip("all-cores variant not built")
    lib = C.CDLL(path)
    dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
    lib.orc_cg_omp.argtypes = [dp, dp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.orc_cg_omp.restype = C.c_double
    for L, d in ((50, 2), (12, 3)):
        n = L ** d
        b = np.zeros(n + 1)
        b[:n] = np.random.default_rng(L).random(n)
        x = np.zeros(n + 1)
        it, th = C.c_int(0), C.c_int(0)
        lib.orc_cg_omp(b, x, L, d, -1, C.byref(it), C.byref(th))
        xs, _, its, _ = oracle.conjugate_gradient(b[:n], L, d)
        assert abs(it.value - its) <= 1 and th.value >= 1
        assert np.linalg.norm(x[:n] - xs) <= 1e-9 * np.linalg.norm(xs)
