"""Replay of the float CG loop (conjugate-gradient_gpu.cu:349-370) at the benchmark
configurations' own sizes, where the CPU oracle is out of reach: the same right-hand side
(cg_fill_uniform, seed 1234 -- what bench.py solves), 25 iterations on the library's solver
against 25 iterations of the same recurrence written with torch tensor operations on the device
(fp32 vectors, dot products accumulated in double and rounded to float as the solver's two-stage
reduction does).  Tolerances: every entry of the r.r history within 1e-4 relative, x within
1e-4 relative L2.  The Morton run gets the same vector permuted into Z-order and must reproduce
the lexicographic replay after permuting x back."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ITERS = 25


def torch_stencil(u, ext):
    """A u with zero Dirichlet boundary; per axis ((-u_plus) + 2 u_0) - u_minus, axes added in
    order 0..d-1 (conjugate-gradient_gpu.cu:172-174).  ext[0] is the contiguous axis."""
    d = len(ext)
    v = u.view(*reversed(ext))          # tensor dim d-1-k is grid axis k
    out = torch.zeros_like(v)
    for k in range(d):
        dim = d - 1 - k
        n = ext[k]
        term = 2.0 * v
        # - u_plus: shift towards lower indices along `dim`
        term.narrow(dim, 0, n - 1).sub_(v.narrow(dim, 1, n - 1))
        term.narrow(dim, 1, n - 1).sub_(v.narrow(dim, 0, n - 1))
        out += term
        del term
    return out.view(-1)


def dot32(a, b):
    """Double-accumulated dot product, rounded to float like the solver's (float)block_read_sum."""
    acc = 0.0
    step = 1 << 26                       # bounded temporaries
    for i in range(0, a.numel(), step):
        acc += float(torch.dot(a[i:i + step].double(), b[i:i + step].double()).item())
    return float(np.float32(acc))


def torch_cg(b, ext, iters):
    """conjugate-gradient_gpu.cu:335-370 from x = 0 with scalars in float; returns (x, history) with
    history[i] = sqrt(r.r at the start of iteration i+1's body), the value the reference prints."""
    f32 = np.float32
    r = b.clone()
    p = b.clone()
    x = torch.zeros_like(b)
    rr = f32(dot32(r, r))
    hist = []
    for _ in range(iters):
        Ap = torch_stencil(p, ext)
        alpha = f32(rr / f32(dot32(p, Ap)))
        x.add_(p, alpha=float(alpha))
        r.add_(Ap, alpha=-float(alpha))
        del Ap
        rr_new = f32(dot32(r, r))
        beta = f32(rr_new / rr)
        p.mul_(float(beta)).add_(r)
        hist.append(float(np.sqrt(np.float64(rr))))
        rr = rr_new
    return x, hist


def morton_to_lex_index(L, d):
    """perm[m] = lexicographic index of the point with Morton index m (bit j of coordinate k is bit
    j*d + k of m: interleave.cuh:14-36,83-91)."""
    bits = L.bit_length() - 1
    N = L ** d
    m = torch.arange(N, device="cuda", dtype=torch.int64)
    lex = torch.zeros(N, device="cuda", dtype=torch.int64)
    for k in range(d):
        c = torch.zeros(N, device="cuda", dtype=torch.int64)
        for j in range(bits):
            c |= ((m >> (j * d + k)) & 1) << j
        lex += c * (L ** k)
        del c
    return lex


def run_solver(lib, b, ext, mode):
    N = b.numel()
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    s = lib.make_solver(ext, index_mode=mode)
    s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
    s.iterate(ITERS)
    done, res, hist = s.finish(hist_len=ITERS)
    name = s.kernel_name
    s.close()
    assert done == ITERS
    return x, hist, name


def check(x, hist, x_ref, hist_ref):
    assert len(hist) == ITERS
    for i, (a, c) in enumerate(zip(hist, hist_ref)):
        # the history holds sqrt(r.r): 1e-4 on r.r is 5e-5 on its square root
        assert abs(a - c) <= 5e-5 * abs(c), (i, a, c)
    err = float((x.double() - x_ref.double()).norm().item())
    ref = float(x_ref.double().norm().item())
    assert err <= 1e-4 * ref, (err, ref)


@pytest.mark.parametrize("ext", [[8192, 8192], [1024, 1024, 128], [128, 128, 128, 128]])
def test_replay_lexicographic(lib, ext):
    N = int(np.prod(ext))
    b = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(C.c_void_p(b.data_ptr()), N, 1234)
    x_ref, hist_ref = torch_cg(b, ext, ITERS)
    x, hist, _ = run_solver(lib, b, ext, 0)
    check(x, hist, x_ref, hist_ref)


@pytest.mark.parametrize("relayout", ["1", "0"])
def test_replay_morton_128_4(lib, monkeypatch, relayout):
    """Config "4-D 128^4 Morton": both the default (the solver permutes into lexicographic order
    internally) and the native Z-curve kernels (CG_MORTON_RELAYOUT=0)."""
    monkeypatch.setenv("CG_MORTON_RELAYOUT", relayout)
    L, d = 128, 4
    ext = [L] * d
    N = L ** d
    b = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(C.c_void_p(b.data_ptr()), N, 1234)
    x_ref, hist_ref = torch_cg(b, ext, ITERS)
    perm = morton_to_lex_index(L, d)
    bm = b[perm].contiguous()           # bm[m] = b[lex(m)]
    xm, hist, name = run_solver(lib, bm, ext, 2)
    assert ("morton" in name) == (relayout == "0"), name
    x = torch.empty_like(xm)
    x[perm] = xm
    check(x, hist, x_ref, hist_ref)


def test_benchmark_of_a_morton_ordered_grid_reports_both_paths(tmp_path):
    """bench.py on vectors in Morton order: `value` is a whole 200-iteration solve with the permutation passes of
    the lexicographic-inside path, and the Z-curve kernels are timed beside it with their own roofline object."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "128x128x128:2", "--steps", "10",
                        "--warmup", "2", "--repeats", "3", "--no-cpu-baseline", "--no-traffic"],
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    m = out["morton"]
    assert out["config"]["index_mode"] == 2 and "lexicographic" in out["config"]["ran"]
    assert out["value"] == m["iterations_per_sec_of_a_200_iteration_solve"] < m["hot_loop_iterations_per_sec"]
    assert m["begin_finish_ms"] > 0 and m["native"]["roofline"]["kernel"].startswith("k_morton")
    assert 0 < m["native"]["hbm_frac_moved"] <= 1 and 0 < out["roofline"]["frac"] <= 1
