"""The paired x update: kernel B of an even iteration leaves x alone and the next kernel B applies both
updates in order (x <- fma(alpha_k, p_k, x), then x <- fma(alpha_k+1, p_k+1, x)), which must give the bits
of the reference's one update per iteration (conjugate-gradient_gpu.cu:356).  Every test runs the same solve
on two solvers, one created with CG_XDEFER=0, and compares them bit for bit at every point where the host
gets control back: after every cg_solver_iterate (a pending update is flushed there), after a stop by the
tolerance at an unknown parity, across a switch to the step interface and back."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

# rows engine in 2-D and 1-D, padded rows, the planes engine, short rows in 3-D, 4-D and 5-D (k_cols; in 5-D
# the second p is loaded where it is used), the one-point-per-lane engine (6-D), a grid small enough for the
# single-workgroup kernel (which does not pair but shares the solver's flush logic)
GRIDS = [[1024, 1024], [1 << 20], [1002, 600], [256, 64, 64], [64, 64, 128], [64, 64, 64, 64], [32] * 5, [8] * 6,
         [48, 40]]


def pair_of_solvers(lib, ext, monkeypatch, index_mode=0):
    monkeypatch.setenv("CG_XDEFER", "0")
    plain = lib.make_solver(ext, index_mode=index_mode)
    monkeypatch.delenv("CG_XDEFER")
    paired = lib.make_solver(ext, index_mode=index_mode)
    assert not plain.paired_x
    assert paired.paired_x, f"{ext}: engine {paired.kernel_name} does not pair x updates"
    return plain, paired


def rhs(lib, N, seed=5, centred=False):
    """u01, or u01 - 1/2: without its mean the right-hand side is dominated by the short waves CG removes
    first, so the residual falls at once (that of u01 grows for hundreds of iterations)."""
    b = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(b.data_ptr(), N, seed)
    if centred:
        b -= 0.5
    torch.cuda.synchronize()
    return b


RELS = [0.7, 0.5, 0.4, 0.3, 0.25, 0.2, 0.15, 0.1, 0.07, 0.05]


@pytest.mark.parametrize("ext", GRIDS)
@pytest.mark.parametrize("plan", [[1], [3, 4], [7, 1, 1, 2], [20, 5], [45]])
def test_x_is_complete_and_bit_identical_after_every_iterate(lib, monkeypatch, ext, plan):
    N = int(np.prod(ext))
    b = rhs(lib, N)
    plain, paired = pair_of_solvers(lib, ext, monkeypatch)
    padded = ext[0] % 4 != 0          # such a solver works on an internal copy of x until finish()
    xs = [torch.zeros(N, dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    for s, x in zip((plain, paired), xs):
        s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
    for i, n in enumerate(plan):
        for s in (plain, paired):
            s.iterate(n, use_graph=(i % 2 == 0))
            s.sync()
        if not padded:
            assert torch.equal(xs[0], xs[1]), f"after {sum(plan[:i + 1])} iterations"
    out = [s.finish(hist_len=sum(plan)) for s in (plain, paired)]
    assert out[0] == out[1] and out[0][0] == sum(plan)
    assert torch.equal(xs[0], xs[1])
    assert float(xs[0].abs().max()) > 0
    plain.close()
    paired.close()


@pytest.mark.parametrize("ext", [[1024, 1024], [256, 64, 64], [64, 64, 64, 64]])
@pytest.mark.parametrize("rel", RELS)
def test_stop_by_tolerance_at_either_parity(lib, monkeypatch, ext, rel):
    N = int(np.prod(ext))
    b = rhs(lib, N, seed=9, centred=True)
    plain, paired = pair_of_solvers(lib, ext, monkeypatch)
    xs = [torch.zeros(N, dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    out = []
    for s, x in zip((plain, paired), xs):
        s.begin(b.data_ptr(), x.data_ptr(), rel, 1)
        s.iterate(200)
        out.append(s.finish(hist_len=200))
    assert out[0] == out[1]
    assert 0 < out[0][0] < 200, "the tolerance is meant to stop the solve early"
    assert torch.equal(xs[0], xs[1])
    plain.close()
    paired.close()


def test_both_parities_were_hit_by_the_tolerance_sweep(lib, monkeypatch):
    """The sweep above is only a test of the flush if some solves stop after an odd count."""
    ext, N = [1024, 1024], 1 << 20
    b = rhs(lib, N, seed=9, centred=True)
    s = lib.make_solver(ext)
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    counts = []
    for rel in RELS:
        x.zero_()
        torch.cuda.synchronize()
        s.begin(b.data_ptr(), x.data_ptr(), rel, 1)
        s.iterate(200)
        counts.append(s.finish()[0])
    s.close()
    assert any(c % 2 for c in counts) and any(c % 2 == 0 for c in counts), counts


@pytest.mark.parametrize("mode,relayout", [(2, "0"), (2, "1"), (0, "1")])
@pytest.mark.parametrize("L,d", [(64, 3), (32, 4), (128, 3)])
def test_morton_layout(lib, monkeypatch, L, d, mode, relayout):
    """The Z-curve kernels (CG_MORTON_RELAYOUT=0) and the internal permutation pair x updates as well."""
    monkeypatch.setenv("CG_MORTON_RELAYOUT", relayout)
    ext, N = [L] * d, L ** d
    b = rhs(lib, N)
    plain, paired = pair_of_solvers(lib, ext, monkeypatch, index_mode=mode)
    xs = [torch.zeros(N, dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    out = []
    for s, x in zip((plain, paired), xs):
        s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
        for n in (3, 4, 20, 2):
            s.iterate(n)
        out.append(s.finish(hist_len=29))
    assert out[0] == out[1] and out[0][0] == 29
    assert torch.equal(xs[0], xs[1])
    plain.close()
    paired.close()


@pytest.mark.parametrize("ext", [[1024, 1024], [256, 64, 64]])
def test_step_interface_between_paired_iterations(lib, monkeypatch, ext):
    """cg_solver_kernel_B always updates x; iterations launched by cg_solver_iterate before and after it
    pair up from wherever the count stands (odd here)."""
    N = int(np.prod(ext))
    b = rhs(lib, N)
    plain, paired = pair_of_solvers(lib, ext, monkeypatch)
    xs = [torch.zeros(N, dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    for s, x in zip((plain, paired), xs):
        s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
        s.iterate(3)
        for _ in range(2):
            s.add_iterations(1)
            s.kernel_A()
            s.kernel_B()
            s.sync()
    assert torch.equal(xs[0], xs[1])           # 5 iterations, no finish() yet
    for s in (plain, paired):
        s.iterate(6)
        s.sync()
    assert torch.equal(xs[0], xs[1])
    out = [s.finish(hist_len=11) for s in (plain, paired)]
    assert out[0] == out[1] and out[0][0] == 11
    assert torch.equal(xs[0], xs[1])
    plain.close()
    paired.close()


def test_second_solve_on_the_same_solver(lib, monkeypatch):
    """begin() resets the pairing state: a solve that ended with an odd count leaves nothing behind."""
    ext, N = [1024, 1024], 1 << 20
    b = rhs(lib, N)
    plain, paired = pair_of_solvers(lib, ext, monkeypatch)
    xs = [torch.zeros(N, dtype=torch.float32, device="cuda") for _ in range(2)]
    for iters in (7, 7, 4):
        for s, x in zip((plain, paired), xs):
            x.zero_()
            torch.cuda.synchronize()
            s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
            s.iterate(iters)
            s.finish()
        assert torch.equal(xs[0], xs[1])
    plain.close()
    paired.close()


@pytest.mark.parametrize("n", [1, 3, 4])
def test_kernel_timing_on_a_single_workgroup_grid_keeps_x_complete(lib, monkeypatch, n):
    """cg_solver_time_kernels launches kernels A and B (which pair x updates) even on a grid whose
    cg_solver_iterate runs the single-workgroup kernel (which updates x every iteration): an update left
    pending by the last timed kernel B must be flushed, and later iterations must continue from a complete x."""
    ext, N = [48, 40], 48 * 40
    b = rhs(lib, N)
    plain, paired = pair_of_solvers(lib, ext, monkeypatch)
    assert paired.kernel_name == "k_small"
    xs = [torch.zeros(N, dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    plain.begin(b.data_ptr(), xs[0].data_ptr(), 0.0, 1)
    plain.iterate(n + 5)
    paired.begin(b.data_ptr(), xs[1].data_ptr(), 0.0, 1)
    paired.time_kernels(n)
    paired.sync()
    paired.iterate(5)
    out = [s.finish(hist_len=n + 5) for s in (plain, paired)]
    assert out[0] == out[1] and out[0][0] == n + 5
    assert torch.equal(xs[0], xs[1])
    plain.close()
    paired.close()
