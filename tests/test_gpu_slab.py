"""Slab code path of the HIP kernels on one GPU: several "virtual ranks" share the device,
ghost planes are copied by hand and the dot products summed on the host.  This exercises
exactly what each rank runs under torch.distributed (ghost-plane reads in both kernels, the
local rebuild of p on ghost planes, all-reduced scalars through bind_sums)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def run_virtual(pkg, lib, L, d, nranks, b, iters, engine=0, heights=None, ahead=False):
    from conjugate_gradient_amd.slab import HipSlabOps
    dev = torch.device("cuda", 0)
    heights = heights or [L // nranks] * nranks
    assert sum(heights) == L and len(heights) == nranks
    plane = L ** (d - 1)
    slabs, xs, bs, n_locs = [], [], [], []
    for r in range(nranks):
        ext = [L] * (d - 1) + [heights[r]]
        n_loc, first = heights[r] * plane, sum(heights[:r]) * plane
        o = HipSlabOps(lib, ext, int(r > 0), int(r < nranks - 1), dev)
        slabs.append(o)
        n_locs.append(n_loc)
        bs.append(torch.from_numpy(b[first:first + n_loc].copy()).cuda())
        xs.append(torch.zeros(n_loc, dtype=torch.float32, device=dev))
    # slabs of different shape launch different numbers of workgroups: every rank reduces the largest count
    most = max(o.nparts for o in slabs)
    for o in slabs:
        if o.nparts != most:
            o.set_nparts(most)

    def sync():
        torch.cuda.synchronize()

    def allreduce(which):
        sync()
        tot = sum(o.sum_tensor(which) for o in slabs)  # element-wise, fixed rank order
        for o in slabs:
            o.sum_tensor(which).copy_(tot)
        sync()

    def exchange(from_edge=False):
        """from_edge: the planes that travel are the ones computed ahead of kernel B (cg_solver_edge_*), which
        must be the bits kernel B left in r's boundary planes."""
        sync()
        for r, o in enumerate(slabs):
            g, n, pl = o.ghost_offset, n_locs[r], o.plane
            if r > 0:
                lo = slabs[r - 1]
                glo, nlo = lo.ghost_offset, n_locs[r - 1]
                plane = lo.r_buf[glo + nlo - pl:glo + nlo]
                if from_edge:
                    assert torch.equal(lo.edge_hi, plane), f"rank {r - 1}: upper plane computed ahead differs from r"
                    plane = lo.edge_hi
                o.r_buf[g - pl:g] = plane
            if r < nranks - 1:
                hi = slabs[r + 1]
                ghi = hi.ghost_offset
                plane = hi.r_buf[ghi:ghi + pl]
                if from_edge:
                    assert torch.equal(hi.edge_lo, plane), f"rank {r + 1}: lower plane computed ahead differs from r"
                    plane = hi.edge_lo
                o.r_buf[g + n:g + n + pl] = plane
        sync()

    for o, bb, xx in zip(slabs, bs, xs):
        o.begin_local(bb, xx)
    allreduce(1)
    for o in slabs:
        o.init_state(1e-6, 1, 0)
    exchange()
    for o in slabs:
        o.add_iterations(iters)
    for _ in range(iters):
        for o in slabs:
            if ahead:
                o.edge_copy()
            o.kernel_A()
        allreduce(0)
        for o in slabs:
            if ahead:
                o.edge_update()
            o.kernel_B()
        allreduce(1)
        exchange(ahead)
    outs = [o.finish(hist_len=iters) for o in slabs]
    sync()
    x = torch.cat(xs).cpu().numpy()
    assert len({o[0] for o in outs}) == 1, "ranks disagree on the iteration count"
    return x, outs[0][0], outs[0][2]


@pytest.mark.parametrize("L,d,nranks", [(16, 2, 2), (64, 2, 4), (16, 3, 2), (16, 3, 4), (10, 2, 2), (6, 3, 3), (8, 4, 2), (32, 3, 2), (16, 4, 2)])
def test_virtual_ranks_match_single_gpu_and_oracle(pkg, lib, oracle, engine_env, L, d, nranks):
    N = L ** d
    rng = np.random.default_rng(L + d)
    xt = rng.random(N).astype(np.float32)
    b = oracle.laplace_f32(xt, d, L)
    x, iters, hist = run_virtual(pkg, lib, L, d, nranks, b, 200)
    xe, rese, iterse, histe = oracle.cg_f32(b, L, d, hist_len=200)
    assert abs(iters - iterse) <= max(2, 0.05 * iterse), (iters, iterse)
    assert np.all(np.abs(x - xt) < 1e-3)
    assert np.linalg.norm(x - xe) <= 1e-4 * np.linalg.norm(xe)
    # single solver on the whole grid
    x1 = torch.zeros(N, dtype=torch.float32, device="cuda")
    bd = torch.from_numpy(b).cuda()
    _, it1, h1 = lib.conjugate_gradient_gpu_ex(bd.data_ptr(), x1.data_ptr(), L, d, hist_len=200)
    assert abs(iters - it1) <= max(2, 0.05 * it1)
    assert np.linalg.norm(x - x1.cpu().numpy()) <= 1e-5 * np.linalg.norm(xe)
    n = min(len(hist), len(h1), 8)
    assert np.allclose(hist[:n], h1[:n], rtol=1e-5)


@pytest.mark.parametrize("L,d,nranks", [(64, 2, 4), (16, 3, 2), (10, 2, 2), (6, 3, 3), (8, 4, 2), (32, 3, 2), (16, 4, 2), (256, 3, 4),
                                        (1024, 2, 2), (16, 5, 2)])
def test_boundary_planes_computed_ahead_of_kernel_B_are_its_bits(pkg, lib, engine_env, L, d, nranks):
    """cg_solver_edge_copy + cg_solver_edge_update produce, before kernel B runs, the two boundary planes of r
    that kernel B then writes -- bit for bit on every rank and in every iteration (run_virtual compares them
    after each kernel B and hands them to the neighbours instead of r's own planes), so the solve is the
    same solve: same history, same x as with r's planes."""
    N = L ** d
    bd = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(bd.data_ptr(), N, 21)
    b = bd.cpu().numpy()
    iters = 12 if N > (1 << 22) else 40
    x0, it0, h0 = run_virtual(pkg, lib, L, d, nranks, b, iters)
    x1, it1, h1 = run_virtual(pkg, lib, L, d, nranks, b, iters, ahead=True)
    assert it0 == it1 and h0 == h1
    assert np.array_equal(x0, x1)


@pytest.mark.parametrize("L,nranks,keep_mb", [(1024, 2, "1"), (1024, 4, "0"), (512, 2, "1")])
def test_slabs_that_keep_only_part_of_r_in_the_cache_change_no_bit(pkg, lib, monkeypatch, L, nranks, keep_mb):
    """2-D slabs larger than the cache budget (CG_KEEP_MB, shrunk here so that small slabs qualify) keep only their
    first rows of r in the Infinity Cache and stream the rest: the same solve bit for bit, ghost rows and the
    planes pushed to the neighbours included."""
    N = L * L
    bd = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(bd.data_ptr(), N, 5)
    b = bd.cpu().numpy()
    x0, it0, h0 = run_virtual(pkg, lib, L, 2, nranks, b, 30, ahead=True)
    monkeypatch.setenv("CG_KEEP_MB", keep_mb)
    monkeypatch.setenv("CG_CACHE_POLICY", "4")
    x1, it1, h1 = run_virtual(pkg, lib, L, 2, nranks, b, 30, ahead=True)
    assert it0 == it1 and h0 == h1
    assert np.array_equal(x0, x1)


@pytest.mark.parametrize("L,d,heights", [(16, 3, [6, 10]), (64, 2, [10, 30, 24]), (32, 3, [4, 20, 8])])
def test_virtual_ranks_of_unequal_height(pkg, lib, oracle, engine_env, L, d, heights):
    """Slabs of different heights launch different numbers of workgroups, so their arrays of partial sums
    differ in length; every rank reduces the longest (cg_solver_bind_partials_n) and the solve agrees with
    one solver on the whole grid."""
    N = L ** d
    rng = np.random.default_rng(3 * L + d)
    xt = rng.random(N).astype(np.float32)
    b = oracle.laplace_f32(xt, d, L)
    x, iters, hist = run_virtual(pkg, lib, L, d, len(heights), b, 200, heights=heights)
    assert np.all(np.abs(x - xt) < 1e-3)
    x1 = torch.zeros(N, dtype=torch.float32, device="cuda")
    bd = torch.from_numpy(b).cuda()
    _, it1, h1 = lib.conjugate_gradient_gpu_ex(bd.data_ptr(), x1.data_ptr(), L, d, hist_len=200)
    assert abs(iters - it1) <= max(2, 0.05 * it1)
    assert np.linalg.norm(x - x1.cpu().numpy()) <= 1e-5 * np.linalg.norm(x)
    n = min(len(hist), len(h1), 8)
    assert np.allclose(hist[:n], h1[:n], rtol=1e-5)


@pytest.mark.parametrize("L,d,nranks,R,stream", [(16, 4, 2, 2, 1), (16, 4, 4, 1, 0), (16, 5, 2, 2, 1), (16, 5, 2, 2, 0),
                                                  (32, 3, 2, 2, 1), (64, 3, 4, 1, 0), (24, 4, 2, 2, 1), (48, 3, 2, 2, 0),
                                                  (20, 4, 2, 1, 0)])
def test_blocked_columns_on_slabs(pkg, lib, monkeypatch, L, d, nranks, R, stream):
    """Slabs of 4-D and 5-D grids with short rows run the columns-in-blocks kernels with ghost planes
    on the marching axis: same solution as one solver on the whole grid."""
    monkeypatch.setenv("CG_ENGINE", "1")
    monkeypatch.setenv("CG_COLS_R", str(R))
    monkeypatch.setenv("CG_COLS_STREAM", str(stream))
    monkeypatch.setenv("CG_COLS_EXACT", "1")   # rows of 24, 48, 20 points: exactly n0/4 lanes each
    N = L ** d
    bd = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(bd.data_ptr(), N, 5)
    b = bd.cpu().numpy()
    x, iters, hist = run_virtual(pkg, lib, L, d, nranks, b, 40)
    monkeypatch.setenv("CG_ENGINE", "2")
    x1 = torch.zeros(N, dtype=torch.float32, device="cuda")
    _, it1, h1 = lib.conjugate_gradient_gpu_ex(bd.data_ptr(), x1.data_ptr(), L, d, maxit=40, rel=1e-6, hist_len=40)
    assert iters == it1
    n = min(len(hist), len(h1))
    assert np.allclose(hist[:n], h1[:n], rtol=1e-4)
    assert np.linalg.norm(x - x1.cpu().numpy()) <= 1e-5 * np.linalg.norm(x)


def run_world_hip(tmp_path, world, L, d, iters, port):
    here = os.path.dirname(os.path.abspath(__file__))
    out = str(tmp_path / f"hip_w{world}_L{L}_d{d}.npz")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(here, "slab_worker.py"), out, str(L), str(d),
                                       str(iters), "hip"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            log, _ = p.communicate(timeout=100)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append((p.returncode, log.decode()[-2000:]))
    assert all(rc == 0 for rc, _ in logs), logs
    return np.load(out)


@pytest.mark.parametrize("L,d,world", [(64, 2, 2), (16, 3, 4)])
def test_processes_sharing_the_gpu(tmp_path, lib, oracle, L, d, world):
    """One process per rank through torch.distributed (SlabCG as bench.py drives it), every
    rank's kernels on cuda:0, planes and partials exchanged over gloo."""
    r = run_world_hip(tmp_path, world, L, d, 200, 29660 + world)
    xe, rese, iterse, histe = oracle.cg_f32(r["b"], L, d, hist_len=200)
    assert abs(int(r["iters"]) - iterse) <= max(2, 0.05 * iterse)
    assert np.all(np.abs(r["x"] - r["xt"]) < 1e-3)
    assert np.linalg.norm(r["x"] - xe) <= 1e-4 * np.linalg.norm(xe)
    bd = torch.from_numpy(r["b"]).cuda()
    x1 = torch.zeros(L ** d, dtype=torch.float32, device="cuda")
    _, it1, h1 = lib.conjugate_gradient_gpu_ex(bd.data_ptr(), x1.data_ptr(), L, d, hist_len=200)
    assert abs(int(r["iters"]) - it1) <= max(2, 0.05 * it1)
    assert np.linalg.norm(r["x"] - x1.cpu().numpy()) <= 1e-5 * np.linalg.norm(xe)
    n = min(len(r["hist"]), len(h1), 8)
    assert np.allclose(r["hist"][:n], h1[:n], rtol=1e-5)
