"""Direct exchange between slab ranks through device memory (csrc/cg_links.hpp): the ranks
write their dot products and r's boundary planes into each other's boxes from a small kernel
after A and B.  On one GPU the "other ranks" are further solvers of the same process (plain
pointers) or further processes (IPC handles); the kernels and the protocol are the ones that
run across xGMI."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(autouse=True)
def short_waits(monkeypatch):
    # a protocol bug must fail the test in seconds, not sit in a kernel for the default budget
    monkeypatch.setenv("CG_LINKS_TIMEOUT_MS", "3000")


def run_links(lib, ext_of, nranks, bs, iters, use_graph, engine=0):
    solvers = [lib.make_solver(ext_of, 0, int(r > 0), int(r < nranks - 1), engine=engine) for r in range(nranks)]
    nbytes = solvers[0].box_bytes()
    boxes = [lib.links_alloc(nbytes) for _ in range(nranks)]
    for r, s in enumerate(solvers):
        s.bind_links(r, boxes)
    xs = [torch.zeros_like(b) for b in bs]
    torch.cuda.synchronize()
    try:
        for s, b, x in zip(solvers, bs, xs):
            s.begin(b.data_ptr(), x.data_ptr(), 1e-6, 1)
        for s in solvers:
            s.iterate(iters, use_graph)
        outs = [s.finish(hist_len=iters) for s in solvers]
        errs = [s.links_error() for s in solvers]
    finally:
        torch.cuda.synchronize()
        for s in solvers:
            s.close()
        for b in boxes:
            lib.links_free(b)
    assert errs == [0] * nranks, errs
    assert len({o[0] for o in outs}) == 1, "ranks disagree on the iteration count"
    assert len({o[1] for o in outs}) == 1, "ranks disagree on the residue"
    return torch.cat(xs).cpu().numpy(), outs[0][0], outs[0][2]


@pytest.mark.parametrize("use_graph", [False, True])
# two ranks only: a process has four hardware queues, and two spinning solvers whose streams share
# one would wait for each other forever (more ranks: the multi-process test below)
@pytest.mark.parametrize("L,d,nranks", [(16, 2, 2), (64, 2, 2), (16, 3, 2), (10, 2, 2), (6, 3, 2), (8, 4, 2), (32, 3, 2)])
def test_linked_ranks_match_single_gpu_and_oracle(lib, oracle, engine_env, L, d, nranks, use_graph):
    N = L ** d
    rng = np.random.default_rng(L + d)
    xt = rng.random(N).astype(np.float32)
    b = oracle.laplace_f32(xt, d, L)
    n_loc = N // nranks
    bs = [torch.from_numpy(b[r * n_loc:(r + 1) * n_loc].copy()).cuda() for r in range(nranks)]
    x, iters, hist = run_links(lib, [L] * (d - 1) + [L // nranks], nranks, bs, 200, use_graph)
    xe, rese, iterse, histe = oracle.cg_f32(b, L, d, hist_len=200)
    assert abs(iters - iterse) <= max(2, 0.05 * iterse), (iters, iterse)
    assert np.all(np.abs(x - xt) < 1e-3)                                   # tests.cu:14-41
    assert np.linalg.norm(x - xe) <= 1e-4 * np.linalg.norm(xe)
    x1 = torch.zeros(N, dtype=torch.float32, device="cuda")
    bd = torch.from_numpy(b).cuda()
    _, it1, h1 = lib.conjugate_gradient_gpu_ex(bd.data_ptr(), x1.data_ptr(), L, d, hist_len=200)
    assert abs(iters - it1) <= max(2, 0.05 * it1)
    assert np.linalg.norm(x - x1.cpu().numpy()) <= 1e-5 * np.linalg.norm(xe)
    n = min(len(hist), len(h1), 8)
    assert np.allclose(hist[:n], h1[:n], rtol=1e-5)


def test_one_linked_rank_is_bit_identical_to_the_plain_solver(lib):
    """world = 1: the exchange adds 0.0 + the same deterministic sum, so nothing may change."""
    L, d = 256, 2
    b = torch.empty(L * L, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(b.data_ptr(), L * L, 7)
    x, iters, hist = run_links(lib, [L, L], 1, [b], 60, True)
    plain = lib.make_solver([L, L])
    x1 = torch.zeros_like(b)
    plain.begin(b.data_ptr(), x1.data_ptr(), 1e-6, 1)
    plain.iterate(60, True)
    n1, _, h1 = plain.finish(hist_len=60)
    plain.close()
    assert iters == n1 == 60
    assert hist == h1
    assert np.array_equal(x, x1.cpu().numpy())


def test_two_solves_in_a_row_and_wide_planes(lib, oracle):
    """Stamps keep counting across solves; planes wider than one workgroup's share use the
    multi-workgroup push (3-D, 128 x 128 planes = 64 KiB -> forced to 4 workgroups)."""
    os.environ["CG_LINKS_BLOCKS"] = "4"
    try:
        L, d, nranks = 128, 3, 2
        ext = [L, L, 8]
        n_loc = L * L * 8
        bs = []
        for r in range(nranks):
            t = torch.empty(n_loc, dtype=torch.float32, device="cuda")
            lib.c.cg_fill_uniform(t.data_ptr(), n_loc, 11 + r)
            bs.append(t)
        solvers = [lib.make_solver(ext, 0, int(r > 0), int(r < nranks - 1)) for r in range(nranks)]
        boxes = [lib.links_alloc(solvers[0].box_bytes()) for _ in range(nranks)]
        for r, s in enumerate(solvers):
            s.bind_links(r, boxes)
        results = []
        for rep in range(2):
            xs = [torch.zeros_like(b) for b in bs]
            torch.cuda.synchronize()
            for s, b, x in zip(solvers, bs, xs):
                s.begin(b.data_ptr(), x.data_ptr(), 1e-6, 1)
            for s in solvers:
                s.iterate(45, True)          # not a multiple of the graph's 20 iterations
            outs = [s.finish(hist_len=45) for s in solvers]
            assert [s.links_error() for s in solvers] == [0, 0]
            results.append((torch.cat(xs).cpu().numpy(), outs[0]))
        assert results[0][1] == results[1][1]
        assert np.array_equal(results[0][0], results[1][0])
        # against one solver on the whole 128 x 128 x 16 grid
        whole = lib.make_solver([L, L, 16])
        bw = torch.cat(bs)
        xw = torch.zeros_like(bw)
        whole.begin(bw.data_ptr(), xw.data_ptr(), 1e-6, 1)
        whole.iterate(45, True)
        nw, resw, hw = whole.finish(hist_len=45)
        whole.close()
        n, res, hist = results[0][1]
        assert n == nw
        assert np.allclose(hist, hw, rtol=1e-4)
        xr = results[0][0]
        assert np.linalg.norm(xr - xw.cpu().numpy()) <= 1e-5 * np.linalg.norm(xr)
        torch.cuda.synchronize()
        for s in solvers:
            s.close()
        for bx in boxes:
            lib.links_free(bx)
    finally:
        del os.environ["CG_LINKS_BLOCKS"]


def test_missing_rank_times_out_and_reports(lib, monkeypatch):
    """A rank whose peer never shows up must come back (bounded wait) with the error flag set."""
    monkeypatch.setenv("CG_LINKS_TIMEOUT_MS", "200")
    ext = [64, 32]
    s = lib.make_solver(ext, 0, 0, 1)
    boxes = [lib.links_alloc(s.box_bytes()) for _ in range(2)]
    s.bind_links(0, boxes)
    b = torch.ones(64 * 32, dtype=torch.float32, device="cuda")
    x = torch.zeros_like(b)
    s.begin(b.data_ptr(), x.data_ptr(), 1e-6, 1)
    s.iterate(5, False)
    n, res, _ = s.finish()
    assert s.links_error() == 1
    assert n == 0                                   # NaN sums stop the solve at once
    s.close()
    for bx in boxes:
        lib.links_free(bx)


def run_world(tmp_path, world, L, d, iters, port):
    out = str(tmp_path / f"links_w{world}_L{L}_d{d}.npz")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", CG_LINKS_TIMEOUT_MS="5000")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "slab_worker.py"), out, str(L), str(d),
                                       str(iters), "links"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
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


@pytest.mark.parametrize("L,d,world", [(64, 2, 2), (16, 3, 4), (6, 3, 3)])
def test_processes_linked_through_ipc(tmp_path, lib, oracle, L, d, world):
    """One process per rank, boxes mapped through HIP IPC handles (the path a multi-GPU node
    takes), all on cuda:0."""
    r = run_world(tmp_path, world, L, d, 200, 29670 + world)
    xe, rese, iterse, histe = oracle.cg_f32(r["b"], L, d, hist_len=200)
    assert abs(int(r["iters"]) - iterse) <= max(2, 0.05 * iterse)
    assert np.all(np.abs(r["x"] - r["xt"]) < 1e-3)
    assert np.linalg.norm(r["x"] - xe) <= 1e-4 * np.linalg.norm(xe)


def test_benchmark_with_two_ranks_on_one_gpu_as_the_launcher_starts_it(tmp_path):
    """bench.py exactly as the multi-GPU launch runs it (torch.distributed.run, one process per rank), with
    both ranks on this box's one GPU and gloo instead of RCCL (which refuses two ranks on a device): the
    collectives are timed, the boxes are probed in child processes, the fastest usable transport runs the
    timed regions, rank 0 prints one JSON line."""
    import json
    env = dict(os.environ, CG_BENCH_DEVICE="0")
    for k in list(env):
        if k.startswith("TORCHELASTIC_") or k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29683", os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "40",
           "--warmup", "5", "--repeats", "3", "--workload", "1024x1024"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "slab2" and out["scaling"] == "weak"
    assert out["config"]["comm"].startswith(("xgmi-links", "collectives")), out["config"]["comm"]
    assert out["value"] > 0 and np.isfinite(out["final_residue"])
    assert abs(out["value"] - 2 * out["iterations_per_sec_global"]) <= 1e-3 * out["value"]


def test_benchmark_started_plainly_with_two_gpus_starts_two_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it (the way the single-GPU benchmark is started) must
    start the two rank processes itself and report n_gpus = 2 -- here with both ranks on this box's one GPU and
    gloo instead of RCCL.  The default multi-rank workload: 8192 x 8192 per GPU as `value`, and the 1024 x 1024 x 128
    slab of the 8-GPU configuration as `config3`, each with the same slab on one GPU beside it."""
    import json
    env = dict(os.environ, CG_BENCH_DEVICE="0")
    for k in list(env):
        if k.startswith("TORCHELASTIC_") or k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k)
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "6",
           "--warmup", "2", "--repeats", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "slab2" and out["scaling"] == "weak"
    assert out["config"]["workload"] == "2d-8192" and out["config"]["global_extents"] == [8192, 16384]
    assert out["value"] > 0 and out["single_gpu_reference"]["value"] > 0
    assert 0 < out["roofline"]["frac"] <= 1 and 0 < out["hbm_frac_moved"] <= 1
    c3 = out["config3"]
    assert c3["config"]["local_extents"] == [1024, 1024, 128] and c3["config"]["global_extents"] == [1024, 1024, 256]
    assert c3["value"] > 0 and c3["single_gpu_reference"]["value"] > 0
    # a mismatch between --gpus and the ranks that exist is an error, never a silent single-GPU run
    env1 = dict(env, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2"], env=env1,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
