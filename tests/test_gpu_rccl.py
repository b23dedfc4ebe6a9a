"""The in-library RCCL transport (csrc/cg_rccl.hip) with one rank: the communicator is created from
a unique id exactly as a multi-rank job does it, the all-reduces of the partial sums run after every
kernel and are captured in the solver's hipGraph.  (RCCL refuses two ranks on one device, so more
ranks cannot be rehearsed on a one-GPU box; the multi-rank logic around it -- ghost planes, common
partial count -- is covered by the virtual-rank and gloo tests.)  With one rank an all-reduce is a
copy, so the bound solver must reproduce the plain solver bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("ext", [[256, 256], [64, 64, 64], [16, 16, 16, 16]])
def test_one_rccl_rank_is_bit_identical_to_the_plain_solver(lib, ext):
    N = int(np.prod(ext))
    b = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(b.data_ptr(), N, 11)
    iters = 45                                     # two 20-iteration graphs and a 5-iteration tail graph
    x0 = torch.zeros(N, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    s0 = lib.make_solver(ext)
    s0.begin(b.data_ptr(), x0.data_ptr(), 0.0, 1)
    s0.iterate(iters)
    n0, res0, hist0 = s0.finish(hist_len=iters)
    s0.close()

    comm = lib.rccl_comm_create(lib.rccl_unique_id(), 0, 1)
    x1 = torch.zeros(N, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    s1 = lib.make_solver(ext)
    s1.bind_rccl(comm, 0, 1)
    for _ in range(2):                              # a second solve on the same binding and graphs
        x1.zero_()
        torch.cuda.synchronize()
        s1.begin(b.data_ptr(), x1.data_ptr(), 0.0, 1)
        s1.iterate(iters)
        n1, res1, hist1 = s1.finish(hist_len=iters)
        assert n1 == n0 == iters
        assert hist1 == hist0 and res1 == res0
        assert torch.equal(x1, x0)
    # unbinding gives the plain solver back
    s1.bind_rccl(None, 0, 1)
    x1.zero_()
    torch.cuda.synchronize()
    s1.begin(b.data_ptr(), x1.data_ptr(), 0.0, 1)
    s1.iterate(iters)
    n2, res2, hist2 = s1.finish(hist_len=iters)
    assert hist2 == hist0 and torch.equal(x1, x0)
    s1.close()
    lib.rccl_comm_destroy(comm)


def mirror_reference(lib, ext, b, plan):
    """The step interface on a slab whose lower and upper neighbour are the slab itself: after every kernel B
    (and at the start) the ghost planes of r receive r's own boundary plane of that side."""
    from conjugate_gradient_amd.slab import HipSlabOps
    N = int(np.prod(ext))
    o = HipSlabOps(lib, ext, 1, 1, torch.device("cuda", 0))
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    g, pl = o.ghost_offset, o.plane

    def mirror():
        with o.stream_ctx():
            o.r_buf[g - pl:g] = o.r_buf[g:g + pl]
            o.r_buf[g + N:g + N + pl] = o.r_buf[g + N - pl:g + N]

    with o.stream_ctx():
        o.begin_local(b, x)
        o.init_state(0.0, 1, 0)
    mirror()
    for n in plan:
        with o.stream_ctx():
            o.add_iterations(n)
        for _ in range(n):
            with o.stream_ctx():
                o.kernel_A()
                o.kernel_B()
            mirror()
    out = o.finish(hist_len=sum(plan))
    torch.cuda.synchronize()
    o.solver.close()
    return x, out


@pytest.mark.parametrize("overlap_bytes", ["0", "1000000000"], ids=["beside_kernel_B", "after_kernel_B"])
@pytest.mark.parametrize("ext", [[256, 256], [1024, 512], [64, 64, 16], [256, 64, 32], [1024, 256, 16], [32, 32, 32, 32], [16, 16, 16, 8, 8]])
def test_ring_of_one_rank_runs_the_whole_multi_rank_machinery(lib, monkeypatch, ext, overlap_bytes):
    """One RCCL rank that is its own lower and upper neighbour (cg_solver_bind_rccl_peers with lo = hi = 0): the
    all-gathers of the per-rank sums, the side stream, the boundary planes computed ahead of kernel B and the
    grouped ncclSend / ncclRecv into the ghost planes all run as they do between GPUs -- from plain launches
    and replayed from the captured graph -- and must reproduce the step interface with hand-copied planes bit
    for bit.  Both forms of the plane exchange: beside kernel B on the second stream (what planes of at least
    CG_RCCL_OVERLAP_BYTES take, 256 KiB by default) and after kernel B on the solver's stream."""
    monkeypatch.setenv("CG_RCCL_OVERLAP_BYTES", overlap_bytes)
    N = int(np.prod(ext))
    b = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(b.data_ptr(), N, 13)
    torch.cuda.synchronize()
    plan = [3, 45, 2]                              # graphs of 20, tails of 3, 5 and 2
    x0, out0 = mirror_reference(lib, ext, b, plan)

    comm = lib.rccl_comm_create(lib.rccl_unique_id(), 0, 1)
    s = lib.make_solver(ext, 0, 1, 1)
    s.bind_rccl(comm, 0, 1, peers=(0, 0))
    assert s.rccl_ranks() == 1
    x1 = torch.zeros(N, dtype=torch.float32, device="cuda")
    for use_graph in (True, False, True):
        x1.zero_()
        torch.cuda.synchronize()
        s.begin(b.data_ptr(), x1.data_ptr(), 0.0, 1)
        for n in plan:
            s.iterate(n, use_graph=use_graph)
        out1 = s.finish(hist_len=sum(plan))
        assert out1 == out0, (use_graph, out1[0], out0[0])
        assert torch.equal(x1, x0)
    msA, msB = s.time_kernels(4)
    assert msA > 0 and msB > 0
    s.bind_rccl(None, 0, 1)
    s.close()
    lib.rccl_comm_destroy(comm)


def test_slab_driver_over_rccl_world_1(tmp_path):
    """bench.py's multi-rank path with one rank over the nccl backend: SlabCG.enable_rccl() carries the
    unique id through torch.distributed, the timed loop runs from the library's graph."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29671", RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-slab", "--comm", "rccl", "--workload",
                        "1024x1024", "--steps", "25", "--warmup", "5", "--repeats", "2", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    import json
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["config"]["comm"].startswith("rccl"), out["config"]
    assert out["value"] > 0 and out["config"]["parallelism"] == "slab1"
