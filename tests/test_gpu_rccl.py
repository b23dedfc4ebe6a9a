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
