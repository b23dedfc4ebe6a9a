"""Slab decomposition over torch.distributed (gloo, CPU, world_size 1 and 2): the driver's
halo exchange and all-reduce placement, with the per-slab compute done by a CPU stand-in."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def run_world(tmp_path, world, L, d, iters, port, heights=None):
    out = str(tmp_path / f"slab_w{world}_L{L}_d{d}.npz")
    extra = ["cpu", ",".join(str(h) for h in heights)] if heights else []
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "slab_worker.py"), out, str(L), str(d),
                                       str(iters)] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        try:
            log, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, log.decode()[-2000:]
    return np.load(out)


@pytest.mark.parametrize("L,d", [(16, 2), (8, 3)])
def test_two_ranks_match_one_rank_and_oracle(tmp_path, oracle, L, d):
    one = run_world(tmp_path, 1, L, d, 200, 29611 + d)
    two = run_world(tmp_path, 2, L, d, 200, 29621 + d)
    xe, rese, iterse, histe = oracle.cg_f32(one["b"], L, d, hist_len=200)
    for r in (one, two):
        assert abs(int(r["iters"]) - iterse) <= max(2, 0.05 * iterse)
        assert np.all(np.abs(r["x"] - r["xt"]) < 1e-3)                      # tests.cu:14-41
        assert np.linalg.norm(r["x"] - xe) <= 1e-4 * np.linalg.norm(xe)
    # the decomposition only re-associates the two dot products
    assert np.linalg.norm(two["x"] - one["x"]) <= 1e-5 * np.linalg.norm(one["x"])
    n = min(len(one["hist"]), len(two["hist"]), 10)
    assert np.allclose(one["hist"][:n], two["hist"][:n], rtol=1e-5)


def test_four_ranks_3d(tmp_path, oracle):
    r = run_world(tmp_path, 4, 8, 3, 200, 29641)
    xe, _, iterse, _ = oracle.cg_f32(r["b"], 8, 3)
    assert abs(int(r["iters"]) - iterse) <= max(2, 0.05 * iterse)
    assert np.linalg.norm(r["x"] - xe) <= 1e-4 * np.linalg.norm(xe)


def test_unequal_slab_heights(tmp_path, oracle):
    """Slabs of different heights produce partial-sum arrays of different lengths; the driver makes every
    rank reduce the longest (an all-reduce over tensors of different length would hang or mis-add)."""
    L, d = 16, 2
    r = run_world(tmp_path, 3, L, d, 200, 29651, heights=[3, 8, 5])
    xe, _, iterse, _ = oracle.cg_f32(r["b"], L, d)
    assert abs(int(r["iters"]) - iterse) <= max(2, 0.05 * iterse)
    assert np.all(np.abs(r["x"] - r["xt"]) < 1e-3)
    assert np.linalg.norm(r["x"] - xe) <= 1e-4 * np.linalg.norm(xe)
