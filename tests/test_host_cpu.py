"""CPU-side tests of the product library: exports, host solver vs the oracle (no GPU calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def test_library_exports_every_declared_symbol(pkg, lib):
    hdr = open(os.path.join(ROOT, "include", "conjugate_gradient_gpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b([a-z_][a-z_0-9A-Z]*)\s*\(", hdr)) - {"defined"}
    declared = {n for n in declared if n.startswith(("cg_", "conjugate", "inner_", "neighbour", "strong_", "time_",
                                                     "add_jl", "laplace_gpu_jl", "preconditioned"))}
    assert declared == set(pkg.binding.EXPORTS)
    for name in declared:
        assert hasattr(lib.c, name), name


def test_neighbour_index_host_entry_matches_oracle(lib, oracle):
    for L, d in [(3, 2), (5, 2), (4, 3), (3, 4), (7, 1)]:
        N = L ** d
        for ind in range(N):
            for k in range(d):
                for a in (-1, 0, 1):
                    want = oracle.neighbour_index_gpu(ind, k, a, L, d, N, 0)
                    assert lib.neighbour_index_gpu(ind, k, a, L, d, N, 0) == want
                    assert lib.neighbour_index_gpu(ind, k, a, L, d, N, 1) == want


def test_neighbour_index_morton_mode(lib, oracle):
    for L, d in [(4, 2), (8, 3), (4, 4)]:
        N = L ** d
        for ind in range(N):
            for k in range(d):
                for a in (-1, 1):
                    nb = oracle.morton_neighbour(ind, d, k, a)
                    want = nb if nb < N else N
                    assert lib.neighbour_index_gpu(ind, k, a, L, d, N, 2) == want


def test_u01_generator_host(lib):
    v = np.array([lib.c.cg_u01_host(1234, i) for i in range(4096)], np.float32)
    assert v.min() >= 0.0 and v.max() < 1.0
    assert abs(v.mean() - 0.5) < 0.02 and abs(v.std() - 0.2887) < 0.02


def _cpu_api(lib):
    """C++-linkage host API of the product library (include/conjugate_gradient_cpu.h)."""
    c = lib.c
    dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
    f = {}
    f["minus_laplace"] = c["_Z13minus_laplacePdS_iii"]
    f["minus_laplace"].argtypes = [dp, dp, C.c_int, C.c_int, C.c_int]
    f["conjugate_gradient"] = c["_Z18conjugate_gradientPKdPdii"]
    f["conjugate_gradient"].argtypes = [dp, dp, C.c_int, C.c_int]
    f["conjugate_gradient"].restype = C.c_double
    f["preconditioned_cg"] = c["_Z17preconditioned_cgPdS_ii"]
    f["preconditioned_cg"].argtypes = [dp, dp, C.c_int, C.c_int]
    f["preconditioned_cg"].restype = C.c_void_p
    return f


@pytest.mark.parametrize("name", ["cg_cpu_L5_d3", "cg_cpu_L10_d2", "cg_cpu_L64_d1", "cg_cpu_L8_d4"])
def test_host_solver_bitwise_equals_reference_fixture(lib, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    L, d = int(g["L"]), int(g["d"])
    N = L ** d
    api = _cpu_api(lib)
    x = np.zeros(N + 1)
    x[:N] = g["x"]
    b = np.zeros(N + 1)
    api["minus_laplace"](b, x, d, L, N)
    assert np.array_equal(b[:N], g["b"])
    x0 = np.zeros(N + 1)
    ret = api["conjugate_gradient"](b, x0, L, d)
    assert ret == float(g["ret"]) and np.array_equal(x0[:N], g["x_solved"])


def test_host_pcg_bitwise_equals_reference_fixture(lib, capfd):
    g = np.load(os.path.join(GOLDEN, "pcg_cpu_L50_d2.npz"))
    api = _cpu_api(lib)
    b = np.zeros(2501)
    b[:2500] = g["b"]
    x = np.zeros(2501)
    api["preconditioned_cg"](b, x, 50, 2)
    assert np.array_equal(x[:2500], g["x_pcg"])
    C.CDLL(None).fflush(None)  # the library prints through C stdio
    out = capfd.readouterr().out
    assert "number of steps (outer) 4" in out and out.count("number of steps inner") == 4


def test_neighbour_index_random_samples_at_config_sizes(lib, oracle):
    """Host entry point against the oracle at the benchmark grids (lexicographic and Morton)."""
    rng = np.random.default_rng(99)
    for L, d in [(8192, 2), (1024, 3), (512, 3), (128, 4), (1 << 20, 1)]:
        N = L ** d
        for ind in rng.integers(0, N, 300).tolist() + [0, N - 1, L - 1, L, N - L]:
            for k in range(d):
                for a in (-1, 0, 1):
                    want = oracle.neighbour_index_gpu(ind, k, a, L, d, N, 0)
                    assert lib.neighbour_index_gpu(ind, k, a, L, d, N, 0) == want
    L, d = 128, 4
    N = L ** d
    for ind in rng.integers(0, N, 500).tolist() + [0, N - 1, 0x9999d9d]:
        for k in range(d):
            for a in (-1, 1):
                nb = oracle.morton_neighbour(ind, d, k, a)
                assert lib.neighbour_index_gpu(ind, k, a, L, d, N, 2) == (nb if nb < N else N)


# ---- bench.py's byte accounting (no GPU needed) ------------------------------------------------------

def _bench_module():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("cg_bench", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_prices_bytes_as_documented():
    """roofline.achieved / frac are priced with the bytes the kernels stream: 32 B per point and iteration
    (A 12, B 20), or 30 B (B 18 on average) when x is updated every second iteration.  The 44 B of the
    three-kernel fusion appear only as an equivalent rate, never as a fraction of the peak."""
    bench = _bench_module()
    N = 1 << 26
    run = {"N": N, "msA": 0.128, "msB": 0.184, "kernel": "k_rows", "paired_x": True}
    roof = bench.roofline_of(run, {"A": 800_000_000, "B": 1_200_000_000}, "note", {})
    assert roof["kernel"] == "k_rows<MODE_B>" and roof["bound"] == "hbm" and roof["unit"] == "GB/s"
    assert roof["algorithmic_bytes"] == 18 * N and roof["algorithmic_bytes_per_point"] == 18
    assert abs(roof["achieved"] - 18 * N / 0.184e-3 / 1e9) < 0.1
    assert abs(roof["frac"] - roof["achieved"] / 8000.0) < 1e-4 and roof["frac"] <= 1.0
    assert abs(roof["three_kernel_equivalent_GBps"] - 24 * N / 0.184e-3 / 1e9) < 0.1
    assert "18 B/point" in roof["pricing"]
    assert roof["traffic"] == 1_200_000_000 and abs(roof["traffic_over_algorithmic"] - 1.2e9 / (18 * N)) < 1e-3
    # frac x peak x launch time gives the algorithmic bytes back, which the counter traffic is compared with
    assert abs(roof["frac"] * 8000e9 * roof["ms_per_launch"] * 1e-3 - roof["algorithmic_bytes"]) < 1e-3 * roof["algorithmic_bytes"]
    assert abs(roof["other_kernel"]["achieved"] - 12 * N / 0.128e-3 / 1e9) < 0.1 and roof["other_kernel"]["frac"] <= 1.0
    run["paired_x"] = False
    assert bench.roofline_of(run)["algorithmic_bytes_per_point"] == 20
    its = 3000.0
    paired, plain = bench.iteration_rates(N, its, True), bench.iteration_rates(N, its, False)
    assert abs(paired["hbm_moved_GBps"] - 30 * N * its / 1e9) < 0.1 and abs(plain["hbm_moved_GBps"] - 32 * N * its / 1e9) < 0.1
    assert paired["three_kernel_equivalent_GBps"] == plain["three_kernel_equivalent_GBps"]
    assert abs(paired["hbm_frac_moved"] - 30 * N * its / 8e12) < 1e-4 and paired["hbm_frac_moved"] <= 1.0

    def fracs(o):
        if isinstance(o, dict):
            for k, v in o.items():
                if "frac" in k and isinstance(v, (int, float)):
                    yield k, v
                yield from fracs(v)
    for o in (roof, paired, plain):
        assert all(v <= 1.0 for _, v in fracs(o)), list(fracs(o))


def test_bench_refuses_a_rank_count_that_does_not_match(tmp_path):
    """--gpus N with a different WORLD_SIZE is an error before anything touches a GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
