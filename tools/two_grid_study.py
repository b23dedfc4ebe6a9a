#!/usr/bin/env python3
"""The "Multigrid" comparison of the assignment the reference implements (tasks.md:32-36): two-grid
cycles (2+2 damped-Jacobi steps, coarse grid solved by the float GPU CG) against plain CG on the same
right-hand side, and CG preconditioned by one such cycle, to a 1e-5 reduction of the residual: cycles /
iterations and wall time.  CSV to stdout.
usage: two_grid_study.py [d] [L ...]   (L odd)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import torch  # noqa: E402

lib = load_package().load_library()
d = int(sys.argv[1]) if len(sys.argv) > 1 else 2
Ls = [int(v) for v in sys.argv[2:]] or [63, 127, 255, 511, 1023, 2047]
rel = 1e-5
print("d,L,N,tg_cycles,tg_coarse_cg_iterations,tg_ms,tg_factor_first_cycles,tgpcg_iterations,tgpcg_ms,mg_levels,mg_cycles,mg_ms,cg_iterations,cg_ms")
for L in Ls:
    N = L ** d
    xt = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(xt.data_ptr(), N, 11)
    f = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.laplace_gpu_jl(f.data_ptr(), xt.data_ptr(), 0.0, d, L, N, 0, (N + 255) // 256, 256)
    out = {}
    for rep in range(2):  # the second pass is timed (allocations, graph capture warmed up)
        u = torch.zeros(N, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cycles, hist, inner = lib.two_grid_gpu(f.data_ptr(), u.data_ptr(), L, d, max_cycles=60, rel=rel)
        t_tg = time.perf_counter() - t0
        v = torch.zeros(N, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pits, _, _ = lib.two_grid_pcg_gpu(f.data_ptr(), v.data_ptr(), L, d, max_iters=60, rel=rel)
        t_pcg = time.perf_counter() - t0
        m = torch.zeros(N, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mcyc, _, mlev, _ = lib.multigrid_gpu(f.data_ptr(), m.data_ptr(), L, d, max_cycles=60, rel=rel)
        t_mg = time.perf_counter() - t0
        x = torch.zeros(N, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, iters, _ = lib.conjugate_gradient_gpu_ex(f.data_ptr(), x.data_ptr(), L, d, maxit=200000, rel=rel, lagged=0)
        t_cg = time.perf_counter() - t0
    fac = (hist[min(cycles, 3)] / hist[0]) ** (1.0 / min(cycles, 3)) if cycles else 0.0
    print(f"{d},{L},{N},{cycles},{inner},{t_tg * 1e3:.2f},{fac:.3f},{pits},{t_pcg * 1e3:.2f},{mlev},{mcyc},{t_mg * 1e3:.2f},{iters},{t_cg * 1e3:.2f}", flush=True)
