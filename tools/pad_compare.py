#!/usr/bin/env python3
"""Whole-solve rate (iterations/s, graph path) of grids whose axis 0 is not a multiple of 4 points,
with rows padded inside the solver (default) and without (CG_PAD_ROWS=0: the one-point-per-lane engine).
usage: pad_compare.py AxBxC ..."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--one":
    from __graft_entry__ import load_package
    import torch
    lib = load_package().load_library()
    ext = [int(v) for v in sys.argv[2].split("x")]
    N = 1
    for e in ext:
        N *= e
    b = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(b.data_ptr(), N, 3)
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    s = lib.make_solver(ext)
    s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
    s.iterate(20)
    s.sync()
    t0 = time.perf_counter()
    s.iterate(100)
    s.sync()
    dt = time.perf_counter() - t0
    print(f"{sys.argv[2]:>20s} pad={os.environ.get('CG_PAD_ROWS', '1')} {s.kernel_name:9s} {100 / dt:9.1f} it/s", flush=True)
    s.finish()
    s.close()
else:
    for shape in sys.argv[1:]:
        for pad in ("1", "0"):
            subprocess.run([sys.executable, __file__, "--one", shape], env=dict(os.environ, CG_PAD_ROWS=pad), check=True)
