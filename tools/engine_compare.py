#!/usr/bin/env python3
"""Kernel A / B device time per engine for a list of grids (engine choice heuristics)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import torch  # noqa: E402

lib = load_package().load_library()
cases = [(2, 64), (2, 128), (2, 256), (2, 512), (3, 32), (3, 64), (3, 128), (3, 256), (4, 16), (4, 32), (4, 64), (4, 128),
         (5, 16), (5, 32), (6, 16), (8, 8)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(",")) if "," in a else (0, [int(v) for v in a.split("x")]) for a in sys.argv[1:]]
for d, L in cases:
    # "d,L" is the cube L^d; "AxBxC" lists the extents, first (contiguous) axis first
    ext = list(L) if d == 0 else [L] * d
    d, L = len(ext), ext[0]
    N = 1
    for e in ext:
        N *= e
    b = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(b.data_ptr(), N, 3)
    line = f"{'x'.join(str(e) for e in ext):>18s} N=2^{N.bit_length() - 1:2d}:"
    for eng in (1, 2, 3):
        if (eng == 3 and d != 3) or (eng != 2 and (L % 4 or d > 16)):
            continue
        x = torch.zeros(N, dtype=torch.float32, device="cuda")
        s = lib.make_solver(ext, engine=eng)
        s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
        s.iterate(5, False)
        a, bb = s.time_kernels(20)
        s.close()
        line += f"  eng{eng}: A {a * 1e3:8.1f} B {bb * 1e3:8.1f} us"
    print(line, flush=True)
