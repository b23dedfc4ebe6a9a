#!/usr/bin/env python3
"""In-kernel timestamps of a debug build of the library (-DCG_DBG=64, CG_LIB_PATH): where inside kernels A and B
of a small grid the time goes.  usage: ktimes.py LxL"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
c = C.CDLL(pkg.binding.LIB_PATH)
ext = [int(v) for v in sys.argv[1].split("x")]
N = 1
for e in ext:
    N *= e
b = torch.empty(N, dtype=torch.float32, device="cuda")
x = torch.zeros(N, dtype=torch.float32, device="cuda")
c.cg_fill_uniform.argtypes = [C.c_void_p, C.c_ulonglong, C.c_uint]
c.cg_fill_uniform(b.data_ptr(), N, 1234)
d = pkg.binding.SolverDesc()
d.d = len(ext)
for k, e in enumerate(ext):
    d.extents[k] = e
c.cg_solver_create.restype = C.c_void_p
c.cg_solver_create.argtypes = [C.c_void_p]
s = c.cg_solver_create(C.byref(d))
vp = C.c_void_p
c.cg_solver_begin.argtypes = [vp, vp, vp, C.c_double, C.c_int]
c.cg_solver_iterate.argtypes = [vp, C.c_int, C.c_int]
c.cg_solver_sync.argtypes = [vp]
c.cg_solver_begin(s, b.data_ptr(), x.data_ptr(), 0.0, 1)
c.cg_solver_iterate(s, 200, 1)
c.cg_solver_sync(s)
t = (C.c_ulonglong * 64)()
c.cg_dbg_times(t)
for name, o in (("A", 0), ("B", 32)):
    for blk, oo in (("first workgroup", 0), ("last workgroup", 16)):
        v = [t[o + oo + i] for i in range(4)]
        print(f"kernel {name} {blk}: prologue {(v[1]-v[0])*10} ns, body {(v[2]-v[1])*10} ns, epilogue {(v[3]-v[2])*10} ns")
    print(f"kernel {name}: first workgroup's entry to last workgroup's end {(t[o+16+3]-t[o+0])*10} ns")
print(f"end of A (last wg) to entry of B (first wg): {(t[32]-t[16+3])*10} ns; end of B to entry of A: {(t[0]-t[32+16+3])*10} ns (negative: A of the next iteration was recorded earlier)")
