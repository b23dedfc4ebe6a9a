#!/usr/bin/env python3
"""Iterations/s and kernel A / B times of one workload with whatever library CG_LIB_PATH names (uses only entry
points every version of the library has)."""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package  # noqa: E402
import ctypes as C  # noqa: E402

pkg = load_package()
lib = pkg.binding.Library.__new__(pkg.binding.Library)
c = C.CDLL(pkg.binding.LIB_PATH)
spec = sys.argv[1]
mode = 0
if ":" in spec:
    spec, m = spec.split(":")
    mode = int(m)
ext = [int(v) for v in spec.split("x")]
N = 1
for e in ext:
    N *= e
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
b = torch.empty(N, dtype=torch.float32, device="cuda")
x = torch.zeros(N, dtype=torch.float32, device="cuda")
c.cg_fill_uniform.argtypes = [C.c_void_p, C.c_ulonglong, C.c_uint]
c.cg_fill_uniform(b.data_ptr(), N, 1234)
torch.cuda.synchronize()
d = pkg.binding.SolverDesc()
d.d = len(ext)
for k, e in enumerate(ext):
    d.extents[k] = e
d.index_mode = mode
c.cg_solver_create.restype = C.c_void_p
c.cg_solver_create.argtypes = [C.c_void_p]
s = c.cg_solver_create(C.byref(d))
vp = C.c_void_p
c.cg_solver_begin.argtypes = [vp, vp, vp, C.c_double, C.c_int]
c.cg_solver_iterate.argtypes = [vp, C.c_int, C.c_int]
c.cg_solver_sync.argtypes = [vp]
c.cg_solver_time_kernels.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
c.cg_solver_finish.argtypes = [vp, vp, vp, C.c_int]
c.cg_solver_begin(s, b.data_ptr(), x.data_ptr(), 0.0, 1)
c.cg_solver_iterate(s, 10, 1)
c.cg_solver_sync(s)
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c.cg_solver_iterate(s, steps, 1)
    c.cg_solver_sync(s)
    ts.append(time.perf_counter() - t0)
a, bb = C.c_float(0), C.c_float(0)
c.cg_solver_time_kernels(s, 20, C.byref(a), C.byref(bb))
c.cg_solver_finish(s, None, None, 0)
print(f"{steps / statistics.median(ts):9.2f} it/s   {statistics.median(ts) / steps * 1e3:.4f} ms/it   A {a.value * 1e3:.1f} us  B {bb.value * 1e3:.1f} us   regions {[round(t / steps * 1e3, 4) for t in ts]}")
