#!/usr/bin/env python3
"""Wall time of whole conjugate_gradient_gpu_ex calls on small grids (what a driver that solves
many small systems pays per call), same and alternating x buffers."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import torch  # noqa: E402

lib = load_package().load_library()
for L, d in ((64, 2), (50, 2), (16, 3), (256, 2), (1024, 2)):
    N = L ** d
    b = torch.rand(N, device="cuda")
    xs = [torch.zeros(N, device="cuda") for _ in range(2)]
    for label, pick in (("same x", lambda i: xs[0]), ("alternating x", lambda i: xs[i & 1])):
        lib.conjugate_gradient_gpu_ex(b.data_ptr(), xs[0].data_ptr(), L, d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps, iters = 20, 0
        for i in range(reps):
            x = pick(i)
            x.zero_()
            _, it, _ = lib.conjugate_gradient_gpu_ex(b.data_ptr(), x.data_ptr(), L, d)
            iters += it
        dt = (time.perf_counter() - t0) / reps
        print(f"L={L} d={d} {label:14s}: {dt * 1e3:7.3f} ms per call, {iters // reps} iterations, "
              f"{dt / (iters / reps) * 1e6:6.1f} us per iteration", flush=True)
