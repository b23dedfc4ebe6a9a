#!/usr/bin/env python3
"""Wall time of inner_product_gpu (launch, two-stage reduction, read-back) on device vectors."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import torch  # noqa: E402

lib = load_package().load_library()
for logn in (20, 24, 26, 28):
    n = 1 << logn
    v = torch.rand(n, device="cuda")
    w = torch.rand(n, device="cuda")
    for name, a, b, nbytes in (("v.v", v, v, 4 * n), ("v.w", v, w, 8 * n)):
        lib.inner_product_gpu(a.data_ptr(), b.data_ptr(), n)
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            lib.inner_product_gpu(a.data_ptr(), b.data_ptr(), n)
        dt = (time.perf_counter() - t0) / reps
        print(f"N=2^{logn} {name}: {dt * 1e6:8.1f} us  {nbytes / dt / 1e9:8.1f} GB/s", flush=True)
