#!/usr/bin/env python3
"""What the multi-rank machinery costs on one GPU: the same slab solved (a) by a plain solver, (b) with one RCCL
rank bound and no neighbours (two all-gathers per iteration), (c) with that rank as its own lower and upper
neighbour (plus the copy of the boundary planes, the kernel that computes them ahead of kernel B, and the grouped
ncclSend / ncclRecv on the second stream).  Iterations replayed from the captured graph; median of 5 regions.
usage: ring_bench.py [L0 L1 ...] (default 1024 1024 128)"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

ext = [int(v) for v in sys.argv[1:]] or [1024, 1024, 128]
steps = 100
lib = load_package().load_library()
N = 1
for e in ext:
    N *= e
b = torch.empty(N, dtype=torch.float32, device="cuda")
lib.c.cg_fill_uniform(b.data_ptr(), N, 13)
x = torch.zeros(N, dtype=torch.float32, device="cuda")


def timed(s):
    x.zero_()
    torch.cuda.synchronize()
    s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
    s.iterate(20)
    s.sync()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.iterate(steps)
        s.sync()
        ts.append(time.perf_counter() - t0)
    msA, msB = s.time_kernels(20)
    s.finish()
    return statistics.median(ts) / steps * 1e3, msA, msB


s = lib.make_solver(ext)
print("plain solver                 %.4f ms/iteration  (A %.1f us, B %.1f us)" % tuple(v * (1 if i == 0 else 1e3) for i, v in enumerate(timed(s))))
s.close()
comm = lib.rccl_comm_create(lib.rccl_unique_id(), 0, 1)
s = lib.make_solver(ext)
s.bind_rccl(comm, 0, 1)
print("one rank, no neighbours      %.4f ms/iteration  (A %.1f us, B %.1f us)" % tuple(v * (1 if i == 0 else 1e3) for i, v in enumerate(timed(s))))
s.bind_rccl(None, 0, 1)
s.close()
s = lib.make_solver(ext, 0, 1, 1)
s.bind_rccl(comm, 0, 1, peers=(0, 0))
print("one rank, its own neighbours %.4f ms/iteration  (A %.1f us, B %.1f us)" % tuple(v * (1 if i == 0 else 1e3) for i, v in enumerate(timed(s))))
s.bind_rccl(None, 0, 1)
s.close()
lib.rccl_comm_destroy(comm)
