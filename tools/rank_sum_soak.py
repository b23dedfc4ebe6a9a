#!/usr/bin/env python3
"""Soak test of the per-rank sum in the epilogue of kernels A and B (the last workgroup to finish adds the
launch's partials up; cg_device.hpp, block_write_partial): with one RCCL rank bound and no neighbours every
iteration goes through it, and the solve must stay bit-identical to the unbound solver's -- any partial read
before it was visible would show up as a different history.  usage: rank_sum_soak.py [solves] [iterations]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

solves = int(sys.argv[1]) if len(sys.argv) > 1 else 40
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 500
lib = load_package().load_library()
comm = lib.rccl_comm_create(lib.rccl_unique_id(), 0, 1)
bad = total = 0
for ext in ([2048, 2048], [8192, 8192], [256, 256, 256], [64, 64, 64, 64], [1 << 22], [1000, 1000]):
    N = 1
    for e in ext:
        N *= e
    plain = lib.make_solver(ext)
    bound = lib.make_solver(ext)
    bound.bind_rccl(comm, 0, 1)
    b = torch.empty(N, dtype=torch.float32, device="cuda")
    xs = [torch.zeros(N, dtype=torch.float32, device="cuda") for _ in range(2)]
    n_solves = solves if N <= (1 << 24) else max(2, solves // 10)
    for k in range(n_solves):
        lib.c.cg_fill_uniform(b.data_ptr(), N, 100 + k)
        out = []
        for s, x in zip((plain, bound), xs):
            x.zero_()
            torch.cuda.synchronize()
            s.begin(b.data_ptr(), x.data_ptr(), 1e-7, 1)
            s.iterate(iters, use_graph=bool(k & 1))
            out.append(s.finish(hist_len=min(iters, 1024)))
        total += out[0][0]
        if out[0] != out[1] or not torch.equal(xs[0], xs[1]):
            bad += 1
            print("MISMATCH", ext, k, out[0][:2], out[1][:2], flush=True)
    print(f"{'x'.join(map(str, ext)):>14s}: {n_solves} solves, {plain.blocks} workgroups, kernel {plain.kernel_name}", flush=True)
    bound.bind_rccl(None, 0, 1)
    bound.close()
    plain.close()
lib.rccl_comm_destroy(comm)
print(f"iterations compared: {total}, mismatching solves: {bad}")
sys.exit(1 if bad else 0)
