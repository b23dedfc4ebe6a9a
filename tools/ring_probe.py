#!/usr/bin/env python3
"""One RCCL rank that is its own neighbour (see tests/test_gpu_rccl.py): run the multi-rank iteration from
plain launches and/or from the captured graph and print what happened.  usage: ring_probe.py plain|graph [L0 L1 ...]"""
import faulthandler
import os
import sys

import torch

faulthandler.enable()
_bt = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dbg", "libsegv_bt.so")
if os.path.exists(_bt):
    import ctypes
    ctypes.CDLL(_bt).install_segv_bt()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

mode = sys.argv[1]
ext = [int(v) for v in sys.argv[2:]] or [256, 256]
lib = load_package().load_library()
N = 1
for e in ext:
    N *= e
b = torch.empty(N, dtype=torch.float32, device="cuda")
lib.c.cg_fill_uniform(b.data_ptr(), N, 13)
x = torch.zeros(N, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
comm = lib.rccl_comm_create(lib.rccl_unique_id(), 0, 1)
print("comm created", flush=True)
s = lib.make_solver(ext, 0, 1, 1)
s.bind_rccl(comm, 0, 1, peers=(0, 0))
print("bound, ranks", s.rccl_ranks(), flush=True)
s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
s.sync()
print("begun", flush=True)
for n in (3, 45, 2):
    s.iterate(n, use_graph=(mode == "graph"))
    s.sync()
    print("iterated", n, flush=True)
print(s.finish(hist_len=5), flush=True)
s.bind_rccl(None, 0, 1)
s.close()
lib.rccl_comm_destroy(comm)
print("done", flush=True)
