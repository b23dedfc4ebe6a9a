#!/usr/bin/env python3
"""Kernel A / B device time of one grid under a list of CG_* environment settings.

    tools/knob_sweep.py 1024x1024x128 "" "CG_PLANES_R=4" "CG_PLANES_WAVES=4,CG_PLANES_BPCU=4"

Every setting is a comma-separated list of NAME=VALUE pairs applied on top of the caller's
environment for one solver (the library reads the switches when a solver is created).  Prints mean
microseconds of kernels A and B (HIP events, 30 iterations) and the moved-bytes rates (12 B per point
for A; 20 for B, or 18 on average when the solver updates x in pairs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import torch  # noqa: E402

lib = load_package().load_library()
spec, _, mode = sys.argv[1].partition(":")
ext = [int(v) for v in spec.split("x")]
mode = int(mode) if mode else 0
N = 1
for e in ext:
    N *= e
b = torch.empty(N, dtype=torch.float32, device="cuda")
lib.c.cg_fill_uniform(b.data_ptr(), N, 3)
base_env = dict(os.environ)
for setting in sys.argv[2:] or [""]:
    os.environ.clear()
    os.environ.update(base_env)
    for kv in filter(None, setting.split(",")):
        k, _, v = kv.partition("=")
        os.environ[k] = v
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    s = lib.make_solver(ext, index_mode=mode)
    s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
    s.iterate(5, False)
    a, bb = s.time_kernels(30)
    name, blocks, bytes_b = s.kernel_name, s.blocks, 18 if s.paired_x else 20
    s.close()
    print(f"{spec:>16s} {name:>14s} grid {blocks:5d}  A {a * 1e3:8.1f} us {12 * N / a / 1e6:6.0f} GB/s   "
          f"B {bb * 1e3:8.1f} us {bytes_b * N / bb / 1e6:6.0f} GB/s ({bytes_b} B/pt)   {1e3 / (a + bb):7.1f} it/s   [{setting}]",
          flush=True)
