#!/usr/bin/env python3
"""Random grids cut into 2 to 4 slabs along the last axis, every slab a "virtual rank" on this GPU (ghost
planes copied by hand, dot products summed on the host in rank order, as tests/test_gpu_slab.py does):
15 CG iterations against one solver on the whole grid with the one-point-per-lane engine.  Exercises the
ghost-plane paths of every engine the automatic choice picks for the slab shapes.
usage: fuzz_slabs.py [count] [seed]   (FUZZ_MIN_N: fewest points per slab; 2^19 and more leave the launch-bound
sizes that all take the one-point-per-lane engine)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import torch  # noqa: E402

pkg = load_package()
lib = pkg.load_library()
from conjugate_gradient_amd.slab import HipSlabOps  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda", 0)
ITERS = 15
bad = 0
for case in range(count):
    while True:
        d = int(rng.integers(2, 6))
        nranks = int(rng.integers(2, 5))
        h = int(rng.integers(1, 40))
        kind = int(rng.integers(0, 3))
        if kind == 0:    # short rows (packed rows, columns in blocks)
            n0 = int(rng.integers(1, 65)) * 4
            ext = [n0] + [int(rng.integers(1, 40)) for _ in range(d - 2)] + [h]
        elif kind == 1:  # long rows (rows / planes engines)
            ext = [int(rng.integers(64, 700)) * 4] + [int(rng.integers(1, 30)) for _ in range(d - 2)] + [h]
        else:            # anything
            ext = [int(rng.integers(2, 300))] + [int(rng.integers(1, 20)) for _ in range(d - 2)] + [h]
        n_loc = int(np.prod(ext, dtype=np.int64))
        if int(os.environ.get("FUZZ_MIN_N", 20000)) <= n_loc and n_loc * nranks <= (1 << 25):
            break
    N = n_loc * nranks
    whole = ext[:-1] + [h * nranks]
    b = torch.empty(N, dtype=torch.float32, device=dev)
    lib.c.cg_fill_uniform(b.data_ptr(), N, 500 + case)
    slabs, xs = [], []
    for r in range(nranks):
        slabs.append(HipSlabOps(lib, ext, int(r > 0), int(r < nranks - 1), dev))
        xs.append(torch.zeros(n_loc, dtype=torch.float32, device=dev))

    def allreduce(which):
        torch.cuda.synchronize()
        for o in slabs:
            o.sync()
        tot = sum(o.sum_tensor(which) for o in slabs)
        for o in slabs:
            o.sum_tensor(which).copy_(tot)
        torch.cuda.synchronize()

    def exchange():
        torch.cuda.synchronize()
        for o in slabs:
            o.sync()
        for r, o in enumerate(slabs):
            g, pl = o.ghost_offset, o.plane
            if r > 0:
                o.r_buf[g - pl:g] = slabs[r - 1].r_buf[g + n_loc - pl:g + n_loc]
            if r < nranks - 1:
                o.r_buf[g + n_loc:g + n_loc + pl] = slabs[r + 1].r_buf[g:g + pl]
        torch.cuda.synchronize()

    for r, o in enumerate(slabs):
        o.begin_local(b[r * n_loc:(r + 1) * n_loc], xs[r])
    allreduce(1)
    for o in slabs:
        o.init_state(0.0, 1, 0)
    exchange()
    for o in slabs:
        o.add_iterations(ITERS)
    for _ in range(ITERS):
        for o in slabs:
            o.kernel_A()
        allreduce(0)
        for o in slabs:
            o.kernel_B()
        allreduce(1)
        exchange()
    outs = [o.finish(hist_len=ITERS) for o in slabs]
    torch.cuda.synchronize()
    x = torch.cat(xs).cpu().numpy()
    names = sorted({o.solver.kernel_name for o in slabs})
    for o in slabs:
        o.solver.close()
    ref = lib.make_solver(whole, engine=2)
    xr = torch.zeros(N, dtype=torch.float32, device=dev)
    ref.begin(b.data_ptr(), xr.data_ptr(), 0.0, 1)
    ref.iterate(ITERS, use_graph=False)
    it, _, hr = ref.finish(hist_len=ITERS)
    ref.close()
    h0 = np.array(outs[0][2])
    ok = (outs[0][0] == it == ITERS and np.allclose(h0, np.array(hr), rtol=1e-4)
          and np.linalg.norm(x - xr.cpu().numpy()) <= 1e-5 * np.linalg.norm(x))
    bad += 0 if ok else 1
    print(f"{case:4d} {'x'.join(map(str, ext)):>24s} x{nranks} ranks {'/'.join(names):18s} {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
