#!/usr/bin/env python3
"""Random grid shapes through the automatic engine choice (rows, packed rows, folded 1-D, padded
rows, planes, single-workgroup, Morton) against the one-point-per-lane engine: 8 to 15 CG iterations in
two calls (so that x updates left pending by the paired update get flushed mid-solve), histories to 1e-5,
solutions to 1e-5 relative L2; and the automatic engine against itself with CG_XDEFER=0 (x updated in every
iteration), which must agree bit for bit.  usage: fuzz_shapes.py [count] [seed]
(FUZZ_MIN_N: smallest grid, to keep away from the launch-bound sizes that all take the scalar engine;
FUZZ_PACKED=1: only grids of 3 to 12 dimensions whose rows have 4 to 256 points)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import torch  # noqa: E402

lib = load_package().load_library()
count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
names = {1: "rows", 2: "scalar", 3: "planes", 4: "morton"}
bad = 0
for case in range(count):
    while True:
        d = int(rng.integers(1, 6))
        kind = rng.integers(0, 4)
        if kind == 0:      # powers of two
            ext = [int(2 ** rng.integers(1, 25 if d == 1 else 9)) for _ in range(d)]
        elif kind == 1:    # arbitrary
            ext = [int(rng.integers(2, 3000 if d <= 2 else 300)) for _ in range(d)]
        elif kind == 2:    # long axis 0
            ext = [int(rng.integers(100, 5000))] + [int(rng.integers(1, 40)) for _ in range(d - 1)]
        else:              # cubic
            ext = [int(rng.integers(2, 200))] * d
        if os.environ.get("FUZZ_PACKED"):  # short rows in d >= 3: packed rows, columns in blocks
            d = int(rng.integers(3, 13))
            n0 = int(rng.integers(1, 65)) * 4 if rng.integers(0, 4) else int(rng.integers(4, 257))
            ext = [n0] + [int(rng.integers(1, 48 if d <= 6 else 7)) for _ in range(d - 1)]
            kind = 4
        N = int(np.prod(ext, dtype=np.int64))
        if int(os.environ.get("FUZZ_MIN_N", 2)) <= N <= (1 << (26 if kind == 4 else 24)):
            break
    mode = 0
    if kind == 0 and len(set(ext)) == 1 and d >= 2 and rng.integers(0, 2):
        mode = 2
    b = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(b.data_ptr(), N, 1000 + case)
    res = []
    iters = int(rng.integers(8, 16))
    first = int(rng.integers(1, iters))
    for engine, xdefer in ((2, "1"), (0, "1"), (0, "0")):
        os.environ["CG_XDEFER"] = xdefer
        s = lib.make_solver(ext, index_mode=mode, engine=engine)
        eng = s.engine
        x = torch.zeros(N, dtype=torch.float32, device="cuda")
        s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
        s.iterate(first, use_graph=bool(case & 1))
        s.iterate(iters - first, use_graph=bool(case & 2))
        it, _, hist = s.finish(hist_len=iters)
        s.close()
        res.append((x.cpu().numpy(), np.array(hist), it, eng))
    same_bits = (res[1][0].view(np.uint32) == res[2][0].view(np.uint32)).all() and res[1][2] == res[2][2] and \
        np.array_equal(res[1][1], res[2][1], equal_nan=True)
    res = [(r[0].astype(np.float64),) + r[1:] for r in res]
    (x0, h0, i0, _), (x1, h1, i1, eng) = res[:2]
    # a grid of a few points reaches r = 0 exactly; with the tolerance 0 used here the next step is 0/0 in
    # every engine alike (as it would be in the reference): NaNs in the same places count as agreement
    # (or infinities, once alpha = rr / pAp has overflowed: the same ones in the same places)
    nan0, nan1 = ~np.isfinite(x0), ~np.isfinite(x1)
    xa, xb = np.where(nan0, 0, x0), np.where(nan1, 0, x1)
    ok = (same_bits and i0 == i1 and np.allclose(h0, h1, rtol=1e-5, equal_nan=True) and np.array_equal(nan0, nan1)
          and np.array_equal(x0[nan0], x1[nan1], equal_nan=True)
          and np.linalg.norm(xb - xa) <= 1e-5 * max(np.linalg.norm(xa), 1e-30))
    bad += 0 if ok else 1
    if not ok:
        print("     same_bits", bool(same_bits), "iters", i0, i1, res[2][2], "hist", h0.tolist(), h1.tolist(), res[2][1].tolist(),
              "x", x0[:8].tolist(), x1[:8].tolist(), res[2][0][:8].tolist(), flush=True)
    print(f"{case:4d} {'x'.join(map(str, ext)):>24s} mode {mode} N={N:8d} auto->{names.get(eng, eng)} {'ok' if ok else 'MISMATCH'}",
          flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
