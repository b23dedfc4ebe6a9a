#!/usr/bin/env python3
"""The reference driver's timing sweeps (main.jl:85-163) as CSV, on the device clock.

  dimension   main.jl:85-116  N ~ 2^24 points, d in 1,2,3,4,6,8,12,24, one thread per block as there
  size        main.jl:118-136 ten grid sizes up to 2e6 points for a given d and threads per block
  1d          main.jl:139-163 1-D, blocks x threads points, GPU time / CPU time

Columns: what the reference measures (a launch of the caller's shape, one point per thread,
wall clock with a host synchronisation, `strong_scaling`), the same launch on HIP events
(`cg_strong_scaling_events`), the library's own tiled stencil for that grid
(`cg_laplace_time_events`), and the CPU stencil (`time_cpu_laplace`) where the sweep uses it.
Output: gpurun_out/sweep_<name>.csv.  No plots: the numbers are what the figures are drawn from.
"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    which = sys.argv[1:] or ["dimension", "size", "1d"]
    lib = load_package().load_library()
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)

    if "dimension" in which:
        rows = []
        for d in (1, 2, 3, 4, 6, 8, 12, 24):
            L = round((2 ** 24) ** (1.0 / d))
            N = L ** d
            nb = min(N, 2 ** 31 - 1)
            rows.append({"d": d, "L": L, "N": N,
                         "ref_shape_wall_us": lib.strong_scaling(nb, 1, N, L, d),
                         "ref_shape_events_us": lib.strong_scaling_events(nb, 1, N, L, d, 3),
                         "tiled_events_us": lib.laplace_time_events(L, d, 0, 5)})
            print(rows[-1], flush=True)
        write(os.path.join(out, "sweep_dimension.csv"), rows)

    if "size" in which:
        rows = []
        for d, nt in ((1, 512), (2, 512), (3, 512), (4, 512)):
            top = round(2000000 ** (1.0 / d))
            for i in range(10):
                L = max(2, round(1 + (top - 1) * i / 9))
                N = L ** d
                nb = (N + nt - 1) // nt
                rows.append({"d": d, "threads": nt, "L": L, "N": N,
                             "ref_shape_wall_us": lib.strong_scaling(nb, nt, N, L, d),
                             "ref_shape_events_us": lib.strong_scaling_events(nb, nt, N, L, d, 5),
                             "tiled_events_us": lib.laplace_time_events(L, d, 0, 5)})
            print(rows[-1], flush=True)
        write(os.path.join(out, "sweep_size.csv"), rows)

    if "1d" in which:
        rows = []
        for nt in (128, 512):
            for nblocks in (1, 10, 100, 250, 500, 750, 1000):
                N = nblocks * nt
                rows.append({"threads": nt, "blocks": nblocks, "N": N,
                             "ref_shape_wall_us": lib.strong_scaling(nblocks, nt, N, N, 1),
                             "ref_shape_events_us": lib.strong_scaling_events(nblocks, nt, N, N, 1, 10),
                             "tiled_events_us": lib.laplace_time_events(N, 1, 0, 10),
                             "cpu_us": lib.time_cpu_laplace(N, N, 1)})
            print(rows[-1], flush=True)
        write(os.path.join(out, "sweep_1d.csv"), rows)


def write(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    print("wrote", path)


if __name__ == "__main__":
    main()
