"""Generate tests/golden/* from the reference's own CPU code (oracle/_ref).

Run here (where /root/reference exists):  python oracle/make_golden.py
The fixtures are data only: inputs and the outputs the compiled reference produced
for them.  Inputs come from the reference's random_array (glibc rand(), srand(1)),
so they are stored as well -- glibc rand is not portable.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    orc.build(ref=True)
    ref = orc.Reference()
    assert ref.available and ref.morton_available, "reference build missing"
    os.makedirs(OUT, exist_ok=True)

    # conjugate_gradient (cpu.cu:216-261) on b = A x_rand, x0 = 0
    for L, d in [(64, 1), (5, 3), (10, 2), (50, 2), (16, 3), (8, 4)]:
        N = L ** d
        x = ref.random_array(N, seed=1)
        b = ref.minus_laplace(x, d, L)
        xs, ret = ref.conjugate_gradient(b, L, d)
        np.savez_compressed(os.path.join(OUT, f"cg_cpu_L{L}_d{d}.npz"), L=L, d=d, x=x, b=b,
                            x_solved=xs, ret=np.float64(ret))
        print(f"cg L={L} d={d}: ret={ret!r} sum(x_solved)={xs.sum()!r} b[0]={b[0]!r}")

    # preconditioner(…,1e-6) and preconditioned_cg (cpu.cu:274-392), 50x50 as tests.cu:320-322
    L, d = 50, 2
    N = L ** d
    x = ref.random_array(N, seed=1)
    b = ref.minus_laplace(x, d, L)
    xp, resp = ref.preconditioner(b, L, d, 1e-6)
    xo = ref.preconditioned_cg(b, L, d)
    np.savez_compressed(os.path.join(OUT, "pcg_cpu_L50_d2.npz"), L=L, d=d, x=x, b=b,
                        x_precond=xp, res_precond=np.float64(resp), x_pcg=xo)

    # Morton index goldens (interleave.cuh + BMI2 pdep/pext)
    rng = np.random.default_rng(20250114)
    mort = {"masks": {}, "cases": []}
    for d in (2, 3, 4, 5, 6):
        mort["masks"][str(d)] = [int(ref.mlib.ref_morton_mask(d, k)) for k in range(d)]
    for d, bits in ((2, 16), (3, 10), (4, 7), (4, 8)):
        Lm = 1 << bits
        pts = [np.zeros(d, np.int32), np.full(d, Lm - 1, np.int32)]
        pts += [rng.integers(0, Lm, d).astype(np.int32) for _ in range(40)]
        if d == 4 and bits == 7:
            pts.append(np.array([127, 0, 5, 127], np.int32))
        for c in pts:
            idx = int(ref.mlib.ref_morton_from_coords(np.ascontiguousarray(c), d))
            back = np.zeros(d, np.int32)
            ref.mlib.ref_morton_to_coords(idx, d, back)
            nb = [[int(ref.mlib.ref_morton_neighbour(idx, d, k, +1)),
                   int(ref.mlib.ref_morton_neighbour(idx, d, k, -1))] for k in range(d)]
            mort["cases"].append({"d": d, "bits": bits, "coords": c.tolist(), "index": idx,
                                  "decoded": back.tolist(), "neighbours": nb})
    pd = []
    for _ in range(64):
        x_, m_ = int(rng.integers(0, 2 ** 32)), int(rng.integers(0, 2 ** 32))
        pd.append([x_, m_, int(ref.mlib.ref_pdep(x_, m_)), int(ref.mlib.ref_pext(x_, m_))])
    mort["pdep_pext"] = pd
    with open(os.path.join(OUT, "morton_index.json"), "w") as f:
        json.dump(mort, f)
    print("morton cases:", len(mort["cases"]))


if __name__ == "__main__":
    main()
