#!/usr/bin/env python3
"""Config "mixed precision, 3-D 512^3": double outer CG (host control, fp64 vectors on the GPU)
with the float GPU CG as inner solve.  Builds b = A x_true in fp64 on the device, solves from
x0 = 0 and reports outer/inner iteration counts, time and the fp64 residual / error."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def lap64(t):
    import torch
    d = t.dim()
    out = 2 * d * t.clone()
    for ax in range(d):
        n = t.shape[ax]
        sl_lo = [slice(None)] * d
        sl_hi = [slice(None)] * d
        sl_lo[ax] = slice(0, n - 1)
        sl_hi[ax] = slice(1, n)
        out[tuple(sl_lo)] -= t[tuple(sl_hi)]
        out[tuple(sl_hi)] -= t[tuple(sl_lo)]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=512)
    ap.add_argument("--d", type=int, default=3)
    a = ap.parse_args()
    import torch
    from __graft_entry__ import load_package
    lib = load_package().load_library()
    L, d = a.L, a.d
    gen = torch.Generator(device="cuda").manual_seed(7)
    xt = torch.rand([L] * d, dtype=torch.float64, device="cuda", generator=gen)
    b = lap64(xt).reshape(-1).contiguous()
    x = torch.zeros_like(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outer, inner = lib.preconditioned_cg_gpu(b.data_ptr(), x.data_ptr(), L, d)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = (b - lap64(x.reshape([L] * d)).reshape(-1)).norm().item() / b.norm().item()
    err = (x - xt.reshape(-1)).abs().max().item()
    print(f"mixed L={L} d={d}: outer={outer} inner={inner} total_inner={sum(inner)} time={dt:.3f}s "
          f"rel_residual={res:.3e} max_abs_err={err:.3e}")


if __name__ == "__main__":
    main()
