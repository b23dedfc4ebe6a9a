#!/usr/bin/env python3
"""Item 4 of the assignment the reference implements (tasks.md:26): CG iterations needed for a given
reduction of the residual norm against the condition number kappa = lambda_max / lambda_min, both
measured on the GPU (power iteration / inverse iteration through the float CG) and compared with the
analytic spectrum and with the bound  k <= ceil(1/2 sqrt(kappa) ln(2/eps))  (Shewchuk, eq. 28 solved
for the iteration count).  Writes CSV to stdout.  usage: kappa_study.py [d] [L ...]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import torch  # noqa: E402

lib = load_package().load_library()
d = int(sys.argv[1]) if len(sys.argv) > 1 else 2
Ls = [int(v) for v in sys.argv[2:]] or [16, 32, 64, 128, 256, 512]
eps = 1e-4
print("d,L,N,lambda_min,lambda_min_analytic,lambda_max,lambda_max_analytic,kappa,cg_iterations_to_1e-4,bound")
for L in Ls:
    N = L ** d
    e = lib.eigen_extremes_gpu(L, d, max_steps=4000, tol=1e-8)
    lo = d * (2 - 2 * math.cos(math.pi / (L + 1)))
    hi = d * (2 + 2 * math.cos(math.pi / (L + 1)))
    kappa = e["lambda_max"] / e["lambda_min"]
    b = torch.empty(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(b.data_ptr(), N, 7)
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    _, iters, _ = lib.conjugate_gradient_gpu_ex(b.data_ptr(), x.data_ptr(), L, d, maxit=100000, rel=eps, lagged=0)
    bound = math.ceil(0.5 * math.sqrt(kappa) * math.log(2 / eps))
    print(f"{d},{L},{N},{e['lambda_min']:.8g},{lo:.8g},{e['lambda_max']:.8g},{hi:.8g},{kappa:.6g},{iters},{bound}", flush=True)
