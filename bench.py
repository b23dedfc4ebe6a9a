#!/usr/bin/env python3
"""Benchmark of the fused Laplace-CG iteration on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 10
    python bench.py --gpus N ...            (starts N rank processes itself, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one CG iteration (kernel A + kernel B) over the whole grid.  Workload at every N:
each GPU owns an 8192 x 8192 float slab (2^26 points, config "2D Laplace 8192^2 float CG");
with N > 1 the slabs stack along axis 1 into an 8192 x (8192 N) grid (weak scaling) with one
halo row of r each way and two dot-product exchanges per iteration (config.comm says which
transport ran).  A multi-rank run also measures, in the same process group and with the same
transport, config "3D Laplace 1024^3 slab-decomposed over 8 GPUs" as weak scaling of its
per-GPU slab (1024 x 1024 x 128 on every GPU: exactly that configuration at N = 8) and reports
it as the object `config3`; both carry `single_gpu_reference`, the same slab solved on one GPU
without any exchange in the same run.  Synthetic data: b = u01 (counter-based generator, seed
1234 + rank), x0 = 0; float CG cannot converge on these grids within any practical count, so
every step does full work (the stop tolerance is set to 0).

Timing: after the warm-up the region "exactly --steps iterations between barrier + synchronize"
is timed --repeats times in a row on the same running solve (every iteration does full work);
value and ms_per_step are the MEDIAN region, min and max are printed beside it.

Bytes per grid point, always named:
  moved      what these kernels stream, the algorithmic traffic of this two-kernel design: A 12 B
             (r, p_old in, p_new out), B 20 B (x, p, r in, x, r out), iteration 32 B.  With the
             paired x update (config.paired_x_update: kernel B of an even iteration leaves x alone,
             the next one reads both p and applies both updates) a pair of B launches moves
             12 + 24 B: B 18 B on average, iteration 30 B.  roofline.achieved / roofline.frac and
             hbm_frac_moved use this price: bandwidth the memory system delivers -- at 8192^2 part
             of it from the 256 MiB Infinity Cache, which holds r; `roofline_hbm_bound` repeats the
             measurement on a grid whose vectors are twice that cache.
  counter    bytes leaving L2 per launch (roofline.traffic), measured in this run by two rocprofv3
             --pmc passes (FETCH_SIZE doubled as the gfx950 counter reports half of wide reads,
             WRITE_SIZE as is); Infinity-Cache hits are included, so it bounds HBM traffic from
             above.  null when the counter passes could not run.
  three-kernel equivalent   the 44 B (A 20, B 24) of the textbook fusion that stores A p (stencil +
             dot, x/r update + dot, p update).  Not a bandwidth: this design never moves those bytes.
             Reported only as `three_kernel_equivalent_GBps`, the rate such a solver would need for
             the same iterations/s.
cpu_baseline: one CG iteration's worth of the reference's own CPU calls (fp64, one core) on the
full workload grid where that takes under a minute, else on a bounded cubic sample.
"""
import argparse
import ctypes as C
import json
import os
import re
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak
# bytes per grid point: "moved" = what this design streams (its algorithmic traffic), "equivalent" = the
# three-kernel fusion that stores A p (only ever reported as an equivalent rate, never as a fraction)
BYTES = {"A": {"equivalent": 20, "moved": 12}, "B": {"equivalent": 24, "moved": 20},
         "iter": {"equivalent": 44, "moved": 32}}
# a solver that updates x every second iteration (two updates at once): B moves 12 and 24 B in turn
BYTES_PAIRED = {"A": BYTES["A"], "B": {"equivalent": 24, "moved": 18}, "iter": {"equivalent": 44, "moved": 30}}


def bytes_of(paired):
    return BYTES_PAIRED if paired else BYTES


BEYOND_CACHE_WORKLOAD = "16384x8192"   # 2^27 points: every vector is twice the Infinity Cache
CONFIG3_SLAB = "3d-1024x128"           # one GPU's share of the 1024^3 grid cut into 8 slabs


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="2d-8192", help="2d-8192 | 3d-1024x128 | 3d-512 | 3d-1024 | 4d-128-morton | mixed-3d-512 | LxLx..[:mode]")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--force-slab", action="store_true", help="run the multi-rank driver even with one rank")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the measured path) | gloo (rehearsal: host-staged)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every GPU owns a --workload grid; strong: the --workload grid is cut into --gpus slabs")
    ap.add_argument("--comm", default="auto", help="multi-rank exchange: auto (the fastest of the library's RCCL calls, "
                    "the direct xGMI boxes and the Python-driven collectives that passes a self-check) | rccl | links | collectives")
    ap.add_argument("--probe-seconds", type=float, default=60.0,
                    help="time limit for all sandboxed transport probes together; what has not answered by then is not used")
    ap.add_argument("--no-config3", action="store_true",
                    help="multi-rank runs: skip the second workload (1024 x 1024 x 128 per GPU, the 8-GPU configuration's slab)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--repeats", type=int, default=9, help="timed regions of --steps iterations; the median is reported "
                    "(the first region of a run also captures the iteration graph, and the clocks settle during the first few)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 counter passes (roofline.traffic = null)")
    ap.add_argument("--no-beyond-cache", action="store_true",
                    help="skip the second, larger 2-D workload whose vectors exceed the Infinity Cache")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--probe-child", default="", help=argparse.SUPPRESS)
    ap.add_argument("--no-sandbox", action="store_true",
                    help="probe the multi-rank transports in this process instead of in child processes")
    return ap.parse_args()


def workload_extents(name):
    mode = 0
    if name == "2d-8192":
        ext = [8192, 8192]
    elif name == "3d-1024x128":
        ext = [1024, 1024, 128]
    elif name == "3d-512":
        ext = [512, 512, 512]
    elif name == "3d-1024":
        ext = [1024, 1024, 1024]
    elif name == "4d-128-morton":
        ext, mode = [128] * 4, 2
    else:
        spec, _, m = name.partition(":")
        ext = [int(v) for v in spec.split("x")]
        mode = int(m) if m else 0
    return ext, mode


def _under_profiler():
    return any("ROCPROF" in k.upper() or "ROCP_" in k.upper() for k in os.environ) or \
        "rocprof" in os.environ.get("LD_PRELOAD", "").lower()


def live_traffic(workload):
    """Bytes leaving L2 per launch of kernels A and B on `workload`, measured now: two child runs of this
    script (six plain iterations each) under `rocprofv3 --pmc`, FETCH_SIZE and WRITE_SIZE in separate
    passes (they do not fit one).  FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of wide
    coalesced reads at 64 bytes), WRITE_SIZE taken as is; both are in KiB.  Returns ({"A": bytes,
    "B": bytes}, note); a kernel is missing from the dict when a pass failed."""
    import csv
    import glob
    import shutil
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}, "rocprofv3 not found", {}
    if _under_profiler():
        return {}, "already running under a profiler", {}
    got = {"fetch": {}, "write": {}}
    notes = []
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        out = tempfile.mkdtemp(prefix=f"cg_pmc_{kind}_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--workload", workload]
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=150)
            if r.returncode != 0:
                notes.append(f"{counter} pass exit {r.returncode}: {r.stderr[-200:]}")
                continue
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            per = {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    m = re.search(r"cgamd::k_\w+<(\d),", row.get("Kernel_Name", ""))
                    if m and m.group(1) in "12" and row.get("Counter_Name", counter) == counter:
                        per.setdefault("A" if m.group(1) == "1" else "B", []).append(float(row["Counter_Value"]))
            for k, v in per.items():
                # drop no-op launches; the mean, because with the paired x update kernel B's launches
                # alternate between two traffic levels (six iterations: three of each)
                full = [x for x in v if x > 0.2 * max(v)]
                got[kind][k] = statistics.mean(full) * 1024.0 * (2.0 if kind == "fetch" else 1.0)
            if not per:
                notes.append(f"{counter} pass: no kernel rows")
        except Exception as e:  # noqa: BLE001 -- the counters are optional evidence, never fatal
            notes.append(f"{counter} pass failed: {e!r}"[:300])
        finally:
            shutil.rmtree(out, ignore_errors=True)
    traffic = {k: int(got["fetch"][k] + got["write"][k]) for k in ("A", "B") if k in got["fetch"] and k in got["write"]}
    detail = {k: {"fetch_bytes": int(got["fetch"][k]), "write_bytes": int(got["write"][k])} for k in traffic}
    return traffic, ("; ".join(notes) if notes else "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run, "
                     "6 plain iterations each; FETCH_SIZE x2 (gfx950), Infinity-Cache hits included"), detail


def pmc_child(args):
    """What the counter passes profile: the solver on the workload, six plain (un-graphed) iterations."""
    import torch
    from __graft_entry__ import load_package
    lib = load_package().load_library()
    ext, mode = workload_extents(args.workload)
    N = 1
    for e in ext:
        N *= e
    b = torch.empty(N, dtype=torch.float32, device="cuda")
    x = torch.zeros(N, dtype=torch.float32, device="cuda")
    lib.c.cg_fill_uniform(C.c_void_p(b.data_ptr()), N, 1234)
    torch.cuda.synchronize()
    s = lib.make_solver(ext, index_mode=mode)
    s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
    s.iterate(6, use_graph=False)
    s.sync()
    s.finish()
    s.close()


def run_single(lib, ext, mode, steps, warmup, repeats, graph, barrier, device):
    """The timed regions on one GPU: warm-up, then `repeats` regions of exactly `steps` iterations of
    the same running solve, then the per-kernel event timing.  Returns a dict."""
    import torch
    N = 1
    for e in ext:
        N *= e
    b = torch.empty(N, dtype=torch.float32, device=device)
    x = torch.zeros(N, dtype=torch.float32, device=device)
    lib.c.cg_fill_uniform(C.c_void_p(b.data_ptr()), N, 1234)
    torch.cuda.synchronize()
    s = lib.make_solver(ext, index_mode=mode)
    s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
    s.iterate(warmup, use_graph=graph)
    s.sync()
    times = []
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        s.iterate(steps, use_graph=graph)
        s.sync()
        barrier()
        times.append(time.perf_counter() - t0)
    done, res, _ = s.finish()
    assert done == warmup + steps * repeats, (done, warmup, steps, repeats)
    # per-kernel device time: HIP events on the solver's stream, outside the timed regions
    msA, msB = s.time_kernels(max(10, min(50, steps)) // 2 * 2)   # whole pairs of the paired x update
    name, paired = s.kernel_name, s.paired_x
    s.close()
    del b, x
    return {"N": N, "times": times, "msA": msA, "msB": msB, "kernel": name, "residue": res, "paired_x": paired}


def roofline_of(run, traffic=None, traffic_note=None, traffic_detail=None):
    """The roofline object of the dominant kernel of one run_single() result.  achieved = the bytes the
    kernel streams per launch (moved price x points) / its mean launch time; frac = achieved / HBM peak."""
    N, msA, msB = run["N"], run["msA"], run["msB"]
    dom, other = ("B", "A") if msB >= msA else ("A", "B")
    ms = {"A": msA, "B": msB}
    BYTES = bytes_of(run.get("paired_x"))

    def rate(k, price):
        return BYTES[k][price] * N / (ms[k] * 1e-3) / 1e9

    mov = rate(dom, "moved")
    roof = {"bound": "hbm", "kernel": run["kernel"] + f"<MODE_{dom}>",
            "achieved": round(mov, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(mov / HBM_PEAK_GBPS, 4),
            "pricing": f"moved: the {BYTES[dom]['moved']} B/point this kernel streams per launch"
                       + (" (x is updated every second iteration: launches of 12 and 24 B/point alternate)"
                          if dom == "B" and run.get("paired_x") else ""),
            "algorithmic_bytes": BYTES[dom]["moved"] * N, "algorithmic_bytes_per_point": BYTES[dom]["moved"],
            "ms_per_launch": round(ms[dom], 5),
            "traffic": (traffic or {}).get(dom), "traffic_note": traffic_note,
            "three_kernel_equivalent_GBps": round(rate(dom, "equivalent"), 1),
            "other_kernel": {"kernel": f"MODE_{other}", "ms_per_launch": round(ms[other], 5),
                             "algorithmic_bytes_per_point": BYTES[other]["moved"],
                             "achieved": round(rate(other, "moved"), 1),
                             "frac": round(rate(other, "moved") / HBM_PEAK_GBPS, 4),
                             "traffic": (traffic or {}).get(other)}}
    if traffic and dom in traffic:
        roof["achieved_counter"] = round(traffic[dom] / (ms[dom] * 1e-3) / 1e9, 1)
        roof["frac_counter"] = round(roof["achieved_counter"] / HBM_PEAK_GBPS, 4)
        roof["traffic_over_algorithmic"] = round(traffic[dom] / (BYTES[dom]["moved"] * N), 4)
    if traffic and other in traffic:
        roof["other_kernel"]["traffic_over_algorithmic"] = round(traffic[other] / (BYTES[other]["moved"] * N), 4)
    if traffic_detail:
        roof["traffic_detail"] = traffic_detail
    return roof


def lap64(t):
    """A t in double with torch tensor operations (t shaped [L] * d), for residual checks."""
    d = t.dim()
    out = 2 * d * t.clone()
    for ax in range(d):
        n = t.shape[ax]
        lo = [slice(None)] * d
        hi = [slice(None)] * d
        lo[ax], hi[ax] = slice(0, n - 1), slice(1, n)
        out[tuple(lo)] -= t[tuple(hi)]
        out[tuple(hi)] -= t[tuple(lo)]
    return out


def run_mixed(args, lib, device):
    """Workload mixed-<d>d-<L> (config "mixed precision, 3-D 512^3"): the outer double CG of
    conjugate-gradient_cpu.cu:334-392 (to 1e-8 |b|) with the float GPU CG as inner solve (to 1e-3),
    b = A x_true in double, x0 = 0.  A timed region is one whole solve; value = inner float CG iterations
    per second over the whole solve, outer fp64 work included."""
    import torch
    m = re.fullmatch(r"mixed-(\d+)d-(\d+)", args.workload)
    if not m:
        raise SystemExit("mixed workloads are named mixed-<d>d-<L>, e.g. mixed-3d-512")
    d, L = int(m.group(1)), int(m.group(2))
    N = L ** d
    gen = torch.Generator(device=device).manual_seed(7)
    xt = torch.rand([L] * d, dtype=torch.float64, device=device, generator=gen)
    b = lap64(xt).reshape(-1).contiguous()
    x = torch.zeros_like(b)
    times, outer, inner = [], 0, []
    for _ in range(max(1, args.warmup > 0) + max(1, min(args.repeats, 3))):
        x.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outer, inner = lib.preconditioned_cg_gpu(b.data_ptr(), x.data_ptr(), L, d)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    times = times[1:] if len(times) > 1 else times          # the first solve warms up (graph capture, allocation)
    dt = statistics.median(times)
    res = float(((b - lap64(x.reshape([L] * d)).reshape(-1)).norm() / b.norm()).item())
    err = float((x - xt.reshape(-1)).abs().max().item())
    total = int(sum(inner))
    out = {"metric": "cg_iterations_per_sec", "value": round(total / dt, 3),
           "unit": "inner float CG iterations/s over the whole mixed-precision solve (outer fp64 work included)",
           "n_gpus": 1, "steps": total, "warmup": 1, "ms_per_step": round(dt / max(total, 1) * 1e3, 5),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 inner / f64 outer",
           "data": "synthetic",
           "config": {"workload": args.workload, "local_extents": [L] * d, "points_per_gpu": N, "parallelism": "single",
                      "rhs": "A x_true in fp64, x_true = torch.rand seed 7", "x0": 0, "outer_tolerance": 1e-8,
                      "inner_tolerance": 1e-3},
           "timing": {"regions": len(times), "statistic": "median", "seconds_per_solve": round(dt, 4),
                      "seconds_all": [round(t, 4) for t in times]},
           "outer_iterations": int(outer), "inner_iterations": [int(v) for v in inner],
           "fp64_relative_residual": res, "max_abs_error": err}
    out.update(iteration_rates(N, total / dt))
    print(json.dumps(out), flush=True)


def iteration_rates(N, its, paired=False):
    """Whole-iteration byte rates of `its` iterations/s on N points."""
    BYTES = bytes_of(paired)
    mov = BYTES["iter"]["moved"] * N * its / 1e9
    return {"hbm_moved_GBps": round(mov, 1), "hbm_frac_moved": round(mov / HBM_PEAK_GBPS, 4),
            "moved_bytes_per_point_per_iteration": BYTES["iter"]["moved"],
            "three_kernel_equivalent_GBps": round(BYTES["iter"]["equivalent"] * N * its / 1e9, 1)}


def cpu_baseline_c1():
    """Config 1 of the baseline list as is: 1-D N = 2^20, fp64, the oracle's conjugate_gradient for
    K = 20 iterations on one core (the reference loop has no cap and would need O(N) iterations)."""
    import numpy as np
    from oracle import oracle as orc
    o = orc.Oracle(native=True)
    n = 1 << 20
    b = np.random.default_rng(1234).random(n)
    t0 = time.perf_counter()
    _, _, done, _ = o.conjugate_gradient(b, n, 1, max_iters=20)
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": f"{done} fp64 CG iterations, 1-D N=2^20 ({dt:.2f} s)"}


def cpu_baseline(ext, budget_s):
    """One CG iteration's worth of the reference's own CPU calls, timed on one core: minus_laplace
    (conjugate-gradient_cpu.cu:97-113) once and inner_product (:150-158) twice from the reference's
    sources compiled unchanged with its -O3 (oracle/_ref/libcg_ref_opt.so), plus the three vector
    updates of its loop (:241-252) as numpy expressions (they are inline statements of
    conjugate_gradient, which has no iteration cap and cannot be stopped after one iteration).
    On the workload's own grid when that is a cube/square whose estimated time is under about
    3 x budget_s; otherwise on a cubic sample of the same dimension sized for budget_s, scaled by
    points (the cost per iteration is linear in points).  Without the reference build the oracle's
    restatement of the same functions is timed instead (kind "port")."""
    import numpy as np
    from oracle import oracle as orc
    ref = orc.Reference(native=True)
    kind = "reference" if ref.available else "port"
    lib = ref.lib if ref.available else orc.Oracle(native=True).lib
    lap = lib.ref_minus_laplace if ref.available else lib.orc_minus_laplace
    dot = lib.ref_inner_product if ref.available else lib.orc_inner_product
    d = len(ext)
    n_full = 1
    for e in ext:
        n_full *= e

    def one_iteration(L):
        n = L ** d
        rng = np.random.default_rng(1)
        p = np.zeros(n + 1)
        p[:n] = rng.random(n)
        r = np.zeros(n + 1)
        r[:n] = rng.random(n)
        x = np.zeros(n + 1)
        Ap = np.zeros(n + 1)
        t0 = time.perf_counter()
        lap(Ap, p, d, L, n)                       # Ap = A p
        t_lap = time.perf_counter() - t0
        rr = float(n) / 3.0                        # the loop carries r.r over from the previous iteration
        alpha = rr / dot(p, Ap, n)                # :243
        x[:n] += alpha * p[:n]                    # :244-247
        r[:n] -= alpha * Ap[:n]
        beta = dot(r, r, n) / rr                  # :248-249
        p[:n] = r[:n] + beta * p[:n]
        return time.perf_counter() - t0, t_lap, n

    L0 = {1: 1 << 16, 2: 256, 3: 40, 4: 16}.get(d, 8)
    t_cal, _, n_cal = one_iteration(L0)
    per_point = t_cal / n_cal
    cubic = all(e == ext[0] for e in ext)
    if cubic and per_point * n_full <= 3.0 * budget_s:
        L = ext[0]
    else:
        L = max(L0, min(int((budget_s / per_point) ** (1.0 / d)), min(ext)))
    t_it, t_lap, n = one_iteration(L)
    where = ("the workload's own grid" if n == n_full else
             f"a {L}^{d} sample ({n} points), scaled by points to the {'x'.join(map(str, ext))} workload")
    return {"value": 1.0 / t_it * n / n_full, "unit": "iterations/s", "cores": 1, "kind": kind, "L": L,
            "sample": f"one CG iteration's worth of calls (1 minus_laplace {t_lap:.1f} s, 2 inner_product, 3 vector "
                      f"updates; {t_it:.1f} s in all, fp64) on {where}",
            "measured_at_full_size": n == n_full}


def cpu_baseline_all_cores(ext, L):
    """The same CPU loops spread over all cores with OpenMP (oracle/cg_oracle_omp.c) on the L^d sample
    grid of cpu_baseline: a second, separately labelled figure.  The reference has no threading; this
    is not the reference."""
    import numpy as np
    path = os.path.join(ROOT, "oracle", "libcg_oracle_omp.so")
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
    lib.orc_cg_omp.argtypes = [dp, dp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.orc_cg_omp.restype = C.c_double
    d = len(ext)
    n_full = 1
    for e in ext:
        n_full *= e
    n = L ** d
    b = np.zeros(n + 1)
    b[:n] = np.random.default_rng(1).random(n)
    x = np.zeros(n + 1)
    it, th = C.c_int(0), C.c_int(0)
    t0 = time.perf_counter()
    lib.orc_cg_omp(b, x, L, d, -1, C.byref(it), C.byref(th))
    dt = time.perf_counter() - t0
    return {"value": it.value / dt * n / n_full, "unit": "iterations/s", "cores": th.value, "kind": "port-omp",
            "sample": f"{it.value} fp64 CG iterations on the {L}^{d} grid with OpenMP on {th.value} threads ({dt:.1f} s), "
                      f"scaled by points; not the reference (which is single-threaded)"}


def cpu_baselines_isolated(ext, budget_s):
    """Run the CPU legs in a child process: a host-side failure there (e.g. an instruction the
    box's CPU lacks) must not take the GPU result down with it."""
    import subprocess
    code = ("import json,sys; sys.path.insert(0, %r); import bench; "
            "cb = bench.cpu_baseline(%r, %r); "
            "L = {1: 1 << 18, 2: 320, 3: 48, 4: 16}.get(len(%r), 8); "
            "print(json.dumps({'cpu_baseline': cb, 'cpu_baseline_all_cores': bench.cpu_baseline_all_cores(%r, L), "
            "'cpu_baseline_port_1d_2e20': bench.cpu_baseline_c1()}))" % (ROOT, list(ext), float(budget_s), list(ext), list(ext)))
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
        if r.returncode == 0:
            return json.loads(r.stdout.strip().splitlines()[-1])
        err = f"exit {r.returncode}: {r.stderr[-300:]}"
    except Exception as e:  # noqa: BLE001
        err = repr(e)
    return {"cpu_baseline": {"value": None, "unit": "iterations/s", "cores": 1, "kind": "port", "sample": "failed: " + err}}


def slab_probe(s, b, x, barrier, rank, world, n=30):
    """A short timed run on the slab driver's current transport: (seconds for n iterations, max over
    ranks; residual history of the first three iterations; the ranks' error messages, empty if none).
    Every rank makes the same collective calls whatever fails on one of them."""
    import torch.distributed as dist
    x.zero_()
    barrier()
    s.begin(b, x, 0.0, 1)
    s.iterate(3)
    s.sync()
    barrier()
    t0 = time.perf_counter()
    s.iterate(n)
    s.sync()
    barrier()
    dt = time.perf_counter() - t0
    err, hist = "", []
    try:
        _, _, hist = s.finish(hist_len=3)
    except RuntimeError as e:
        err = f"rank {rank}: {e}"
    seen = [None] * world
    if world > 1:
        dist.all_gather_object(seen, (dt, err))
    else:
        seen = [(dt, err)]
    return max(t for t, _ in seen), hist, "; ".join(e for _, e in seen if e)


def check_candidate(s, name, hist_c, b, x, barrier, rank, world):
    """Set transport `name` ("rccl" | "links") up, run the probe on it, compare with the collectives'
    residual history hist_c and the neighbours' boundary planes, and take it down again.
    Returns (usable on every rank, probe seconds, reason if not usable).  Collective."""
    up = s.enable_rccl() if name == "rccl" else s.enable_links()
    if not up:
        return False, None, (s.rccl_reason if name == "rccl" else s.links_reason) or "could not be set up"
    t_new, hist_n, err_n = slab_probe(s, b, x, barrier, rank, world)
    ok = int(not err_n and len(hist_n) == 3 and all(abs(a - c) <= 1e-5 * abs(c) for a, c in zip(hist_n, hist_c)))
    why = "" if ok else (err_n or f"self-check mismatch {hist_n} vs {hist_c}")
    mine = ok
    ok = s._all_agree(ok)
    if mine and not ok:
        why = "self-check failed on another rank"
    if ok and not s._all_agree(s.ghosts_match_neighbours()):
        ok, why = False, "ghost planes differ from the neighbours' boundary planes"
    if name == "rccl":
        s.disable_rccl()
    else:
        s.disable_links()
    return bool(ok), t_new, why


def probe_child(args):
    """`--probe-child NAME`: the sandbox of one transport.  A process group of its own (the parent passes a
    different MASTER_PORT), the slab driver on the same workload, the collectives' probe and then
    check_candidate(NAME).  A transport that aborts or hangs does so here, not in the benchmark process,
    which only uses what came back clean from every rank.  Prints one line "PROBE_RESULT {json}"."""
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("CG_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(args.backend)
    lib = load_package().load_library()
    ext, _ = workload_extents(args.workload)
    if args.scaling == "strong" and world > 1:
        ext = ext[:-1] + [ext[-1] // world]
    N = 1
    for e in ext:
        N *= e
    from conjugate_gradient_amd.slab import SlabCG
    b = torch.empty(N, dtype=torch.float32, device=device)
    x = torch.zeros(N, dtype=torch.float32, device=device)
    lib.c.cg_fill_uniform(C.c_void_p(b.data_ptr()), N, 1234 + rank)
    torch.cuda.synchronize()

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    s = SlabCG(lib, ext, rank=rank, world=world, device=device)
    t_coll, hist_c, err_c = slab_probe(s, b, x, barrier, rank, world)
    if err_c:
        out = {"ok": False, "why": "collectives probe failed: " + err_c}
    else:
        ok, t_new, why = check_candidate(s, args.probe_child, hist_c, b, x, barrier, rank, world)
        out = {"ok": ok, "seconds": t_new, "collectives_seconds": t_coll, "why": why}
    print("PROBE_RESULT " + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def probe_in_sandbox(name, args, port_offset, timeout_s=60.0):
    """Run probe_child(name) as a child process of this rank (every rank does; the children form their own
    process group on MASTER_PORT + port_offset).  Returns (ok, seconds, why) of THIS rank's child; a child
    that crashes, hangs (killed after timeout_s) or prints nothing counts as not ok."""
    env = dict(os.environ, MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + port_offset))
    # under torch.distributed.run the ranks are told to use the launcher's store on MASTER_PORT; the children
    # rendezvous on a port of their own, so their rank 0 has to host the store itself
    for k in [k for k in env if k.startswith("TORCHELASTIC_")]:
        del env[k]
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--probe-child", name, "--workload", args.workload,
           "--scaling", args.scaling, "--backend", args.backend, "--gpus", str(args.gpus)]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
        for line in reversed(r.stdout.splitlines()):
            if line.startswith("PROBE_RESULT "):
                out = json.loads(line[len("PROBE_RESULT "):])
                return bool(out.get("ok")), out.get("seconds"), out.get("why", "")
        return False, None, f"child exit {r.returncode}: {(r.stderr or r.stdout)[-200:]}"
    except subprocess.TimeoutExpired:
        return False, None, "child timed out"
    except Exception as e:  # noqa: BLE001
        return False, None, repr(e)[:200]


def begin_finish_ms(lib, ext, mode, device):
    """Milliseconds of cg_solver_begin + cg_solver_finish with no iteration between them: what a solve costs
    beyond its iterations (r = b, p = 0, |b|; for a Morton grid kept in lexicographic order inside the solver
    also the three permutation passes: b and x in, x out).  Median of three after one untimed round."""
    import torch
    N = 1
    for e in ext:
        N *= e
    b = torch.empty(N, dtype=torch.float32, device=device)
    x = torch.zeros(N, dtype=torch.float32, device=device)
    lib.c.cg_fill_uniform(C.c_void_p(b.data_ptr()), N, 1234)
    s = lib.make_solver(ext, index_mode=mode)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
        s.finish()
        ts.append((time.perf_counter() - t0) * 1e3)
    s.close()
    return statistics.median(ts[1:])


def morton_both_ways(args, lib, ext, mode, run, barrier, device):
    """A Morton-ordered grid is solved in lexicographic order inside the solver (b, x permuted in at begin, x
    permuted back at finish) because those kernels move fewer bytes per iteration.  Report both: the rate of
    a whole 200-iteration solve (the reference's cap, conjugate-gradient_gpu.cu:348) with the permutations
    included, and the kernels that work on the Z-curve directly (CG_MORTON_RELAYOUT=0)."""
    cap = 200
    its = args.steps / statistics.median(run["times"])
    over = begin_finish_ms(lib, ext, mode, device)
    had = os.environ.get("CG_MORTON_RELAYOUT")
    os.environ["CG_MORTON_RELAYOUT"] = "0"
    try:
        over_n = begin_finish_ms(lib, ext, mode, device)
        run_n = run_single(lib, ext, mode, args.steps, args.warmup, max(3, args.repeats // 2 + 1), not args.no_graph, barrier, device)
    finally:
        if had is None:
            os.environ.pop("CG_MORTON_RELAYOUT")
        else:
            os.environ["CG_MORTON_RELAYOUT"] = had
    its_n = args.steps / statistics.median(run_n["times"])

    def solve_rate(rate, over_ms):
        return cap / (cap / rate + over_ms * 1e-3)

    return {"morton": {
        "ran": f"lexicographic order inside the solver ({run['kernel']}); b and x are permuted at begin, x back at finish",
        "hot_loop_iterations_per_sec": round(its, 3), "begin_finish_ms": round(over, 4),
        "iterations_per_sec_of_a_200_iteration_solve": round(solve_rate(its, over), 3),
        "native": dict({"ran": f"the Z-curve kernels ({run_n['kernel']}, CG_MORTON_RELAYOUT=0)",
                        "hot_loop_iterations_per_sec": round(its_n, 3), "begin_finish_ms": round(over_n, 4),
                        "iterations_per_sec_of_a_200_iteration_solve": round(solve_rate(its_n, over_n), 3),
                        "ms_per_step": round(1e3 / its_n, 5),
                        "roofline": roofline_of(run_n, None, "not collected for this run")},
                       **iteration_rates(run_n["N"], its_n, run_n["paired_x"]))}}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N rank processes here (fresh interpreters
    with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one per GPU), before this process has made any GPU
    call, pass rank 0's output through, and exit non-zero if any rank does."""
    import socket
    import threading
    n = args.gpus
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), CG_BENCH_SPAWNED="1")
        for k in [k for k in env if k.startswith("TORCHELASTIC_")]:
            del env[k]
        # rank 0 prints the result line; whatever the other ranks print is diagnostics
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))

    def relay():
        for line in procs[0].stdout:
            sys.stdout.write(line)
            sys.stdout.flush()

    t = threading.Thread(target=relay, daemon=True)
    t.start()
    rc = 0
    left = set(range(n))
    while left:
        for r in sorted(left):
            c = procs[r].poll()
            if c is None:
                continue
            left.discard(r)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 1
                print(f"bench.py: rank {r} exited with {c}; stopping the other ranks", file=sys.stderr, flush=True)
                for q in left:
                    procs[q].terminate()
        time.sleep(0.05)
    t.join(timeout=5)
    return rc


def pick_transport(args, s, b, x, barrier, rank, world):
    """Which exchange carries the multi-rank run: the library's own RCCL calls (graph-captured), the direct
    xGMI boxes, or the Python-driven collectives.  A candidate is used only if it could be set up on every
    rank, reproduced the collectives' residual history, delivered ghost planes that equal the neighbours'
    boundary planes bit for bit, and (with --comm auto) was the fastest in a 30-iteration probe.  The
    candidates are tried in child processes first (probe_in_sandbox): a transport that aborts or hangs on
    this machine must not take the benchmark down with it; all probes together get --probe-seconds.
    Returns (pick, note, probes)."""
    import torch.distributed as dist
    candidates = {"auto": ["rccl", "links"], "rccl": ["rccl"], "links": ["links"], "collectives": []}[args.comm]
    if args.backend != "nccl" and world > 1:
        candidates = [c for c in candidates if c != "rccl"]   # rehearsal backends share one GPU: RCCL refuses that
    probes, notes = {}, []
    if not candidates:
        return "collectives", "", probes
    t_coll, hist_c, err_c = slab_probe(s, b, x, barrier, rank, world)
    if err_c:
        raise SystemExit(f"probe over the collectives failed: {err_c}")
    probes["collectives"] = t_coll
    deadline = time.monotonic() + args.probe_seconds
    for k, name in enumerate(candidates):
        if args.no_sandbox:
            ok, t_new, why = check_candidate(s, name, hist_c, b, x, barrier, rank, world)
        else:
            # every rank allows its child the same time: rank 0's clock decides
            left = [deadline - time.monotonic()]
            if world > 1:
                dist.broadcast_object_list(left, src=0)
            if left[0] < 8.0:
                notes.append(f"{name}: not probed (the --probe-seconds budget was used up)")
                continue
            ok, t_new, why = probe_in_sandbox(name, args, 101 + 7 * k, left[0])
            seen = [None] * world
            if world > 1:
                dist.all_gather_object(seen, (ok, t_new, why))
            else:
                seen = [(ok, t_new, why)]
            ok = all(o for o, _, _ in seen)
            t_new = max((t for _, t, _ in seen if t is not None), default=None)
            why = "; ".join(sorted({w for o, _, w in seen if not o and w}))
        if ok and t_new is not None:
            probes[name] = t_new
        else:
            notes.append(f"{name}: {why}")
    usable = [c for c in candidates if c in probes]
    if args.comm == "auto":
        pick = min(["collectives"] + usable, key=lambda c: probes[c])
    elif usable:
        pick = usable[0]
    else:
        raise SystemExit(f"--comm {args.comm} unavailable: {'; '.join(notes)}")
    note = "probe seconds for 30 iterations: " + ", ".join(f"{k} {v:.4f}" for k, v in probes.items())
    if notes:
        note += "; " + "; ".join(notes)
    return pick, note, probes


COMM_NAMES = {"rccl": "rccl-in-library", "links": "xgmi-links", "collectives": "collectives"}


def run_slab(args, lib, ext, mode, rank, world, device, barrier, pick=None):
    """One multi-rank workload: every rank owns an `ext` slab.  pick: the transport to use (None: choose it,
    see pick_transport).  Returns a dict with the median-region time (max over ranks), the transport that
    ran, the roofline of rank 0's dominant kernel and the single-GPU reference of the same slab."""
    import torch
    import torch.distributed as dist
    from conjugate_gradient_amd.slab import SlabCG
    N = 1
    for e in ext:
        N *= e
    b = torch.empty(N, dtype=torch.float32, device=device)
    x = torch.zeros(N, dtype=torch.float32, device=device)
    lib.c.cg_fill_uniform(C.c_void_p(b.data_ptr()), N, 1234 + rank)   # rank-distinct synthetic right-hand side
    torch.cuda.synchronize()
    s = SlabCG(lib, ext, rank=rank, world=world, device=device)
    note = ""
    if pick is None:
        pick, note, _ = pick_transport(args, s, b, x, barrier, rank, world)
    if pick == "rccl" and not s.enable_rccl():
        pick, note = "collectives", note + f"; rccl could not be set up in the benchmark process: {s.rccl_reason}"
    if pick == "links" and not s.enable_links():
        pick, note = "collectives", note + f"; links could not be set up in the benchmark process: {s.links_reason}"
    comm = COMM_NAMES[pick]
    x.zero_()

    def timed():
        x.zero_()
        barrier()
        s.begin(b, x, 0.0, 1)
        s.iterate(args.warmup)
        s.sync()
        ts = []
        for _ in range(args.repeats):
            barrier()
            t0 = time.perf_counter()
            s.iterate(args.steps)
            s.sync()
            barrier()
            ts.append(time.perf_counter() - t0)
        done, res, _ = s.finish()
        if done != args.warmup + args.steps * args.repeats:
            raise RuntimeError(f"{done} iterations ran instead of {args.warmup + args.steps * args.repeats}")
        return ts, res

    ok = 1
    try:
        times, res = timed()
    except RuntimeError as e:
        if comm == "collectives":
            raise
        ok, note = 0, note + f"; timed run over {comm} failed: " + str(e)
    if comm != "collectives" and not s._all_agree(ok):
        # a rank lost the fast path under load: everyone goes back to the collectives
        s.disable_links()
        s.disable_rccl()
        comm, pick = "collectives", "collectives"
        times, res = timed()
    rccl_ranks = s.ops.solver.rccl_ranks() if comm == "rccl-in-library" else None
    nk = max(10, min(50, args.steps)) // 2 * 2      # whole pairs of the paired x update
    if comm != "collectives":
        with s._ctx():
            msA, msB = s.ops.solver.time_kernels(nk)
            s.sync()
        xA, xB = s.ops.solver.links_times()
        exchange = {"after_A": round(xA, 5), "after_B": round(xB, 5)}
    else:
        # over the collectives the kernels cannot be timed in place without the all-reduces between
        # them: time the same two kernels on this rank's slab with a stand-alone solver
        had = os.environ.get("CG_XDEFER")
        os.environ["CG_XDEFER"] = "0"     # as the step interface runs them: x updated by every kernel B
        t = lib.make_solver(ext, index_mode=mode)
        if had is None:
            os.environ.pop("CG_XDEFER")
        else:
            os.environ["CG_XDEFER"] = had
        x.zero_()
        torch.cuda.synchronize()
        t.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
        t.iterate(5, use_graph=False)
        msA, msB = t.time_kernels(nk)
        t.close()
        exchange = None
    # the step interface the Python-driven collectives use updates x in every kernel B
    paired = s.ops.solver.paired_x and comm != "collectives"
    roof = roofline_of({"N": N, "msA": msA, "msB": msB, "kernel": s.ops.solver.kernel_name, "paired_x": paired},
                       None, "not collected in multi-rank runs")
    roof["exchange_ms"] = exchange
    roof["rank"] = 0
    if comm == "rccl-in-library":
        s.disable_rccl()
    elif comm == "xgmi-links":
        s.disable_links()
    s.ops.solver.close()
    del s
    # the same slab on one GPU with nothing to exchange, in the same run: what N ranks would each do alone
    barrier()
    x.zero_()
    solo = lib.make_solver(ext, index_mode=mode)
    solo.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
    solo.iterate(args.warmup, use_graph=not args.no_graph)
    solo.sync()
    ts1 = []
    for _ in range(max(3, args.repeats // 2 + 1)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solo.iterate(args.steps, use_graph=not args.no_graph)
        solo.sync()
        ts1.append(time.perf_counter() - t0)
    solo.finish()
    solo.close()
    del b, x
    both = torch.tensor(times + [statistics.median(ts1)], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(both, op=dist.ReduceOp.MAX)      # every region's time is the maximum over ranks
    both = [float(v) for v in both.tolist()]
    return {"N": N, "times": both[:-1], "solo_seconds": both[-1], "res": res, "comm": comm, "pick": pick, "note": note.strip("; "),
            "roof": roof, "paired": bool(paired), "rccl_ranks": rccl_ranks}


def slab_line(args, ext, mode, workload, world, r):
    """The result fields of one run_slab() outcome."""
    N, times = r["N"], r["times"]
    dt = statistics.median(times)
    its = args.steps / dt
    solo = args.steps / r["solo_seconds"]
    weak = args.scaling == "weak"
    out = {"metric": "cg_iterations_per_sec", "value": round(its * (world if weak else 1), 3),
           "unit": ("CG iterations/s of the whole grid (strong scaling)" if not weak else
                    "CG iterations/s per 2^%d-point slab, summed over GPUs" % (N.bit_length() - 1)
                    if (N & (N - 1)) == 0 else "CG iterations/s per slab, summed over GPUs"),
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt / args.steps * 1e3, 5), "higher_is_better": True, "scaling": args.scaling,
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": workload, "local_extents": ext, "global_extents": ext[:-1] + [ext[-1] * world],
                      "index_mode": mode, "points_per_gpu": N, "parallelism": f"slab{world}", "graph": True,
                      "backend": args.backend, "comm": r["comm"] + (f" ({r['note']})" if r["note"] else ""),
                      "rccl_ranks": r["rccl_ranks"], "paired_x_update": r["paired"], "rhs": "u01 seed 1234+rank", "x0": 0},
           "timing": {"regions": args.repeats, "statistic": "median, every region the maximum over ranks",
                      "ms_per_step_min_max": [round(min(times) / args.steps * 1e3, 5), round(max(times) / args.steps * 1e3, 5)],
                      "ms_per_step_all": [round(t / args.steps * 1e3, 5) for t in times]},
           "iterations_per_sec_global": round(its, 3),
           "single_gpu_reference": {"value": round(solo, 3), "ms_per_step": round(r["solo_seconds"] / args.steps * 1e3, 5),
                                    "note": "the same slab on one GPU with no neighbours and no exchange, timed in this run "
                                            "after the multi-rank regions (median region, maximum over ranks)"},
           "final_residue": r["res"]}
    out.update(iteration_rates(N * world, its, r["paired"]))   # all ranks' points move once per global iteration
    out["roofline"] = r["roof"]
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "CG_BENCH_DEVICE" in os.environ:  # rehearsal on a one-GPU box: every rank on the same device
        local_rank = int(os.environ["CG_BENCH_DEVICE"])
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank process per GPU is needed "
                         "(plain `python bench.py --gpus N` starts them itself; so does torch.distributed.run)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank} needs GPU {local_rank}, this node shows {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or args.force_slab:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    pkg = load_package()
    lib = pkg.load_library()
    if args.workload.startswith("mixed-"):
        if world > 1:
            raise SystemExit("the mixed-precision workload is a single-GPU configuration")
        run_mixed(args, lib, device)
        return
    ext, mode = workload_extents(args.workload)
    if args.scaling == "strong" and world > 1:
        if ext[-1] % world:
            raise SystemExit(f"the slowest axis ({ext[-1]}) must be divisible by --gpus {world}")
        ext = ext[:-1] + [ext[-1] // world]

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    graph = not args.no_graph
    if world == 1 and not args.force_slab:
        run = run_single(lib, ext, mode, args.steps, args.warmup, args.repeats, graph, barrier, device)
        N, times = run["N"], run["times"]
        traffic, tnote, tdetail = ({}, "counter passes skipped (--no-traffic)", {}) if args.no_traffic else live_traffic(args.workload)
        dt = statistics.median(times)
        its = args.steps / dt
        out = {
            "metric": "cg_iterations_per_sec", "value": round(its, 3),
            "unit": "CG iterations/s per 2^%d-point slab, summed over GPUs" % (N.bit_length() - 1)
                    if (N & (N - 1)) == 0 else "CG iterations/s per slab, summed over GPUs",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 5), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "local_extents": ext, "index_mode": mode,
                       "points_per_gpu": N, "parallelism": "single", "graph": graph, "backend": None, "comm": None,
                       "kernel": run["kernel"], "paired_x_update": bool(run["paired_x"]), "rhs": "u01 seed 1234+rank", "x0": 0},
            "timing": {"regions": args.repeats, "statistic": "median",
                       "ms_per_step_min_max": [round(min(times) / args.steps * 1e3, 5), round(max(times) / args.steps * 1e3, 5)],
                       "ms_per_step_all": [round(t / args.steps * 1e3, 5) for t in times]},
            "iterations_per_sec_global": round(its, 3),
            "final_residue": run["residue"],
        }
        out.update(iteration_rates(N, its, run["paired_x"]))
        out["roofline"] = roofline_of(run, traffic, tnote, tdetail)
        if mode == 2 and run["kernel"] not in ("k_morton_tile", "k_morton4"):
            # vectors in Morton order, solved in lexicographic order inside the solver: the figure of merit is a
            # whole solve with the permutation passes in it; the hot loop alone and the Z-curve kernels beside it
            out.update(morton_both_ways(args, lib, ext, mode, run, barrier, device))
            out["value"] = out["morton"]["iterations_per_sec_of_a_200_iteration_solve"]
            out["ms_per_step"] = round(1e3 / out["value"], 5)
            out["unit"] = "CG iterations/s of a 200-iteration solve on Morton-ordered vectors, the three permutation passes included"
            out["config"]["ran"] = out["morton"]["ran"]
        if not args.no_beyond_cache and args.workload == "2d-8192":
            # The same kernels on a 2-D grid whose vectors are twice the Infinity Cache, with the cache taken out of
            # play (CG_KEEP_MB=0: no part of r is kept, every stream carries the non-temporal hint): the HBM-bound
            # rate.  The solver's own choice on this grid -- half of r kept -- is timed beside it.
            ext2, mode2 = workload_extents(BEYOND_CACHE_WORKLOAD)
            reps2 = max(3, args.repeats // 2 + 1)
            run2d = run_single(lib, ext2, mode2, args.steps, args.warmup, reps2, graph, barrier, device)
            had = os.environ.get("CG_KEEP_MB")
            os.environ["CG_KEEP_MB"] = "0"
            try:
                run2 = run_single(lib, ext2, mode2, args.steps, args.warmup, reps2, graph, barrier, device)
                tr2, tn2, td2 = ({}, "counter passes skipped (--no-traffic)", {}) if args.no_traffic else live_traffic(BEYOND_CACHE_WORKLOAD)
            finally:
                if had is None:
                    os.environ.pop("CG_KEEP_MB")
                else:
                    os.environ["CG_KEEP_MB"] = had
            t2 = statistics.median(run2["times"])
            its2 = args.steps / t2
            t2d = statistics.median(run2d["times"])
            out["roofline_hbm_bound"] = dict(
                {"workload": BEYOND_CACHE_WORKLOAD, "points": run2["N"], "iterations_per_sec": round(its2, 3),
                 "why": "every vector of this grid is 512 MiB, twice the Infinity Cache, and the run keeps nothing in it "
                        "(CG_KEEP_MB=0): what HBM alone sustains for this traffic mix",
                 "with_default_policy": {"iterations_per_sec": round(args.steps / t2d, 3),
                                         "ms_per_step": round(t2d / args.steps * 1e3, 5),
                                         "kernel_ms": {"A": round(run2d["msA"], 5), "B": round(run2d["msB"], 5)},
                                         "note": "the solver's own choice on this grid: the 256 MiB of r that fit the "
                                                 "Infinity Cache are kept in it, the other half streams"},
                 "ms_per_step": round(t2 / args.steps * 1e3, 5),
                 "ms_per_step_min_max": [round(min(run2["times"]) / args.steps * 1e3, 5),
                                         round(max(run2["times"]) / args.steps * 1e3, 5)],
                 "roofline": roofline_of(run2, tr2, tn2, td2)},
                **iteration_rates(run2["N"], its2, run2["paired_x"]))
        if not args.no_cpu_baseline:
            out.update(cpu_baselines_isolated(ext, args.cpu_seconds))
        print(json.dumps(out), flush=True)
    else:
        r = run_slab(args, lib, ext, mode, rank, world, device, barrier)
        out = slab_line(args, ext, mode, args.workload, world, r)
        # config "3D 1024^3 slab-decomposed over 8 GPUs", as weak scaling of its per-GPU slab, over the same transport
        if world > 1 and not args.no_config3 and args.workload == "2d-8192" and args.scaling == "weak":
            ext3, mode3 = workload_extents(CONFIG3_SLAB)
            try:
                r3 = run_slab(args, lib, ext3, mode3, rank, world, device, barrier, pick=r["pick"])
                c3 = slab_line(args, ext3, mode3, CONFIG3_SLAB, world, r3)
                for k in ("metric", "higher_is_better", "vs_baseline", "dtype", "data", "steps", "warmup", "n_gpus"):
                    c3.pop(k, None)
                out["config3"] = c3
            except RuntimeError as e:
                out["config3"] = {"failed": str(e)[:300]}
        if rank == 0:
            print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    if "--pmc-child" in sys.argv:
        pmc_child(parse())
    elif "--probe-child" in sys.argv:
        probe_child(parse())
    else:
        main()
