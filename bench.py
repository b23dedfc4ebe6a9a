#!/usr/bin/env python3
"""Benchmark of the fused Laplace-CG iteration on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one CG iteration (kernel A + kernel B) over the whole grid.  Workload at every N:
each GPU owns an 8192 x 8192 float slab (2^26 points, config "2D Laplace 8192^2 float CG");
with N > 1 the slabs stack along axis 1 into an 8192 x (8192 N) grid (weak scaling) with one
halo row of r each way and two dot-product exchanges per iteration: written straight into the
peers' HBM over xGMI when every rank can map every other rank's box and a three-iteration
self-check reproduces the collectives' residuals, otherwise over RCCL (config.comm says which).  Synthetic data: b = u01
(counter-based generator, seed 1234), x0 = 0; float CG cannot converge on this grid within
any practical count, so every step does full work (the stop tolerance is set to 0).

value = slab-iterations per second summed over ranks (= CG iterations/s at N = 1).
roofline: algorithmic HBM bytes of the dominant kernel (B: 24 B/point) over its mean
duration measured with HIP events on the solver's stream.  cpu_baseline: the oracle's
own CPU solver built from its sources (fp64, one core) on a bounded sample, beside the oracle's
restatement of it.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak
BYTES_A, BYTES_B = 20, 24       # algorithmic bytes/point: A = stencil+dot (8) + p update (12); B = x,r update + dot
BYTES_ITER = 44


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="2d-8192", help="2d-8192 | 3d-1024x128 | 3d-512 | 3d-1024 | 4d-128-morton | LxLx..[:mode]")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--force-slab", action="store_true", help="run the multi-rank driver even with one rank")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the measured path) | gloo (rehearsal: host-staged)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every GPU owns a --workload grid; strong: the --workload grid is cut into --gpus slabs")
    ap.add_argument("--comm", default="auto", help="multi-rank exchange: auto (direct xGMI boxes if they pass a "
                    "self-check, else collectives) | links | collectives")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
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


def measured_traffic(workload, kernel_mode):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 counter
    passes (profiles/*_summary.json, written by tools/summarize_profile.py); None if no
    profile of this workload is in the tree."""
    import glob
    tag = workload.replace("-", "")
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}_summary.json")))
    if not files:
        return None
    try:
        ks = json.load(open(files[-1]))["kernels"]
        for name, k in ks.items():
            if any(f"{k}<{kernel_mode}," in name for k in ("k_rows", "k_cols", "k_scalar", "k_planes", "k_morton4", "k_morton_tile")):
                return k.get("traffic_bytes")
    except Exception:
        return None
    return None


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
    """The reference's own conjugate_gradient (conjugate-gradient_cpu.cu:216-261, fp64, one thread),
    compiled unchanged from its sources with its -O3 (oracle/_ref/libcg_ref_opt.so), on a bounded
    sample: a square/cubic grid of the workload's dimension sized so that the solve to the
    reference's own tolerance takes roughly budget_s / 2.  That function has no iteration counter;
    the count comes from the oracle's restatement run on the same input, which is also timed
    ("port") and whose solution must equal the reference's bit for bit.  iterations/s is scaled to
    the workload's point count (cost per iteration is linear in points).  Without the reference
    build the port alone is reported."""
    import numpy as np
    from oracle import oracle as orc
    o = orc.Oracle(native=True)
    d = len(ext)
    n_full = 1
    for e in ext:
        n_full *= e
    # calibrate on a small grid: iterations grow like L, cost per iteration like L^d
    L0 = {1: 192, 2: 64, 3: 20, 4: 10}.get(d, 6)
    b0 = np.random.default_rng(0).random(L0 ** d)
    t0 = time.perf_counter()
    o.conjugate_gradient(b0, L0, d)
    t_cal = max(time.perf_counter() - t0, 1e-4)
    L = int(L0 * (0.5 * budget_s / t_cal) ** (1.0 / (d + 1)))
    L = max(L0, min(L, min(ext)))
    n = L ** d
    b = np.random.default_rng(1).random(n)
    t0 = time.perf_counter()
    x_port, _, done, _ = o.conjugate_gradient(b, L, d)
    dt_port = time.perf_counter() - t0
    scale = n / n_full
    port = {"value": done / dt_port * scale, "unit": "iterations/s", "cores": 1, "kind": "port", "L": L,
            "sample": f"{done} fp64 CG iterations to 1e-6 on a {L}^{d} grid ({n} points, {dt_port:.1f} s), "
                      f"scaled by points to the {'x'.join(map(str, ext))} workload"}
    ref = orc.Reference(native=True)
    if not ref.available:
        return port
    t0 = time.perf_counter()
    x_ref, _ = ref.conjugate_gradient(b, L, d)
    dt_ref = time.perf_counter() - t0
    return {"value": done / dt_ref * scale, "unit": "iterations/s", "cores": 1, "kind": "reference",
            "sample": f"the reference's conjugate_gradient to its 1e-6 tolerance on a {L}^{d} grid ({n} points): "
                      f"{done} iterations in {dt_ref:.1f} s, scaled by points to the {'x'.join(map(str, ext))} workload",
            "bit_identical_to_port": bool(np.array_equal(x_ref, x_port)),
            "port": port}


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
            "L = cb.get('L') or cb.get('port', {}).get('L'); "
            "print(json.dumps({'cpu_baseline': cb, 'cpu_baseline_all_cores': bench.cpu_baseline_all_cores(%r, L), "
            "'cpu_baseline_1d_2e20': bench.cpu_baseline_c1()}))" % (ROOT, list(ext), float(budget_s), list(ext)))
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
        if r.returncode == 0:
            return json.loads(r.stdout.strip().splitlines()[-1])
        err = f"exit {r.returncode}: {r.stderr[-300:]}"
    except Exception as e:  # noqa: BLE001
        err = repr(e)
    return {"cpu_baseline": {"value": None, "unit": "iterations/s", "cores": 1, "kind": "port", "sample": "failed: " + err}}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "CG_BENCH_DEVICE" in os.environ:  # rehearsal on a one-GPU box: every rank on the same device
        local_rank = int(os.environ["CG_BENCH_DEVICE"])
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
    ext, mode = workload_extents(args.workload)
    if args.scaling == "strong" and world > 1:
        if ext[-1] % world:
            raise SystemExit(f"the slowest axis ({ext[-1]}) must be divisible by --gpus {world}")
        ext = ext[:-1] + [ext[-1] // world]
    N = 1
    for e in ext:
        N *= e

    b = torch.empty(N, dtype=torch.float32, device=device)
    x = torch.zeros(N, dtype=torch.float32, device=device)
    # rank-distinct synthetic right-hand side: one global u01 stream cut into slabs
    lib.c.cg_fill_uniform(C.c_void_p(b.data_ptr()), N, 1234 + rank)
    torch.cuda.synchronize()

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    roof = None
    if world == 1 and not args.force_slab:
        s = lib.make_solver(ext, index_mode=mode)
        s.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
        s.iterate(args.warmup, use_graph=not args.no_graph)
        s.sync()
        barrier()
        t0 = time.perf_counter()
        s.iterate(args.steps, use_graph=not args.no_graph)
        s.sync()
        barrier()
        dt = time.perf_counter() - t0
        done, res, _ = s.finish()
        assert done == args.warmup + args.steps, (done, args.warmup, args.steps)
        # per-kernel device time, HIP events on the solver's stream, outside the timed region
        msA, msB = s.time_kernels(max(10, min(50, args.steps)))
        gA = BYTES_A * N / (msA * 1e-3) / 1e9
        gB = BYTES_B * N / (msB * 1e-3) / 1e9
        dom = ("B", msB, gB, BYTES_B) if msB >= msA else ("A", msA, gA, BYTES_A)
        roof = {"bound": "hbm", "kernel": s.kernel_name + f"<MODE_{dom[0]}>",
                "achieved": round(dom[2], 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(dom[2] / HBM_PEAK_GBPS, 4),
                "traffic": measured_traffic(args.workload, 2 if dom[0] == "B" else 1),
                "algorithmic_bytes": dom[3] * N,
                "algorithmic_bytes_per_point": dom[3], "ms_per_launch": round(dom[1], 5),
                "other_kernel": {"ms_per_launch": round(msA if dom[0] == "B" else msB, 5),
                                 "achieved": round(gA if dom[0] == "B" else gB, 1)}}
        parallelism = "single"
    else:
        from conjugate_gradient_amd.slab import SlabCG
        s = SlabCG(lib, ext, rank=rank, world=world, device=device)
        comm, comm_note = "collectives", ""

        def probe(n=30):
            """A short timed run on the current transport: (seconds for n iterations, max over
            ranks; residual history of the first three iterations; the ranks' error messages, empty
            if none).  Every rank makes the same collective calls whatever fails on one of them."""
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

        if args.comm in ("auto", "links"):
            # the direct path is used only if every rank mapped every box, the boxes reproduce the
            # collectives' residual history and ghost planes bit for bit, and it is not slower
            t_coll, hist_c, err_c = probe()
            if err_c:
                raise SystemExit(f"probe over the collectives failed: {err_c}")
            if s.enable_links():
                t_link, hist_l, err_l = probe()
                ok = int(not err_l and len(hist_l) == 3 and
                         all(abs(a - c) <= 1e-5 * abs(c) for a, c in zip(hist_l, hist_c)))
                if not ok:
                    comm_note = err_l or f"self-check mismatch {hist_l} vs {hist_c}"
                mine = ok
                ok = s._all_agree(ok)
                if mine and not ok:
                    comm_note = "self-check failed on another rank"
                if ok and not s._all_agree(s.ghosts_match_neighbours()):
                    ok, comm_note = False, "self-check: ghost planes differ from the neighbours' boundary planes"
                if ok and args.comm == "auto" and t_link > t_coll:
                    ok, comm_note = False, f"boxes slower than collectives in the probe ({t_link:.4f} s vs {t_coll:.4f} s)"
                if ok:
                    comm = "xgmi-links"
                    comm_note = f"probe: {t_link:.4f} s vs {t_coll:.4f} s over collectives for 30 iterations"
                else:
                    s.disable_links()
            else:
                comm_note = s.links_reason
            if args.comm == "links" and comm != "xgmi-links":
                raise SystemExit(f"direct links unavailable: {comm_note}")
            x.zero_()
        def timed():
            x.zero_()
            barrier()
            s.begin(b, x, 0.0, 1)
            s.iterate(args.warmup)
            s.sync()
            barrier()
            t0 = time.perf_counter()
            s.iterate(args.steps)
            s.sync()
            barrier()
            dt = time.perf_counter() - t0
            done, res, _ = s.finish()
            if done != args.warmup + args.steps:
                raise RuntimeError(f"{done} iterations ran instead of {args.warmup + args.steps}")
            return dt, res

        ok = 1
        try:
            dt, res = timed()
        except RuntimeError as e:
            if comm != "xgmi-links":
                raise
            ok, comm_note = 0, "timed run over the boxes failed: " + str(e)
        if comm == "xgmi-links" and not s._all_agree(ok):
            # a rank lost the direct path under load: everyone goes back to the collectives
            s.disable_links()
            comm = "collectives"
            dt, res = timed()
        parallelism = f"slab{world}"
        if comm == "xgmi-links":
            with s._ctx():
                msA, msB = s.ops.solver.time_kernels(max(10, min(50, args.steps)))
                s.sync()
            xA, xB = s.ops.solver.links_times()
            exchange = {"after_A": round(xA, 5), "after_B": round(xB, 5)}
        else:
            # over the collectives the kernels cannot be timed in place without the all-reduces between
            # them: time the same two kernels on this rank's slab with a stand-alone solver
            t = lib.make_solver(ext, index_mode=mode)
            x.zero_()
            torch.cuda.synchronize()
            t.begin(b.data_ptr(), x.data_ptr(), 0.0, 1)
            t.iterate(5, use_graph=False)
            msA, msB = t.time_kernels(max(10, min(50, args.steps)))
            t.close()
            exchange = None
        gB = BYTES_B * N / (msB * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "MODE_B", "achieved": round(gB, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(gB / HBM_PEAK_GBPS, 4), "traffic": measured_traffic(args.workload, 2),
                "algorithmic_bytes": BYTES_B * N, "ms_per_launch": round(msB, 5),
                "other_kernel": {"ms_per_launch": round(msA, 5)}, "exchange_ms": exchange, "rank": 0}

    tmax = torch.tensor([dt], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        its = args.steps / dt
        out = {
            "metric": "cg_iterations_per_sec", "value": round(its * (world if args.scaling == "weak" else 1), 3),
            "unit": ("CG iterations/s of the whole grid (strong scaling)" if args.scaling == "strong" else
                     "CG iterations/s per 2^%d-point slab, summed over GPUs" % (N.bit_length() - 1)
                     if (N & (N - 1)) == 0 else "CG iterations/s per slab, summed over GPUs"),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 5), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "local_extents": ext, "index_mode": mode,
                       "points_per_gpu": N, "parallelism": parallelism, "graph": not args.no_graph,
                       "backend": args.backend if world > 1 else None,
                       "comm": (comm + (f" ({comm_note})" if comm_note else "")) if parallelism != "single" else None,
                       "rhs": "u01 seed 1234+rank", "x0": 0},
            "iterations_per_sec_global": round(its, 3),
            "hbm_algorithmic_GBps": round(BYTES_ITER * N * world * its / 1e9, 1),
            "hbm_frac_of_peak": round(BYTES_ITER * N * its / 1e9 / HBM_PEAK_GBPS, 4),
            "final_residue": res,
        }
        if roof:
            out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            out.update(cpu_baselines_isolated(ext, args.cpu_seconds))
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
