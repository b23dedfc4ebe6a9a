"""ctypes binding of libconjugate_gradient_gpu.so (see include/conjugate_gradient_gpu.h).

Function names, argument order and meaning follow the reference's Julia wrappers
(/root/reference/src/main.jl:16-58), so a call site reads like the reference's own.
Device vectors are passed as integer addresses (e.g. torch_tensor.data_ptr()).
"""
import ctypes as C
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CG_LIB_PATH") or os.path.join(PKG_DIR, "lib", "libconjugate_gradient_gpu.so")

EXPORTS = [
    # part 1: the reference's symbols
    "conjugate_gradient_gpu", "inner_product_gpu", "neighbour_index_gpu", "strong_scaling",
    "time_cpu_laplace", "add_jl", "laplace_gpu_jl",
    # part 2: additions
    "cg_allocate_field", "cg_free_field", "conjugate_gradient_gpu_ex", "preconditioned_cg_gpu",
    "cg_eigen_extremes_gpu", "cg_two_grid_gpu", "cg_two_grid_pcg_gpu", "cg_multigrid_gpu",
    "cg_laplace_device", "cg_fill_uniform", "cg_u01_host", "cg_strong_scaling_events", "cg_laplace_time_events",
    "cg_solver_buffer_len", "cg_solver_ghost_offset", "cg_solver_plane", "cg_solver_create",
    "cg_solver_destroy", "cg_solver_engine", "cg_solver_blocks", "cg_solver_paired_x", "cg_solver_kernel_name", "cg_solver_begin", "cg_solver_iterate",
    "cg_solver_finish", "cg_solver_sync", "cg_solver_time_kernels", "cg_solver_begin_local",
    "cg_solver_reduce", "cg_solver_bind_sums", "cg_solver_bind_partials", "cg_solver_init_state", "cg_solver_add_iterations",
    "cg_solver_kernel_A", "cg_solver_kernel_B", "cg_solver_r", "cg_solver_p",
    "cg_links_box_bytes", "cg_links_alloc", "cg_links_free", "cg_links_peer_ok", "cg_links_export", "cg_links_open",
    "cg_links_close", "cg_links_poke", "cg_links_peek", "cg_solver_bind_links", "cg_solver_exchange", "cg_solver_links_error",
    "cg_solver_links_times", "cg_noalloc_prepare", "cg_inner_product_noalloc",
    "cg_solver_bind_partials_n", "cg_solver_blocks_of", "cg_release_cache", "cg_rccl_unique_id", "cg_rccl_comm_create", "cg_rccl_comm_destroy", "cg_solver_bind_rccl", "cg_solver_rccl_ranks",
    "cg_solver_bind_rccl_peers", "cg_solver_edge_copy", "cg_solver_edge_update", "cg_solver_edge_plane",
]


class SolverDesc(C.Structure):
    _fields_ = [("d", C.c_int), ("extents", C.c_uint * 32), ("index_mode", C.c_int),
                ("has_lo", C.c_int), ("has_hi", C.c_int), ("stream", C.c_void_p),
                ("r_buf", C.c_void_p), ("p0_buf", C.c_void_p), ("p1_buf", C.c_void_p),
                ("engine", C.c_int)]


def build_library():
    """Compile the HIP library in-tree (hipcc cross-compiles for gfx950 without a GPU)."""
    subprocess.run(["make", "-s", "-j8", "-C", PKG_DIR], check=True)
    return LIB_PATH


def _ptr(p):
    return C.c_void_p(int(p) if p else None)


class Library:
    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `make -C {PKG_DIR}` "
                "(there is no fallback implementation)")
        self.path = path
        # torch ships its own HIP runtime; when both live in one process it has to be the
        # one already loaded, or device pointers from torch are foreign to this library
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        self.c = C.CDLL(path)
        c = self.c
        fp, vp, u32, i32 = C.c_void_p, C.c_void_p, C.c_uint, C.c_int
        c.conjugate_gradient_gpu.argtypes = [fp, fp, i32, i32]
        c.conjugate_gradient_gpu.restype = C.c_float
        c.inner_product_gpu.argtypes = [fp, fp, u32]
        c.inner_product_gpu.restype = C.c_float
        c.neighbour_index_gpu.argtypes = [u32, i32, i32, u32, u32, u32, i32]
        c.neighbour_index_gpu.restype = u32
        c.strong_scaling.argtypes = [i32] * 5
        c.strong_scaling.restype = C.c_float
        c.time_cpu_laplace.argtypes = [i32] * 3
        c.time_cpu_laplace.restype = C.c_float
        c.add_jl.argtypes = [fp, fp, i32, i32]
        c.add_jl.restype = None
        c.laplace_gpu_jl.argtypes = [fp, fp, C.c_float, i32, i32, i32, u32, u32, u32]
        c.laplace_gpu_jl.restype = None
        c.cg_allocate_field.argtypes = [i32]
        c.cg_allocate_field.restype = C.c_void_p
        c.cg_free_field.argtypes = [fp]
        c.cg_free_field.restype = None
        c.conjugate_gradient_gpu_ex.argtypes = [fp, fp, i32, i32, i32, i32, C.c_double, i32, i32,
                                                C.POINTER(C.c_int), vp, i32]
        c.conjugate_gradient_gpu_ex.restype = C.c_float
        c.preconditioned_cg_gpu.argtypes = [vp, vp, i32, i32, vp, i32, i32]
        c.preconditioned_cg_gpu.restype = i32
        c.cg_eigen_extremes_gpu.argtypes = [i32, i32, i32, C.c_double, C.c_uint, vp, vp, vp, vp, vp]
        c.cg_eigen_extremes_gpu.restype = i32
        c.cg_two_grid_gpu.argtypes = [vp, vp, i32, i32, i32, i32, C.c_double, i32, C.c_double, C.c_double, vp, i32, vp]
        c.cg_two_grid_gpu.restype = i32
        c.cg_two_grid_pcg_gpu.argtypes = [vp, vp, i32, i32, i32, C.c_double, i32, C.c_double, C.c_double, vp, i32, vp]
        c.cg_two_grid_pcg_gpu.restype = i32
        c.cg_multigrid_gpu.argtypes = [vp, vp, i32, i32, i32, i32, i32, C.c_double, i32, C.c_double, C.c_double, vp, i32,
                                       vp, vp]
        c.cg_multigrid_gpu.restype = i32
        c.cg_laplace_device.argtypes = [fp, fp, i32, i32, i32, vp]
        c.cg_laplace_device.restype = None
        c.cg_fill_uniform.argtypes = [fp, C.c_ulonglong, u32]
        c.cg_fill_uniform.restype = None
        c.cg_u01_host.argtypes = [u32, C.c_ulonglong]
        c.cg_u01_host.restype = C.c_float
        c.cg_strong_scaling_events.argtypes = [i32] * 6
        c.cg_strong_scaling_events.restype = C.c_float
        c.cg_laplace_time_events.argtypes = [i32] * 4
        c.cg_laplace_time_events.restype = C.c_float
        dp = C.POINTER(SolverDesc)
        for f in ("cg_solver_buffer_len", "cg_solver_ghost_offset", "cg_solver_plane"):
            getattr(c, f).argtypes = [dp]
            getattr(c, f).restype = C.c_size_t
        c.cg_solver_create.argtypes = [dp]
        c.cg_solver_create.restype = C.c_void_p
        c.cg_solver_destroy.argtypes = [vp]
        c.cg_solver_destroy.restype = None
        for f in ("cg_solver_engine", "cg_solver_blocks", "cg_solver_paired_x"):
            getattr(c, f).argtypes = [vp]
            getattr(c, f).restype = i32
        c.cg_solver_blocks_of.argtypes = [vp, i32]
        c.cg_solver_blocks_of.restype = i32
        c.cg_release_cache.argtypes = []
        c.cg_release_cache.restype = None
        c.cg_solver_kernel_name.argtypes = [vp]
        c.cg_solver_kernel_name.restype = C.c_char_p
        c.cg_solver_begin.argtypes = [vp, fp, fp, C.c_double, i32]
        c.cg_solver_begin.restype = None
        c.cg_solver_iterate.argtypes = [vp, i32, i32]
        c.cg_solver_iterate.restype = None
        c.cg_solver_finish.argtypes = [vp, C.POINTER(C.c_float), vp, i32]
        c.cg_solver_finish.restype = i32
        c.cg_solver_sync.argtypes = [vp]
        c.cg_solver_sync.restype = None
        c.cg_solver_time_kernels.argtypes = [vp, i32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        c.cg_solver_time_kernels.restype = None
        c.cg_solver_begin_local.argtypes = [vp, fp, fp]
        c.cg_solver_begin_local.restype = None
        c.cg_solver_reduce.argtypes = [vp, i32, vp]
        c.cg_solver_reduce.restype = None
        c.cg_solver_bind_sums.argtypes = [vp, vp, vp]
        c.cg_solver_bind_sums.restype = None
        c.cg_solver_bind_partials.argtypes = [vp, vp, vp]
        c.cg_solver_bind_partials.restype = None
        c.cg_solver_init_state.argtypes = [vp, C.c_double, i32, i32]
        c.cg_solver_init_state.restype = None
        c.cg_solver_add_iterations.argtypes = [vp, i32]
        c.cg_solver_add_iterations.restype = None
        for f in ("cg_solver_kernel_A", "cg_solver_kernel_B"):
            getattr(c, f).argtypes = [vp]
            getattr(c, f).restype = None
        c.cg_solver_r.argtypes = [vp]
        c.cg_solver_r.restype = C.c_void_p
        c.cg_solver_p.argtypes = [vp, i32]
        c.cg_solver_p.restype = C.c_void_p
        c.cg_links_box_bytes.argtypes = [dp]
        c.cg_links_box_bytes.restype = C.c_size_t
        c.cg_links_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
        c.cg_links_export.argtypes = [vp, C.c_char_p]
        c.cg_links_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        for f in ("cg_links_free", "cg_links_close"):
            getattr(c, f).argtypes = [vp]
        for f in ("cg_links_alloc", "cg_links_free", "cg_links_export", "cg_links_open", "cg_links_close"):
            getattr(c, f).restype = i32
        c.cg_links_poke.argtypes = [vp, i32, C.c_ulonglong]
        c.cg_links_poke.restype = i32
        c.cg_links_peek.argtypes = [vp, i32, C.POINTER(C.c_ulonglong)]
        c.cg_links_peek.restype = i32
        c.cg_links_peer_ok.argtypes = [i32]
        c.cg_links_peer_ok.restype = i32
        c.cg_solver_bind_links.argtypes = [vp, i32, i32, C.POINTER(C.c_void_p)]
        c.cg_solver_bind_links.restype = None
        c.cg_solver_exchange.argtypes = [vp, i32, i32]
        c.cg_solver_exchange.restype = None
        c.cg_solver_links_error.argtypes = [vp]
        c.cg_solver_links_error.restype = i32
        c.cg_solver_links_times.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        c.cg_solver_links_times.restype = None
        c.cg_solver_bind_partials_n.argtypes = [vp, vp, vp, i32]
        c.cg_solver_bind_partials_n.restype = None
        c.cg_rccl_unique_id.argtypes = [C.c_char_p]
        c.cg_rccl_unique_id.restype = i32
        c.cg_rccl_comm_create.argtypes = [C.c_char_p, i32, i32, C.POINTER(C.c_void_p)]
        c.cg_rccl_comm_create.restype = i32
        c.cg_rccl_comm_destroy.argtypes = [vp]
        c.cg_rccl_comm_destroy.restype = i32
        c.cg_solver_bind_rccl.argtypes = [vp, vp, i32, i32]
        c.cg_solver_bind_rccl.restype = i32
        c.cg_solver_bind_rccl_peers.argtypes = [vp, vp, i32, i32, i32, i32]
        c.cg_solver_bind_rccl_peers.restype = i32
        for f in ("cg_solver_edge_copy", "cg_solver_edge_update"):
            getattr(c, f).argtypes = [vp]
            getattr(c, f).restype = None
        c.cg_solver_edge_plane.argtypes = [vp, i32]
        c.cg_solver_edge_plane.restype = C.c_void_p
        c.cg_solver_rccl_ranks.argtypes = [vp]
        c.cg_solver_rccl_ranks.restype = i32
        c.cg_noalloc_prepare.argtypes = []
        c.cg_noalloc_prepare.restype = i32
        c.cg_inner_product_noalloc.argtypes = [fp, fp, C.c_ulonglong, fp, vp]
        c.cg_inner_product_noalloc.restype = i32

    # ---- the reference's entry points (main.jl:16-58) --------------------------------
    def conjugate_gradient_gpu(self, b, x, L, d):
        return self.c.conjugate_gradient_gpu(_ptr(b), _ptr(x), L, d)

    def inner_product_gpu(self, v, w, N):
        return self.c.inner_product_gpu(_ptr(v), _ptr(w), N)

    def neighbour_index_gpu(self, ind, direction, amount, L, d, N, index_mode):
        return self.c.neighbour_index_gpu(ind, direction, amount, L, d, N, index_mode)

    def strong_scaling(self, nblocks, threads_per_block, N, L, d):
        return self.c.strong_scaling(nblocks, threads_per_block, N, L, d)

    def time_cpu_laplace(self, N, L, d):
        return self.c.time_cpu_laplace(N, L, d)

    def add_jl(self, A, B, blocks, threads):
        self.c.add_jl(_ptr(A), _ptr(B), blocks, threads)

    def laplace_gpu_jl(self, ddf, u, dx, d, L, N, index_mode, blocks, threads):
        self.c.laplace_gpu_jl(_ptr(ddf), _ptr(u), dx, d, L, N, index_mode, blocks, threads)

    # ---- additions -------------------------------------------------------------------
    def release_cache(self):
        """Free the solver conjugate_gradient_gpu keeps between calls."""
        self.c.cg_release_cache()

    def inner_product_noalloc(self, v, w, n, result_dev, stream=None):
        """Device pointers only; (float) v.w lands in result_dev[0], asynchronously on `stream`.
        Returns the entry point's status (0 = success)."""
        return self.c.cg_inner_product_noalloc(_ptr(v), _ptr(w), n, _ptr(result_dev), _ptr(stream))

    def strong_scaling_events(self, nblocks, threads_per_block, N, L, d, reps=5):
        return self.c.cg_strong_scaling_events(nblocks, threads_per_block, N, L, d, reps)

    def laplace_time_events(self, L, d, index_mode=0, reps=5):
        return self.c.cg_laplace_time_events(L, d, index_mode, reps)

    def conjugate_gradient_gpu_ex(self, b, x, L, d, index_mode=0, maxit=200, rel=1e-6, lagged=1,
                                  verbose=0, hist_len=0):
        """Returns (residue, iterations, history)."""
        it = C.c_int(0)
        hist = (C.c_float * max(hist_len, 1))()
        res = self.c.conjugate_gradient_gpu_ex(_ptr(b), _ptr(x), L, d, index_mode, maxit, rel, lagged,
                                               verbose, C.byref(it), C.cast(hist, C.c_void_p) if hist_len else None,
                                               hist_len)
        return res, it.value, list(hist)[:min(hist_len, it.value)]

    def preconditioned_cg_gpu(self, b, x, L, d, verbose=0):
        """b, x: addresses of N doubles (host or device). Returns (outer_iterations, [inner steps])."""
        inner = (C.c_int * 1002)()
        outer = self.c.preconditioned_cg_gpu(_ptr(b), _ptr(x), L, d, C.cast(inner, C.c_void_p), 1002, verbose)
        # one preconditioner call before the loop and one per outer iteration that did not stop on the
        # tolerance (conjugate-gradient_cpu.cu:348,374-376); the library ends the list with -1 because a
        # legitimate count can be 0
        counts = list(inner)
        return outer, counts[:counts.index(-1)] if -1 in counts else counts

    def eigen_extremes_gpu(self, L, d, max_steps=1000, tol=1e-7, seed=1, want_max=True, want_min=True):
        """Returns dict(lambda_max, steps_max, lambda_min, steps_min, cg_iterations) of the L^d operator."""
        lmax, lmin = C.c_double(0), C.c_double(0)
        smax, smin, cgit = C.c_int(0), C.c_int(0), C.c_int(0)
        self.c.cg_eigen_extremes_gpu(L, d, max_steps, tol, seed, C.byref(lmax) if want_max else None, C.byref(smax),
                                     C.byref(lmin) if want_min else None, C.byref(smin), C.byref(cgit))
        return dict(lambda_max=lmax.value, steps_max=smax.value, lambda_min=lmin.value, steps_min=smin.value,
                    cg_iterations=cgit.value)

    def two_grid_gpu(self, f, u, L, d, nu1=2, nu2=2, omega=0.0, max_cycles=50, rel=1e-6, coarse_rel=1e-4):
        """f, u: device addresses of N floats.  Returns (cycles, [residual norms], coarse CG iterations)."""
        hist = (C.c_float * (max_cycles + 1))()
        inner = C.c_int(0)
        cycles = self.c.cg_two_grid_gpu(_ptr(f), _ptr(u), L, d, nu1, nu2, omega, max_cycles, rel, coarse_rel,
                                        C.cast(hist, C.c_void_p), max_cycles + 1, C.byref(inner))
        return cycles, list(hist)[:cycles + 1], inner.value

    def multigrid_gpu(self, f, u, L, d, levels=0, nu1=2, nu2=2, omega=0.0, max_cycles=50, rel=1e-6, coarse_rel=1e-4):
        """V-cycles.  Returns (cycles, [residual norms], levels in use, CG iterations on the coarsest grid)."""
        hist = (C.c_float * (max_cycles + 1))()
        used, inner = C.c_int(0), C.c_int(0)
        cycles = self.c.cg_multigrid_gpu(_ptr(f), _ptr(u), L, d, levels, nu1, nu2, omega, max_cycles, rel, coarse_rel,
                                         C.cast(hist, C.c_void_p), max_cycles + 1, C.byref(used), C.byref(inner))
        return cycles, list(hist)[:cycles + 1], used.value, inner.value

    def two_grid_pcg_gpu(self, f, u, L, d, nu=2, omega=0.0, max_iters=50, rel=1e-6, coarse_rel=1e-4):
        """CG preconditioned by one two-grid cycle.  Returns (iterations, [residual norms], coarse CG iterations)."""
        hist = (C.c_float * (max_iters + 1))()
        inner = C.c_int(0)
        its = self.c.cg_two_grid_pcg_gpu(_ptr(f), _ptr(u), L, d, nu, omega, max_iters, rel, coarse_rel,
                                         C.cast(hist, C.c_void_p), max_iters + 1, C.byref(inner))
        return its, list(hist)[:its + 1], inner.value

    # ---- boxes of the direct xGMI path (all raise on a HIP error) ----------------------
    def links_alloc(self, nbytes):
        box = C.c_void_p()
        e = self.c.cg_links_alloc(nbytes, C.byref(box))
        if e or not box.value:
            raise RuntimeError(f"cg_links_alloc({nbytes}) failed with hipError {e}")
        return box.value

    def links_peer_ok(self, peer_device):
        return bool(self.c.cg_links_peer_ok(int(peer_device)))

    def links_free(self, box):
        self.c.cg_links_free(_ptr(box))

    def links_export(self, box):
        buf = C.create_string_buffer(64)
        e = self.c.cg_links_export(_ptr(box), buf)
        if e:
            raise RuntimeError(f"cg_links_export failed with hipError {e}")
        return buf.raw

    def links_open(self, handle):
        box = C.c_void_p()
        e = self.c.cg_links_open(C.create_string_buffer(bytes(handle), 64), C.byref(box))
        if e or not box.value:
            raise RuntimeError(f"cg_links_open failed with hipError {e}")
        return box.value

    def links_poke(self, box, slot, value):
        e = self.c.cg_links_poke(_ptr(box), slot, value)
        if e:
            raise RuntimeError(f"cg_links_poke failed with hipError {e}")

    def links_peek(self, box, slot):
        v = C.c_ulonglong(0)
        e = self.c.cg_links_peek(_ptr(box), slot, C.byref(v))
        if e:
            raise RuntimeError(f"cg_links_peek failed with hipError {e}")
        return v.value

    def links_close(self, box):
        self.c.cg_links_close(_ptr(box))

    # ---- RCCL communicator of the in-library multi-GPU path -------------------------------
    def rccl_unique_id(self):
        buf = C.create_string_buffer(128)
        e = self.c.cg_rccl_unique_id(buf)
        if e:
            raise RuntimeError(f"cg_rccl_unique_id failed with ncclResult {e}")
        return buf.raw

    def rccl_comm_create(self, unique_id, rank, world):
        comm = C.c_void_p()
        e = self.c.cg_rccl_comm_create(C.create_string_buffer(bytes(unique_id), 128), rank, world, C.byref(comm))
        if e or not comm.value:
            raise RuntimeError(f"cg_rccl_comm_create failed with ncclResult {e}")
        return comm.value

    def rccl_comm_destroy(self, comm):
        self.c.cg_rccl_comm_destroy(_ptr(comm))

    def make_solver(self, extents, index_mode=0, has_lo=0, has_hi=0, stream=None, r_buf=None, p0_buf=None,
                    p1_buf=None, engine=0):
        return Solver(self, extents, index_mode, has_lo, has_hi, stream, r_buf, p0_buf, p1_buf, engine)

    def desc(self, extents, index_mode=0, has_lo=0, has_hi=0, stream=None, r_buf=None, p0_buf=None, p1_buf=None,
             engine=0):
        d = SolverDesc()
        d.d = len(extents)
        for k, e in enumerate(extents):
            d.extents[k] = int(e)
        d.index_mode, d.has_lo, d.has_hi = index_mode, int(has_lo), int(has_hi)
        d.stream = int(stream) if stream else None
        d.r_buf = int(r_buf) if r_buf else None
        d.p0_buf = int(p0_buf) if p0_buf else None
        d.p1_buf = int(p1_buf) if p1_buf else None
        d.engine = engine
        return d


class Solver:
    """The fused two-kernel CG iteration on device-resident vectors (cg_solver_* entry points)."""

    def __init__(self, lib, extents, index_mode, has_lo, has_hi, stream, r_buf, p0_buf, p1_buf, engine):
        self.lib = lib
        self.extents = [int(e) for e in extents]
        self._desc = lib.desc(extents, index_mode, has_lo, has_hi, stream, r_buf, p0_buf, p1_buf, engine)
        self.h = lib.c.cg_solver_create(C.byref(self._desc))
        if not self.h:
            raise RuntimeError("cg_solver_create failed")
        self.N = 1
        for e in self.extents:
            self.N *= e

    def close(self):
        if self.h:
            self.lib.c.cg_solver_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def engine(self):
        return self.lib.c.cg_solver_engine(self.h)

    @property
    def kernel_name(self):
        return self.lib.c.cg_solver_kernel_name(self.h).decode()

    @property
    def blocks(self):
        return self.lib.c.cg_solver_blocks(self.h)

    def blocks_of(self, which):
        """Partial sums kernel A (which = 0) / kernel B (1) writes per launch."""
        return self.lib.c.cg_solver_blocks_of(self.h, which)

    @property
    def paired_x(self):
        """True if cg_solver_iterate updates x every second iteration (two updates at once, same bits)."""
        return bool(self.lib.c.cg_solver_paired_x(self.h))

    def begin(self, b, x, rel=1e-6, lagged=1):
        self.lib.c.cg_solver_begin(self.h, _ptr(b), _ptr(x), rel, lagged)

    def iterate(self, n, use_graph=True):
        self.lib.c.cg_solver_iterate(self.h, n, int(use_graph))

    def sync(self):
        self.lib.c.cg_solver_sync(self.h)

    def finish(self, hist_len=0):
        res = C.c_float(0)
        hist = (C.c_float * max(hist_len, 1))()
        n = self.lib.c.cg_solver_finish(self.h, C.byref(res), C.cast(hist, C.c_void_p) if hist_len else None, hist_len)
        return n, res.value, list(hist)[:min(hist_len, n)]

    def time_kernels(self, n):
        a, b = C.c_float(0), C.c_float(0)
        self.lib.c.cg_solver_time_kernels(self.h, n, C.byref(a), C.byref(b))
        return a.value, b.value

    # step interface (slab driver)
    def begin_local(self, b, x):
        self.lib.c.cg_solver_begin_local(self.h, _ptr(b), _ptr(x))

    def reduce(self, which, out):
        self.lib.c.cg_solver_reduce(self.h, which, _ptr(out))

    def bind_sums(self, pAp, rr):
        self.lib.c.cg_solver_bind_sums(self.h, _ptr(pAp), _ptr(rr))

    def bind_partials(self, pAp_parts, rr_parts, n=None):
        """n: number of addends every rank reduces (the largest `blocks` over the ranks)."""
        if n is None:
            self.lib.c.cg_solver_bind_partials(self.h, _ptr(pAp_parts), _ptr(rr_parts))
        else:
            self.lib.c.cg_solver_bind_partials_n(self.h, _ptr(pAp_parts), _ptr(rr_parts), n)

    def bind_rccl(self, comm, rank, world, peers=None):
        """comm: an ncclComm_t address (Library.rccl_comm_create); None unbinds.  peers: (lo_rank, hi_rank) of
        the slabs below / above, -1 for none; default rank - 1 / rank + 1.  Collective."""
        if peers is None:
            e = self.lib.c.cg_solver_bind_rccl(self.h, _ptr(comm), rank, world)
        else:
            e = self.lib.c.cg_solver_bind_rccl_peers(self.h, _ptr(comm), rank, world, int(peers[0]), int(peers[1]))
        if e:
            raise RuntimeError(f"cg_solver_bind_rccl failed with ncclResult {e}")

    # r's boundary planes ahead of kernel B (step interface)
    def edge_copy(self):
        self.lib.c.cg_solver_edge_copy(self.h)

    def edge_update(self):
        self.lib.c.cg_solver_edge_update(self.h)

    def edge_plane(self, side):
        return self.lib.c.cg_solver_edge_plane(self.h, side)

    def rccl_ranks(self):
        """Ranks of the bound RCCL communicator as RCCL itself counts them (0: none bound)."""
        return self.lib.c.cg_solver_rccl_ranks(self.h)

    def init_state(self, rel, lagged, maxit):
        self.lib.c.cg_solver_init_state(self.h, rel, lagged, maxit)

    def add_iterations(self, n):
        self.lib.c.cg_solver_add_iterations(self.h, n)

    def kernel_A(self):
        self.lib.c.cg_solver_kernel_A(self.h)

    def kernel_B(self):
        self.lib.c.cg_solver_kernel_B(self.h)

    # direct xGMI path
    def box_bytes(self):
        return self.lib.c.cg_links_box_bytes(C.byref(self._desc))

    def bind_links(self, rank, boxes):
        """boxes: every rank's box as mapped in this process; an empty list unbinds."""
        arr = (C.c_void_p * max(1, len(boxes)))(*[int(b) for b in boxes])
        self.lib.c.cg_solver_bind_links(self.h, rank, len(boxes), arr)

    def exchange(self, which, force=False):
        self.lib.c.cg_solver_exchange(self.h, which, int(force))

    def links_error(self):
        return self.lib.c.cg_solver_links_error(self.h)

    def links_times(self):
        a, b = C.c_float(0), C.c_float(0)
        self.lib.c.cg_solver_links_times(self.h, C.byref(a), C.byref(b))
        return a.value, b.value


_LIB = None


def load_library():
    """The process-wide Library; raises if the shared object has not been built."""
    global _LIB
    if _LIB is None:
        _LIB = Library()
    return _LIB
