"""ctypes doors into the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  It loads oracle/libcg_oracle.so (the plain-C restatement, cg_oracle.c) and,
when present, oracle/_ref/*.so (the reference's own CPU sources compiled unchanged
by oracle/Makefile).  Nothing here touches a GPU.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_fp = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(ref=False):
    """Compile the oracle (and the reference build when its sources exist)."""
    targets = ["all"]
    if ref and os.path.isdir("/root/reference/src"):
        targets.append("ref")
    subprocess.run(["make", "-s", "-C", HERE] + targets, check=True)


def _load(name):
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        return None
    return C.CDLL(path)


class Oracle:
    """The restatement.  native=True loads the -O3 -march=x86-64-v3 timing build."""

    def __init__(self, native=False):
        name = "libcg_oracle_opt.so" if native else "libcg_oracle.so"
        lib = _load(name)
        if lib is None:
            build()
            lib = _load(name)
        self.lib = lib
        L = lib
        L.orc_get_index.argtypes = [_ip, C.c_int, C.c_int, C.c_int]
        L.orc_get_index.restype = C.c_int
        L.orc_neighbour_index.argtypes = [_ip, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_neighbour_index.restype = C.c_int
        L.orc_index_to_cords.argtypes = [_ip, C.c_int, C.c_int, C.c_int]
        L.orc_index_to_cords_gpu.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_index_to_cords_gpu.restype = C.c_int
        L.orc_neighbour_index_gpu.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_uint, C.c_int]
        L.orc_neighbour_index_gpu.restype = C.c_uint
        L.orc_minus_laplace.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_int]
        L.orc_norm.argtypes = [_dp, C.c_int]
        L.orc_norm.restype = C.c_double
        L.orc_inner_product.argtypes = [_dp, _dp, C.c_int]
        L.orc_inner_product.restype = C.c_double
        L.orc_conjugate_gradient_ex.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_int,
                                                C.POINTER(C.c_int), C.c_void_p, C.c_int]
        L.orc_conjugate_gradient_ex.restype = C.c_double
        L.orc_preconditioner_ex.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int), C.c_int]
        L.orc_preconditioner_ex.restype = C.c_double
        L.orc_preconditioned_cg_ex.argtypes = [_dp, _dp, C.c_int, C.c_int, C.POINTER(C.c_int),
                                               C.c_void_p, C.c_int, C.c_int]
        L.orc_preconditioned_cg_ex.restype = C.c_void_p
        L.orc_laplace_f32.argtypes = [_fp, _fp, C.c_uint, C.c_uint, C.c_uint]
        L.orc_inner_product_f32.argtypes = [_fp, _fp, C.c_uint]
        L.orc_inner_product_f32.restype = C.c_float
        L.orc_cg_f32.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                 C.POINTER(C.c_int), C.c_void_p, C.c_int]
        L.orc_cg_f32.restype = C.c_float
        for f in ("orc_pdep", "orc_pext"):
            getattr(L, f).argtypes = [C.c_uint, C.c_uint]
            getattr(L, f).restype = C.c_uint
        L.orc_morton_base_mask.argtypes = [C.c_int]
        L.orc_morton_base_mask.restype = C.c_uint
        L.orc_morton_mask.argtypes = [C.c_int, C.c_int]
        L.orc_morton_mask.restype = C.c_uint
        for f in ("orc_morton_dec", "orc_morton_inc"):
            getattr(L, f).argtypes = [C.c_uint, C.c_int, C.c_int]
            getattr(L, f).restype = C.c_uint
        L.orc_morton_from_coords.argtypes = [_ip, C.c_int]
        L.orc_morton_from_coords.restype = C.c_uint
        L.orc_morton_to_coords.argtypes = [C.c_uint, C.c_int, _ip]
        L.orc_laplace_morton_f32.argtypes = [_fp, _fp, C.c_uint, C.c_uint]
        L.orc_lex_to_morton.argtypes = [C.c_uint, C.c_uint, C.c_int]
        L.orc_lex_to_morton.restype = C.c_uint

    # -- index mapping -------------------------------------------------------
    def get_index(self, cords, L, d, N):
        return self.lib.orc_get_index(np.ascontiguousarray(cords, np.int32), L, d, N)

    def neighbour_index(self, cords, direction, amount, L, d, N):
        return self.lib.orc_neighbour_index(np.ascontiguousarray(cords, np.int32), direction, amount, L, d, N)

    def index_to_cords(self, index, L, d):
        out = np.zeros(d, np.int32)
        self.lib.orc_index_to_cords(out, index, L, d)
        return out

    def index_to_cords_gpu(self, index, L, axis):
        return self.lib.orc_index_to_cords_gpu(index, L, axis)

    def neighbour_index_gpu(self, ind, direction, amount, L, d, N, index_mode=0):
        return self.lib.orc_neighbour_index_gpu(ind, direction, amount, L, d, N, index_mode)

    # -- fp64 ----------------------------------------------------------------
    def minus_laplace(self, u, d, L):
        """u: N values (boundary slot appended here). Returns A u (N values)."""
        N = L ** d
        uu = np.zeros(N + 1, np.float64)
        uu[:N] = u
        out = np.zeros(N + 1, np.float64)
        self.lib.orc_minus_laplace(out, uu, d, L, N)
        return out[:N].copy()

    def inner_product(self, x, y):
        x = np.ascontiguousarray(x, np.float64)
        y = np.ascontiguousarray(y, np.float64)
        return self.lib.orc_inner_product(x, y, x.size)

    def norm(self, x):
        x = np.ascontiguousarray(x, np.float64)
        return self.lib.orc_norm(x, x.size)

    def conjugate_gradient(self, b, L, d, x0=None, max_iters=-1, hist_len=0):
        """Returns (x, returned_residue, iterations, history)."""
        N = L ** d
        bb = np.zeros(N + 1, np.float64)
        bb[:N] = b
        x = np.zeros(N + 1, np.float64)
        if x0 is not None:
            x[:N] = x0
        it = C.c_int(0)
        hist = np.zeros(max(hist_len, 1), np.float64)
        res = self.lib.orc_conjugate_gradient_ex(bb, x, L, d, max_iters, C.byref(it),
                                                 hist.ctypes.data if hist_len else None, hist_len)
        return x[:N].copy(), res, it.value, hist[:min(hist_len, it.value)]

    def preconditioner(self, b, L, d, errtol, verbose=False):
        N = L ** d
        bb = np.zeros(N + 1, np.float64)
        bb[:N] = b
        x = np.zeros(N + 1, np.float64)
        it = C.c_int(0)
        res = self.lib.orc_preconditioner_ex(bb, x, L, d, errtol, C.byref(it), int(verbose))
        return x[:N].copy(), res, it.value

    def preconditioned_cg(self, b, L, d, x0=None, verbose=False):
        """Returns (x, outer_iterations, [inner step counts])."""
        N = L ** d
        bb = np.zeros(N + 1, np.float64)
        bb[:N] = b
        x = np.zeros(N + 1, np.float64)
        if x0 is not None:
            x[:N] = x0
        outer = C.c_int(0)
        inner = np.zeros(1002, np.int32)
        self.lib.orc_preconditioned_cg_ex(bb, x, L, d, C.byref(outer), inner.ctypes.data, inner.size, int(verbose))
        n_calls = int(np.count_nonzero(inner)) if outer.value else 0
        return x[:N].copy(), outer.value, inner[:max(n_calls, 1)].tolist()

    # -- fp32 ----------------------------------------------------------------
    def laplace_f32(self, u, d, L):
        u = np.ascontiguousarray(u, np.float32)
        out = np.zeros(u.size, np.float32)
        self.lib.orc_laplace_f32(out, u, d, L, u.size)
        return out

    def inner_product_f32(self, v, w):
        v = np.ascontiguousarray(v, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        return self.lib.orc_inner_product_f32(v, w, v.size)

    def cg_f32(self, b, L, d, x0=None, maxit=200, rel=1e-6, lagged=True, hist_len=0):
        """fp32 emulation of the reference GPU loop. Returns (x, residue, iterations, history)."""
        N = L ** d
        b = np.ascontiguousarray(b, np.float32)
        x = np.zeros(N, np.float32) if x0 is None else np.array(x0, np.float32)
        it = C.c_int(0)
        hist = np.zeros(max(hist_len, 1), np.float32)
        res = self.lib.orc_cg_f32(b, x, L, d, maxit, rel, int(lagged), C.byref(it),
                                  hist.ctypes.data if hist_len else None, hist_len)
        return x, res, it.value, hist[:min(hist_len, it.value)]

    # -- Morton --------------------------------------------------------------
    def pdep(self, x, m):
        return self.lib.orc_pdep(x, m)

    def pext(self, x, m):
        return self.lib.orc_pext(x, m)

    def morton_mask(self, d, dim):
        return self.lib.orc_morton_mask(d, dim)

    def morton_neighbour(self, idx, d, dim, direction):
        f = self.lib.orc_morton_inc if direction > 0 else self.lib.orc_morton_dec
        return f(idx, d, dim)

    def morton_from_coords(self, coords):
        c = np.ascontiguousarray(coords, np.int32)
        return self.lib.orc_morton_from_coords(c, c.size)

    def morton_to_coords(self, idx, d):
        out = np.zeros(d, np.int32)
        self.lib.orc_morton_to_coords(idx, d, out)
        return out

    def laplace_morton_f32(self, u, d):
        u = np.ascontiguousarray(u, np.float32)
        out = np.zeros(u.size, np.float32)
        self.lib.orc_laplace_morton_f32(out, u, d, u.size)
        return out

    def lex_to_morton_perm(self, L, d):
        """perm[lex] = morton index, for L = 2^m."""
        N = L ** d
        return np.array([self.lib.orc_lex_to_morton(i, L, d) for i in range(N)], dtype=np.int64)


class Reference:
    """The reference's own CPU code (oracle/_ref).  None-like when not built."""

    def __init__(self, native=False):
        """native=True loads the -O3 -march=x86-64-v3 timing build of the same sources."""
        self.lib = _load("_ref/libcg_ref_opt.so" if native else "_ref/libcg_ref.so")
        self.mlib = _load("_ref/libcg_ref_morton.so")
        if self.lib is not None:
            L = self.lib
            L.ref_minus_laplace.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_int]
            L.ref_conjugate_gradient.argtypes = [_dp, _dp, C.c_int, C.c_int]
            L.ref_conjugate_gradient.restype = C.c_double
            L.ref_preconditioner.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_double]
            L.ref_preconditioner.restype = C.c_double
            L.ref_inner_product.argtypes = [_dp, _dp, C.c_int]
            L.ref_inner_product.restype = C.c_double
            L.ref_preconditioned_cg.argtypes = [_dp, _dp, C.c_int, C.c_int]
            L.ref_preconditioned_cg.restype = C.c_void_p
            L.ref_index_to_cords.argtypes = [_ip, C.c_int, C.c_int, C.c_int]
            L.ref_norm.argtypes = [_dp, C.c_int]
            L.ref_norm.restype = C.c_double
            L.ref_get_index.argtypes = [_ip, C.c_int, C.c_int, C.c_int]
            L.ref_neighbour_index.argtypes = [_ip, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
            L.ref_random_array.argtypes = [_dp, C.c_int, C.c_int, C.c_int]
            L.ref_srand.argtypes = [C.c_uint]
        if self.mlib is not None:
            M = self.mlib
            M.ref_morton_from_coords.argtypes = [_ip, C.c_int]
            M.ref_morton_from_coords.restype = C.c_uint
            M.ref_morton_to_coords.argtypes = [C.c_uint, C.c_int, _ip]
            M.ref_morton_neighbour.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int]
            M.ref_morton_neighbour.restype = C.c_uint
            M.ref_morton_mask.argtypes = [C.c_int, C.c_int]
            M.ref_morton_mask.restype = C.c_uint
            for f in ("ref_pdep", "ref_pext"):
                getattr(M, f).argtypes = [C.c_uint, C.c_uint]
                getattr(M, f).restype = C.c_uint

    @property
    def available(self):
        return self.lib is not None

    @property
    def morton_available(self):
        return self.mlib is not None

    def random_array(self, N, seed=1):
        self.lib.ref_srand(seed)
        r = np.zeros(N + 1, np.float64)
        self.lib.ref_random_array(r, 0, 0, N)
        return r[:N].copy()

    def minus_laplace(self, u, d, L):
        N = L ** d
        uu = np.zeros(N + 1, np.float64)
        uu[:N] = u
        out = np.zeros(N + 1, np.float64)
        self.lib.ref_minus_laplace(out, uu, d, L, N)
        return out[:N].copy()

    def conjugate_gradient(self, b, L, d):
        N = L ** d
        bb = np.zeros(N + 1, np.float64)
        bb[:N] = b
        x = np.zeros(N + 1, np.float64)
        res = self.lib.ref_conjugate_gradient(bb, x, L, d)
        return x[:N].copy(), res

    def preconditioner(self, b, L, d, errtol):
        N = L ** d
        bb = np.zeros(N + 1, np.float64)
        bb[:N] = b
        x = np.zeros(N + 1, np.float64)
        res = self.lib.ref_preconditioner(bb, x, L, d, errtol)
        return x[:N].copy(), res

    def preconditioned_cg(self, b, L, d):
        N = L ** d
        bb = np.zeros(N + 1, np.float64)
        bb[:N] = b
        x = np.zeros(N + 1, np.float64)
        self.lib.ref_preconditioned_cg(bb, x, L, d)
        return x[:N].copy()
