"""CPU stand-in for the per-slab compute steps -- TEST INFRASTRUCTURE.

Implements the interface of conjugate_gradient_amd.slab.HipSlabOps on CPU torch tensors so
that the slab driver's exchange logic (ghost-plane views, send/recv pairing, placement of
the two all-reduces) can run under the gloo backend without a GPU.  The arithmetic is fp32
in the reference's operation order (conjugate-gradient_gpu.cu:173,298,307); the iteration
control mirrors the device-side state machine of the HIP kernels.
"""
import numpy as np
import torch


class FakeSlabOps:
    def __init__(self, extents, has_lo, has_hi):
        self.ext = [int(e) for e in extents]
        self.d = len(self.ext)
        self.N = int(np.prod(self.ext))
        self.plane = self.N // self.ext[-1]
        self.ghost_offset = (self.plane + 63) // 64 * 64
        self.buffer_len = self.ghost_offset + self.N + self.plane + 64
        self.has_lo, self.has_hi = bool(has_lo), bool(has_hi)
        self.r_buf = torch.zeros(self.buffer_len, dtype=torch.float32)
        self.p_buf = [torch.zeros(self.buffer_len, dtype=torch.float32) for _ in range(2)]
        # like the HIP kernels, a slab hands over an array of partial sums whose length depends on its
        # shape; the driver must make all ranks reduce arrays of one length
        self.nparts_own = 1 + self.ext[-1] % 3
        self.nparts = self.nparts_own
        self.dots = torch.zeros(2, 8, dtype=torch.float64)
        self.part = [0.0, 0.0]
        self.x = None
        self.it = 0
        self.done = False
        self.active = False
        self.maxit = 0
        self.hist = []

    # -- helpers ---------------------------------------------------------------------
    def _ghosted(self, buf):
        """numpy view [lo ghost | owned | hi ghost] shaped (n_last + 2, ..., n0)."""
        g, n, pl = self.ghost_offset, self.N, self.plane
        a = buf.numpy()[g - pl:g + n + pl]
        return a.reshape([self.ext[-1] + 2] + self.ext[-2::-1])

    def _stencil(self, pg):
        """A p on the owned points from the ghosted array pg (axis order: last ... 0)."""
        c = pg[1:-1]
        lv = np.zeros_like(c)
        nd = c.ndim
        for k in range(self.d):          # reference order: axis 0 first
            ax = nd - 1 - k
            if k == self.d - 1:          # slab axis: neighbours come from the ghost planes
                up, um = pg[2:], pg[:-2]
            else:
                up = np.zeros_like(c)
                um = np.zeros_like(c)
                sl_hi = [slice(None)] * nd
                sl_lo = [slice(None)] * nd
                sl_hi[ax] = slice(1, None)
                sl_lo[ax] = slice(None, -1)
                up[tuple(sl_lo)] = c[tuple(sl_hi)]
                um[tuple(sl_hi)] = c[tuple(sl_lo)]
            lv = (lv + ((-up + np.float32(2) * c) - um)).astype(np.float32)
        return lv

    @staticmethod
    def _fma(a, b, c):
        return (np.float64(a) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)

    # -- interface used by SlabCG ------------------------------------------------------
    def begin_local(self, b, x):
        self.x = x
        g, n = self.ghost_offset, self.N
        self.r_buf.zero_()
        self.r_buf[g:g + n] = b
        self.p_buf[0].zero_()
        rr = self.r_buf[g:g + n].numpy().astype(np.float64)
        self.part[1] = float(np.dot(rr, rr))

    def set_nparts(self, n):
        assert self.nparts_own <= n <= self.dots.shape[1]
        self.nparts = int(n)

    def sum_tensor(self, which):
        self.dots[which].zero_()
        self.dots[which, self.nparts_own - 1] = self.part[which]   # the last slot this slab owns
        return self.dots[which, :self.nparts]

    def _dot(self, which):
        return self.dots[which, :self.nparts].sum().item()

    def init_state(self, rel, lagged, maxit):
        rr = np.float32(self._dot(1))
        self.it, self.done, self.active, self.maxit, self.lagged = 0, False, False, maxit, bool(lagged)
        self.reltol = np.float32(rel * float(np.float32(np.sqrt(np.float64(rr)))))
        self.hist = [rr]

    def add_iterations(self, n):
        self.maxit = self.it + n

    def kernel_A(self):
        k = self.it
        rr = np.float32(self._dot(1))
        rr_prev = self.hist[k - 1] if k > 0 else None
        test = rr_prev if (self.lagged and k > 0) else rr
        residue = np.float32(np.sqrt(np.float64(test)))
        conv = self.done or not (residue > self.reltol)
        self.done = conv
        self.active = (not conv) and k < self.maxit
        if len(self.hist) <= k:
            self.hist.append(rr)
        else:
            self.hist[k] = rr
        if not self.active:
            return
        beta = np.float32(rr / rr_prev) if k > 0 else np.float32(0)
        po, pn = self.p_buf[k & 1], self.p_buf[(k & 1) ^ 1]
        pog, png, rg = self._ghosted(po), self._ghosted(pn), self._ghosted(self.r_buf)
        new = self._fma(beta, pog, rg)
        lo = 0 if self.has_lo else 1
        hi = new.shape[0] if self.has_hi else new.shape[0] - 1
        png[lo:hi] = new[lo:hi]           # ghost planes of p are rebuilt locally, faces stay zero
        ap = self._stencil(png)
        self.part[0] = float(np.sum(png[1:-1].astype(np.float64) * ap.astype(np.float64)))

    # r's boundary planes ahead of kernel B (cg_solver_edge_copy / cg_solver_edge_update)
    def edge_copy(self):
        g, n, pl = self.ghost_offset, self.N, self.plane
        self.edge_lo = self.r_buf[g:g + pl].clone()
        self.edge_hi = self.r_buf[g + n - pl:g + n].clone()

    def edge_update(self):
        if not self.active:
            return
        k = self.it
        alpha = np.float32(self.hist[k] / np.float32(self._dot(0)))
        ap = self._stencil(self._ghosted(self.p_buf[(k & 1) ^ 1]))
        self.edge_lo = torch.from_numpy(self._fma(-alpha, ap[0].reshape(-1), self.edge_lo.numpy()))
        self.edge_hi = torch.from_numpy(self._fma(-alpha, ap[-1].reshape(-1), self.edge_hi.numpy()))

    def kernel_B(self):
        if not self.active:
            return
        k = self.it
        pAp = np.float32(self._dot(0))
        alpha = np.float32(self.hist[k] / pAp)
        pg = self._ghosted(self.p_buf[(k & 1) ^ 1])
        ap = self._stencil(pg)
        rg = self._ghosted(self.r_buf)
        xs = self.x.numpy().reshape(pg[1:-1].shape)
        xs[...] = self._fma(alpha, pg[1:-1], xs)
        rg[1:-1] = self._fma(-alpha, ap, rg[1:-1])
        self.part[1] = float(np.sum(rg[1:-1].astype(np.float64) ** 2))
        self.it = k + 1
        if hasattr(self, "edge_lo"):   # the planes computed ahead are what this kernel leaves in r
            g, n, pl = self.ghost_offset, self.N, self.plane
            assert torch.equal(self.edge_lo, self.r_buf[g:g + pl]) and torch.equal(self.edge_hi, self.r_buf[g + n - pl:g + n])

    def finish(self, hist_len=0):
        n = self.it
        last = n - 1 if n > 0 else 0
        res = float(np.sqrt(np.float64(self.hist[last])))
        h = [float(np.sqrt(np.float64(v))) for v in self.hist[:min(n, hist_len)]]
        return n, res, h

    def sync(self):
        pass
