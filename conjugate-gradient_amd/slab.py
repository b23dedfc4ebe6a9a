"""Slab-decomposed CG across ranks: one process per GPU, torch.distributed (RCCL over xGMI).

The grid is cut along its slowest axis into contiguous slabs, one per rank.  Per CG
iteration the ranks exchange

  * one hyperplane of r with each neighbour (the only vector whose neighbours' values are
    needed: kernel A rebuilds p = r + beta*p on its ghost planes locally), and
  * the two dot products p.Ap and r.r, by all-reduce of the kernels' per-workgroup partial sums
    (a few KiB of device doubles, no separate reduction launch); alpha and beta are then derived
    on every rank's device from identical bits, so all ranks take the same path.

Three transports carry that traffic, the kernels are the same:

  * `enable_rccl()` -- the library issues the RCCL calls itself (csrc/cg_rccl.hip: ncclAllGather of one
    double per rank for each dot product; the planes, computed ahead of kernel B, travel by grouped
    ncclSend / ncclRecv on a second stream while kernel B runs), captured in the same hipGraph as the
    kernels: no host work inside the loop.  Never run on more than one GPU so far: callers enable it only
    after a self-check against the collectives below (bench.py does).
  * `enable_links()` -- the ranks of one node write the same data straight into each other's HBM over
    xGMI from a small kernel after A and after B (csrc/cg_links.hpp); four launches per iteration,
    replayed from a hipGraph.
  * neither -- this module drives the steps and calls torch.distributed between them (one
    batch_isend_irecv and two all_reduce per iteration from Python): the portable fallback, and what
    the gloo rehearsals on CPU tensors exercise.

Everything is enqueued asynchronously on the current stream; the host never waits inside
the loop.  The compute steps are the HIP library's kernels (binding.Solver); a different
`make_ops` can be injected so that the exchange logic can be exercised on CPU tensors with
the gloo backend.
"""
import contextlib
import os
import socket

import torch
import torch.distributed as dist


class HipSlabOps:
    """The compute steps of one slab on the GPU: thin calls into libconjugate_gradient_gpu.so."""

    def __init__(self, lib, extents, has_lo, has_hi, device):
        self.lib = lib
        self.device = device
        desc = lib.desc(extents, 0, has_lo, has_hi)
        self.buffer_len = lib.c.cg_solver_buffer_len(desc)
        self.ghost_offset = lib.c.cg_solver_ghost_offset(desc)
        self.plane = lib.c.cg_solver_plane(desc)
        self.r_buf = torch.zeros(self.buffer_len, dtype=torch.float32, device=device)
        self.p0_buf = torch.zeros(self.buffer_len, dtype=torch.float32, device=device)
        self.p1_buf = torch.zeros(self.buffer_len, dtype=torch.float32, device=device)
        # per-workgroup partials of [p.Ap, r.r]: all-reduced element-wise across ranks, then summed
        # in a fixed order by every workgroup of the consuming kernel
        self.parts = torch.zeros(2, 4096, dtype=torch.float64, device=device)
        # kernels and collectives share one non-default stream: torch orders its RCCL work
        # against the stream that is current when a collective is issued
        self.stream = torch.cuda.Stream(device)
        torch.cuda.synchronize(device)
        self.solver = lib.make_solver(extents, 0, has_lo, has_hi, stream=self.stream.cuda_stream,
                                      r_buf=self.r_buf.data_ptr(), p0_buf=self.p0_buf.data_ptr(),
                                      p1_buf=self.p1_buf.data_ptr())
        self.solver.bind_partials(self.parts[0].data_ptr(), self.parts[1].data_ptr())
        self.blocks = self.solver.blocks     # partial sums this rank's kernels write (the larger of A's and B's)
        self.blocks_of = [self.solver.blocks_of(0), self.solver.blocks_of(1)]
        self.nparts = self.blocks            # ... and the number every rank reduces
        # the planes r's boundary will hold after the coming kernel B, computed ahead of it by the library
        # (cg_solver_edge_copy / cg_solver_edge_update): library memory, viewed as tensors for the collectives
        self.edge_lo = self._view(self.solver.edge_plane(0), self.plane) if has_lo else None
        self.edge_hi = self._view(self.solver.edge_plane(1), self.plane) if has_hi else None

    def set_nparts(self, n):
        """Ranks whose slabs differ in shape own different numbers of workgroups; all of them reduce
        n = the largest count (the tail of a shorter array stays zero)."""
        if n > self.parts.shape[1]:
            raise ValueError(f"{n} partial sums exceed the {self.parts.shape[1]} slots")
        self.nparts = int(n)
        self.rebind_partials()

    def rebind_partials(self):
        self.parts.zero_()
        torch.cuda.synchronize(self.device)
        self.solver.bind_partials(self.parts[0].data_ptr(), self.parts[1].data_ptr(), self.nparts)

    def _view(self, ptr, n):
        class Raw:
            __cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(Raw(), device=self.device)

    def edge_copy(self):
        self.solver.edge_copy()

    def edge_update(self):
        self.solver.edge_update()

    def stream_ctx(self):
        return torch.cuda.stream(self.stream)

    def begin_local(self, b, x):
        # b and x were produced on the caller's (default) stream
        self.stream.wait_stream(torch.cuda.default_stream(self.device))
        self.solver.begin_local(b.data_ptr(), x.data_ptr())

    def begin_linked(self, b, x, rel, lagged):
        self.stream.wait_stream(torch.cuda.default_stream(self.device))
        self.solver.begin(b.data_ptr(), x.data_ptr(), rel, lagged)

    def sum_tensor(self, which):
        """The tensor whose element-wise sum over ranks the next kernel consumes.  Kernel A writes this
        rank's first blocks_of[0] entries of row 0, kernel B blocks_of[1] of row 1; the rest of the common
        array holds the previous reduction's result and is cleared here, before the next in-place reduction
        adds it up again."""
        wrote = self.blocks_of[which]
        if self.nparts > wrote:
            self.parts[which, wrote:self.nparts].zero_()
        return self.parts[which, :self.nparts]

    def init_state(self, rel, lagged, maxit):
        self.solver.init_state(rel, lagged, maxit)

    def add_iterations(self, n):
        self.solver.add_iterations(n)

    def kernel_A(self):
        self.solver.kernel_A()

    def kernel_B(self):
        self.solver.kernel_B()

    def finish(self, hist_len=0):
        return self.solver.finish(hist_len)

    def sync(self):
        self.solver.sync()


class SlabCG:
    """CG on this rank's slab of an extents[0] x ... x (sum of slab heights) grid.

    extents: this rank's local extents (axis 0 contiguous, last axis = slab axis).
    """

    def __init__(self, lib, extents, rank=None, world=None, group=None, device=None, make_ops=None):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.extents = [int(e) for e in extents]
        if len(self.extents) < 2 and self.world > 1:
            raise ValueError("slab decomposition needs d >= 2")
        self.links = False
        self.links_reason = ""
        self.rccl = False
        self.rccl_reason = ""
        self._comm = None
        self.has_lo = self.rank > 0
        self.has_hi = self.rank < self.world - 1
        self.N = 1
        for e in self.extents:
            self.N *= e
        if make_ops is None:
            if lib is None:
                raise RuntimeError("SlabCG needs the HIP library (no fallback)")
            self.ops = HipSlabOps(lib, self.extents, int(self.has_lo), int(self.has_hi), device)
        else:
            self.ops = make_ops(self.extents, int(self.has_lo), int(self.has_hi))
        o = self.ops
        # every rank all-reduces the same number of partial sums (slabs of different shape launch
        # different numbers of workgroups)
        if self.world > 1 and hasattr(o, "nparts"):
            counts = [None] * self.world
            dist.all_gather_object(counts, int(o.nparts), group=self.group)
            if len(set(counts)) > 1:
                o.set_nparts(max(counts))
        g, n, pl = o.ghost_offset, self.N, o.plane
        # views used by the halo exchange
        self.r_ghost_lo = o.r_buf[g - pl:g]
        self.r_first = o.r_buf[g:g + pl]
        self.r_last = o.r_buf[g + n - pl:g + n]
        self.r_ghost_hi = o.r_buf[g + n:g + n + pl]

    # -- communication ------------------------------------------------------------------
    def _global_rank(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def exchange_r(self, ahead=False):
        """Boundary planes of r to the neighbours' ghost planes (both directions at once).  ahead: send the
        planes computed ahead of kernel B (ops.edge_lo / edge_hi: the same bits) instead of r's own."""
        if self.world == 1:
            return
        first = self.ops.edge_lo if ahead and self.has_lo else self.r_first
        last = self.ops.edge_hi if ahead and self.has_hi else self.r_last
        if self._host_staged(self.r_first):
            self._exchange_r_host_staged(first, last)
            return
        ops = []
        if self.has_lo:
            ops.append(dist.P2POp(dist.isend, first, self._global_rank(self.rank - 1), self.group))
            ops.append(dist.P2POp(dist.irecv, self.r_ghost_lo, self._global_rank(self.rank - 1), self.group))
        if self.has_hi:
            ops.append(dist.P2POp(dist.isend, last, self._global_rank(self.rank + 1), self.group))
            ops.append(dist.P2POp(dist.irecv, self.r_ghost_hi, self._global_rank(self.rank + 1), self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()

    def allreduce(self, which):
        t = self.ops.sum_tensor(which)
        if self.world > 1:
            if self._host_staged(t):
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    # -- rehearsal path: device tensors over a backend that only moves host memory (gloo) -----
    # Lets several ranks share ONE GPU (RCCL refuses that) to rehearse the multi-rank flow; it
    # synchronises with the host on every exchange and is never used with the nccl backend.
    def _host_staged(self, t):
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def _exchange_r_host_staged(self, first, last):
        reqs, recvs = [], []
        if self.has_lo:
            reqs.append(dist.isend(first.cpu(), self._global_rank(self.rank - 1), self.group))
            buf = torch.empty(self.r_ghost_lo.shape, dtype=self.r_ghost_lo.dtype)
            reqs.append(dist.irecv(buf, self._global_rank(self.rank - 1), self.group))
            recvs.append((self.r_ghost_lo, buf))
        if self.has_hi:
            reqs.append(dist.isend(last.cpu(), self._global_rank(self.rank + 1), self.group))
            buf = torch.empty(self.r_ghost_hi.shape, dtype=self.r_ghost_hi.dtype)
            reqs.append(dist.irecv(buf, self._global_rank(self.rank + 1), self.group))
            recvs.append((self.r_ghost_hi, buf))
        for r in reqs:
            r.wait()
        for dst, buf in recvs:
            dst.copy_(buf)

    # -- RCCL issued by the library ---------------------------------------------------------
    def enable_rccl(self):
        """Create an RCCL communicator over this group's ranks (unique id from rank 0, carried by
        torch.distributed) and bind it to the solver.  Collective; True on all ranks or False on all."""
        if self.rccl:
            return True
        if self.links:
            self.disable_links()
        lib, solver = self.ops.lib, self.ops.solver
        ok, uid = 1, [None]
        if self.rank == 0:
            try:
                uid = [lib.rccl_unique_id()]
            except Exception as e:  # noqa: BLE001 -- any failure means "use another transport"
                ok, self.rccl_reason = 0, repr(e)
        src = self._global_rank(0)
        dist.broadcast_object_list(uid, src=src, group=self.group)
        if uid[0] is None:
            ok = 0
        if self._all_agree(ok):
            try:
                with self._ctx():
                    self._comm = lib.rccl_comm_create(uid[0], self.rank, self.world)
            except Exception as e:  # noqa: BLE001
                ok, self.rccl_reason = 0, repr(e)
        else:
            ok = 0
        if not self._all_agree(ok):
            self._release_rccl()
            return False
        try:
            solver.bind_rccl(self._comm, self.rank, self.world)
        except Exception as e:  # noqa: BLE001
            ok, self.rccl_reason = 0, repr(e)
        if not self._all_agree(ok):
            self._release_rccl()
            return False
        self.rccl = True
        return True

    def disable_rccl(self):
        """Back to the Python-driven collectives (collective call)."""
        if not self.rccl:
            return
        self.ops.solver.bind_rccl(None, 0, 1)       # waits for this rank's stream
        self.ops.rebind_partials()
        self.rccl = False
        dist.barrier(group=self.group)
        self._release_rccl()

    def _release_rccl(self):
        if self._comm:
            self.ops.lib.rccl_comm_destroy(self._comm)
            self._comm = None

    # -- direct xGMI path ---------------------------------------------------------------
    def enable_links(self):
        """Map every rank's box into this process and bind them to the solver.  Collective: all
        ranks call it; returns True on all ranks or False on all ranks (then nothing changed)."""
        if self.links:
            return True
        if self.rccl:
            self.disable_rccl()
        lib, solver = self.ops.lib, self.ops.solver
        ok, handle = 1, b""
        try:
            self._box = lib.links_alloc(solver.box_bytes())
            handle = lib.links_export(self._box)
        except Exception as e:  # noqa: BLE001 -- any failure means "use the collectives"
            ok, self.links_reason = 0, repr(e)
        gathered = [None] * self.world
        dev_index = getattr(getattr(self.ops, "device", None), "index", None)
        if dev_index is None:
            dev_index = torch.cuda.current_device()
        me = (ok, socket.gethostname(), os.getpid(), dev_index, handle)
        dist.all_gather_object(gathered, me, group=self.group)
        boxes, self._opened = [], []
        if not all(g[0] for g in gathered):
            ok = 0
        elif any(g[1] != me[1] for g in gathered):
            ok, self.links_reason = 0, "ranks on different hosts"
        else:
            try:
                for j, (_, _, pid, dev, h) in enumerate(gathered):
                    if j == self.rank:
                        boxes.append(self._box)
                        continue
                    if not lib.links_peer_ok(dev):
                        raise RuntimeError(f"device {me[3]} cannot write the memory of device {dev}")
                    # (an IPC handle cannot be opened by the process that made it: own box above)
                    p = lib.links_open(h)
                    self._opened.append(p)
                    boxes.append(p)
            except Exception as e:  # noqa: BLE001
                ok, self.links_reason = 0, repr(e)
        if not self._all_agree(ok):
            self._release_links()
            return False
        # every rank writes a token into every box through the copy engines and reads back what
        # the others wrote into its own: a mapping that does not work shows up here as an error
        # code or a wrong word, not as a fault in the first kernel that stores through it
        try:
            for j, bx in enumerate(boxes):
                lib.links_poke(bx, self.rank, 0xC0FFEE00 + 64 * j + self.rank)
        except Exception as e:  # noqa: BLE001
            ok, self.links_reason = 0, repr(e)
        ok = self._all_agree(ok)          # also: all pokes have landed before anybody peeks
        if ok:
            try:
                for j in range(self.world):
                    if lib.links_peek(self._box, j) != 0xC0FFEE00 + 64 * self.rank + j:
                        raise RuntimeError(f"probe word from rank {j} did not arrive")
                    lib.links_poke(self._box, j, 0)
            except Exception as e:  # noqa: BLE001
                ok, self.links_reason = 0, repr(e)
            ok = self._all_agree(ok)
        if not ok:
            self._release_links()
            return False
        solver.bind_links(self.rank, boxes)
        self.links = True
        return True

    def disable_links(self):
        """Back to collectives (collective call)."""
        if not self.links:
            return
        self.ops.solver.bind_links(self.rank, [])   # waits for this rank's stream
        self.ops.rebind_partials()
        self.links = False
        dist.barrier(group=self.group)                # nobody writes into a box any more
        self._release_links()

    def links_error(self):
        return self.ops.solver.links_error() if self.links else 0

    def ghosts_match_neighbours(self):
        """After finish(): fetch the neighbours' boundary planes of r once more through
        torch.distributed and compare them bit for bit with this rank's ghost planes (what the
        last exchange of the solve must have delivered).  Collective; returns this rank's verdict."""
        self.sync()
        if self.world == 1:
            return True
        staged = self._host_staged(self.r_first)
        dev = "cpu" if staged else self.r_first.device
        lo = torch.empty(self.r_ghost_lo.shape, dtype=self.r_ghost_lo.dtype, device=dev)
        hi = torch.empty(self.r_ghost_hi.shape, dtype=self.r_ghost_hi.dtype, device=dev)
        first = self.r_first.cpu() if staged else self.r_first.clone()
        last = self.r_last.cpu() if staged else self.r_last.clone()
        ops = []
        if self.has_lo:
            ops.append(dist.P2POp(dist.isend, first, self._global_rank(self.rank - 1), self.group))
            ops.append(dist.P2POp(dist.irecv, lo, self._global_rank(self.rank - 1), self.group))
        if self.has_hi:
            ops.append(dist.P2POp(dist.isend, last, self._global_rank(self.rank + 1), self.group))
            ops.append(dist.P2POp(dist.irecv, hi, self._global_rank(self.rank + 1), self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if not staged:
            torch.cuda.synchronize(self.r_first.device)
        ok = True
        if self.has_lo:
            ok = ok and torch.equal(lo.to(self.r_ghost_lo.device), self.r_ghost_lo)
        if self.has_hi:
            ok = ok and torch.equal(hi.to(self.r_ghost_hi.device), self.r_ghost_hi)
        return ok

    def _all_agree(self, ok):
        votes = [None] * self.world
        dist.all_gather_object(votes, int(ok), group=self.group)
        return all(votes)

    def _release_links(self):
        lib = self.ops.lib
        for p in getattr(self, "_opened", []):
            lib.links_close(p)
        self._opened = []
        if getattr(self, "_box", None):
            lib.links_free(self._box)
            self._box = None

    # -- solve --------------------------------------------------------------------------
    def _ctx(self):
        return self.ops.stream_ctx() if hasattr(self.ops, "stream_ctx") else contextlib.nullcontext()

    def begin(self, b, x, rel=1e-6, lagged=1):
        """b, x: this rank's slab (N floats each).  r = b, p = 0, global r.r, ghost planes of r."""
        with self._ctx():
            if self.links or self.rccl:
                self.ops.begin_linked(b, x, rel, lagged)
                return
            self.ops.begin_local(b, x)
            self.allreduce(1)
            self.ops.init_state(rel, lagged, 0)
            self.exchange_r()

    def iterate(self, n):
        with self._ctx():
            if self.links or self.rccl:
                self.ops.solver.iterate(n, True)
                return
            self.ops.add_iterations(n)
            # the planes that travel are the ones computed ahead of kernel B (as in the library's RCCL path,
            # where that lets them travel beside kernel B; here everything shares one stream)
            ahead = hasattr(self.ops, "edge_update") and self.world > 1
            for _ in range(n):
                if ahead:
                    self.ops.edge_copy()
                self.ops.kernel_A()
                self.allreduce(0)
                if ahead:
                    self.ops.edge_update()
                self.ops.kernel_B()
                self.allreduce(1)
                self.exchange_r(ahead)

    def finish(self, hist_len=0):
        out = self.ops.finish(hist_len)
        if self.links and self.ops.solver.links_error():
            raise RuntimeError("direct xGMI exchange timed out waiting for another rank; the result is invalid")
        return out

    def sync(self):
        self.ops.sync()
