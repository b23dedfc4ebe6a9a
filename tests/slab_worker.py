"""Worker for the multi-process slab tests:
python slab_worker.py <out.npz> <L> <d> <iters> [cpu|hip|links] [h0,h1,...: slab heights, default equal].
Reads RANK / WORLD_SIZE / MASTER_* from the environment (gloo).  Without `hip` the per-slab
compute is the CPU stand-in; with `hip` every rank runs the HIP kernels on cuda:0 (the ranks
share the one GPU, the exchange is staged through host memory because RCCL refuses two
ranks on one device); with `links` the ranks exchange through each other's device memory
(IPC-mapped boxes) and gloo only carries the handles."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from conftest import load_package  # noqa: E402
from fake_slab_ops import FakeSlabOps  # noqa: E402


def main():
    out, L, d, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    mode = sys.argv[5] if len(sys.argv) > 5 else "cpu"      # cpu | hip | links
    hip = mode in ("hip", "links")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pkg = load_package()
    from conjugate_gradient_amd.slab import SlabCG
    if len(sys.argv) > 6:
        heights = [int(v) for v in sys.argv[6].split(",")]
        assert len(heights) == world and sum(heights) == L
    else:
        assert L % world == 0
        heights = [L // world] * world
    h = heights[rank]
    ext = [L] * (d - 1) + [h]
    N = L ** d
    rng = np.random.default_rng(123)
    xt = rng.random(N).astype(np.float32)            # same on every rank
    from oracle import oracle as orc
    b = orc.Oracle().laplace_f32(xt, d, L)
    plane = N // L
    n_loc, first = h * plane, sum(heights[:rank]) * plane
    b_loc = torch.from_numpy(b[first:first + n_loc].copy())
    x_loc = torch.zeros(n_loc, dtype=torch.float32)
    if hip:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        b_loc, x_loc = b_loc.to(dev), x_loc.to(dev)
        s = SlabCG(pkg.load_library(), ext, device=dev)
        if mode == "links":
            assert s.enable_links(), s.links_reason
    else:
        s = SlabCG(None, ext, make_ops=lambda e, lo, hi: FakeSlabOps(e, lo, hi))
    s.begin(b_loc, x_loc, 1e-6, 1)
    s.iterate(iters)
    n, res, hist = s.finish(hist_len=iters)
    if hip:
        assert s.ghosts_match_neighbours()
    x_loc = x_loc.cpu()
    gathered = [None] * world
    dist.all_gather_object(gathered, x_loc.numpy())
    if rank == 0:
        np.savez(out, x=np.concatenate(gathered), xt=xt, b=b, iters=n, res=res, hist=np.array(hist))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
