"""Host-side cost per slab iteration: tiny grid (GPU work negligible), collectives forced even at world size 1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from __graft_entry__ import load_package
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
pkg = load_package(); lib = pkg.load_library()
from conjugate_gradient_amd.slab import SlabCG
ext = [256, 256]
N = 256 * 256
b = torch.rand(N, device="cuda"); x = torch.zeros(N, device="cuda")
s = SlabCG(lib, ext, rank=0, world=1, device=torch.device("cuda", 0))
s.begin(b, x, 0.0, 1)
s.iterate(20); s.sync()
def timed(label, f, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{label}: host {1e6*(t1-t0)/n:.1f} us/call, with drain {1e6*(t2-t0)/n:.1f} us/call")
o = s.ops
with o.stream_ctx():
    timed("kernel_A", o.kernel_A)
    timed("all_reduce(1 rank)", lambda: dist.all_reduce(o.sum_tensor(0)))
    timed("iterate(1) eager", lambda: s.iterate(1))
    # graph capture of one iteration including a collective
    try:
        g = torch.cuda.CUDAGraph()
        o.add_iterations(10**6)
        with torch.cuda.graph(g, stream=o.stream):
            for _ in range(10):
                o.kernel_A(); dist.all_reduce(o.sum_tensor(0)); o.kernel_B(); dist.all_reduce(o.sum_tensor(1))
        timed("graph replay (10 iterations)", g.replay, 50)
    except Exception as e:
        print("graph capture failed:", repr(e)[:300])
dist.destroy_process_group()
