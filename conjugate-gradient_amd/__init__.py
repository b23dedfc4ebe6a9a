"""MI355X-native Laplace conjugate-gradient hot path.

Python is only glue here: `binding` loads lib/libconjugate_gradient_gpu.so (HIP, gfx950)
through ctypes and mirrors the reference's entry points by name; `slab` drives the same
kernels across ranks (one process per GPU, torch.distributed / RCCL).  There is no CPU
fallback: every compute call goes into the HIP library and fails loudly without it.
"""
from .binding import Library, LIB_PATH, build_library, load_library  # noqa: F401
