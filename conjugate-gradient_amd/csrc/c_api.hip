// c_api.hip -- the extern "C" boundary of libconjugate_gradient_gpu.so and the C++ host
// functions of conjugate-gradient_gpu.cuh.  Part 1 re-creates the reference's exports
// (conjugate-gradient_gpu.cu:34-42,128-148,239-270,327-386; julia.cu:16-60) on top of the
// fused HIP solver; part 2 exposes the solver object.
#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <chrono>
#include <memory>
#include <mutex>

#include "../../include/conjugate_gradient_cpu.h"
#include "../../include/conjugate_gradient_gpu.hpp"
#include "cg_solver.hpp"
#include "hip_check.hpp"

extern char **environ;

using namespace cgamd;

void cgamd_launch_add(float *A, float *B, int blocks, int threads);
void cgamd_launch_laplace_compat(float *ddf, float *u, unsigned d, unsigned L, unsigned N, unsigned index_mode,
                                 unsigned blocks, unsigned threads);

namespace
{

enum PtrKind
{
  PK_HOST,
  PK_MANAGED,
  PK_DEVICE
};

PtrKind classify(const void *p)
{
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess)
  {
    (void)hipGetLastError();
    return PK_HOST;
  }
  if (at.type == hipMemoryTypeDevice)
    return PK_DEVICE;
  if (at.type == hipMemoryTypeManaged || at.isManaged)
    return PK_MANAGED;
  return PK_HOST;
}

// A vector the hot loop can stream from HBM: the caller's own pointer when it is device
// memory, otherwise a hipMalloc copy (managed and host memory are served over the host
// link, which would bound the solver instead of HBM).
struct Staged
{
  float *dev = nullptr;
  float *orig = nullptr;
  size_t n = 0;
  bool copy = false;
  Staged(const float *p, size_t count) : orig(const_cast<float *>(p)), n(count)
  {
    if (classify(p) == PK_DEVICE)
      dev = orig;
    else
    {
      copy = true;
      CHECK(hipMalloc((void **)&dev, n * sizeof(float)));
      CHECK(hipMemcpy(dev, orig, n * sizeof(float), hipMemcpyDefault));
    }
  }
  void write_back()
  {
    if (copy)
      CHECK(hipMemcpy(orig, dev, n * sizeof(float), hipMemcpyDefault));
  }
  ~Staged()
  {
    if (copy)
      (void)hipFree(dev);
  }
};

unsigned long long grid_points(int L, int d)
{
  CG_REQUIRE(L >= 1 && d >= 1 && d <= CG_MAXD, "need L >= 1 and 1 <= d <= 32");
  unsigned long long n = 1;
  for (int k = 0; k < d; k++)
  {
    n *= (unsigned long long)L;
    CG_REQUIRE(n < (1ull << 31), "L^d must fit an int, as in the reference (conjugate-gradient_gpu.cu:330)");
  }
  return n;
}

// one cached solver: repeated solves on the same grid (the preconditioner pattern,
// benchmarks) reuse r/p buffers and the captured graph
struct Cache
{
  int L = 0, d = 0, mode = -1, device = -1;
  unsigned forced = 0;
  std::unique_ptr<Solver> solver;
  Solver &get(int L_, int d_, int mode_)
  {
    const int m = mode_ == 2 ? 2 : 0;
    int dev = 0;
    CHECK(hipGetDevice(&dev));
    // part of the key: the CG_* environment switches change what a new solver picks (FNV-1a over them)
    unsigned f = 2166136261u;
    for (char **e = environ; e && *e; e++)
      if (strncmp(*e, "CG_", 3) == 0)
        for (const char *c = *e; *c; c++)
          f = (f ^ (unsigned char)*c) * 16777619u;
    if (!solver || L != L_ || d != d_ || mode != m || device != dev || forced != f)
    {
      device = dev;
      forced = f;
      solver.reset();
      SolverDesc desc;
      desc.d = d_;
      for (int k = 0; k < d_; k++)
        desc.extents[k] = (unsigned)L_;
      desc.index_mode = m;
      solver.reset(new Solver(desc));
      L = L_;
      d = d_;
      mode = m;
    }
    return *solver;
  }
  // The reference allocates r, p, Ap inside the call and frees them before it returns
  // (conjugate-gradient_gpu.cu:335-339,371-373).  A cached solver is kept between calls only while it is
  // small (CG_CACHE_KEEP_MB, default 1024 MiB of device memory): the next call on the same grid then skips
  // the allocations and the graph capture.  Anything larger is given back on return.
  void release_if_large()
  {
    const char *e = getenv("CG_CACHE_KEEP_MB");
    const size_t keep = (size_t)((e && *e) ? atoll(e) : 1024) << 20;
    if (solver && solver->device_bytes() > keep)
      release();
  }
  void release()
  {
    solver.reset();
    L = d = 0;
    mode = device = -1;
  }
};
// never destroyed: tearing HIP objects down from a static destructor at process exit is unsafe
Cache &cache()
{
  static Cache *c = new Cache;
  return *c;
}

uint64_t now_us()
{
  using namespace std::chrono;
  return duration_cast<microseconds>(steady_clock::now().time_since_epoch()).count();
}

} // namespace

// =============================== Part 1 ===============================================

// host entry point; device code uses the same inline functions of index_map.hpp directly
extern "C" unsigned int neighbour_index_gpu(unsigned int ind, int direction, int amount, unsigned int L,
                                            unsigned int d, unsigned int N, int index_mode)
{
  assert(amount == 1 || amount == -1 || amount == 0);
  if (index_mode == 2)
  {
    if (amount == 0)
      return ind;
    const unsigned int m = morton_mask((int)d, direction);
    const unsigned int nb = amount > 0 ? morton_up(ind, m, N) : morton_dn(ind, m, N);
    return nb < N ? nb : N;
  }
  return lex_neighbour(ind, direction, amount, L, N);
}

float *cuda_allocate_field(int N)
{
  CG_REQUIRE(N >= 0, "field size must not be negative");
  float *field = nullptr;
  const size_t bytes = ((size_t)N + 1) * sizeof(float);
  int managed = 0;
  int dev = 0;
  CHECK(hipGetDevice(&dev));
  CHECK(hipDeviceGetAttribute(&managed, hipDeviceAttributeManagedMemory, dev));
  if (managed)
  {
    CHECK(hipMallocManaged((void **)&field, bytes));
  }
  else
  {
    // same contract without managed-memory support: pinned host memory mapped into the device
    CHECK(hipHostMalloc((void **)&field, bytes, hipHostMallocMapped));
  }
  CHECK(hipMemset(field, 0, bytes));
  CHECK(hipDeviceSynchronize());
  return field;
}

extern "C" float *cg_allocate_field(int N) { return cuda_allocate_field(N); }

extern "C" void cg_free_field(float *field)
{
  if (!field)
    return;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, field) == hipSuccess && at.type == hipMemoryTypeHost)
  {
    CHECK(hipHostFree(field));
  }
  else
  {
    (void)hipGetLastError();
    CHECK(hipFree(field));
  }
}

extern "C" float inner_product_gpu(float *v, float *w, unsigned int N)
{
  if (N == 0)
    return 0.0f;
  const bool same = v == w;
  Staged sv(v, N);
  if (same)
  {
    const double s = dot_device(sv.dev, sv.dev, N, nullptr);
    return (float)s;
  }
  Staged sw(w, N);
  return (float)dot_device(sv.dev, sw.dev, N, nullptr);
}

float norm(float *v, int N) { return sqrt(inner_product_gpu(v, v, (unsigned)N)); }

float *random_array(float *r, int L, int d, int N)
{
  (void)L;
  (void)d;
  for (int i = 0; i < N; i++)
    r[i] = (float)rand() / (float)RAND_MAX;
  return r;
}

extern "C" float conjugate_gradient_gpu_ex(const float *b, float *x, int L, int d, int index_mode, int maxit, double rel,
                                           int lagged, int verbose, int *iters_out, float *hist, int hist_len)
{
  const unsigned long long N = grid_points(L, d);
  assert(N > 1); // conjugate-gradient_gpu.cu:331
  Solver &s = cache().get(L, d, index_mode);
  Staged sb(b, N), sx(x, N);
  s.begin(sb.dev, sx.dev, rel, lagged);
  s.run(maxit, true);
  float residue = 0;
  std::unique_ptr<float[]> own;
  float *h = hist;
  int hl = hist_len;
  if (verbose && !hist)
  {
    hl = maxit >= 0 ? maxit : CG_HIST;
    own.reset(new float[(size_t)hl > 0 ? hl : 1]);
    h = own.get();
  }
  const int n = s.finish(&residue, h, hl);
  if (verbose)
  {
    // what conjugate-gradient_gpu.cu:369 prints inside its loop
    for (int i = 0; i < n && i < hl; i++)
      printf("%d,%f\n", i + 1, h[i]);
  }
  sx.write_back();
  if (iters_out)
    *iters_out = n;
  cache().release_if_large();
  return residue;
}

extern "C" void cg_release_cache(void) { cache().release(); }

extern "C" float conjugate_gradient_gpu(float *b, float *x, int L, int d)
{
  const char *q = getenv("CG_GPU_QUIET");
  const int verbose = !(q && *q && *q != '0');
  return conjugate_gradient_gpu_ex(b, x, L, d, 0, 200, 1e-6, 1, verbose, nullptr, nullptr, 0);
}

extern "C" void laplace_gpu_jl(float *ddf, float *u, float dx, int d, int L, int N, unsigned int index_mode,
                               unsigned int blocks, unsigned int threads)
{
  (void)dx;
  const unsigned long long covered = (unsigned long long)blocks * threads;
  const unsigned long long full = grid_points(L, d);
  if (covered >= (unsigned long long)N && (unsigned long long)N == full && classify(u) != PK_HOST &&
      classify(ddf) != PK_HOST)
  {
    // the launch covers the grid: use the tiled kernel
    laplace_apply(ddf, u, d, (unsigned)L, (int)index_mode, nullptr);
  }
  else
  {
    // partial launches keep the reference's meaning: thread i handles point i
    Staged su(u, (size_t)N), sd(ddf, (size_t)N);
    cgamd_launch_laplace_compat(sd.dev, su.dev, (unsigned)d, (unsigned)L, (unsigned)N, index_mode, blocks, threads);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    sd.write_back();
    return;
  }
  CHECK(hipDeviceSynchronize());
}

extern "C" void add_jl(float *A, float *B, int blocks, int threads)
{
  cgamd_launch_add(A, B, blocks, threads);
  CHECK(hipGetLastError());
}

extern "C" float strong_scaling(int nblocks, int threads_per_block, int N, int L, int d)
{
  CG_REQUIRE(N >= 1 && nblocks >= 1 && threads_per_block >= 1 && threads_per_block <= 1024,
             "need N, nblocks >= 1 and 1 <= threads_per_block <= 1024");
  float *x = nullptr, *b = nullptr;
  CHECK(hipMalloc((void **)&x, ((size_t)N + 1) * sizeof(float)));
  CHECK(hipMalloc((void **)&b, ((size_t)N + 1) * sizeof(float)));
  fill_uniform(x, (unsigned long long)N, 1u, nullptr);
  CHECK(hipMemset(b, 0, ((size_t)N + 1) * sizeof(float)));
  CHECK(hipDeviceSynchronize());
  const uint64_t t0 = now_us();
  cgamd_launch_laplace_compat(b, x, (unsigned)d, (unsigned)L, (unsigned)N, 1u, (unsigned)nblocks,
                              (unsigned)threads_per_block);
  CHECK(hipDeviceSynchronize());
  const uint64_t t1 = now_us();
  CHECK(hipGetLastError());
  CHECK(hipFree(x));
  CHECK(hipFree(b));
  return (float)(t1 - t0);
}

extern "C" float time_cpu_laplace(int N, int L, int d)
{
  double *x = allocate_field(N);
  double *b = allocate_field(N);
  random_array(x, L, d, N);
  const uint64_t t0 = now_us();
  minus_laplace(b, x, d, L, N);
  const uint64_t t1 = now_us();
  free(x);
  free(b);
  return (float)(t1 - t0);
}

// =============================== Part 2 ===============================================

// ---- inner product without allocation -----------------------------------------------------------
// Scratch of cg_inner_product_noalloc: the per-workgroup partial sums of the dot kernel.  Every GPU the
// library runs on holds its own copy of this array (32 KiB) until the process ends; the name below is
// only a host-side handle, the device address is looked up once per device.
static __device__ __align__(256) double scratch[2 * CG_MAX_BLOCKS];

namespace
{
constexpr int NOALLOC_MAXDEV = 64;
double *scratch_of[NOALLOC_MAXDEV] = {nullptr};

// device address of this device's copy of `scratch` (looked up on first use, with that device current)
int scratch_ptr(double **out)
{
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess)
    return (int)e;
  if (dev < 0 || dev >= NOALLOC_MAXDEV)
    return (int)hipErrorInvalidDevice;
  static std::mutex guard; // two host threads may make their first call on a device at the same time
  std::lock_guard<std::mutex> lock(guard);
  if (!scratch_of[dev])
  {
    double *ptr = nullptr;
    e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr)
    {
      fprintf(stderr, "cg_inner_product_noalloc: hipGetSymbolAddress failed on device %d, code: %d, reason: %s\n", dev,
              (int)e, hipGetErrorString(e));
      return e != hipSuccess ? (int)e : (int)hipErrorInvalidSymbol;
    }
    scratch_of[dev] = ptr;
  }
  *out = scratch_of[dev];
  return 0;
}
} // namespace

extern "C" int cg_noalloc_prepare(void)
{
  double *ptr = nullptr;
  return scratch_ptr(&ptr);
}

extern "C" int cg_inner_product_noalloc(const float *v, const float *w, unsigned long long n, float *result_dev,
                                        void *stream)
{
  if (!v || !w || !result_dev)
  {
    fprintf(stderr, "cg_inner_product_noalloc: null pointer argument\n");
    return (int)hipErrorInvalidValue;
  }
  if (n == 0 || n > CG_NOALLOC_MAX_N)
  {
    fprintf(stderr, "cg_inner_product_noalloc: n = %llu is outside [1, %llu]\n", n, (unsigned long long)CG_NOALLOC_MAX_N);
    return (int)hipErrorInvalidValue;
  }
  double *part = nullptr;
  const int e = scratch_ptr(&part);
  if (e)
    return e;
  // every partial the second stage reads is written by the first stage of the same call, so nothing left
  // in the scratch by an earlier (possibly larger) call is ever read and nothing needs clearing
  dot_device_into(v, w, n, part, result_dev, (hipStream_t)stream);
  return 0;
}

// strong_scaling on the device clock: HIP events around `reps` launches of the caller's shape
// (after one untimed launch), device-resident data, no host synchronisation inside the window
extern "C" float cg_strong_scaling_events(int nblocks, int threads_per_block, int N, int L, int d, int reps)
{
  CG_REQUIRE(N >= 1 && nblocks >= 1 && threads_per_block >= 1 && threads_per_block <= 1024 && reps >= 1,
             "need N, nblocks, reps >= 1 and 1 <= threads_per_block <= 1024");
  float *x = nullptr, *b = nullptr;
  CHECK(hipMalloc((void **)&x, ((size_t)N + 1) * sizeof(float)));
  CHECK(hipMalloc((void **)&b, ((size_t)N + 1) * sizeof(float)));
  fill_uniform(x, (unsigned long long)N, 1u, nullptr);
  CHECK(hipMemset(b, 0, ((size_t)N + 1) * sizeof(float)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  cgamd_launch_laplace_compat(b, x, (unsigned)d, (unsigned)L, (unsigned)N, 1u, (unsigned)nblocks,
                              (unsigned)threads_per_block);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < reps; i++)
    cgamd_launch_laplace_compat(b, x, (unsigned)d, (unsigned)L, (unsigned)N, 1u, (unsigned)nblocks,
                                (unsigned)threads_per_block);
  CHECK(hipEventRecord(e1, nullptr));
  CHECK(hipEventSynchronize(e1));
  CHECK(hipGetLastError());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  CHECK(hipFree(x));
  CHECK(hipFree(b));
  return ms * 1000.0f / (float)reps;
}

// the same measurement for the tiled stencil the library itself uses for a whole grid
extern "C" float cg_laplace_time_events(int L, int d, int index_mode, int reps)
{
  const unsigned long long N = grid_points(L, d);
  CG_REQUIRE(reps >= 1, "need reps >= 1");
  float *x = nullptr, *b = nullptr;
  CHECK(hipMalloc((void **)&x, (N + 1) * sizeof(float)));
  CHECK(hipMalloc((void **)&b, (N + 1) * sizeof(float)));
  fill_uniform(x, N, 1u, nullptr);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  laplace_apply(b, x, d, (unsigned)L, index_mode, nullptr);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < reps; i++)
    laplace_apply(b, x, d, (unsigned)L, index_mode, nullptr);
  CHECK(hipEventRecord(e1, nullptr));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  CHECK(hipFree(x));
  CHECK(hipFree(b));
  return ms * 1000.0f / (float)reps;
}

extern "C" void cg_laplace_device(float *ddf, const float *u, int d, int L, int index_mode, void *stream)
{
  laplace_apply(ddf, u, d, (unsigned)L, index_mode, (hipStream_t)stream);
}

extern "C" void cg_fill_uniform(float *dst, unsigned long long n, unsigned int seed)
{
  fill_uniform(dst, n, seed, nullptr);
  CHECK(hipDeviceSynchronize());
}

extern "C" float cg_u01_host(unsigned int seed, unsigned long long i) { return u01_hash(seed, i); }

struct cg_solver
{
  Solver impl;
  explicit cg_solver(const SolverDesc &d) : impl(d) {}
};

Solver &cg_solver_impl(cg_solver *s) { return s->impl; }

static SolverDesc to_desc(const cg_solver_desc *c)
{
  CG_REQUIRE(c != nullptr, "null solver description");
  CG_REQUIRE(c->d >= 1 && c->d <= CG_MAXD, "dimension must be in [1,32]");
  SolverDesc d;
  d.d = c->d;
  for (int k = 0; k < c->d; k++)
    d.extents[k] = c->extents[k];
  d.index_mode = c->index_mode;
  d.has_lo = c->has_lo;
  d.has_hi = c->has_hi;
  d.stream = (hipStream_t)c->stream;
  d.r_buf = c->r_buf;
  d.p0_buf = c->p0_buf;
  d.p1_buf = c->p1_buf;
  d.engine = c->engine;
  return d;
}

extern "C" size_t cg_solver_buffer_len(const cg_solver_desc *desc) { return Solver::buffer_len_of(to_desc(desc)); }
extern "C" size_t cg_solver_ghost_offset(const cg_solver_desc *desc) { return Solver::ghost_offset_of(to_desc(desc)); }
extern "C" size_t cg_solver_plane(const cg_solver_desc *desc) { return Solver::plane_of(to_desc(desc)); }

extern "C" cg_solver *cg_solver_create(const cg_solver_desc *desc) { return new cg_solver(to_desc(desc)); }
extern "C" void cg_solver_destroy(cg_solver *s) { delete s; }
extern "C" int cg_solver_engine(const cg_solver *s) { return s->impl.engine(); }
extern "C" int cg_solver_blocks(const cg_solver *s) { return s->impl.blocks(); }
extern "C" int cg_solver_blocks_of(const cg_solver *s, int which) { return s->impl.blocks_of(which); }
extern "C" int cg_solver_paired_x(const cg_solver *s) { return s->impl.paired_x() ? 1 : 0; }
extern "C" const char *cg_solver_kernel_name(const cg_solver *s) { return s->impl.kernel_name(); }

extern "C" void cg_solver_begin(cg_solver *s, const float *b, float *x, double rel, int lagged)
{
  s->impl.begin(b, x, rel, lagged);
}
extern "C" void cg_solver_iterate(cg_solver *s, int n_iters, int use_graph) { s->impl.iterate(n_iters, use_graph != 0); }
extern "C" int cg_solver_finish(cg_solver *s, float *residue, float *hist, int hist_len)
{
  return s->impl.finish(residue, hist, hist_len);
}
extern "C" void cg_solver_sync(cg_solver *s) { CHECK(hipStreamSynchronize(s->impl.stream())); }
extern "C" void cg_solver_time_kernels(cg_solver *s, int n_iters, float *ms_A, float *ms_B)
{
  s->impl.time_kernels(n_iters, ms_A, ms_B);
}

extern "C" void cg_solver_begin_local(cg_solver *s, const float *b, float *x) { s->impl.begin_local(b, x); }
extern "C" void cg_solver_reduce(cg_solver *s, int which, double *out) { s->impl.reduce_partials(which, out); }
extern "C" void cg_solver_bind_sums(cg_solver *s, const double *pAp, const double *rr)
{
  s->impl.bind_sums(pAp, 1, rr, 1);
}
extern "C" void cg_solver_bind_partials(cg_solver *s, double *pAp_parts, double *rr_parts)
{
  s->impl.bind_partials(pAp_parts, rr_parts);
}
extern "C" void cg_solver_bind_partials_n(cg_solver *s, double *pAp_parts, double *rr_parts, int n)
{
  CG_REQUIRE(n >= s->impl.blocks() && n <= 2 * CG_MAX_BLOCKS, "addend count must cover this solver's workgroups");
  s->impl.bind_partials(pAp_parts, rr_parts);
  s->impl.bind_sums(pAp_parts, n, rr_parts, n);
}
extern "C" void cg_solver_init_state(cg_solver *s, double rel, int lagged, int maxit)
{
  s->impl.init_state(rel, lagged, maxit);
}
extern "C" void cg_solver_add_iterations(cg_solver *s, int n) { s->impl.add_iterations(n); }
extern "C" void cg_solver_kernel_A(cg_solver *s) { s->impl.launch_A(); }
extern "C" void cg_solver_kernel_B(cg_solver *s) { s->impl.launch_B(false); } // x complete after every call
extern "C" void cg_solver_edge_copy(cg_solver *s) { s->impl.edge_copy(s->impl.stream()); }
extern "C" void cg_solver_edge_update(cg_solver *s) { s->impl.edge_update(s->impl.stream()); }
extern "C" float *cg_solver_edge_plane(cg_solver *s, int side) { return s->impl.edge_plane(side); }
extern "C" float *cg_solver_r(cg_solver *s) { return s->impl.r(); }
extern "C" float *cg_solver_p(cg_solver *s, int which) { return s->impl.p(which); }

// ---- direct xGMI path ------------------------------------------------------------------------

extern "C" size_t cg_links_box_bytes(const cg_solver_desc *desc) { return links_box_bytes(Solver::plane_of(to_desc(desc))); }
extern "C" int cg_links_alloc(size_t bytes, void **box) { return links_alloc(bytes, box); }
extern "C" int cg_links_free(void *box) { return links_free(box); }
extern "C" int cg_links_peer_ok(int peer_device) { return links_peer_ok(peer_device); }
extern "C" int cg_links_export(void *box, unsigned char *handle64) { return links_export(box, handle64); }
extern "C" int cg_links_open(const unsigned char *handle64, void **box) { return links_open(handle64, box); }
extern "C" int cg_links_close(void *box) { return links_close(box); }
extern "C" int cg_links_poke(void *box, int slot, unsigned long long value) { return links_poke(box, slot, value); }
extern "C" int cg_links_peek(void *box, int slot, unsigned long long *value) { return links_peek(box, slot, value); }
extern "C" void cg_solver_bind_links(cg_solver *s, int rank, int world, void *const *boxes)
{
  s->impl.bind_links(rank, world, boxes);
}
extern "C" void cg_solver_exchange(cg_solver *s, int which, int force) { s->impl.launch_exchange(which, force != 0); }
extern "C" int cg_solver_links_error(cg_solver *s) { return s->impl.links_error(); }
extern "C" void cg_solver_links_times(cg_solver *s, float *ms_xA, float *ms_xB) { s->impl.links_times(ms_xA, ms_xB); }
