/* conjugate_gradient_gpu.h -- C-ABI of libconjugate_gradient_gpu.so (MI355X / gfx950).
 *
 * Part 1 is the drop-in boundary: exactly the unmangled symbols the reference's shared
 * library exports and its Julia driver binds with dlsym + @ccall
 * (/root/reference/src/main.jl:13-58), with the same argument meaning, return values and
 * error behaviour (runtime errors print "Error: file:line, code: N, reason: ..." to
 * stderr and exit(1), common.h:7-17; non-convergence is not an error).
 * Part 2 are additions that the reference API cannot express (layout choice, fixed
 * iteration counts, device-resident benchmarking, slab decomposition, mixed precision).
 *
 * Pointers are plain float / double pointers.  Device, managed and host pointers are all
 * accepted by part 1 (non-device memory is staged through HBM); part 2 documents what
 * each argument must be.  All functions are synchronous on return unless stated.
 * Single host thread per solver object; one GPU (the current HIP device).
 */
#ifndef CONJUGATE_GRADIENT_GPU_H
#define CONJUGATE_GRADIENT_GPU_H

#include <stddef.h>

#ifdef __cplusplus
extern "C"
{
#endif

  /* ===================== Part 1: the reference's exports ============================ */

  /* conjugate-gradient_gpu.cuh:47, conjugate-gradient_gpu.cu:327-376; main.jl:41-44.
   * Float CG for A x = b on the L^d grid, A = discrete Laplacian with zero Dirichlet
   * boundary.  r = b (x is assumed 0 and is accumulated into), reltol = 1e-6*|b|, at most
   * 200 iterations, the loop tests the residual of the previous iteration and that lagged
   * value is returned.  Prints "i,residue" per iteration to stdout as the reference does
   * (after the solve; set CG_GPU_QUIET=1 to suppress). */
  float conjugate_gradient_gpu(float *b, float *x, int L, int d);

  /* conjugate-gradient_gpu.cuh:26, conjugate-gradient_gpu.cu:239-255; main.jl:28-32.
   * sum_i v[i]*w[i] over N floats.  Deterministic (two-stage, no atomics). */
  float inner_product_gpu(float *v, float *w, unsigned int N);

  /* conjugate-gradient_gpu.cuh:16-18, conjugate-gradient_gpu.cu:128-148; main.jl:34-37.
   * Host-callable.  Neighbour of `ind` at amount in {-1,0,1} along `direction`; N if the
   * step leaves the grid.  index_mode 0 and 1: lexicographic (identical, as in the
   * reference, where the argument is ignored); 2: Morton (interleave.cuh:43-67). */
  unsigned int neighbour_index_gpu(unsigned int ind, int direction, int amount, unsigned int L, unsigned int d,
                                   unsigned int N, int index_mode);

  /* julia.cu:16-30; main.jl:47-51.  Microseconds of one stencil launch of
   * nblocks x threads_per_block threads (one grid point per thread) plus synchronisation,
   * on device-resident random data. */
  float strong_scaling(int nblocks, int threads_per_block, int N, int L, int d);

  /* julia.cu:32-42; main.jl:54-58.  Microseconds of one CPU (double) stencil. */
  float time_cpu_laplace(int N, int L, int d);

  /* julia.cu:44-53; main.jl:16-19.  A[i] += B[i] for i < blocks*threads (no bounds check,
   * as in the reference).  Asynchronous. */
  void add_jl(float *A, float *B, int blocks, int threads);

  /* julia.cu:55-60; main.jl:22-25.  ddf = A u for the first min(N, blocks*threads) points;
   * dx is ignored (the discrete operator carries no dx, conjugate-gradient_gpu.cu:175).
   * u may have N or N+1 entries: out-of-grid neighbours are literal zeros. */
  void laplace_gpu_jl(float *ddf, float *u, float dx, int d, int L, int N, unsigned int index_mode,
                      unsigned int blocks, unsigned int threads);

  /* ===================== Part 2: additions =========================================== */

  /* cuda_allocate_field (conjugate-gradient_gpu.cuh:10, .cu:34-42) with C linkage: N+1
   * zeroed floats that host and device can both dereference; slot N is the boundary. */
  float *cg_allocate_field(int N);
  void cg_free_field(float *field);

  /* conjugate_gradient_gpu with its constants exposed.  index_mode as above (2 = vectors in
   * Morton order); maxit < 0: no cap; rel: relative tolerance; lagged 1: reference GPU stop
   * rule, 0: stop on the fresh residual (the CPU preconditioner's rule,
   * conjugate-gradient_cpu.cu:299,317); verbose: replay the reference's per-iteration
   * printf.  iters_out / hist (hist_len floats, the printed residues) may be NULL. */
  float conjugate_gradient_gpu_ex(const float *b, float *x, int L, int d, int index_mode, int maxit, double rel,
                                  int lagged, int verbose, int *iters_out, float *hist, int hist_len);

  /* conjugate_gradient_gpu / conjugate_gradient_gpu_ex keep their solver (r, two p buffers, the captured
   * iteration graph) between calls on the same grid while it holds at most CG_CACHE_KEEP_MB (environment,
   * default 1024) MiB of device memory; a larger one is freed before the call returns, as the reference
   * frees r, Ap, p (conjugate-gradient_gpu.cu:371-373).  cg_release_cache() frees a kept solver now. */
  void cg_release_cache(void);

  /* Mixed precision (conjugate-gradient_cpu.cu:334-392 with the float GPU solver as the
   * inner solve, the scheme the reference's readme.md:81 describes): outer CG in double
   * to 1e-8*|b| (cap 1000), M^-1 r by conjugate_gradient_gpu_ex(rel 1e-3, fresh-residual
   * rule).  b, x: N doubles, host or device; x is the initial guess and the result.
   * Returns the number of outer iterations; inner_iters (inner_len ints, may be NULL) receives
   * the inner step counts of the preconditioner calls in order, followed by -1 if there is room. */
  int preconditioned_cg_gpu(const double *b, double *x, int L, int d, int *inner_iters, int inner_len,
                            int verbose);

  /* Largest and smallest eigenvalue of the L^d Laplace operator by power iteration "together with CG"
   * (item 3 of the assignment the reference implements, tasks.md:10-24; the reference itself has no
   * such function).  lambda_max: q <- A q / |A q|; lambda_min: inverse iteration, q <- A^-1 q / |A^-1 q|
   * with the float GPU CG (to 1e-6) as A^-1 and the Rayleigh quotient of the iterate as the estimate.
   * Both start from u01(seed) vectors and stop after max_steps steps or when the estimate changes by
   * less than tol (relative).  Either result pointer may be NULL (that half is skipped); the steps_*
   * and cg_iterations (inner CG iterations, summed) pointers may be NULL.  Returns 0. */
  int cg_eigen_extremes_gpu(int L, int d, int max_steps, double tol, unsigned seed, double *lambda_max,
                            int *steps_max, double *lambda_min, int *steps_min, int *cg_iterations);

  /* Two-grid correction scheme (the "Multigrid" item of the same assignment, tasks.md:32-36; no such
   * function in the reference): u <- TG(u) until |f - A u| <= rel * |f| or max_cycles cycles.  Per cycle:
   * nu1 damped-Jacobi steps u += omega/(2d) (f - A u), full-weighting restriction of the residual to the
   * grid of Lc = (L-1)/2 points per axis, coarse solve A_c e_c = 4 R r by the float GPU CG (to coarse_rel,
   * relative to its right-hand side), d-linear prolongation, nu2 Jacobi steps.  L must be odd, d <= 8;
   * omega <= 0 selects 2d/(2d+1).  f, u: device pointers, N floats; u is the initial guess and the result.
   * hist (hist_len floats, may be NULL) receives |f - A u| before the first and after every cycle;
   * coarse_iterations (may be NULL) the CG iterations of all coarse solves.  Returns the cycles done. */
  int cg_two_grid_gpu(const float *f, float *u, int L, int d, int nu1, int nu2, double omega, int max_cycles,
                      double rel, double coarse_rel, float *hist, int hist_len, int *coarse_iterations);
  /* The same scheme with the coarse problem handed to the scheme itself: V-cycles over `levels` grids
   * (0: as many as the grid allows; coarsening needs an odd number of points per axis on every level but
   * the last, e.g. L = 2^k - 1), CG only on the coarsest.  levels_used (may be NULL): the grids in use. */
  int cg_multigrid_gpu(const float *f, float *u, int L, int d, int levels, int nu1, int nu2, double omega,
                       int max_cycles, double rel, double coarse_rel, float *hist, int hist_len, int *levels_used,
                       int *coarse_iterations);
  /* CG preconditioned by one such cycle from a zero guess (nu steps before and after the coarse correction,
   * tasks.md:36): u is the initial guess and the result, iterations until |f - A u| <= rel * |f| or
   * max_iters.  hist: the residual norm before the first and after every iteration.  Returns the
   * iterations done. */
  int cg_two_grid_pcg_gpu(const float *f, float *u, int L, int d, int nu, double omega, int max_iters, double rel,
                          double coarse_rel, float *hist, int hist_len, int *coarse_iterations);

  /* y = A u on device pointers, asynchronous on `stream` (a hipStream_t, may be NULL). */
  void cg_laplace_device(float *ddf, const float *u, int d, int L, int index_mode, void *stream);

  /* dst[i] = u01(seed, i): counter-based uniform [0,1) generator, identical on host
   * (cg_u01_host) and device.  dst: device pointer. */
  void cg_fill_uniform(float *dst, unsigned long long n, unsigned int seed);
  float cg_u01_host(unsigned int seed, unsigned long long i);

  /* strong_scaling (julia.cu:16-30) on the device clock: mean microseconds of one stencil launch of
   * the caller's shape over `reps` back-to-back launches, HIP events, one untimed launch first.
   * cg_laplace_time_events: the same for the tiled whole-grid stencil (what laplace_gpu_jl runs
   * when the launch covers the grid). */
  float cg_strong_scaling_events(int nblocks, int threads_per_block, int N, int L, int d, int reps);
  float cg_laplace_time_events(int L, int d, int index_mode, int reps);

  /* --- an entry point that allocates nothing ------------------------------------------------
   * No counterpart in the reference.  For callers that cannot allocate device memory: code under
   * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  The same two-stage
   * dot product as inner_product_gpu (same kernels, same launch shape, same bits), but: v, w must
   * be device pointers (nothing is staged), the result (float) goes to result_dev[0] in device
   * memory, the call is asynchronous on `stream` (a hipStream_t, NULL = the default stream) and
   * neither allocates nor frees nor synchronises.  The per-workgroup partial sums live in a static
   * 32 KiB device array that every GPU the library is used on holds until the process ends; all
   * calls on one GPU share it, so calls that overlap in time on one GPU (different streams,
   * several host threads) race -- keep them on one stream or order them.  n must be in
   * [1, CG_NOALLOC_MAX_N]; outside that, or with a null pointer, an error is printed to stderr and
   * a non-zero hipError_t value returned (0 = success).  cg_noalloc_prepare() looks the array's
   * address up for the current device ahead of time (otherwise the first call on a device does);
   * call it before a stream capture starts. */
#define CG_NOALLOC_MAX_N 4294967295ull
  int cg_noalloc_prepare(void);
  int cg_inner_product_noalloc(const float *v, const float *w, unsigned long long n, float *result_dev, void *stream);

  /* --- solver object: the fused two-kernel iteration on device-resident vectors -------- */

  typedef struct cg_solver cg_solver;

  typedef struct cg_solver_desc
  {
    int d;                 /* dimensions, 1..32 */
    unsigned extents[32];  /* points per axis, axis 0 contiguous; last axis = slab axis */
    int index_mode;        /* 0,1 lexicographic; 2 Morton */
    int has_lo, has_hi;    /* a neighbouring slab exists below / above the last axis */
    void *stream;          /* hipStream_t to enqueue on; NULL = private stream */
    float *r_buf;          /* optional caller-owned buffers of cg_solver_buffer_len() */
    float *p0_buf;         /*   floats (16-byte aligned device memory); NULL = allocate. */
    float *p1_buf;         /*   Layout: [pad | lo ghost plane | N owned | hi ghost plane] */
    int engine;            /* 0 auto, 1 rows, 2 scalar, 3 planes (3-D), 4 Morton quads */
  } cg_solver_desc;

  size_t cg_solver_buffer_len(const cg_solver_desc *desc);   /* floats per buffer */
  size_t cg_solver_ghost_offset(const cg_solver_desc *desc); /* floats before the first owned element */
  size_t cg_solver_plane(const cg_solver_desc *desc);        /* floats in one ghost plane */

  cg_solver *cg_solver_create(const cg_solver_desc *desc);
  void cg_solver_destroy(cg_solver *s);
  int cg_solver_engine(const cg_solver *s);
  int cg_solver_blocks(const cg_solver *s);                  /* partial sums per dot product: the larger of the two below */
  int cg_solver_blocks_of(const cg_solver *s, int which);    /* partials kernel A (0) / kernel B (1) writes per launch */
  /* name of the kernel template the iterations run ("k_rows", "k_cols", "k_planes", "k_morton4",
     "k_morton_tile", "k_scalar", "k_small"), as it appears in a profiler's kernel trace; static storage */
  const char *cg_solver_kernel_name(const cg_solver *s);
  /* 1 if the iterations launched by cg_solver_iterate update x in pairs: kernel B of an even iteration
     leaves x alone and the next one applies both updates (same bits as updating every time); such a solver
     streams 30 instead of 32 bytes per point and iteration.  CG_XDEFER=0 turns it off. */
  int cg_solver_paired_x(const cg_solver *s);

  /* whole solve, asynchronous until cg_solver_finish.  b, x: device pointers, N floats. */
  void cg_solver_begin(cg_solver *s, const float *b, float *x, double rel, int lagged);
  void cg_solver_iterate(cg_solver *s, int n_iters, int use_graph);
  int cg_solver_finish(cg_solver *s, float *residue, float *hist, int hist_len);
  void cg_solver_sync(cg_solver *s);

  /* average device milliseconds of kernel A and kernel B over n further iterations
   * (HIP events on the solver's stream, plain launches) */
  void cg_solver_time_kernels(cg_solver *s, int n_iters, float *ms_A, float *ms_B);

  /* step interface for slab decomposition: the caller all-reduces the two dot products
   * between the steps (device doubles) and exchanges r's boundary planes.  All calls are
   * asynchronous on the solver's stream. */
  void cg_solver_begin_local(cg_solver *s, const float *b, float *x);   /* r = b, p = 0, local r.r partials */
  void cg_solver_reduce(cg_solver *s, int which, double *out);          /* out[0] = local p.Ap (0) or r.r (1) */
  void cg_solver_bind_sums(cg_solver *s, const double *pAp, const double *rr); /* all-reduced values */
  /* alternative to reduce + bind_sums: the kernels write their cg_solver_blocks() per-workgroup
   * partials into caller-owned device arrays and read the sums back from them, so the caller
   * all-reduces the arrays element-wise and no separate reduction launch is needed */
  void cg_solver_bind_partials(cg_solver *s, double *pAp_parts, double *rr_parts);
  void cg_solver_init_state(cg_solver *s, double rel, int lagged, int maxit);
  void cg_solver_add_iterations(cg_solver *s, int n);
  void cg_solver_kernel_A(cg_solver *s);
  void cg_solver_kernel_B(cg_solver *s);
  /* r's boundary planes ahead of kernel B, for callers that want the halo exchange to run beside kernel B
   * instead of after it.  cg_solver_edge_copy: two internal planes take r's first / last owned hyperplane
   * (call it while r is as the previous kernel B left it: before, or beside, kernel A).
   * cg_solver_edge_update (after the sum of p.Ap is bound, i.e. after kernel A's dot product has been
   * exchanged): the planes become what kernel B of the coming iteration will leave in r there, bit for bit;
   * it never touches r.  cg_solver_edge_plane(s, 0 / 1): the plane for the rank below / above (device
   * pointer, cg_solver_plane() floats) -- send it instead of r's own plane.  Asynchronous on the solver's
   * stream.  Lexicographic slabs only. */
  void cg_solver_edge_copy(cg_solver *s);
  void cg_solver_edge_update(cg_solver *s);
  float *cg_solver_edge_plane(cg_solver *s, int side);
  /* first owned element of r / p.  (A stand-alone grid whose axis 0 is not a multiple of 4 points may be
   * kept with rows padded to 4 floats inside the solver: cg_solver_plane() / cg_solver_buffer_len() do
   * not describe that internal layout, and x is only complete in the caller's array after
   * cg_solver_finish.  CG_PAD_ROWS=0 switches the padding off.  Likewise a Morton-ordered grid of at
   * least 2^19 points is kept in lexicographic order inside the solver: r and p are then in that order,
   * and x is in the caller's (Morton) order only after cg_solver_finish; CG_MORTON_RELAYOUT=0 switches
   * that off.) */
  float *cg_solver_r(cg_solver *s);
  float *cg_solver_p(cg_solver *s, int which);

  /* alternative to cg_solver_bind_partials for ranks whose solvers have different cg_solver_blocks():
   * the kernels write their own cg_solver_blocks() partials but read n addends (n = the largest
   * cg_solver_blocks() over the ranks, at most 4096), so that all ranks can all-reduce arrays of one
   * length.  Kernel A writes pAp_parts[0 .. cg_solver_blocks_of(s, 0)), kernel B (and cg_solver_begin_local)
   * rr_parts[0 .. cg_solver_blocks_of(s, 1)); the caller keeps the rest of each array, up to n, zero before
   * every reduction (an in-place all-reduce leaves the previous sum there) */
  void cg_solver_bind_partials_n(cg_solver *s, double *pAp_parts, double *rr_parts, int n);

  /* --- multi-GPU on RCCL (over xGMI inside a node): the halo exchange and the two dot-product
   * exchanges of the slab decomposition, issued by the library itself (no counterpart in the
   * reference, which is single-GPU).  One rank per GPU, rank r owns slab r along the slowest axis
   * (has_lo = r > 0, has_hi = r + 1 < world in its cg_solver_desc).  Rank 0 makes a unique id
   * (cg_rccl_unique_id) and hands the 128 bytes to the other ranks over any channel; every rank, with
   * its GPU current, creates its communicator (cg_rccl_comm_create, collective) -- or brings an
   * ncclComm_t it already has -- and binds it (cg_solver_bind_rccl, collective; it splits a second
   * communicator off for the boundary planes).  From then on cg_solver_begin / cg_solver_iterate /
   * cg_solver_finish run the multi-rank solve: per iteration kernel A, ncclAllGather of one double per
   * rank (its p.Ap), kernel B, ncclAllGather of r.r; beside kernel B, on a second stream, the boundary
   * planes r will have after the iteration are computed ahead (cg_solver_edge_update) and travel by a
   * grouped ncclSend / ncclRecv into the neighbours' ghost planes -- all replayed from one hipGraph, no
   * host work inside the loop.  Every rank passes the same rel / lagged / iteration counts.  At most
   * 1024 ranks.  All return an ncclResult_t value, 0 = success (a message goes to stderr otherwise);
   * cg_solver_bind_rccl(s, NULL, 0, 1) unbinds.
   * cg_solver_bind_rccl_peers names the ranks that own the slab below / above explicitly (-1: none; must
   * agree with has_lo / has_hi) for rank layouts other than "slab r on rank r"; a rank may be its own
   * neighbour on both sides (one rank, has_lo = has_hi = 1: every ghost plane receives the rank's own
   * boundary plane on that side, which is how this path is tested on one GPU).
   * cg_solver_rccl_ranks: the ranks of the bound communicator as RCCL counts them, 0 if none is bound. */
#define CG_RCCL_ID_BYTES 128
  int cg_rccl_unique_id(unsigned char *id128);
  int cg_rccl_comm_create(const unsigned char *id128, int rank, int world, void **comm);
  int cg_rccl_comm_destroy(void *comm);
  int cg_solver_bind_rccl(cg_solver *s, void *comm, int rank, int world);
  int cg_solver_bind_rccl_peers(cg_solver *s, void *comm, int rank, int world, int lo_rank, int hi_rank);
  int cg_solver_rccl_ranks(cg_solver *s);

  /* --- direct xGMI path: slab neighbours write into each other's HBM, no collective library ---
   * (no counterpart in the reference, which is single-GPU).  Every rank allocates a box,
   * passes its handle to the other ranks of the node (any channel: the 64 bytes are plain
   * data), opens theirs and binds all `world` boxes to its solver.  From then on
   * cg_solver_begin / cg_solver_iterate / cg_solver_finish run the multi-rank solve by
   * themselves: after kernel A and kernel B a small launch publishes this rank's dot product
   * (and, after B, r's boundary planes) into the other ranks' boxes and waits for theirs, so
   * an iteration is four launches, replayed from a hipGraph, and no host work.  The solver's
   * has_lo / has_hi must match its rank.  cg_links_* return a hipError_t value, 0 = success
   * (they never terminate the process, so a caller can fall back to collectives). */
  size_t cg_links_box_bytes(const cg_solver_desc *desc);
  int cg_links_alloc(size_t bytes, void **box);            /* device memory no cache keeps, zeroed */
  int cg_links_free(void *box);
  int cg_links_peer_ok(int peer_device);                   /* 1 if the current device can write that device's memory (enables it) */
  int cg_links_export(void *box, unsigned char *handle64); /* for the other processes of the node */
  int cg_links_open(const unsigned char *handle64, void **box);
  int cg_links_close(void *box);
  /* host-side write / read (copy engines) of probe word `slot` (0..63) at the start of a box's staging
   * area: lets the ranks check every mapping before the first kernel stores through it */
  int cg_links_poke(void *box, int slot, unsigned long long value);
  int cg_links_peek(void *box, int slot, unsigned long long *value);
  /* boxes[0..world): every rank's box as mapped in this process; boxes[rank] is this rank's own.
   * Binding restarts the exchange sequence numbers at zero on every rank, so the boxes must be
   * freshly allocated (all zero) and all ranks must bind before any of them begins a solve.
   * world = 0 unbinds (waits for the solver's stream first). */
  void cg_solver_bind_links(cg_solver *s, int rank, int world, void *const *boxes);
  void cg_solver_exchange(cg_solver *s, int which, int force); /* step interface: enqueue one exchange */
  /* 1 if a wait for another rank ever ran out of time (CG_LINKS_TIMEOUT_MS, default 10 s): the
   * sums were then poisoned with NaN, every rank stopped, and the result is invalid.  Synchronises. */
  int cg_solver_links_error(cg_solver *s);
  /* average device milliseconds of the two exchange launches during the last cg_solver_time_kernels */
  void cg_solver_links_times(cg_solver *s, float *ms_xA, float *ms_xB);

#ifdef __cplusplus
}
#endif
#endif
