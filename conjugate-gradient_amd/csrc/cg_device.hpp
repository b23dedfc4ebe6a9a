// cg_device.hpp -- the CDNA4 (gfx950) kernels of the fused Laplace-CG iteration.
//
// One CG iteration of the reference (conjugate-gradient_gpu.cu:349-370) is five
// kernels, two blocking dot products and three axpy passes.  Here it is two kernels:
//
//   kernel A (MODE_A):  p_new = r + beta*p_old            (reference muladd3, :302-308,:363)
//                       partial sums of  p_new . (A p_new) (laplace_gpu :164-178 + the dot :355)
//   kernel B (MODE_B):  x += alpha*p ; r -= alpha*(A p)    (muladd :293-299,:356-357)
//                       partial sums of  r . r             (:359)
//
// A p is never stored: kernel B recomputes the stencil from p, which costs no HBM
// traffic beyond the halo and saves one store and one load of a whole vector.  HBM
// words per grid point per iteration: A reads r,p_old and writes p_new (3), B reads
// x,p,r and writes x,r (5).  p is double-buffered (p0/p1, selected by iteration
// parity) because kernel A recomputes r + beta*p_old at halo points that belong to
// other workgroups.
//
// Paired x update: both p buffers are intact while kernel B runs, so x need not be touched
// every iteration.  Of two consecutive launches of kernel B the first skips x (3 words), the
// second reads both p and applies x = fma(alpha_k, p_k, x), x = fma(alpha_k+1, p_k+1, x) in
// that order (6 words): the bits of two single updates, 4.5 instead of 5 words per iteration
// on average.  The host pairs the launches of one request (the last launch of a request
// never skips), so an update can only be left pending by a stop on the tolerance between the
// two launches of a pair; a flush launch applies it whenever the host gets control back
// (CgState::xdef_k).
//
// alpha, beta, the iteration counter and the stop test live on the device
// (CgState).  Dot products are deterministic: every workgroup writes one double
// partial; the consumer kernel's wave 0 sums the partials in a fixed order, so all
// workgroups derive bit-identical scalars and no atomics are used
// (the reference accumulates with float atomicAdd, :227-228).
//
// Two engines implement the same three modes:
//   k_rows   -- lexicographic layout, 4 points per lane along axis 0 (16-byte
//               accesses), workgroup marches along axis 1 keeping rows y-1,y,y+1 in
//               registers, x-neighbours through cross-lane shifts.
//   k_scalar -- one point per lane; any extent, any d, and the Morton layout.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "index_map.hpp"

namespace cgamd
{

// Measurement-only switches (tools/rowsbench.hip compiles the rows kernels with parts removed
// to price them); always 0 in the library.
#ifndef CG_DBG
#define CG_DBG 0
#endif
// -DCG_DBG=64: the one-point-per-lane kernels of the first and the last workgroup leave time stamps (100 MHz
// clock, after waiting for their outstanding memory operations) at entry, after the prologue, after the body and
// at the end; cg_dbg_times() copies them out (tools/dbg/ktimes.py).  Not compiled otherwise.
#if CG_DBG & 64
__device__ unsigned long long cg_dbg_t[64];
#define CG_T(i)                                                                                          \
  do                                                                                                     \
  {                                                                                                      \
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                            \
    {                                                                                                    \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                        \
      cg_dbg_t[(i) + (blockIdx.x == 0 ? 0 : 16) + (MODE == MODE_B ? 32 : 0)] = wall_clock64();           \
    }                                                                                                    \
  } while (0)
#else
#define CG_T(i)
#endif

constexpr int CG_MAXD = 32;        // scalar engine
constexpr int CG_ROWS_MAXD = 14;   // rows engine: axes 2..15 (rows of >= 4 points and N < 2^32 allow no more)
constexpr int CG_MAX_BLOCKS = 2048;
constexpr int CG_RCCL_MAX_RANKS = 1024; // slots of the gathered per-rank dot products
constexpr int CG_HIST = 1024;      // ring of r.r values, indexed by iteration mod CG_HIST

enum
{
  MODE_STENCIL = 0,
  MODE_A = 1,
  MODE_B = 2
};
enum
{
  LAYOUT_LEX = 0,
  LAYOUT_MORTON = 2
};

// Iteration control block in device memory.  Kernel A only reads side A and writes
// side B; kernel B only reads side B and writes side A; so no launch reads a word
// that the same launch writes.
struct CgState
{
  int itA;   // iteration about to start
  int doneA; // stop test has fired (sticky)
  int itB;
  int doneB;
  int activeB; // kernel A ran its body for iteration itB
  int maxit;   // iterations >= maxit pause (not sticky)
  int lagged;  // 1: test sqrt(rr_{k-1}) as conjugate-gradient_gpu.cu:349,366; 0: test sqrt(rr_k) as cpu.cu:299,317
  float reltol;
  float alpha, beta;
  float pAp;
  // Paired x update: a kernel B launched with VecArgs::xdefer leaves out x += alpha_k p_k (no read, no write
  // of x) and records xdef_k = k; kernel B of iteration k + 1 (launched without xdefer: the host never sets it
  // twice in a row) then applies both updates, in order, from the two p buffers.  A flush launch applies a
  // pending update where the host needs x and records the iteration count it did so at.  Kernel B of
  // iteration k reads alpha_hist[(k - 1) & 1] and writes alpha_hist[k & 1].
  int xdef_k;       // last iteration whose x update was left out (-2: none)
  int xflushed_at;  // iteration count at which a flush applied a pending update (-1: never)
  float alpha_hist[2];
  // r.r of the last two iterations, rr_pp[k & 1] = r.r entering iteration k: the copy of rr_hist's newest two
  // entries that the prologues read.  Both slots are loaded unconditionally together with everything else the
  // prologue needs (one trip to memory) and selected once the iteration number has arrived; an address that
  // depends on it (rr_hist[k % CG_HIST]) would cost a second trip, about 0.6 us per kernel.
  float rr_pp[2];
  float rr_hist[CG_HIST];
};

struct VecArgs
{
  const float *u; // MODE_STENCIL input
  float *out;     // MODE_STENCIL output
  float *x;
  float *r;
  float *p0;
  float *p1;
  CgState *st;
  const double *sumA; // p.Ap as nA addends (per-block partials, or 1 all-reduced value)
  int nA;
  const double *sumB; // r.r as nB addends
  int nB;
  double *part_out; // this launch's per-block partials
  // Direct xGMI path, kernel B only: r's first / last owned hyperplane also goes straight into the staging
  // area of the rank below / above (null: nothing to push), so that the exchange launch after B only has to
  // publish stamps and collect.  push_seq: exchanges of this kind completed so far (device memory); its
  // parity selects the staging buffer, push_stride floats apart.
  float *push_lo, *push_hi;
  const unsigned long long *push_seq;
  unsigned long long push_stride;
  int xdefer; // this kernel B leaves x alone; the next one applies both updates (see CgState::xdef_k)
  // RCCL path: the last workgroup of the launch to finish also adds all of the launch's partials up (fixed
  // order) into rank_sum[0], so that one double per rank travels; ticket counts the finished workgroups
  // (device memory, zero between launches).  Null: nobody does.
  double *rank_sum;
  unsigned int *ticket;
};

// staging buffers of this launch: the exchange that follows kernel B is number *push_seq + 1
__device__ __forceinline__ void push_targets(const VecArgs &a, float *&lo, float *&hi)
{
  lo = nullptr;
  hi = nullptr;
  if (a.push_seq)
  {
    const unsigned long long par = (a.push_seq[0] + 1ull) & 1ull;
    lo = a.push_lo ? a.push_lo + par * a.push_stride : nullptr;
    hi = a.push_hi ? a.push_hi + par * a.push_stride : nullptr;
  }
}

struct RowsGeom
{
  int d;
  unsigned n0, n1;
  unsigned s0;     // floats between consecutive rows of axis 1: n0, or n0 rounded up to 4 (padded rows)
  unsigned nxc;    // columns per row of axis 0
  unsigned bpc;    // segments per column
  unsigned nitems; // columns * bpc
  unsigned keep_y; // POL_PART: segments that end at or below this row keep r in the cache, the others stream it
  unsigned nk[CG_ROWS_MAXD];
  unsigned long long sk[CG_ROWS_MAXD];
  int has_lo, has_hi;
  // packed mode (rows shorter than a wave's 256 points): `sub` lanes hold one row, a workgroup
  // holds `cpg` columns side by side; `ncols` columns in all
  unsigned sub, cpg, ncols;
  unsigned long long sy; // packed mode: stride of the last axis, along which packed columns march
  // blocked columns (k_cols, cg_cols.hpp): a lane owns R columns that are neighbours along axis d-2
  int R;                 // 0: one column per lane (k_rows)
  unsigned nb;           // extent of axis d-2
  unsigned long long sb; // its stride
  int stream;            // k_cols: single-use traffic carries the streaming hint
  int xl;                // k_cols: a row takes exactly n0/4 lanes (x neighbours through LDS too)
  int lds_a, lds_b;      // k_cols, kernels A (and the plain stencil) / B: axis-1 neighbours inside a workgroup travel through LDS
  unsigned xcd;          // k_cols, tiled slots: patches per XCD along axis 1 (0: patches are dealt to the XCDs in turn)
  unsigned t2;           // k_cols in 4-D through LDS: slots of a workgroup along the blocks of axis d-2 (1: a strip along axis 1)
};

struct ScalarGeom
{
  int d;
  unsigned long long N;
  unsigned n[CG_MAXD];
  unsigned long long stride[CG_MAXD];
  unsigned mmask[CG_MAXD];
  int has_lo, has_hi;
};

// ---- reductions ----------------------------------------------------------------------

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    v += __shfl_xor(v, o);
  return v;
}

// Sum of src[0..n) handed to every thread of the workgroup, the same bits for every
// workgroup shape: 256 virtual lanes each sum a strided subset (8 loads in flight at a
// time), four virtual waves are butterfly-reduced, and their totals are added in a fixed
// order.  A workgroup with fewer than four waves lets each wave play several virtual ones.
__device__ __forceinline__ double block_read_sum(const double *__restrict__ src, int n, double *sh)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  for (int vw = wave; vw < 4; vw += nw)
  {
    const int v = vw * 64 + lane;
    double s = 0;
    for (int i0 = 0; i0 < n; i0 += 2048)
    {
      double t[8];
#pragma unroll
      for (int j = 0; j < 8; j++)
      {
        const int i = i0 + v + 256 * j;
        t[j] = i < n ? src[i] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 8; j++)
        s += t[j];
    }
    s = wave_sum(s);
    if (lane == 0)
      sh[vw] = s;
  }
  __syncthreads();
  const double v = ((sh[0] + sh[1]) + sh[2]) + sh[3];
  __syncthreads();
  return v;
}

__device__ __forceinline__ void block_write_partial(float acc, double *__restrict__ part_out, double *sh)
{
  double v = wave_sum((double)acc);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0)
    sh[w] = v;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    double s = 0;
    for (int i = 0; i < nw; i++)
      s += sh[i];
    part_out[blockIdx.x] = s;
  }
}

// block_read_sum of partials other workgroups of THIS launch wrote: every load is a device-scope atomic, which
// no cache of this workgroup's XCD serves from a stale line
__device__ __forceinline__ double block_read_sum_fresh(const double *src, int n, double *sh)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  for (int vw = wave; vw < 4; vw += nw)
  {
    const int v = vw * 64 + lane;
    double s = 0;
    for (int i0 = 0; i0 < n; i0 += 2048)
    {
      double t[8];
#pragma unroll
      for (int j = 0; j < 8; j++)
      {
        const int i = i0 + v + 256 * j;
        t[j] = i < n ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 8; j++)
        s += t[j];
    }
    s = wave_sum(s);
    if (lane == 0)
      sh[vw] = s;
  }
  __syncthreads();
  const double v = ((sh[0] + sh[1]) + sh[2]) + sh[3];
  __syncthreads();
  return v;
}

// The epilogue of kernels A and B: this workgroup's partial, and -- with a.rank_sum set -- the sum of all the
// launch's partials by whichever workgroup finishes last, in the order block_read_sum adds them (so
// rank_sum[0] has the bits a consumer's block_read_sum over the same partials would produce).
// The partial and the ticket are both device-scope atomics, which are performed where every XCD sees them
// (they bypass the per-XCD L2s), and the partial's store has completed before the ticket is drawn (the
// workgroup-scope fence is a wait for the wave's outstanding memory operations).  A device-scope release
// fence would order them too, but it writes back every dirty line of the XCD's L2 and is issued once per
// workgroup: 1024 of them made kernels A and B 40 us slower each at 8192^2.
__device__ __forceinline__ void block_write_partial(float acc, const VecArgs &a, double *sh)
{
  if (!a.rank_sum) // uniform over the launch
  {
    block_write_partial(acc, a.part_out, sh);
    return;
  }
  double v = wave_sum((double)acc);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0)
    sh[w] = v;
  __syncthreads();
  __shared__ int is_last;
  if (threadIdx.x == 0)
  {
    double s = 0;
    for (int i = 0; i < nw; i++)
      s += sh[i];
    __hip_atomic_store(a.part_out + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    const unsigned t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == gridDim.x - 1u);
  }
  __syncthreads();
  if (!is_last)
    return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const double s = block_read_sum_fresh(a.part_out, (int)gridDim.x, sh);
  if (threadIdx.x == 0)
  {
    a.rank_sum[0] = s;
    __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // for the next launch
  }
}

// ---- iteration control ---------------------------------------------------------------

struct IterCtl
{
  bool active;
  float coef; // beta for kernel A, alpha for kernel B
  int k;
  // kernel B, paired x update: 0 = x += alpha p as usual, 1 = leave x alone this time,
  // 2 = x += alpha_prev p_prev first (the update iteration k - 1 left out), then x += alpha p
  int xm;
  float alpha_prev;
};

__device__ __forceinline__ IterCtl prologue_A(const VecArgs &a, double *sh)
{
  CgState *st = a.st;
  // everything the state has to say is requested before the partial sums are, so that it all travels together
  const int k = st->itA;
  const int done = st->doneA;
  const int lagged = st->lagged, maxit = st->maxit;
  const float reltol = st->reltol;
  const float rr_p0 = st->rr_pp[0], rr_p1 = st->rr_pp[1];
  const float rr = (CG_DBG & 4) ? 1.0f : (float)block_read_sum(a.sumB, a.nB, sh);
  const float rr_prev = (k & 1) ? rr_p0 : rr_p1; // rr_pp[(k - 1) & 1]
  const float rr_test = (lagged && k > 0) ? rr_prev : rr;
  const float residue = (float)sqrt((double)rr_test);
  const bool conv = done || !(residue > reltol);
  IterCtl c;
  c.k = k;
  c.active = !conv && k < maxit;
  c.coef = k > 0 ? rr / rr_prev : 0.0f;
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    st->itB = k;
    st->doneB = conv;
    st->activeB = c.active;
    st->rr_hist[k % CG_HIST] = rr;
    st->rr_pp[k & 1] = rr;
    st->beta = c.coef;
  }
  return c;
}

// XD: this kernel implements the paired x update
template <bool XD = false>
__device__ __forceinline__ IterCtl prologue_B(const VecArgs &a, double *sh)
{
  CgState *st = a.st;
  IterCtl c;
  // as in prologue_A: all of the state in flight before the partial sums
  c.k = st->itB;
  c.active = st->activeB != 0;
  const int done = st->doneB;
  const float rr_p0 = st->rr_pp[0], rr_p1 = st->rr_pp[1];
  const int xdef_k = XD ? st->xdef_k : 0, xflushed_at = XD ? st->xflushed_at : 0;
  const float ah0 = XD ? st->alpha_hist[0] : 0.0f, ah1 = XD ? st->alpha_hist[1] : 0.0f;
  const float pAp = (CG_DBG & 4) ? 1e9f : (float)block_read_sum(a.sumA, a.nA, sh);
  c.coef = ((c.k & 1) ? rr_p1 : rr_p0) / pAp; // rr_pp[k & 1], what prologue_A of this iteration stored
  c.xm = 0;
  c.alpha_prev = 0.0f;
  if constexpr (XD)
  {
    // a launch that may leave x alone only writes xdef_k, every other launch only reads it
    if (a.xdefer)
      c.xm = 1;
    else if (xdef_k == c.k - 1 && xflushed_at != c.k)
    {
      c.xm = 2;
      c.alpha_prev = ((c.k - 1) & 1) ? ah1 : ah0;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    st->itA = c.active ? c.k + 1 : c.k;
    st->doneA = done;
    if (c.active)
    {
      st->alpha = c.coef;
      st->pAp = pAp;
      if constexpr (XD)
      {
        st->alpha_hist[c.k & 1] = c.coef;
        if (c.xm == 1)
          st->xdef_k = c.k;
      }
    }
  }
  return c;
}

// ---- cache policy ------------------------------------------------------------------------
//
// MI355X has a 256 MiB memory-side Infinity Cache.  Loads and stores issued with the
// non-temporal hint stream past it; everything else allocates in it.  POL_KEEP_R issues
// every stream of the iteration non-temporally except the loads of r: r is touched three
// times per iteration (read and written by kernel B, read by kernel A), so a grid whose r
// fits the cache re-reads and re-writes it on-die and HBM carries 5 instead of 8 words per
// point.  POL_PLAIN uses ordinary accesses throughout.
enum
{
  POL_PLAIN = 0,
  POL_KEEP_R = 1,
  // as POL_KEEP_R, but p is loaded through the caches as well (x and all stores still stream): the
  // halo rows and planes of p that neighbouring tiles re-read, and part of what kernel B read last
  // and kernel A reads first, are then served on-die.  The planes engine's default (3-D,
  // 1024 x 1024 x 128: kernel A 342 -> 316 us, kernel B 447 -> 456 us).
  POL_KEEP_RP = 2,
  // everything streams, r included: for grids whose vectors exceed the cache several times over, where keeping
  // r resident buys nothing (rows engine only)
  POL_STREAM = 3,
  // Vectors larger than the cache: a cyclic sweep over more bytes than the cache holds finds nothing of the last
  // sweep in it, so only a cache-sized part of r is kept.  Work items of that part run under POL_KEEP_R, the
  // others under POL_STREAM (1-D and 2-D rows).
  POL_PART = 4
};

typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ float4 load4(const float *p)
{
  if constexpr (NT)
  {
    const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
    return make_float4(t.x, t.y, t.z, t.w);
  }
  else
    return *reinterpret_cast<const float4 *>(p);
}

template <bool NT>
__device__ __forceinline__ float load1(const float *p)
{
  if constexpr (NT)
    return __builtin_nontemporal_load(p);
  else
    return *p;
}

template <bool NT>
__device__ __forceinline__ void store4(float *p, float4 v)
{
  if constexpr (NT)
  {
    const v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(p));
  }
  else
    *reinterpret_cast<float4 *>(p) = v;
}

// ---- the stencil input of each mode ----------------------------------------------------

template <int MODE, int POL>
struct Src
{
  // p streams past the cache under POL_KEEP_R (in every mode but the plain stencil, whose
  // input is re-read by neighbouring rows of other workgroups); r is what the cache keeps
  static constexpr bool NT_P = (POL == POL_KEEP_R || POL == POL_STREAM) && (MODE != MODE_STENCIL);
  static constexpr bool NT_R = POL == POL_STREAM;
  const float *__restrict__ p;
  const float *__restrict__ r;
  float beta;
  __device__ __forceinline__ float ld1(long long i) const
  {
    float v = load1<NT_P>(p + i);
    if constexpr (MODE == MODE_A)
      v = fmaf(beta, v, load1<NT_R>(r + i));
    return v;
  }
  __device__ __forceinline__ float4 ld4(long long i) const
  {
    float4 v = load4<NT_P>(p + i);
    if constexpr (MODE == MODE_A)
    {
      const float4 q = load4<NT_R>(r + i);
      v.x = fmaf(beta, v.x, q.x);
      v.y = fmaf(beta, v.y, q.y);
      v.z = fmaf(beta, v.z, q.z);
      v.w = fmaf(beta, v.w, q.w);
    }
    return v;
  }
};

// one axis' term in the reference's operation order: ((-u_plus) + 2*u_0) - u_minus
// (conjugate-gradient_gpu.cu:173)
__device__ __forceinline__ float axis_term(float up, float c, float um) { return (-up + 2.0f * c) - um; }

// ---- rows engine -----------------------------------------------------------------------
//
// Work item = (column, segment): a column is 4*blockDim.x consecutive points of axis 0 at
// fixed coordinates of axes >= 2; a segment is a run of rows (axis 1) of that column.  The
// host cuts every column into bpc near-equal segments so that the number of items matches
// the number of workgroups the chip holds at once (no second, partial round of workgroups).
// A workgroup marches down its segment with rows y-1, y, y+1 of the stencil input in
// registers; the loads of row y+2 (and of x, r for row y+1) are issued before row y is
// computed, so two rows' worth of bytes per lane are in flight.
// Loads are never predicated: out-of-range rows and lanes are redirected to a valid
// address and the value replaced by zero afterwards.
// GHOST: the vectors own one extra hyperplane below and above the last axis (zero on a
// global face, the neighbour rank's data on a slab face, has_lo/has_hi).
// D: compile-time dimension (1..3), 0 = run-time g.d (>= 4).
// PACK (d >= 3, 4 <= n0 <= 256): a row of axis 0 occupies only n0/4 lanes (rounded up to a power of
// two, the lanes past the end of the row idle), so the lanes of
// a workgroup hold several columns side by side; a row never crosses a wave, its two ends are grid
// faces, and everything that depends on the column (base offset, face masks) is per lane instead
// of per workgroup.  Packed columns march along the LAST axis (g.n1 rows g.sy apart) and are
// numbered along axes 1 .. d-2 (g.nk, g.sk): the neighbours with the longest stride then come from
// registers, and the directly loaded ones are at most one plane of axes 0..d-2 away, where
// neighbouring lanes and workgroups have just put them into the caches.

__device__ __forceinline__ float4 sel4(bool c, float4 v)
{
  return c ? v : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

template <int MODE, bool GHOST, int D, int POL, bool PACK = false>
__global__ void __launch_bounds__(PACK ? 1024 : 256) k_rows( // packed columns: up to 1024 lanes (512 by default)
    const RowsGeom g, const VecArgs a)
{
  static_assert(!PACK || D == 0 || D == 3, "packed rows exist for d >= 3 only");
  __shared__ double sh[16];
  const int tid = threadIdx.x, lane = tid & 63;
  // pointers and beta of the stencil input; a work item loads through Src<MODE, its own policy>
  Src<MODE, POL_PLAIN> src0;
  float *__restrict__ p_new = nullptr;
  float coef = 0;
  // paired x update (kernel B): xm and the p of the previous iteration
  int xm = 0;
  float alpha_prev = 0;
  const float *__restrict__ p_prev = nullptr;
  if constexpr (MODE == MODE_A)
  {
    const IterCtl c = prologue_A(a, sh);
    if (!c.active)
      return;
    src0.p = (c.k & 1) ? a.p1 : a.p0;
    p_new = (c.k & 1) ? a.p0 : a.p1;
    src0.r = a.r;
    src0.beta = c.coef;
  }
  else if constexpr (MODE == MODE_B)
  {
    const IterCtl c = prologue_B<true>(a, sh);
    if (!c.active)
      return;
    src0.p = (c.k & 1) ? a.p0 : a.p1;
    p_prev = (c.k & 1) ? a.p1 : a.p0;
    coef = c.coef;
    xm = c.xm;
    alpha_prev = c.alpha_prev;
  }
  else
  {
    src0.p = a.u;
  }
  const float alpha = coef;
  float acc = 0;
  float *push_lo = nullptr, *push_hi = nullptr;
  if constexpr (MODE == MODE_B && GHOST)
    push_targets(a, push_lo, push_hi);
  const float4 zero4 = make_float4(0, 0, 0, 0);
  const int d = D > 0 ? D : g.d;
  // distance between consecutive rows of a march: packed columns march along the LAST axis
  const long long rs = PACK ? (long long)g.sy : (long long)g.s0;
  // padded rows (n0 % 4 != 0): the last lane of a row owns fewer than four points; the others of
  // its float4 are padding that holds zeros in r, p and x and must stay zero
  const bool tail = (g.n0 & 3u) != 0;
  const long long n1 = g.n1;
  // axis 1 is the ghosted (last) axis only in 2-D
  // packed columns march along the last axis, so (as in 2-D) the ghosted axis is the marching one
  const bool rowghost = GHOST && (PACK || d == 2);

  // one work item under cache policy P (POL itself, or under POL_PART the policy of the item's part of the grid)
  auto work = [&](auto pol_c, const unsigned item) {
    constexpr int P = decltype(pol_c)::value;
    constexpr bool NT = P == POL_KEEP_R || P == POL_STREAM; // x, the stores of r and of p_new
    constexpr bool NT_R = P == POL_STREAM;                  // kernel B's loads of r
    Src<MODE, P> src;
    src.p = src0.p;
    src.r = src0.r;
    src.beta = src0.beta;
    const unsigned seg = item % g.bpc;
    const unsigned col = item / g.bpc;
    unsigned xc = 0, hi, x0;
    bool act;
    if constexpr (PACK)
    {
      const unsigned cs = (unsigned)tid / g.sub;      // column within this workgroup's group
      const unsigned c = col * g.cpg + cs;
      x0 = ((unsigned)tid % g.sub) * 4u;
      // a row holds n0/4 lanes rounded up to a power of two: the lanes past its end idle
      act = c < g.ncols && x0 < g.n0;
      hi = c < g.ncols ? c : g.ncols - 1u;
    }
    else
    {
      xc = col % g.nxc;
      hi = col / g.nxc;
      x0 = (xc * blockDim.x + tid) * 4u;
      act = x0 < g.n0;
    }
    long long base = 0;
    unsigned okm = 0, okp = 0;
    bool st_lo = false, st_hi = false;
    for (int k = 2; k < d; k++)
    {
      const unsigned nk = g.nk[k - 2];
      const unsigned ck = hi % nk;
      hi /= nk;
      base += (long long)ck * (long long)g.sk[k - 2];
      const bool last = !PACK && GHOST && (k == d - 1);
      // ghost planes exist in memory and hold zeros on a global face
      if (ck > 0 || last)
        okm |= 1u << (k - 2);
      if (ck + 1 < nk || last)
        okp |= 1u << (k - 2);
      if (last && ck == 0 && g.has_lo)
        st_lo = true;
      if (last && ck + 1 == nk && g.has_hi)
        st_hi = true;
    }
    const long long y0 = (long long)((unsigned long long)seg * g.n1 / g.bpc);
    const long long y1 = (long long)((unsigned long long)(seg + 1) * g.n1 / g.bpc);
    if (y0 >= y1)
      return;
    const long long col0 = base + (x0 < g.n0 ? x0 : ((g.n0 + 3u) & ~3u) - 4u);
    const bool el = act && lane == 0 && x0 > 0;
    const bool er = act && lane == 63 && x0 + 4 < g.n0;
    const bool edge = (el || er) && !(CG_DBG & 1);
    const long long eoff = (lane == 0) ? -1 : 4;
    const bool rightmost = (lane == 63) || (x0 + 4 >= g.n0);

    // stencil input of row yy (a zero row outside the grid unless it is a ghost row)
    auto ldrow = [&](long long yy) -> float4 {
      const bool ok = (rowghost || (yy >= 0 && yy < n1)) && (!(CG_DBG & 16) || (yy >= y0 && yy < y1));
      if ((CG_DBG & 16) && !ok)
        return make_float4(0, 0, 0, 0);
      if constexpr (D == 1)
      {
        if (!ok) // a 1-D grid has a single row: nothing above or below to fetch
          return make_float4(0, 0, 0, 0);
      }
      const float4 v = src.ld4(col0 + (ok ? yy : y0) * rs);
      return sel4(ok, v);
    };
    // the element just outside this wave's span of row yy (rows inside the grid only)
    auto ldedge = [&](long long yy) -> float {
      float e = 0.0f;
      if constexpr (D == 1)
      {
        // a long 1-D grid is cut into "rows" of n0 points only to march through it: the rows are
        // contiguous in memory and the ends of a row are ordinary neighbours, except at the two
        // ends of the whole grid
        // (a row need not fill its last wave: the last lane of a row is an edge lane as well)
        const bool e1 = act && ((lane == 0 && (x0 > 0 || yy > 0)) || (lane != 0 && rightmost && (x0 + 4 < g.n0 || yy + 1 < n1)));
        if (e1 && !(CG_DBG & 1))
          e = src.ld1(col0 + yy * rs + eoff);
      }
      else
      {
        if (edge)
          e = src.ld1(col0 + yy * rs + eoff);
      }
      return e;
    };

    float4 pm = ldrow(y0 - 1);
    float4 pc = ldrow(y0);
    float4 pn = ldrow(y0 + 1);
    float ec = ldedge(y0);
    float en = ldedge(y0 + 1 < y1 ? y0 + 1 : y0);
    float4 xv = zero4, rv = zero4, qv = zero4;
    if constexpr (MODE == MODE_B)
    {
      if (xm != 1)
        xv = load4<NT>(a.x + col0 + y0 * rs);
      if (xm == 2)
        qv = load4<Src<MODE, P>::NT_P>(p_prev + col0 + y0 * rs);
      rv = load4<NT_R>(a.r + col0 + y0 * rs);
    }
    if constexpr (MODE == MODE_A)
    {
      if (rowghost && g.has_lo && y0 == 0 && act)
        store4<NT>(p_new + col0 - rs, pm);
    }

    for (long long y = y0; y < y1; y++)
    {
      const long long idx = col0 + y * rs;
      // ---- prefetch for the next row -------------------------------------------------
      const long long y2 = y + 2 <= y1 ? y + 2 : y1;       // row y1 is the last one ever needed
      const float4 pnn = ldrow(y2);
      const float enn = ldedge(y + 2 < y1 ? y + 2 : y1 - 1);
      float4 xn = zero4, rn = zero4, qn = zero4;
      if constexpr (MODE == MODE_B)
      {
        if (y + 1 < y1) // wave-uniform
        {
          if (xm != 1)
            xn = load4<NT>(a.x + col0 + (y + 1) * rs);
          if (xm == 2)
            qn = load4<Src<MODE, P>::NT_P>(p_prev + col0 + (y + 1) * rs);
          rn = load4<NT_R>(a.r + col0 + (y + 1) * rs);
        }
      }

      // ---- row y ---------------------------------------------------------------------
      float lft = (CG_DBG & 2) ? 0.0f : __shfl_up(pc.w, 1);
      float rgt = (CG_DBG & 2) ? 0.0f : __shfl_down(pc.x, 1);
      if (lane == 0)
        lft = ec;
      if (rightmost)
        rgt = (lane == 63 || D == 1) ? ec : 0.0f;
      if constexpr (PACK)
      {
        if (x0 == 0) // first lane of a packed row: the grid face, not the previous lane's row
          lft = 0.0f;
      }

      float4 lv;
      lv.x = 0.0f + axis_term(pc.y, pc.x, lft);
      lv.y = 0.0f + axis_term(pc.z, pc.y, pc.x);
      lv.z = 0.0f + axis_term(pc.w, pc.z, pc.y);
      lv.w = 0.0f + axis_term(rgt, pc.w, pc.z);
      // terms are added in axis order 0, 1, ..., d-1 (conjugate-gradient_gpu.cu:172-174): the marching
      // axis is axis 1, or the last axis for packed columns
      if (!PACK && d >= 2)
      {
        lv.x += axis_term(pn.x, pc.x, pm.x);
        lv.y += axis_term(pn.y, pc.y, pm.y);
        lv.z += axis_term(pn.z, pc.z, pm.z);
        lv.w += axis_term(pn.w, pc.w, pm.w);
      }
      for (int k = 2; k < d; k++)
      {
        const long long s = (long long)g.sk[k - 2];
        const bool hp = (okp >> (k - 2)) & 1u, hm = (okm >> (k - 2)) & 1u;
        const float4 up = sel4(hp, src.ld4(idx + (hp ? s : 0)));
        const float4 um = sel4(hm, src.ld4(idx - (hm ? s : 0)));
        lv.x += axis_term(up.x, pc.x, um.x);
        lv.y += axis_term(up.y, pc.y, um.y);
        lv.z += axis_term(up.z, pc.z, um.z);
        lv.w += axis_term(up.w, pc.w, um.w);
        if constexpr (MODE == MODE_A && GHOST)
        {
          if (k == d - 1 && act)
          {
            if (st_lo)
              store4<NT>(p_new + idx - s, um);
            if (st_hi)
              store4<NT>(p_new + idx + s, up);
          }
        }
      }
      if constexpr (PACK)
      {
        lv.x += axis_term(pn.x, pc.x, pm.x);
        lv.y += axis_term(pn.y, pc.y, pm.y);
        lv.z += axis_term(pn.z, pc.z, pm.z);
        lv.w += axis_term(pn.w, pc.w, pm.w);
      }
      // lanes past the end of the row duplicate the last lane's data: no contribution
      if (!act)
        lv = zero4;
      if (tail)
      {
        // A p is not zero at the padding (its left neighbour is a grid point): keep it out of r
        if (x0 + 1 >= g.n0)
          lv.y = 0.0f;
        if (x0 + 2 >= g.n0)
          lv.z = 0.0f;
        if (x0 + 3 >= g.n0)
          lv.w = 0.0f;
      }

      if constexpr (MODE == MODE_STENCIL)
      {
        if (act)
          store4<false>(a.out + idx, lv);
      }
      else if constexpr (MODE == MODE_A)
      {
        if (act)
          store4<NT>(p_new + idx, pc);
        acc = fmaf(pc.x, lv.x, acc);
        acc = fmaf(pc.y, lv.y, acc);
        acc = fmaf(pc.z, lv.z, acc);
        acc = fmaf(pc.w, lv.w, acc);
      }
      else
      {
        if (xm == 2)
        {
          xv.x = fmaf(alpha_prev, qv.x, xv.x);
          xv.y = fmaf(alpha_prev, qv.y, xv.y);
          xv.z = fmaf(alpha_prev, qv.z, xv.z);
          xv.w = fmaf(alpha_prev, qv.w, xv.w);
        }
        xv.x = fmaf(alpha, pc.x, xv.x);
        xv.y = fmaf(alpha, pc.y, xv.y);
        xv.z = fmaf(alpha, pc.z, xv.z);
        xv.w = fmaf(alpha, pc.w, xv.w);
        rv.x = fmaf(-alpha, lv.x, rv.x);
        rv.y = fmaf(-alpha, lv.y, rv.y);
        rv.z = fmaf(-alpha, lv.z, rv.z);
        rv.w = fmaf(-alpha, lv.w, rv.w);
        if (act)
        {
          if (xm != 1)
            store4<NT>(a.x + idx, xv);
          store4<NT>(a.r + idx, rv);
          if constexpr (GHOST)
          {
            // boundary hyperplanes of the slab: also into the neighbours' staging areas
            if (rowghost)
            {
              if (push_lo && y == 0)
                store4<false>(push_lo + col0, rv);
              if (push_hi && y + 1 == n1)
                store4<false>(push_hi + col0, rv);
            }
            else if (d >= 3)
            {
              const int kl = d >= 3 ? d - 3 : 0; // (d < 3 never gets here)
              const long long s = (long long)g.sk[kl];
              if (push_lo && st_lo)
                store4<false>(push_lo + idx, rv);
              if (push_hi && st_hi)
                store4<false>(push_hi + (idx - (long long)(g.nk[kl] - 1u) * s), rv);
            }
          }
        }
        else
          rv = zero4;
        acc = fmaf(rv.x, rv.x, acc);
        acc = fmaf(rv.y, rv.y, acc);
        acc = fmaf(rv.z, rv.z, acc);
        acc = fmaf(rv.w, rv.w, acc);
      }
      pm = pc;
      pc = pn;
      pn = pnn;
      ec = en;
      en = enn;
      xv = xn;
      rv = rn;
      qv = qn;
    }
    if constexpr (MODE == MODE_A)
    {
      // pc now holds row y1
      if (rowghost && g.has_hi && y1 == n1 && act)
        store4<NT>(p_new + col0 + y1 * rs, pc);
    }
  };
  for (unsigned item = blockIdx.x; item < g.nitems; item += gridDim.x)
  {
    if constexpr (POL == POL_PART)
    {
      // a work item takes the policy of the part of the grid in which its last row lies
      const unsigned long long y1 = (unsigned long long)(item % g.bpc + 1u) * g.n1 / g.bpc;
      if (y1 <= g.keep_y)
        work(std::integral_constant<int, POL_KEEP_R>{}, item);
      else
        work(std::integral_constant<int, POL_STREAM>{}, item);
    }
    else
      work(std::integral_constant<int, POL>{}, item);
  }
  if constexpr (MODE != MODE_STENCIL)
    block_write_partial(acc, a, sh);
}

// ---- scalar engine ---------------------------------------------------------------------

template <int MODE, bool GHOST, int LAYOUT>
__global__ void __launch_bounds__(256) k_scalar(const ScalarGeom g, const VecArgs a)
{
  __shared__ double sh[8];
  CG_T(0);
  Src<MODE, POL_PLAIN> src;
  float *__restrict__ p_new = nullptr;
  float coef = 0;
  int xm = 0; // paired x update, as in k_rows
  float alpha_prev = 0;
  const float *__restrict__ p_prev = nullptr;
  if constexpr (MODE == MODE_A)
  {
    const IterCtl c = prologue_A(a, sh);
    if (!c.active)
      return;
    src.p = (c.k & 1) ? a.p1 : a.p0;
    p_new = (c.k & 1) ? a.p0 : a.p1;
    src.r = a.r;
    src.beta = c.coef;
  }
  else if constexpr (MODE == MODE_B)
  {
    const IterCtl c = prologue_B<true>(a, sh);
    if (!c.active)
      return;
    src.p = (c.k & 1) ? a.p0 : a.p1;
    p_prev = (c.k & 1) ? a.p1 : a.p0;
    coef = c.coef;
    xm = c.xm;
    alpha_prev = c.alpha_prev;
  }
  else
  {
    src.p = a.u;
  }
  const float alpha = coef;
  float acc = 0;
  CG_T(1);
  float *push_lo = nullptr, *push_hi = nullptr;
  if constexpr (MODE == MODE_B && GHOST)
    push_targets(a, push_lo, push_hi);
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < g.N; i += step)
  {
    const float c = src.ld1((long long)i);
    float lv = 0.0f;
    long long off_lo = -1, off_hi = -1; // position inside the first / last owned hyperplane
    if constexpr (LAYOUT == LAYOUT_LEX)
    {
      // N < 2^32, so 32-bit division.  (Replacing it by shifts or a multiply-high made the
      // kernels slower on MI355X: they are bound by the 2d neighbour loads, not by the ALU.)
      unsigned rem = (unsigned)i;
      for (int k = 0; k < g.d; k++)
      {
        const unsigned nk = g.n[k];
        const unsigned ck = rem % nk;
        rem /= nk;
        const long long s = (long long)g.stride[k];
        const bool last = GHOST && (k == g.d - 1);
        const bool hp = (ck + 1 < nk) || (last && g.has_hi);
        const bool hm = (ck > 0) || (last && g.has_lo);
        const float up = hp ? src.ld1((long long)i + s) : 0.0f;
        const float um = hm ? src.ld1((long long)i - s) : 0.0f;
        lv = lv + axis_term(up, c, um);
        if constexpr (MODE == MODE_A && GHOST)
        {
          if (last && g.has_lo && ck == 0)
            p_new[(long long)i - s] = um;
          if (last && g.has_hi && ck + 1 == nk)
            p_new[(long long)i + s] = up;
        }
        if constexpr (MODE == MODE_B && GHOST)
        {
          if (last && ck == 0)
            off_lo = (long long)i;
          if (last && ck + 1 == nk)
            off_hi = (long long)i - (long long)ck * s;
        }
      }
    }
    else
    {
      // Morton: a step out of the grid lands on an index >= N (needs L = 2^m)
      const unsigned ii = (unsigned)i;
      for (int k = 0; k < g.d; k++)
      {
        const unsigned m = g.mmask[k];
        const unsigned ip = morton_up(ii, m, g.N);
        const unsigned im = morton_dn(ii, m, g.N);
        const float up = ip < g.N ? src.ld1((long long)ip) : 0.0f;
        const float um = im < g.N ? src.ld1((long long)im) : 0.0f;
        lv = lv + axis_term(up, c, um);
      }
    }
    if constexpr (MODE == MODE_STENCIL)
    {
      a.out[i] = lv;
    }
    else if constexpr (MODE == MODE_A)
    {
      p_new[i] = c;
      acc = fmaf(c, lv, acc);
    }
    else
    {
      const float rv = fmaf(-alpha, lv, a.r[i]);
      if (xm != 1)
      {
        float xv = a.x[i];
        if (xm == 2)
          xv = fmaf(alpha_prev, p_prev[i], xv);
        a.x[i] = fmaf(alpha, c, xv);
      }
      a.r[i] = rv;
      if constexpr (GHOST)
      {
        if (push_lo && off_lo >= 0)
          push_lo[off_lo] = rv;
        if (push_hi && off_hi >= 0)
          push_hi[off_hi] = rv;
      }
      acc = fmaf(rv, rv, acc);
    }
  }
  CG_T(2);
  if constexpr (MODE != MODE_STENCIL)
    block_write_partial(acc, a, sh);
  CG_T(3);
}

// ---- small kernels (defined in cg_solver.hip) ------------------------------------------

// per-block partials of v.w; 16-byte loads when both pointers allow it
__global__ void k_dot(const float *__restrict__ v, const float *__restrict__ w, unsigned long long n, int vec4,
                      double *__restrict__ part_out);
// out[0] = sum(part[0..n)) in the same order block_read_sum uses
__global__ void k_sum_partials(const double *__restrict__ part, int n, double *__restrict__ out);
// start of a solve: iteration 0, reltol = rel * |b|
__global__ void k_init_state(CgState *st, const double *__restrict__ sumB, int nB, double rel, int lagged, int maxit);
__global__ void k_set_maxit(CgState *st, int add);
__global__ void k_fill_uniform(float *__restrict__ dst, unsigned long long n, unsigned int seed);

// counter-based uniform [0,1): the same integer hash on host and device
__host__ __device__ __forceinline__ float u01_hash(unsigned int seed, unsigned long long i)
{
  unsigned int h = (unsigned int)i * 0x9E3779B9u + seed + (unsigned int)(i >> 32) * 0x85EBCA6Bu;
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}

} // namespace cgamd
