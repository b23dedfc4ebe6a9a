// cg_small.hpp -- whole CG iterations of a small grid inside ONE workgroup.
//
// The reference's own tests and demos solve grids of a few thousand points (50^2, 16^3, 8^4 ...),
// where the two-kernel iteration is bound by launch latency (about 11 us per iteration on MI355X).
// For grids of at most CG_SMALL_MAX points a single workgroup of 1024 lanes keeps x and r in
// registers and p in LDS and runs iteration after iteration without leaving the kernel: the
// stencil reads its neighbours from LDS, the two dot products are block reductions, alpha and beta
// never leave the CU.  State conventions are exactly those of kernels A and B (cg_device.hpp): it
// starts from CgState, r, p (the buffer iteration parity selects) and the bound r.r addends, and
// leaves them as kernel B would after the same number of iterations, so launches of either kind
// can follow each other.
//
// Same arithmetic as the big kernels per point (operation order of conjugate-gradient_gpu.cu:173,
// float alpha and beta, FMA updates); the dot products add the same fp32 per-lane partials in a
// different, equally fixed order, so results agree to rounding, not bit for bit.
#pragma once
#include "cg_device.hpp"

namespace cgamd
{

constexpr int CG_SMALL_PPT = 16;                      // points per lane, at most (8 for grids of up to 8192 points)
constexpr int CG_SMALL_THREADS = 1024;
constexpr int CG_SMALL_MAX = CG_SMALL_PPT * CG_SMALL_THREADS;
constexpr int CG_SMALL_MAXD = 8;                      // 2 validity bits per axis in a 16-bit mask

struct SmallGeom
{
  int d;
  unsigned N;
  unsigned n[CG_SMALL_MAXD];
  unsigned stride[CG_SMALL_MAXD];
  unsigned mmask[CG_SMALL_MAXD];
};

// sum over the workgroup, the same double in every lane; fixed order (wave butterflies, then waves 0..15)
__device__ __forceinline__ double small_block_sum(float acc, double *sh)
{
  const double v = wave_sum((double)acc);
  const int w = threadIdx.x >> 6;
  __syncthreads(); // sh may still be read from the previous reduction
  if ((threadIdx.x & 63) == 0)
    sh[w] = v;
  __syncthreads();
  double s = 0;
#pragma unroll
  for (int i = 0; i < CG_SMALL_THREADS / 64; i++)
    s += sh[i];
  return s;
}

// D: compile-time dimension (1..4) so that the axis loops unroll; 0 = run-time g.d
// PPT: points per lane (8: up to 8192 points, 2.75 us per iteration; 16: up to 16384, about twice that, still
// below the two launches of the two-kernel iteration)
template <int LAYOUT, int D, int PPT>
__global__ void __launch_bounds__(CG_SMALL_THREADS) k_small(const SmallGeom g, const VecArgs a, double *partB, int nB)
{
  static_assert(PPT <= CG_SMALL_PPT, "more points per lane than the geometry allows");
  __shared__ float pl[PPT * CG_SMALL_THREADS];
  __shared__ double sh[CG_SMALL_THREADS / 64];
  __shared__ double shb[8];
  CgState *st = a.st;
  const int tid = threadIdx.x;
  const int d = D > 0 ? D : g.d;
  const int nj = (int)((g.N + CG_SMALL_THREADS - 1) / CG_SMALL_THREADS); // point slots in use
  int k = st->itA;
  bool done = st->doneA != 0;
  const int maxit = st->maxit;
  const int lagged = st->lagged;
  const float reltol = st->reltol;
  float rr = (float)block_read_sum(a.sumB, a.nB, shb);
  float rr_prev = st->rr_hist[(k + CG_HIST - 1) % CG_HIST];
  const int k_in = k;

  // ---- load this lane's points; neighbour validity is computed once ------------------------
  float x[PPT], r[PPT], p[PPT];
  unsigned okbits[PPT];
  const float *p_in = (k & 1) ? a.p1 : a.p0;
#pragma unroll
  for (int j = 0; j < PPT; j++)
  {
    const unsigned i = (unsigned)tid + (unsigned)CG_SMALL_THREADS * j;
    const bool in = i < g.N;
    x[j] = in ? a.x[i] : 0.0f;
    r[j] = in ? a.r[i] : 0.0f;
    p[j] = in ? p_in[i] : 0.0f;
    unsigned bits = 0;
    if (in)
    {
      if constexpr (LAYOUT == LAYOUT_LEX)
      {
        unsigned rem = i;
        for (int ax = 0; ax < d; ax++)
        {
          const unsigned c = rem % g.n[ax];
          rem /= g.n[ax];
          if (c + 1 < g.n[ax])
            bits |= 1u << (2 * ax);
          if (c > 0)
            bits |= 2u << (2 * ax);
        }
      }
      else
      {
        for (int ax = 0; ax < d; ax++)
        {
          if (morton_up(i, g.mmask[ax], g.N) < g.N)
            bits |= 1u << (2 * ax);
          if (morton_dn(i, g.mmask[ax], g.N) < g.N)
            bits |= 2u << (2 * ax);
        }
      }
    }
    okbits[j] = bits;
  }

  for (;;)
  {
    // ---- what the prologue of kernel A decides (cg_device.hpp, prologue_A) -------------------
    const float rr_test = (lagged && k > 0) ? rr_prev : rr;
    const float residue = (float)sqrt((double)rr_test);
    const bool conv = done || !(residue > reltol);
    const bool active = !conv && k < maxit;
    const float beta = k > 0 ? rr / rr_prev : 0.0f;
    if (tid == 0)
    {
      st->rr_hist[k % CG_HIST] = rr;
      st->rr_pp[k & 1] = rr;
      st->beta = beta;
    }
    done = conv;
    if (!active)
      break;

    // ---- p = r + beta p, into LDS -------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < PPT; j++)
    {
      p[j] = fmaf(beta, p[j], r[j]);
      pl[tid + CG_SMALL_THREADS * j] = p[j];
    }
    __syncthreads();

    // ---- A p and p . A p ----------------------------------------------------------------------
    float ap[PPT];
    float acc = 0;
#pragma unroll
    for (int j = 0; j < PPT; j++)
    {
      ap[j] = 0.0f;
      if (j >= nj) // no lane has a point in this slot (uniform)
        continue;
      const unsigned i = (unsigned)tid + (unsigned)CG_SMALL_THREADS * j;
      const float c = p[j];
      float lv = 0.0f;
      for (int ax = 0; ax < d; ax++)
      {
        const bool hp = (okbits[j] >> (2 * ax)) & 1u, hm = (okbits[j] >> (2 * ax + 1)) & 1u;
        unsigned ip, im;
        if constexpr (LAYOUT == LAYOUT_LEX)
        {
          ip = i + g.stride[ax];
          im = i - g.stride[ax];
        }
        else
        {
          ip = morton_inc(i, g.mmask[ax]);
          im = morton_dec(i, g.mmask[ax]);
        }
        const float up = hp ? pl[ip] : 0.0f;
        const float um = hm ? pl[im] : 0.0f;
        lv = lv + axis_term(up, c, um);
      }
      ap[j] = lv;
      acc = fmaf(c, lv, acc);
    }
    const float pAp = (float)small_block_sum(acc, sh);
    const float alpha = rr / pAp;

    // ---- x += alpha p ; r -= alpha A p ; r . r --------------------------------------------------
    acc = 0;
#pragma unroll
    for (int j = 0; j < PPT; j++)
    {
      x[j] = fmaf(alpha, p[j], x[j]);
      r[j] = fmaf(-alpha, ap[j], r[j]);
      acc = fmaf(r[j], r[j], acc);
    }
    const float rr_new = (float)small_block_sum(acc, sh); // its barriers also fence this step's reads of pl
    if (tid == 0)
    {
      st->alpha = alpha;
      st->pAp = pAp;
    }
    rr_prev = rr;
    rr = rr_new;
    k++;
  }

  // ---- leave everything as kernel B would have --------------------------------------------------
  float *p_out = (k & 1) ? a.p1 : a.p0;
#pragma unroll
  for (int j = 0; j < PPT; j++)
  {
    const unsigned i = (unsigned)tid + (unsigned)CG_SMALL_THREADS * j;
    if (i < g.N)
    {
      a.x[i] = x[j];
      a.r[i] = r[j];
      if (k != k_in)
        p_out[i] = p[j];
    }
  }
  for (int i = tid; i < nB; i += CG_SMALL_THREADS)
    partB[i] = (i == 0) ? (double)rr : 0.0;
  if (tid == 0)
  {
    st->itA = k;
    st->doneA = done ? 1 : 0;
    st->itB = k;
    st->doneB = done ? 1 : 0;
    st->activeB = 0;
  }
}

} // namespace cgamd
