// cg_morton.hpp -- the Morton (bit-interleaved) engine, 4 points per lane.
//
// Vectors are stored in Z-order (interleave.cuh:14-118): coordinate k owns bits k, k+d,
// k+2d, ... of the index.  Four consecutive indices i..i+3 (i % 4 == 0, d >= 2) differ only
// in bit 0 of coordinate 0 and bit 0 of coordinate 1, so
//   * along axes >= 2 all four points step together: one 16-byte load at inc/dec(i);
//   * along axis 0 (axis 1) two of the four neighbours are inside the lane's own quad and
//     the other two sit in ONE other quad (the masked add/subtract of interleave.cuh:43-67
//     applied to the quad's odd / even element), again one 16-byte load.
// A step that leaves the grid lands on an index >= N (needs L = 2^m) and reads as zero.
// Every workgroup walks a contiguous range of the Z-curve, i.e. a compact block of the
// grid, so most neighbour quads are re-used from its own XCD's L2.
#pragma once
#include "cg_device.hpp"

namespace cgamd
{

struct MortonGeom
{
  int d;
  unsigned long long N;
  unsigned mmask[CG_MAXD];
  int interleave; // tile order of the tiled kernel: 0 contiguous range per workgroup, 1 XCD-chunked round-robin
};

template <int MODE, int POL>
__global__ void __launch_bounds__(256) k_morton4(const MortonGeom g, const VecArgs a)
{
  __shared__ double sh[8];
  constexpr bool NT = POL == POL_KEEP_R;
  // p is re-read through the caches by neighbouring quads: never stream it past them
  Src<MODE, POL_PLAIN> src;
  float *__restrict__ p_new = nullptr;
  float coef = 0;
  // paired x update (kernel B, see CgState::xdef_k)
  int xmode = 0;
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
    xmode = c.xm;
    alpha_prev = c.alpha_prev;
  }
  else
  {
    src.p = a.u;
  }
  const float alpha = coef;
  float acc = 0;
  const float4 zero4 = make_float4(0, 0, 0, 0);
  const unsigned N = (unsigned)g.N;
  const unsigned quads = N >> 2;
  const unsigned q0 = (unsigned)((unsigned long long)blockIdx.x * quads / gridDim.x);
  const unsigned q1 = (unsigned)((unsigned long long)(blockIdx.x + 1) * quads / gridDim.x);
  const unsigned m0 = g.mmask[0], m1 = g.mmask[1];

  // quad holding index j (zero if j is outside the grid)
  auto quad = [&](unsigned j) -> float4 {
    const bool ok = j < N;
    return sel4(ok, src.ld4((long long)(ok ? (j & ~3u) : 0u)));
  };

  for (unsigned q = q0 + threadIdx.x; q < q1; q += blockDim.x)
  {
    const unsigned i = q << 2;
    const float4 c = src.ld4((long long)i);
    float4 xv = zero4, rv = zero4, qv = zero4;
    if constexpr (MODE == MODE_B)
    {
      if (xmode != 1)
        xv = load4<NT>(a.x + i);
      if (xmode == 2)
        qv = load4<NT>(p_prev + i);
      rv = load4<false>(a.r + i);
    }
    // axis 0: elements 0,2 have bit 0 clear, elements 1,3 have it set
    const float4 xp = quad(morton_up(i + 1u, m0, N)); // +1 of elements 1 and 3: its elements 0 and 2
    const float4 xm = quad(morton_dn(i, m0, N));      // -1 of elements 0 and 2: its elements 1 and 3
    // axis 1: elements 0,1 have bit 1 clear, elements 2,3 have it set
    const float4 yp = quad(morton_up(i + 2u, m1, N)); // +1 of elements 2 and 3: its elements 0 and 1
    const float4 ym = quad(morton_dn(i, m1, N));      // -1 of elements 0 and 1: its elements 2 and 3
    float4 lv;
    lv.x = 0.0f + axis_term(c.y, c.x, xm.y);
    lv.y = 0.0f + axis_term(xp.x, c.y, c.x);
    lv.z = 0.0f + axis_term(c.w, c.z, xm.w);
    lv.w = 0.0f + axis_term(xp.z, c.w, c.z);
    lv.x += axis_term(c.z, c.x, ym.z);
    lv.y += axis_term(c.w, c.y, ym.w);
    lv.z += axis_term(yp.x, c.z, c.x);
    lv.w += axis_term(yp.y, c.w, c.y);
    for (int k = 2; k < g.d; k++)
    {
      const unsigned m = g.mmask[k];
      const float4 up = quad(morton_up(i, m, N));
      const float4 um = quad(morton_dn(i, m, N));
      lv.x += axis_term(up.x, c.x, um.x);
      lv.y += axis_term(up.y, c.y, um.y);
      lv.z += axis_term(up.z, c.z, um.z);
      lv.w += axis_term(up.w, c.w, um.w);
    }
    if constexpr (MODE == MODE_STENCIL)
    {
      store4<false>(a.out + i, lv);
    }
    else if constexpr (MODE == MODE_A)
    {
      store4<false>(p_new + i, c);
      acc = fmaf(c.x, lv.x, acc);
      acc = fmaf(c.y, lv.y, acc);
      acc = fmaf(c.z, lv.z, acc);
      acc = fmaf(c.w, lv.w, acc);
    }
    else
    {
      if (xmode == 2)
      {
        xv.x = fmaf(alpha_prev, qv.x, xv.x);
        xv.y = fmaf(alpha_prev, qv.y, xv.y);
        xv.z = fmaf(alpha_prev, qv.z, xv.z);
        xv.w = fmaf(alpha_prev, qv.w, xv.w);
      }
      xv.x = fmaf(alpha, c.x, xv.x);
      xv.y = fmaf(alpha, c.y, xv.y);
      xv.z = fmaf(alpha, c.z, xv.z);
      xv.w = fmaf(alpha, c.w, xv.w);
      rv.x = fmaf(-alpha, lv.x, rv.x);
      rv.y = fmaf(-alpha, lv.y, rv.y);
      rv.z = fmaf(-alpha, lv.z, rv.z);
      rv.w = fmaf(-alpha, lv.w, rv.w);
      if (xmode != 1)
        store4<NT>(a.x + i, xv);
      store4<false>(a.r + i, rv);
      acc = fmaf(rv.x, rv.x, acc);
      acc = fmaf(rv.y, rv.y, acc);
      acc = fmaf(rv.z, rv.z, acc);
      acc = fmaf(rv.w, rv.w, acc);
    }
  }
  if constexpr (MODE != MODE_STENCIL)
    block_write_partial(acc, a, sh);
}


// ---- LDS-tiled variant -------------------------------------------------------------------
//
// A tile is CG_MORTON_TILE consecutive indices of the Z-curve: a compact box of the grid
// (16x8x8x8 in 4-D with the default 8192 indices; 4 quads per thread).  The workgroup first stages the tile's stencil input in LDS (for kernel A
// that is r + beta*p, formed once per point), then every quad takes the neighbour quads that
// lie inside the tile from LDS and only the others (one face per axis and direction) from
// memory.  Two LDS buffers alternate, so one barrier per tile suffices.  Measured at 128^4:
// 4096-index tiles with 256 threads 451 it/s, 8192 with 512 threads 470, 16384 with 1024 threads
// (one buffer) 460; more than 4 quads per thread costs registers and loses a quarter.
#ifndef CG_MORTON_TILE_BITS_DEF
#define CG_MORTON_TILE_BITS_DEF 13
#endif
#ifndef CG_MORTON_BT
#define CG_MORTON_BT 512
#endif
#ifndef CG_MORTON_NBUF
#define CG_MORTON_NBUF 2
#endif
constexpr unsigned CG_MORTON_TILE_BITS = CG_MORTON_TILE_BITS_DEF;
constexpr unsigned CG_MORTON_TILE = 1u << CG_MORTON_TILE_BITS;

template <int MODE, int POL>
__global__ void __launch_bounds__(CG_MORTON_BT) k_morton_tile(const MortonGeom g, const VecArgs a)
{
  __shared__ double sh[CG_MORTON_BT / 64 < 8 ? 8 : CG_MORTON_BT / 64];
  __shared__ float4 tile[CG_MORTON_NBUF][CG_MORTON_TILE / 4];
  constexpr bool NT = POL == POL_KEEP_R;
  constexpr int QPT = CG_MORTON_TILE / 4 / CG_MORTON_BT; // quads per thread and tile
  Src<MODE, POL_PLAIN> src;
  float *__restrict__ p_new = nullptr;
  float coef = 0;
  // paired x update (kernel B, see CgState::xdef_k)
  int xmode = 0;
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
    xmode = c.xm;
    alpha_prev = c.alpha_prev;
  }
  else
  {
    src.p = a.u;
  }
  const float alpha = coef;
  float acc = 0;
  const float4 zero4 = make_float4(0, 0, 0, 0);
  const unsigned N = (unsigned)g.N;
  const unsigned ntiles = N >> CG_MORTON_TILE_BITS;
  unsigned t0 = (unsigned)((unsigned long long)blockIdx.x * ntiles / gridDim.x);
  unsigned t1 = (unsigned)((unsigned long long)(blockIdx.x + 1) * ntiles / gridDim.x);
  unsigned tstep = 1;
  if (g.interleave && gridDim.x % 8 == 0 && ntiles % 8 == 0)
  {
    // workgroups b, b+8, ... share an XCD (round-robin dispatch): give each XCD one eighth of the
    // Z-curve and let its workgroups sweep it side by side, so that the tiles in flight are
    // neighbours and their faces meet in that XCD's L2
    const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3, per = gridDim.x >> 3, chunk = ntiles >> 3;
    t0 = xcd * chunk + j;
    t1 = (xcd + 1) * chunk;
    tstep = per;
  }
  const unsigned m0 = g.mmask[0], m1 = g.mmask[1];

  unsigned par = 0;
  for (unsigned t = t0; t < t1; t += tstep, par ^= 1u)
  {
    const unsigned base = t << CG_MORTON_TILE_BITS;
    float4(&buf)[CG_MORTON_TILE / 4] = tile[CG_MORTON_NBUF == 2 ? par : 0];
    float4 c[QPT], xv[QPT], rv[QPT], qv[QPT];
#pragma unroll
    for (int s = 0; s < QPT; s++)
    {
      const unsigned ql = threadIdx.x + (unsigned)CG_MORTON_BT * s;
      const unsigned i = base + (ql << 2);
      c[s] = src.ld4((long long)i);
      if constexpr (MODE == MODE_B)
      {
        xv[s] = qv[s] = zero4;
        if (xmode != 1)
          xv[s] = load4<NT>(a.x + i);
        if (xmode == 2)
          qv[s] = load4<NT>(p_prev + i);
        rv[s] = load4<false>(a.r + i);
      }
    }
#pragma unroll
    for (int s = 0; s < QPT; s++)
      buf[threadIdx.x + (unsigned)CG_MORTON_BT * s] = c[s];
    __syncthreads();

    // quad holding index j: LDS if j is in this tile, memory otherwise, zero outside the grid
    auto quad = [&](unsigned j) -> float4 {
      float4 v = zero4;
      if ((j >> CG_MORTON_TILE_BITS) == t)
        v = buf[(j & (CG_MORTON_TILE - 1u)) >> 2];
      else if (j < N)
        v = src.ld4((long long)(j & ~3u));
      return v;
    };

#pragma unroll
    for (int s = 0; s < QPT; s++)
    {
      const unsigned ql = threadIdx.x + (unsigned)CG_MORTON_BT * s;
      const unsigned i = base + (ql << 2);
      const float4 cc = c[s];
      const float4 xp = quad(morton_up(i + 1u, m0, N));
      const float4 xm = quad(morton_dn(i, m0, N));
      const float4 yp = quad(morton_up(i + 2u, m1, N));
      const float4 ym = quad(morton_dn(i, m1, N));
      float4 lv;
      lv.x = 0.0f + axis_term(cc.y, cc.x, xm.y);
      lv.y = 0.0f + axis_term(xp.x, cc.y, cc.x);
      lv.z = 0.0f + axis_term(cc.w, cc.z, xm.w);
      lv.w = 0.0f + axis_term(xp.z, cc.w, cc.z);
      lv.x += axis_term(cc.z, cc.x, ym.z);
      lv.y += axis_term(cc.w, cc.y, ym.w);
      lv.z += axis_term(yp.x, cc.z, cc.x);
      lv.w += axis_term(yp.y, cc.w, cc.y);
      for (int k = 2; k < g.d; k++)
      {
        const unsigned m = g.mmask[k];
        const float4 up = quad(morton_up(i, m, N));
        const float4 um = quad(morton_dn(i, m, N));
        lv.x += axis_term(up.x, cc.x, um.x);
        lv.y += axis_term(up.y, cc.y, um.y);
        lv.z += axis_term(up.z, cc.z, um.z);
        lv.w += axis_term(up.w, cc.w, um.w);
      }
      if constexpr (MODE == MODE_STENCIL)
      {
        store4<false>(a.out + i, lv);
      }
      else if constexpr (MODE == MODE_A)
      {
        store4<NT>(p_new + i, cc); // streams: kernel B re-reads it only after the whole vector was written
        acc = fmaf(cc.x, lv.x, acc);
        acc = fmaf(cc.y, lv.y, acc);
        acc = fmaf(cc.z, lv.z, acc);
        acc = fmaf(cc.w, lv.w, acc);
      }
      else
      {
        float4 xo = xv[s], ro = rv[s];
        if (xmode == 2)
        {
          xo.x = fmaf(alpha_prev, qv[s].x, xo.x);
          xo.y = fmaf(alpha_prev, qv[s].y, xo.y);
          xo.z = fmaf(alpha_prev, qv[s].z, xo.z);
          xo.w = fmaf(alpha_prev, qv[s].w, xo.w);
        }
        xo.x = fmaf(alpha, cc.x, xo.x);
        xo.y = fmaf(alpha, cc.y, xo.y);
        xo.z = fmaf(alpha, cc.z, xo.z);
        xo.w = fmaf(alpha, cc.w, xo.w);
        ro.x = fmaf(-alpha, lv.x, ro.x);
        ro.y = fmaf(-alpha, lv.y, ro.y);
        ro.z = fmaf(-alpha, lv.z, ro.z);
        ro.w = fmaf(-alpha, lv.w, ro.w);
        if (xmode != 1)
          store4<NT>(a.x + i, xo);
        store4<false>(a.r + i, ro);
        acc = fmaf(ro.x, ro.x, acc);
        acc = fmaf(ro.y, ro.y, acc);
        acc = fmaf(ro.z, ro.z, acc);
        acc = fmaf(ro.w, ro.w, acc);
      }
    }
    if (CG_MORTON_NBUF == 1)
      __syncthreads(); // the single buffer is overwritten by the next tile
  }
  if constexpr (MODE != MODE_STENCIL)
    block_write_partial(acc, a, sh);
}

} // namespace cgamd
