// cg_cols.hpp -- the rows engine for short rows in d = 3, 4 and 5, with R columns per lane.
//
// As the packed mode of k_rows (cg_device.hpp): a row of axis 0 occupies n0/4 lanes rounded up to a power
// of two (n0 = 4 ... 256),
// the lanes of a workgroup hold several columns side by side and march along the LAST axis with rows
// y-1, y, y+1 in registers.  Two things differ:
//  * every lane owns R columns that are neighbours along axis d-2, so R-1 of every R neighbour pairs
//    of that axis are register values of the same lane and only the two columns just outside the
//    lane's block are loaded;
//  * the neighbours that are loaded (axes 1 .. d-3, the two block-edge columns) are fetched for row
//    y+2 while row y is computed, in the same step in which the owners of those rows prefetch them.
//    All uses of a cache line then fall into one step of the march.  One step of all resident
//    workgroups moves about as many bytes as the eight L2s hold (128^4: 24 MiB against 32 MiB), so
//    neighbours loaded at use, two steps after the owner's prefetch, came from memory again: 2.8
//    times the ideal number of bytes left L2 in kernel A at 128^4, 1.4 times with this kernel.
//
// Columns are numbered along axes 1 .. d-3 (g.nk, g.sk) and then along blocks of R coordinates of
// axis d-2 (g.nb coordinates, g.sb apart).  Terms are added in axis order 0, 1, ..., d-1
// (conjugate-gradient_gpu.cu:172-174), so the result has the same bits as every other engine's.
#pragma once
#include "cg_device.hpp"

namespace cgamd
{

// STREAM: everything that is touched once per kernel (x, the centre loads of r in kernel B, all
// stores) carries the streaming hint, so that L2 and the Infinity Cache keep what neighbours re-read.
// For grids whose vectors exceed that cache (128^4: kernel A 829 -> 781 us, B 1017 -> 946 us); on
// smaller grids the cache serves the next kernel and plain accesses are faster (64^4: B 65 -> 72 us).
// LDS: the packed columns of a workgroup are neighbours along axis 1 (as long as they lie on one line of
// that axis), so a lane's axis-1 neighbours are the current rows of the lanes one column slot below and
// above: every lane puts its rows of step y into LDS (double-buffered by the step's parity, one barrier
// per step) and only the first and last slot of the workgroup load that neighbour from memory.
// Measured, kernels A / B in us without -> with: 128^4 780 / 946 -> 725 / 895, 64x64x64x512 388 / 578 ->
// 360 / 463, 64^4 x 16 1271 / 1300 -> 1384 / 1156, 96^4 406 / 452 -> 500 / 449, 48^5 1229 / 1125 -> 1383 /
// 1081: kernel B always gains, kernel A only in 4-D with rows that fill their lanes (the host chooses).
// Exact rows (g.xl, both kernels through LDS): a row whose n0/4 lanes are not a power of two would idle up
// to half of its slot; instead it takes exactly n0/4 lanes, rows cross wave boundaries, and the x
// neighbours of a lane come from LDS as well instead of from lane shifts.
// TILE (4-D through LDS only): the column slots of a workgroup form a patch of axis 1 x blocks of axis d-2
// (g.cpg / g.t2 by g.t2 slots) instead of a strip along axis 1, so that block-edge neighbours along axis d-2
// inside the patch are rows of this workgroup as well.
// TILE = 2: the same in workgroups of up to 1024 lanes (twice the slots).
template <int MODE, bool GHOST, int D, int R, bool STREAM, bool LDS, int TILE = 0>
__global__ void __launch_bounds__(TILE == 2 ? 1024 : 512, (MODE == MODE_A && D == 4 && STREAM && LDS) ? 4 : 1) k_cols(const RowsGeom g, const VecArgs a)
{
  static_assert(!TILE || (D == 4 && LDS), "tiled slots exist in 4-D with the LDS hand-off only");
  constexpr int BLK = TILE == 2 ? 1024 : 512; // lanes a workgroup can have: the LDS hand-off's row pitch
  __shared__ float4 hand[LDS ? 2 * R * BLK : 1];
  static_assert(D == 3 || D == 4 || D == 5, "columns in blocks: d = 3, 4 and 5");
  static_assert(R == 1 || R == 2, "columns per lane");
  __shared__ double sh[16];
  const int tid = threadIdx.x;
  Src<MODE, POL_PLAIN> src;
  float *__restrict__ p_new = nullptr;
  float coef = 0;
  // paired x update (kernel B, see CgState::xdef_k)
  int xm = 0;
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
  float *push_lo = nullptr, *push_hi = nullptr;
  if constexpr (MODE == MODE_B && GHOST)
    push_targets(a, push_lo, push_hi);
  const float4 zero4 = make_float4(0, 0, 0, 0);
  constexpr int NA = D - 3; // column axes other than the blocked one
  // the second p of a paired x update is fetched a row ahead like x, except in 5-D, where kernel B has no
  // registers left for it (256 with two columns per lane) and loads it where it is used
  constexpr bool QLATE = D >= 5;
  const long long rs = (long long)g.sy; // between consecutive rows of the march (the last axis)
  const long long sb = (long long)g.sb; // between the R columns of a lane (axis d-2)
  const long long n1 = g.n1;
  // padded rows (n0 % 4 != 0): the last lane of a row owns fewer than four points; the others of its
  // float4 are padding that holds zeros in r, p and x and must stay zero
  const bool tail = (g.n0 & 3u) != 0;

  for (unsigned item = blockIdx.x; item < g.nitems; item += gridDim.x)
  {
    const unsigned seg = item % g.bpc;
    const unsigned col = item / g.bpc;
    const unsigned cs = (unsigned)tid / g.sub; // column slot within the workgroup
    const unsigned x0 = ((unsigned)tid % g.sub) * 4u;
    // a row holds n0/4 lanes rounded up to a power of two; idle lanes (past the end of the row, past
    // the last column) shadow a valid quad and store nothing
    // (exact rows, g.xl: a row takes exactly n0/4 lanes and the lanes left over at the end of the
    // workgroup form no slot)
    const bool slot = cs < g.cpg;
    long long base = x0 < g.n0 ? x0 : ((g.n0 + 3u) & ~3u) - 4u;
    unsigned okm = 0, okp = 0;
    unsigned hi;
    bool valid;
    // tiled slots (4-D through LDS, g.t2 > 1): the slots of a workgroup form a t1 x t2 patch of axis 1 x blocks
    // of axis d-2 instead of a strip along axis 1, so that block-edge neighbours along axis d-2 are rows of
    // this workgroup as well; s2 = the slot's position along that axis
    unsigned t1 = g.cpg, s2 = 0;
    if constexpr (TILE)
    {
      t1 = g.cpg / g.t2;
      const unsigned nt1 = (g.nk[0] + t1 - 1) / t1;
      // workgroups b, b + 8, ... run on one XCD (round-robin dispatch) and share its L2: with g.xcd set, the
      // patches of one XCD are g.xcd neighbours along axis 1 (and all of axis d-2) instead of every eighth
      unsigned cc = col;
      if (g.xcd)
      {
        const unsigned x = col % 8u, q = col / 8u;
        cc = (q / g.xcd) * nt1 + x * g.xcd + q % g.xcd;
      }
      const unsigned c1 = (cc % nt1) * t1 + cs % t1;
      s2 = cs / t1;
      const unsigned blk = (cc / nt1) * g.t2 + s2, nblk = g.nb / (unsigned)R;
      valid = slot && c1 < g.nk[0] && blk < nblk;
      const unsigned ck = c1 < g.nk[0] ? c1 : g.nk[0] - 1u;
      hi = blk < nblk ? blk : nblk - 1u;
      base += (long long)ck * (long long)g.sk[0];
      if (ck > 0)
        okm |= 1u;
      if (ck + 1 < g.nk[0])
        okp |= 1u;
    }
    else
    {
      const unsigned c = col * g.cpg + cs;
      valid = slot && c < g.ncols;
      hi = valid ? c : g.ncols - 1u;
#pragma unroll
      for (int k = 0; k < NA; k++)
      {
        const unsigned nk = g.nk[k];
        const unsigned ck = hi % nk;
        hi /= nk;
        base += (long long)ck * (long long)g.sk[k];
        if (ck > 0)
          okm |= 1u << k;
        if (ck + 1 < nk)
          okp |= 1u << k;
      }
    }
    const bool act = valid && x0 < g.n0;
    const unsigned cb = hi * (unsigned)R; // first coordinate of the lane's block along axis d-2
    base += (long long)cb * sb;
    const bool has_bm = cb > 0, has_bp = cb + R < g.nb;
    const long long y0 = (long long)((unsigned long long)seg * g.n1 / g.bpc);
    const long long y1 = (long long)((unsigned long long)(seg + 1) * g.n1 / g.bpc);
    if (y0 >= y1)
      continue;
    const bool first = x0 == 0, rightmost = x0 + 4 >= g.n0;
    // axis 1 (g.nk[0]; in 3-D the blocked axis itself): the neighbour is the next slot of this workgroup
    // unless the slot is the first / last one; it exists at all (okm / okp, has_bm / has_bp) only on the
    // same line of axis 1, where slot +-1 is column (block) +-1
    const unsigned s1 = TILE ? cs % t1 : cs; // the slot's position along axis 1
    const bool lds_m = LDS && s1 > 0, lds_p = LDS && s1 + 1 < t1;
    // tiled slots: the block below / above along axis d-2 is the slot t1 places down / up
    const bool lds_bm = TILE && s2 > 0, lds_bp = TILE && s2 + 1 < g.t2;
    if constexpr (LDS)
      __syncthreads(); // the previous item's last step may still be read

    // stencil input of row yy of column j (a zero row outside the grid unless it is a ghost row)
    auto ldrow = [&](int j, long long yy) -> float4 {
      const bool ok = GHOST || (yy >= 0 && yy < n1);
      const float4 v = src.ld4(base + j * sb + (ok ? yy : y0) * rs);
      return sel4(ok, v);
    };

    // the directly loaded neighbours of one row
    struct Nb
    {
      float4 bm, bp;
      float4 um[NA > 0 ? NA * R : 1], up[NA > 0 ? NA * R : 1];
    };
    auto ldnb = [&](long long yy) -> Nb {
      Nb nb;
      const long long i = base + (yy < y1 ? yy : y1 - 1) * rs;
      if constexpr (LDS && NA == 0)
      {
        // 3-D: the columns just outside the lane's block belong to the next slots of the workgroup
        nb.bm = zero4;
        nb.bp = zero4;
        if (has_bm && !lds_m)
          nb.bm = src.ld4(i - sb);
        if (has_bp && !lds_p)
          nb.bp = src.ld4(i + R * sb);
      }
      else if constexpr (TILE)
      {
        nb.bm = zero4;
        nb.bp = zero4;
        if (has_bm && !lds_bm)
          nb.bm = src.ld4(i - sb);
        if (has_bp && !lds_bp)
          nb.bp = src.ld4(i + R * sb);
      }
      else
      {
        nb.bm = sel4(has_bm, src.ld4(i - (has_bm ? sb : 0)));
        nb.bp = sel4(has_bp, src.ld4(i + (R - 1) * sb + (has_bp ? sb : 0)));
      }
#pragma unroll
      for (int k = 0; k < NA; k++)
      {
        const long long s = (long long)g.sk[k];
        const bool hp = (okp >> k) & 1u, hm = (okm >> k) & 1u;
#pragma unroll
        for (int j = 0; j < R; j++)
        {
          if constexpr (LDS)
          {
            // predicated: most slots take this neighbour from LDS and skip the load
            nb.up[k * R + j] = zero4;
            nb.um[k * R + j] = zero4;
            if (hp && !(k == 0 && lds_p))
              nb.up[k * R + j] = src.ld4(i + j * sb + s);
            if (hm && !(k == 0 && lds_m))
              nb.um[k * R + j] = src.ld4(i + j * sb - s);
          }
          else
          {
            // never predicated (clamped address, select)
            nb.up[k * R + j] = sel4(hp, src.ld4(i + j * sb + (hp ? s : 0)));
            nb.um[k * R + j] = sel4(hm, src.ld4(i + j * sb - (hm ? s : 0)));
          }
        }
      }
      return nb;
    };

    // Tiled slots: a lane loads at most one block-edge row (the patch's first slots along axis d-2 the row
    // below their block, the last slots the row above) and at most R axis-1 neighbours (the first slots of the
    // patch along axis 1 the rows below, the last the rows above); everything else comes from LDS.  Half the
    // registers of Nb per row in flight, which is what lets kernel A keep the deep pipeline (neighbours of row
    // y + 2 fetched in the very step in which their owners prefetch that row, while both are in L2) at two
    // workgroups per CU.
    struct NbT
    {
      float4 bx;
      float4 ux[R];
    };
    const bool bx_on = TILE && ((s2 == 0 && has_bm) || (s2 + 1 == g.t2 && s2 != 0 && has_bp));
    const long long bx_off = (TILE && s2 == 0) ? -sb : (long long)R * sb;
    const bool ux_lo = TILE && s1 == 0; // this lane's loaded axis-1 neighbours are the rows below
    const bool ux_on = TILE && (ux_lo ? (okm & 1u) != 0 : (s1 + 1 == t1 && (okp & 1u) != 0));
    const long long ux_off = ux_lo ? -(long long)g.sk[0] : (long long)g.sk[0];
    auto ldnbT = [&](long long yy) -> NbT {
      NbT nb;
      const long long i = base + (yy < y1 ? yy : y1 - 1) * rs;
      nb.bx = zero4;
      if (bx_on)
        nb.bx = src.ld4(i + bx_off);
#pragma unroll
      for (int j = 0; j < R; j++)
      {
        nb.ux[j] = zero4;
        if (ux_on)
          nb.ux[j] = src.ld4(i + j * sb + ux_off);
      }
      return nb;
    };

    float4 pm[R], pc[R], pn[R], xv[R], rv[R], qv[R];
#pragma unroll
    for (int j = 0; j < R; j++)
    {
      pm[j] = ldrow(j, y0 - 1);
      pc[j] = ldrow(j, y0);
      pn[j] = ldrow(j, y0 + 1);
      xv[j] = zero4;
      rv[j] = zero4;
      qv[j] = zero4;
      if constexpr (MODE == MODE_B)
      {
        if (xm != 1)
          xv[j] = load4<STREAM>(a.x + base + j * sb + y0 * rs);
        if (xm == 2 && !QLATE)
          qv[j] = load4<STREAM>(p_prev + base + j * sb + y0 * rs);
        rv[j] = load4<STREAM>(a.r + base + j * sb + y0 * rs);
      }
      if constexpr (MODE == MODE_A && GHOST)
      {
        if (g.has_lo && y0 == 0 && act)
          store4<STREAM>(p_new + base + j * sb - rs, pm[j]);
      }
    }

    // Depth of the neighbour pipeline: the neighbours of rows y, y+1, y+2 in flight, or (kernel A of large 4-D
    // grids) only of y and y+1.  Kernel A loads every neighbour twice (r and p), so the deep pipeline cost it 155
    // VGPRs and left one 512-lane workgroup per CU; the shallow one fits 128 (two workgroups per CU, as the
    // launch bounds above demand: 8 bytes of spills per lane).  128^4: kernel A 744 -> 665 us, 96^4 317 -> 248,
    // 64^3 x 512 360 -> 340; 3-D and 5-D grids and grids the caches hold lose a few percent and keep the deep one.
    constexpr bool SHALLOW = (MODE == MODE_A) && D == 4 && STREAM && !TILE;
    Nb nb0, nb1;
    NbT t0, t1r;
    if constexpr (TILE)
    {
      t0 = ldnbT(y0);
      t1r = ldnbT(y0 + 1);
    }
    else
    {
      nb0 = ldnb(y0);
      nb1 = SHALLOW ? nb0 : ldnb(y0 + 1);
    }

    for (long long y = y0; y < y1; y++)
    {
      const long long idx = base + y * rs;
      // ---- prefetch for the next rows ----------------------------------------------------
      const long long y2 = y + 2 <= y1 ? y + 2 : y1; // row y1 is the last one ever needed
      float4 pnn[R], xn[R], rn[R], qn[R];
#pragma unroll
      for (int j = 0; j < R; j++)
      {
        pnn[j] = ldrow(j, y2);
        xn[j] = zero4;
        rn[j] = zero4;
        qn[j] = zero4;
        if constexpr (MODE == MODE_B)
        {
          if (y + 1 < y1) // wave-uniform
          {
            if (xm != 1)
              xn[j] = load4<STREAM>(a.x + idx + j * sb + rs);
            if (xm == 2 && !QLATE)
              qn[j] = load4<STREAM>(p_prev + idx + j * sb + rs);
            rn[j] = load4<STREAM>(a.r + idx + j * sb + rs);
          }
        }
      }
      Nb nb2;
      NbT t2r;
      if constexpr (TILE)
      {
        t2r = ldnbT(y + 2);
        // the neighbours of this row that came from memory; LDS supplies the others below
        nb0.bm = (s2 == 0) ? t0.bx : zero4;
        nb0.bp = (s2 == 0) ? zero4 : t0.bx;
#pragma unroll
        for (int j = 0; j < R; j++)
        {
          nb0.um[j] = ux_lo ? t0.ux[j] : zero4;
          nb0.up[j] = ux_lo ? zero4 : t0.ux[j];
        }
      }
      else
        nb2 = ldnb(SHALLOW ? y + 1 : y + 2);
      float4 bm = nb0.bm, bp = nb0.bp;
      if constexpr (LDS)
      {
        float4 *mine = hand + (size_t)(y & 1) * R * BLK + tid;
#pragma unroll
        for (int j = 0; j < R; j++)
          mine[j * BLK] = pc[j];
        __syncthreads();
        if constexpr (NA > 0)
        {
#pragma unroll
          for (int j = 0; j < R; j++)
          {
            if (lds_p && (okp & 1u))
              nb0.up[j] = mine[j * BLK + (int)g.sub];
            if (lds_m && (okm & 1u))
              nb0.um[j] = mine[j * BLK - (int)g.sub];
          }
          if constexpr (TILE)
          {
            if (lds_bm && has_bm)
              bm = mine[(R - 1) * BLK - (int)(t1 * g.sub)];
            if (lds_bp && has_bp)
              bp = mine[(int)(t1 * g.sub)];
          }
        }
        else
        {
          if (lds_m && has_bm)
            bm = mine[(R - 1) * BLK - (int)g.sub];
          if (lds_p && has_bp)
            bp = mine[(int)g.sub];
        }
      }

#pragma unroll
      for (int j = 0; j < R; j++)
      {
        const long long ij = idx + j * sb;
        const float4 q = pc[j];
        float lft, rgt;
        if (LDS && g.xl)
        {
          // exact rows cross wave boundaries: the x neighbours come from the rows just handed to LDS
          const float4 *mine = hand + (size_t)(y & 1) * R * BLK + tid + j * BLK;
          lft = (first || !slot) ? 0.0f : mine[-1].w;
          rgt = (rightmost || !slot) ? 0.0f : mine[1].x;
        }
        else
        {
          lft = __shfl_up(q.w, 1);
          rgt = __shfl_down(q.x, 1);
          if (first) // the grid face, not the previous lane's row
            lft = 0.0f;
          if (rightmost)
            rgt = 0.0f;
        }
        float4 lv;
        lv.x = 0.0f + axis_term(q.y, q.x, lft);
        lv.y = 0.0f + axis_term(q.z, q.y, q.x);
        lv.z = 0.0f + axis_term(q.w, q.z, q.y);
        lv.w = 0.0f + axis_term(rgt, q.w, q.z);
#pragma unroll
        for (int k = 0; k < NA; k++)
        {
          const float4 up = nb0.up[k * R + j], um = nb0.um[k * R + j];
          lv.x += axis_term(up.x, q.x, um.x);
          lv.y += axis_term(up.y, q.y, um.y);
          lv.z += axis_term(up.z, q.z, um.z);
          lv.w += axis_term(up.w, q.w, um.w);
        }
        {
          const float4 up = j + 1 < R ? pc[j + 1 < R ? j + 1 : j] : bp;
          const float4 um = j > 0 ? pc[j > 0 ? j - 1 : j] : bm;
          lv.x += axis_term(up.x, q.x, um.x);
          lv.y += axis_term(up.y, q.y, um.y);
          lv.z += axis_term(up.z, q.z, um.z);
          lv.w += axis_term(up.w, q.w, um.w);
        }
        lv.x += axis_term(pn[j].x, q.x, pm[j].x);
        lv.y += axis_term(pn[j].y, q.y, pm[j].y);
        lv.z += axis_term(pn[j].z, q.z, pm[j].z);
        lv.w += axis_term(pn[j].w, q.w, pm[j].w);
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
            store4<STREAM>(a.out + ij, lv);
        }
        else if constexpr (MODE == MODE_A)
        {
          if (act)
            store4<STREAM>(p_new + ij, q);
          acc = fmaf(q.x, lv.x, acc);
          acc = fmaf(q.y, lv.y, acc);
          acc = fmaf(q.z, lv.z, acc);
          acc = fmaf(q.w, lv.w, acc);
        }
        else
        {
          float4 xo = xv[j], ro = rv[j];
          if (xm == 2)
          {
            const float4 qq = QLATE ? load4<STREAM>(p_prev + ij) : qv[j];
            xo.x = fmaf(alpha_prev, qq.x, xo.x);
            xo.y = fmaf(alpha_prev, qq.y, xo.y);
            xo.z = fmaf(alpha_prev, qq.z, xo.z);
            xo.w = fmaf(alpha_prev, qq.w, xo.w);
          }
          xo.x = fmaf(alpha, q.x, xo.x);
          xo.y = fmaf(alpha, q.y, xo.y);
          xo.z = fmaf(alpha, q.z, xo.z);
          xo.w = fmaf(alpha, q.w, xo.w);
          ro.x = fmaf(-alpha, lv.x, ro.x);
          ro.y = fmaf(-alpha, lv.y, ro.y);
          ro.z = fmaf(-alpha, lv.z, ro.z);
          ro.w = fmaf(-alpha, lv.w, ro.w);
          if (act)
          {
            if (xm != 1)
              store4<STREAM>(a.x + ij, xo);
            store4<STREAM>(a.r + ij, ro);
            if constexpr (GHOST)
            {
              // boundary hyperplanes of the slab (the march runs along the slab axis)
              if (push_lo && y == 0)
                store4<false>(push_lo + base + j * sb, ro);
              if (push_hi && y + 1 == n1)
                store4<false>(push_hi + base + j * sb, ro);
            }
          }
          else
            ro = zero4;
          acc = fmaf(ro.x, ro.x, acc);
          acc = fmaf(ro.y, ro.y, acc);
          acc = fmaf(ro.z, ro.z, acc);
          acc = fmaf(ro.w, ro.w, acc);
        }
      }
#pragma unroll
      for (int j = 0; j < R; j++)
      {
        pm[j] = pc[j];
        pc[j] = pn[j];
        pn[j] = pnn[j];
        xv[j] = xn[j];
        rv[j] = rn[j];
        qv[j] = qn[j];
      }
      if constexpr (TILE)
      {
        t0 = t1r;
        t1r = t2r;
      }
      else if constexpr (SHALLOW)
        nb0 = nb2;
      else
      {
        nb0 = nb1;
        nb1 = nb2;
      }
    }
    if constexpr (MODE == MODE_A && GHOST)
    {
      // pc now holds row y1
      if (g.has_hi && y1 == n1 && act)
      {
#pragma unroll
        for (int j = 0; j < R; j++)
          store4<STREAM>(p_new + base + j * sb + y1 * rs, pc[j]);
      }
    }
  }
  if constexpr (MODE != MODE_STENCIL)
    block_write_partial(acc, a, sh);
}

} // namespace cgamd
