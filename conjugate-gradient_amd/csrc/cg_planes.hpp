// cg_planes.hpp -- the 3-D engine: workgroups march along the slowest axis.
//
// In 3-D the rows engine fetches the z-1 and z+1 neighbours of every point straight from
// memory, which triples the L2-side traffic of the stencil input (measured at 1024x1024x128:
// 5.8 and 6.4 words per point leave L2 in kernels A and B where 3 and 5 are needed).  Here
// each element of the stencil input is loaded once per work item:
//
//   * a work item is a tile of 256 points of axis 0 by TY = waves*R rows of axis 1, marched
//     along a segment of axis 2;
//   * a lane owns 4 consecutive points of axis 0 in R consecutive rows; planes z-1, z, z+1 of
//     its rows stay in registers (the z neighbours), rows y-1 / y+1 inside the R rows are the
//     lane's own registers, and the two rows just outside a wave's R rows come from the
//     neighbouring wave through LDS (one 16-byte write and read per lane, row and plane) or,
//     at the tile's edge, from memory;
//   * x neighbours are lane shifts, the two elements outside the wave's span a 2-lane load.
//
// One workgroup barrier per plane orders the LDS hand-off (double-buffered by plane parity).
// The modes, the cache policy and the iteration control are those of the rows engine.
#pragma once
#include "cg_device.hpp"

namespace cgamd
{

struct PlanesGeom
{
  unsigned n0, n1, n2;
  unsigned s0; // floats between consecutive rows: n0, or n0 rounded up to 4 (padded rows)
  unsigned nxc, nyt, nzs; // tiles along axis 0 (256 points), axis 1 (TY rows), segments of axis 2
  unsigned nitems;
  int has_lo, has_hi;
  // marching direction along axis 2 of kernels A and B (+1: ascending, -1: descending).  Opposite
  // directions let each kernel start where the other one stopped, on planes the Infinity Cache still holds.
  int dir_a, dir_b;
  // nitems / 8 when work items are dealt to the XCDs in contiguous runs (0: item = tile index).  Workgroup b runs
  // on XCD b % 8, so item i becomes tile (i % 8) * xcd_run + i / 8, tiles counted along axis 0 first, then axis 1,
  // then the segments of axis 2: the workgroups resident on one XCD hold tiles that are neighbours along axes 0
  // and 1, and the lines two tiles share (x edges, the rows just outside a tile) are fetched into that L2 once.
  unsigned xcd_run;
};

constexpr int CG_PLANES_MAX_WAVES = 8;

template <int MODE, bool GHOST, int POL, int R, bool ASC = true>
__global__ void __launch_bounds__(512) k_planes(const PlanesGeom g, const VecArgs a)
{
  __shared__ double sh[8];
  // [plane parity][wave][0 = first row, 1 = last row][lane]
  __shared__ float4 halo[2][CG_PLANES_MAX_WAVES][2][64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
  constexpr bool NT = POL != POL_PLAIN; // x and all stores
  constexpr bool NTS = NT;
  Src<MODE, POL> src;
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
  const long long n1 = g.n1, n2 = g.n2;
  const long long sy = g.s0, sz = sy * n1;
  const bool tail = (g.n0 & 3u) != 0; // padded rows: see k_rows
  const unsigned TY = (unsigned)nw * R;
  // planes are visited in order z = zs + dir * step; "behind" (pm) is z - dir, "ahead" (pn) is z + dir
  constexpr long long dir = ASC ? 1 : -1;
  constexpr bool asc = ASC;

  for (unsigned item = blockIdx.x; item < g.nitems; item += gridDim.x)
  {
    unsigned zseg, xc, yt;
    if (g.xcd_run)
    {
      unsigned t = (item & 7u) * g.xcd_run + (item >> 3);
      xc = t % g.nxc;
      t /= g.nxc;
      yt = t % g.nyt;
      zseg = t / g.nyt;
    }
    else
    {
      zseg = item % g.nzs;
      const unsigned t = item / g.nzs;
      xc = t % g.nxc;
      yt = t / g.nxc;
    }
    const long long z0 = (long long)((unsigned long long)zseg * g.n2 / g.nzs);
    const long long z1 = (long long)((unsigned long long)(zseg + 1) * g.n2 / g.nzs);
    if (z0 >= z1)
      continue;
    const unsigned x0 = xc * 256u + (unsigned)lane * 4u;
    const bool act = x0 < g.n0;
    const long long xcl = act ? x0 : ((g.n0 + 3u) & ~3u) - 4u;
    const long long yb = (long long)yt * TY + (long long)w * R; // this wave's first row
    const bool el = act && lane == 0 && x0 > 0;
    const bool er = act && lane == 63 && x0 + 4 < g.n0;
    const bool edge = el || er;
    const long long eoff = (lane == 0) ? -1 : 4;
    const bool rightmost = (lane == 63) || (x0 + 4 >= g.n0);

    auto zvalid = [&](long long z) { return GHOST || (z >= 0 && z < n2); };
    auto zaddr = [&](long long z) { return GHOST ? z : (z < 0 ? 0 : (z >= n2 ? n2 - 1 : z)); };
    // one row of the stencil input (zero outside the grid; ghost planes are real memory)
    auto ldrow = [&](long long y, long long z) -> float4 {
      const bool ok = y >= 0 && y < n1 && zvalid(z);
      const long long yc = y < 0 ? 0 : (y >= n1 ? n1 - 1 : y);
      const float4 v = src.ld4(zaddr(z) * sz + yc * sy + xcl);
      return sel4(ok, v);
    };
    auto ldplane = [&](long long z, float4(&out)[R]) {
#pragma unroll
      for (int j = 0; j < R; j++)
        out[j] = ldrow(yb + j, z);
    };
    auto ldedges = [&](long long z, float(&e)[R]) {
#pragma unroll
      for (int j = 0; j < R; j++)
      {
        e[j] = 0.0f;
        if (edge && yb + j < n1)
          e[j] = src.ld1(z * sz + (yb + j) * sy + xcl + eoff);
      }
    };

    float4 pm[R], pc[R], pn[R];
    float ec[R], en[R];
    float4 hu_ext = zero4, hd_ext = zero4; // rows outside the tile, for the first / last wave
    const long long zs = asc ? z0 : z1 - 1, nz = z1 - z0;
    // a plane just outside the slab that this rank must fill in p_new (a neighbouring rank owns it)
    auto is_ghost = [&](long long z) { return (z == -1 && g.has_lo) || (z == n2 && g.has_hi); };
    auto store_ghost = [&](long long z, const float4(&v)[R]) {
      if constexpr (MODE == MODE_A && GHOST)
      {
        if (is_ghost(z) && act)
        {
#pragma unroll
          for (int j = 0; j < R; j++)
            if (yb + j < n1)
              store4<NTS>(p_new + z * sz + (yb + j) * sy + xcl, v[j]);
        }
      }
    };
    ldplane(zs - dir, pm);
    ldplane(zs, pc);
    ldedges(zs, ec);
    if (w == 0)
      hu_ext = ldrow(yb - 1, zs);
    if (w == nw - 1)
      hd_ext = ldrow(yb + R, zs);
    halo[0][w][0][lane] = pc[0];
    halo[0][w][1][lane] = pc[R - 1];
    store_ghost(zs - dir, pm);
    __syncthreads();

    for (long long step = 0; step < nz; step++)
    {
      const long long z = zs + dir * step;
      const int par = (int)(step & 1);
      const bool more = step + 1 < nz;
      // ---- loads for the next plane and this plane's x, r ---------------------------------
      ldplane(z + dir, pn);
      float4 hu_n = zero4, hd_n = zero4;
#pragma unroll
      for (int j = 0; j < R; j++)
        en[j] = 0.0f;
      if (more)
      {
        ldedges(z + dir, en);
        if (w == 0)
          hu_n = ldrow(yb - 1, z + dir);
        if (w == nw - 1)
          hd_n = ldrow(yb + R, z + dir);
      }
      float4 xv[R], rv[R], qv[R];
      if constexpr (MODE == MODE_B)
      {
#pragma unroll
        for (int j = 0; j < R; j++)
        {
          const long long yc = yb + j < n1 ? yb + j : n1 - 1;
          xv[j] = qv[j] = zero4;
          if (xm != 1)
            xv[j] = load4<NT>(a.x + z * sz + yc * sy + xcl);
          if (xm == 2)
            qv[j] = load4<NT>(p_prev + z * sz + yc * sy + xcl);
          rv[j] = load4<false>(a.r + z * sz + yc * sy + xcl);
        }
      }
      // ---- rows just outside this wave's rows, plane z ------------------------------------
      const float4 hu = (w == 0) ? hu_ext : halo[par][w - 1][1][lane];
      const float4 hd = (w == nw - 1) ? hd_ext : halo[par][w + 1][0][lane];

#pragma unroll
      for (int j = 0; j < R; j++)
      {
        const bool rowok = act && (yb + j < n1);
        const long long idx = z * sz + (yb + j) * sy + xcl;
        const float4 c = pc[j];
        const float4 yu = (j + 1 < R) ? pc[(j + 1 < R) ? j + 1 : j] : hd; // y + 1
        const float4 yd = (j > 0) ? pc[(j > 0) ? j - 1 : j] : hu;         // y - 1
        const float4 zu = asc ? pn[j] : pm[j];                            // z + 1
        const float4 zd = asc ? pm[j] : pn[j];                            // z - 1
        float lft = __shfl_up(c.w, 1);
        float rgt = __shfl_down(c.x, 1);
        if (lane == 0)
          lft = ec[j];
        if (rightmost)
          rgt = (lane == 63) ? ec[j] : 0.0f;
        float4 lv;
        lv.x = 0.0f + axis_term(c.y, c.x, lft);
        lv.y = 0.0f + axis_term(c.z, c.y, c.x);
        lv.z = 0.0f + axis_term(c.w, c.z, c.y);
        lv.w = 0.0f + axis_term(rgt, c.w, c.z);
        lv.x += axis_term(yu.x, c.x, yd.x);
        lv.y += axis_term(yu.y, c.y, yd.y);
        lv.z += axis_term(yu.z, c.z, yd.z);
        lv.w += axis_term(yu.w, c.w, yd.w);
        lv.x += axis_term(zu.x, c.x, zd.x);
        lv.y += axis_term(zu.y, c.y, zd.y);
        lv.z += axis_term(zu.z, c.z, zd.z);
        lv.w += axis_term(zu.w, c.w, zd.w);
        if (!rowok)
          lv = zero4;
        if (tail)
        {
          if (x0 + 1 >= g.n0)
            lv.y = 0.0f;
          if (x0 + 2 >= g.n0)
            lv.z = 0.0f;
          if (x0 + 3 >= g.n0)
            lv.w = 0.0f;
        }

        if constexpr (MODE == MODE_STENCIL)
        {
          if (rowok)
            store4<false>(a.out + idx, lv);
        }
        else if constexpr (MODE == MODE_A)
        {
          if (rowok)
            store4<NTS>(p_new + idx, c);
          acc = fmaf(c.x, lv.x, acc);
          acc = fmaf(c.y, lv.y, acc);
          acc = fmaf(c.z, lv.z, acc);
          acc = fmaf(c.w, lv.w, acc);
        }
        else
        {
          float4 xo = xv[j], ro = rv[j];
          if (xm == 2)
          {
            xo.x = fmaf(alpha_prev, qv[j].x, xo.x);
            xo.y = fmaf(alpha_prev, qv[j].y, xo.y);
            xo.z = fmaf(alpha_prev, qv[j].z, xo.z);
            xo.w = fmaf(alpha_prev, qv[j].w, xo.w);
          }
          xo.x = fmaf(alpha, c.x, xo.x);
          xo.y = fmaf(alpha, c.y, xo.y);
          xo.z = fmaf(alpha, c.z, xo.z);
          xo.w = fmaf(alpha, c.w, xo.w);
          ro.x = fmaf(-alpha, lv.x, ro.x);
          ro.y = fmaf(-alpha, lv.y, ro.y);
          ro.z = fmaf(-alpha, lv.z, ro.z);
          ro.w = fmaf(-alpha, lv.w, ro.w);
          if (rowok)
          {
            if (xm != 1)
              store4<NT>(a.x + idx, xo);
            store4<NTS>(a.r + idx, ro);
            if constexpr (GHOST)
            {
              // boundary planes of the slab: also into the neighbours' staging areas
              if (push_lo && z == 0)
                store4<false>(push_lo + (yb + j) * sy + xcl, ro);
              if (push_hi && z + 1 == n2)
                store4<false>(push_hi + (yb + j) * sy + xcl, ro);
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
      // ---- hand this wave's boundary rows of the next plane to its neighbours ---------------
      if (more)
      {
        halo[par ^ 1][w][0][lane] = pn[0];
        halo[par ^ 1][w][1][lane] = pn[R - 1];
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < R; j++)
      {
        pm[j] = pc[j];
        pc[j] = pn[j];
        ec[j] = en[j];
      }
      hu_ext = hu_n;
      hd_ext = hd_n;
    }
    // pc now holds the plane just past the end of the march
    store_ghost(zs + dir * nz, pc);
  }
  if constexpr (MODE != MODE_STENCIL)
    block_write_partial(acc, a, sh);
}

} // namespace cgamd
