// cg_solver.hip -- host side of the fused CG iteration (see cg_solver.hpp, cg_device.hpp).
#include "cg_solver.hpp"

#include <assert.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "hip_check.hpp"

namespace cgamd
{

// ---- small kernels -------------------------------------------------------------------------

// per-block partials of v.w; 16-byte loads when both pointers allow it.  Four independent
// 16-byte loads per vector are in flight per lane and feed four accumulators that are added in a
// fixed order at the end, so the result depends on the launch shape only (the callers fix it).
__global__ void __launch_bounds__(256) k_dot(const float *__restrict__ v, const float *__restrict__ w,
                                             unsigned long long n, int vec4, double *__restrict__ part_out)
{
  __shared__ double sh[8];
  float acc = 0;
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec4)
  {
    const unsigned long long n4 = n >> 2;
    const float4 *v4 = reinterpret_cast<const float4 *>(v);
    const float4 *w4 = reinterpret_cast<const float4 *>(w);
    const bool same = v == w;
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    unsigned long long j = i;
    for (; j + 3 * step < n4; j += 4 * step)
    {
      const float4 p0 = load4<true>(v + 4 * j), p1 = load4<true>(v + 4 * (j + step));
      const float4 p2 = load4<true>(v + 4 * (j + 2 * step)), p3 = load4<true>(v + 4 * (j + 3 * step));
      const float4 q0 = same ? p0 : load4<true>(w + 4 * j), q1 = same ? p1 : load4<true>(w + 4 * (j + step));
      const float4 q2 = same ? p2 : load4<true>(w + 4 * (j + 2 * step));
      const float4 q3 = same ? p3 : load4<true>(w + 4 * (j + 3 * step));
      a0 = fmaf(p0.w, q0.w, fmaf(p0.z, q0.z, fmaf(p0.y, q0.y, fmaf(p0.x, q0.x, a0))));
      a1 = fmaf(p1.w, q1.w, fmaf(p1.z, q1.z, fmaf(p1.y, q1.y, fmaf(p1.x, q1.x, a1))));
      a2 = fmaf(p2.w, q2.w, fmaf(p2.z, q2.z, fmaf(p2.y, q2.y, fmaf(p2.x, q2.x, a2))));
      a3 = fmaf(p3.w, q3.w, fmaf(p3.z, q3.z, fmaf(p3.y, q3.y, fmaf(p3.x, q3.x, a3))));
    }
    for (; j < n4; j += step)
    {
      const float4 p0 = v4[j], q0 = w4[j];
      a0 = fmaf(p0.w, q0.w, fmaf(p0.z, q0.z, fmaf(p0.y, q0.y, fmaf(p0.x, q0.x, a0))));
    }
    acc = (a0 + a1) + (a2 + a3);
    for (unsigned long long k = (n4 << 2) + i; k < n; k += step)
      acc = fmaf(v[k], w[k], acc);
  }
  else
  {
    for (unsigned long long j = i; j < n; j += step)
      acc = fmaf(v[j], w[j], acc);
  }
  block_write_partial(acc, part_out, sh);
}

// out[0] = sum(part[0..n)) in the same order block_read_sum uses
__global__ void __launch_bounds__(64) k_sum_partials(const double *__restrict__ part, int n, double *__restrict__ out)
{
  __shared__ double sh[8];
  const double s = block_read_sum(part, n, sh);
  if (threadIdx.x == 0)
    out[0] = s;
}

// Start of a solve: iteration 0, reltol = rel * |b| with |b| = sqrt(b.b) rounded to float the
// way conjugate-gradient_gpu.cu:267-270,333 does it.
__global__ void __launch_bounds__(64) k_init_state(CgState *st, const double *__restrict__ sumB, int nB, double rel,
                                                   int lagged, int maxit)
{
  __shared__ double sh[8];
  const float rr = (float)block_read_sum(sumB, nB, sh);
  if (threadIdx.x == 0)
  {
    const float nrm = (float)sqrt((double)rr);
    st->itA = 0;
    st->doneA = 0;
    st->itB = 0;
    st->doneB = 0;
    st->activeB = 0;
    st->maxit = maxit;
    st->lagged = lagged;
    st->reltol = (float)(rel * (double)nrm);
    st->alpha = 0;
    st->beta = 0;
    st->pAp = 0;
    st->xdef_k = -2;
    st->xflushed_at = -1;
    st->alpha_hist[0] = st->alpha_hist[1] = 0;
    st->rr_hist[0] = rr;
    st->rr_pp[0] = rr;
    st->rr_pp[1] = 0;
  }
}

// Paired x update (CgState::xdef_k): applies the update the last kernel B left out, if any.  The decision
// is read from words no workgroup of this launch writes; k_flush_mark records it afterwards.
// The pending update belongs to iteration j = itA - 1: its p is the buffer kernel B of iteration j read.
__global__ void __launch_bounds__(256) k_flush_x(const CgState *__restrict__ st, float *__restrict__ x,
                                                 const float *__restrict__ p0, const float *__restrict__ p1,
                                                 unsigned long long n, int vec4)
{
  const int it = st->itA, j = it - 1;
  if (st->xdef_k != j || st->xflushed_at == it)
    return;
  const float alpha = st->alpha_hist[j & 1];
  const float *__restrict__ p = (j & 1) ? p0 : p1;
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec4)
  {
    for (unsigned long long i = t; i < n / 4; i += step)
    {
      float4 xv = reinterpret_cast<float4 *>(x)[i];
      const float4 pv = reinterpret_cast<const float4 *>(p)[i];
      xv.x = fmaf(alpha, pv.x, xv.x);
      xv.y = fmaf(alpha, pv.y, xv.y);
      xv.z = fmaf(alpha, pv.z, xv.z);
      xv.w = fmaf(alpha, pv.w, xv.w);
      reinterpret_cast<float4 *>(x)[i] = xv;
    }
    for (unsigned long long i = n / 4 * 4 + t; i < n; i += step)
      x[i] = fmaf(alpha, p[i], x[i]);
  }
  else
    for (unsigned long long i = t; i < n; i += step)
      x[i] = fmaf(alpha, p[i], x[i]);
}

__global__ void __launch_bounds__(64) k_flush_mark(CgState *st)
{
  const int it = st->itA;
  if (threadIdx.x == 0 && st->xdef_k == it - 1)
    st->xflushed_at = it;
}

__global__ void __launch_bounds__(64) k_set_maxit(CgState *st, int add)
{
  if (threadIdx.x == 0)
    st->maxit = st->itA + add;
}

// r's first / last owned hyperplane as kernel B of the coming iteration will leave it, computed ahead of that
// kernel from a copy of the planes: edge[side][j] = fma(-alpha, (A p)[point], edge[side][j]) with the stencil
// terms added in axis order and alpha formed exactly as kernel B's prologue forms it, so the bits are kernel
// B's.  Reads p (which kernel B only reads) and words of the state kernel A wrote; never touches r.
// Nothing happens when the iteration is inactive: the copy then is what r keeps.
struct EdgeArgs
{
  const float *p0, *p1;
  const CgState *st;
  const double *sumA;
  int nA;
  float *edge;
  unsigned long long plane, pad;
};

__global__ void __launch_bounds__(256) k_edge_r(const ScalarGeom g, const EdgeArgs e)
{
  __shared__ double sh[8];
  const CgState *st = e.st;
  const int k = st->itB;
  const bool active = st->activeB != 0;
  const float pAp = (float)block_read_sum(e.sumA, e.nA, sh);
  if (!active)
    return;
  const float alpha = st->rr_hist[k % CG_HIST] / pAp;
  const float *__restrict__ p = (k & 1) ? e.p0 : e.p1;
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (int side = 0; side < 2; side++)
  {
    if (side == 0 ? !g.has_lo : !g.has_hi)
      continue;
    const unsigned long long first = side == 0 ? 0ull : g.N - e.plane;
    float *__restrict__ out = e.edge + (size_t)side * e.pad;
    for (unsigned long long j = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; j < e.plane; j += step)
    {
      const long long i = (long long)(first + j);
      const float c = p[i];
      float lv = 0.0f;
      unsigned rem = (unsigned)i;
      for (int a = 0; a < g.d; a++)
      {
        const unsigned nk = g.n[a];
        const unsigned ck = rem % nk;
        rem /= nk;
        const long long s = (long long)g.stride[a];
        const bool last = a == g.d - 1; // the slab axis: ghost planes hold the neighbours' values
        const bool hp = (ck + 1 < nk) || (last && g.has_hi);
        const bool hm = (ck > 0) || (last && g.has_lo);
        const float up = hp ? p[i + s] : 0.0f;
        const float um = hm ? p[i - s] : 0.0f;
        lv = lv + axis_term(up, c, um);
      }
      out[j] = fmaf(-alpha, lv, out[j]);
    }
  }
}

__global__ void __launch_bounds__(256) k_fill_uniform(float *__restrict__ dst, unsigned long long n, unsigned int seed)
{
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step)
    dst[i] = u01_hash(seed, i);
}


static int env_int(const char *name, int dflt)
{
  const char *s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}

static int max_blocks()
{
  int mb = env_int("CG_MAX_BLOCKS", CG_MAX_BLOCKS);
  return std::max(1, std::min(mb, 2 * CG_MAX_BLOCKS));
}

// Number of workgroups the chip keeps resident at once for the rows kernels: CUs times
// workgroups per CU (kernel B needs 82 VGPRs: 5 workgroups of 256 fit; 4 measured fastest).  A grid of exactly this size runs as one round with no partial second round.
static int resident_blocks(int per_cu_override = 0)
{
  static int cus = 0;
  if (!cus)
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    else
      cus = 256;
    (void)hipGetLastError();
  }
  const int per_cu = per_cu_override > 0 ? per_cu_override : std::max(1, env_int("CG_BLOCKS_PER_CU", 4));
  return std::min(cus * per_cu, max_blocks());
}

static bool is_pow2(unsigned v) { return v && !(v & (v - 1)); }

// rows of at most a wave's 256 points, which the rows engine packs several to a workgroup (a row takes
// n0/4 lanes rounded up to a power of two, so at least half of the lanes work).  Down to rows of 8 and
// of 4 points (one lane per row) this beats the one-point-per-lane engine two to one (8^8: kernels A / B
// 107 / 98 against 229 / 182 us; 4^12: 160 / 135 against 308 / 223).
static bool rows_packable(int d, unsigned n0) { return d >= 3 && n0 >= (unsigned)env_int("CG_PACK_MIN", 4) && n0 <= 256; }

// Segments per column of a march (rows engine, planes engine): the work items should fill the resident
// workgroups in whole rounds (a last round that is nearly empty, or fewer items than the chip holds,
// leaves CUs idle: 96^3 x 192 with 288 column groups on 512 slots ran kernels A / B in 803 / 884 us, cut
// into 4 to 8 segments in 575 / 623), while every segment pays for two extra rows and the start of its
// pipeline.  `length` rows (planes) along the march, at least `shortest` per segment; `cost`: what a
// segment pays, in rows (planes) of the march.
static unsigned long long choose_segments(unsigned long long ncols, unsigned long long target, unsigned length,
                                          unsigned shortest, double cost)
{
  // more columns than resident workgroups: the dispatcher balances the rounds by itself (48^5, 1728
  // column groups: cutting them in two cost 9 %)
  if (ncols >= target)
    return 1;
  const unsigned long long most = std::max(1u, length / shortest);
  const unsigned long long last = std::min(most, std::max<unsigned long long>(1, (4 * target + ncols - 1) / ncols));
  unsigned long long segs = 1;
  double best = -1;
  for (unsigned long long c = 1; c <= last; c++)
  {
    const unsigned long long n = ncols * c, rounds = (n + target - 1) / target;
    const double fill = (double)n / (double)(rounds * target);
    const double score = fill / (1.0 + cost / (double)(length / c));
    if (score > best * 1.0001)
    {
      best = score;
      segs = c;
    }
  }
  return segs;
}

// How much of r a solve keeps in the Infinity Cache (256 MiB on MI355X).  Measured on 2-D grids beyond the cache,
// share of r kept against iterations/s: 16384 x 8192 (512 MiB per vector) 0 % 1330 (all of r streaming in kernel
// A, all of it "kept" in B: nothing survives a sweep), 40 % 1404, 50 % 1444, 55 % 1422, 60 % 1408; 12288 x 8192
// (384 MiB) 0 % 1540, 50 % 1722, 65 % 1778, 70 % 1672; 10240 x 8192 (320 MiB) 0 % 2105, 80 % 2400, 90 % 2064;
// 16384^2 (1 GiB) 0 % 641, 25 % 670, 30 % 659: the best share is always the cache's size, and a little more
// costs more than a little less.
static unsigned long long keep_bytes() { return (unsigned long long)std::max(0, env_int("CG_KEEP_MB", 256)) << 20; }

// columns x segments decomposition of the rows engine (see cg_device.hpp)
static void make_rows_geom(int d, const unsigned *n, int has_lo, int has_hi, RowsGeom &g, int &block, int &grid,
                           unsigned s0 = 0, int per_cu = 0, bool for_kernel_a = false)
{
  memset(&g, 0, sizeof(g));
  g.d = d;
  g.n0 = n[0];
  g.s0 = s0 ? s0 : n[0]; // row stride (padded rows: n0 rounded up to 4)
  g.n1 = d >= 2 ? n[1] : 1;
  // a long 1-D grid marches through "rows" of up to 1024 points (see ldedge in k_rows): consecutive rows
  // of a segment are contiguous memory, and the next row's loads are in flight while one is computed.
  // The row length is the largest multiple of 4 from 1024 down to 512 that divides the grid (the
  // reference's own driver solves 100000 x 312 points, main.cu:40-50: rows of 1000).
  unsigned folded[2] = {0u, 0u};
  if (d == 1 && n[0] >= (1u << 20) && env_int("CG_ROWS_FOLD_1D", 1))
    for (unsigned w = 1024u; w >= 512u && !folded[0]; w -= 4u)
      if (n[0] % w == 0)
        folded[0] = w;
  if (folded[0])
  {
    folded[1] = n[0] / folded[0];
    g.n0 = folded[0];
    g.s0 = folded[0];
    g.n1 = folded[1];
    n = folded;
  }
  const unsigned lanes = (n[0] + 3) / 4;
  block = (int)std::min(256u, (lanes + 63) / 64 * 64);
  block = std::max(64, std::min(256, env_int("CG_ROWS_BLOCK", block) / 64 * 64)); // k_rows is bounded to 256 threads
  g.nxc = (lanes + block - 1) / block;
  unsigned long long ncols = g.nxc, stride = (unsigned long long)g.s0 * g.n1;
  for (int k = 2; k < d; k++)
  {
    g.nk[k - 2] = n[k];
    g.sk[k - 2] = stride;
    stride *= n[k];
    ncols *= n[k];
  }
  // rows shorter than a wave: several columns side by side in one workgroup, marching along the
  // last axis (see k_rows, PACK): columns are numbered along axes 1 .. d-2
  if (rows_packable(d, n[0]) && env_int("CG_ROWS_PACK", 1))
  {
    unsigned long long nc = 1, st = g.s0;
    unsigned sub = 1;
    while (sub < lanes)
      sub *= 2;
    for (int k = 1; k + 1 < d; k++)
    {
      g.nk[k - 1] = n[k];
      g.sk[k - 1] = st;
      st *= n[k];
      nc *= n[k];
    }
    if (nc < (1ull << 32))
    {
      g.sy = st;            // stride of axis d-1
      g.n1 = n[d - 1];
      // 512 lanes (16 columns of 128 points: 8 KiB of consecutive memory per step of the march) measured
      // best: 128^4 kernels A / B 1189 / 1140 us with 256 lanes, 1021 / 1087 with 512, 1289 / 1136 with
      // 1024; rows of 16 points are better off with 256 (16^6: 82 / 82 against 90 / 84)
      block = std::max(256, std::min(1024, env_int("CG_ROWS_PACK_BLOCK", n[0] >= 32 ? 512 : 256) / 256 * 256));
      g.sub = sub;
      g.cpg = (unsigned)block / sub;
      g.ncols = (unsigned)nc;
      g.nxc = 1;
      // d = 3, 4, 5: two columns per lane, neighbours along axis d-2 (k_cols, cg_cols.hpp); that axis leaves
      // the column numbering, which then runs over axes 1 .. d-3 and the blocks of two.  Measured,
      // kernels A / B in us against one column per lane: 128^4 781 / 946 against 1022 / 1089,
      // 64x64x64x512 410 / 534 against 482 / 549, 64^4 53 / 65 against 59 / 69, 64^4 x 16 1283 / 1281
      // against 1956 / 1446, 128 x 1024 x 1024 313 / 483 against 452 / 525; small 3-D and 5-D grids gain
      // nothing.  Four columns per lane spill.
      unsigned long long N = 1;
      for (int k = 0; k < d; k++)
        N *= n[k];
      const bool big = N >= (1ull << 25); // the vectors no longer fit the Infinity Cache together
      int R = env_int("CG_COLS_R", (d == 4 || ((d == 3 || d == 5) && big)) ? 2 : 0);
      R = R >= 2 ? 2 : R >= 1 ? 1 : 0;
      if (R == 2 && n[d - 2] % 2u)
        R = env_int("CG_COLS_ODD", 1); // an odd extent leaves one column per lane (1), or k_rows (0)
      if (R && d <= 5 && n[d - 2] % (unsigned)R == 0)
      {
        g.R = R;
        g.nb = n[d - 2];
        g.sb = g.sk[d - 3];
        g.ncols = (unsigned)(nc / R);
        nc = g.ncols;
        block = std::min(block, 512); // k_cols is bounded to 512 lanes (tiled slots: see below)
        g.cpg = (unsigned)block / sub;
        g.stream = env_int("CG_COLS_STREAM", big ? 1 : 0);
        // neighbours along axis 1 through LDS (see k_cols): kernel B always, kernel A where it measured faster
        g.lds_b = env_int("CG_COLS_LDS_B", 1);
        g.lds_a = env_int("CG_COLS_LDS_A", (d == 4 && sub == lanes) ? 1 : 0);
        // rows that leave more than a tenth of their power-of-two slot idle take exactly n0/4 lanes, on large
        // grids in d <= 4 (kernels A / B: 100^4 387 / 390 -> 345 / 358 us, 80^4 153 / 175 -> 144 / 148, 96^4
        // 313 / 340 -> 317 / 302; 48^5 loses 8 %, grids of a few million points up to 20 % in kernel A)
        if (n[0] % 4 == 0 && lanes * 10 < sub * 9 && env_int("CG_COLS_EXACT", (d <= 4 && big) ? 1 : 0))
        {
          g.xl = 1;
          g.lds_a = g.lds_b = 1;
          g.sub = lanes;
          g.cpg = (unsigned)block / lanes;
        }
      }
      ncols = (nc + g.cpg - 1) / g.cpg; // column groups
      // 4-D with the neighbours through LDS: the slots of a workgroup form a patch of axis 1 x blocks of axis 2
      // (t1 x t2) instead of a strip along axis 1, so that half (t2 = 2) of the block-edge neighbours along
      // axis 2 are rows of the same workgroup too.  For kernel A of large grids, which fetches every neighbour
      // from two arrays (128^4: 662 -> 614 us); kernel B, whose neighbour loads mostly hit L2 anyway, loses
      // (797 -> 847 us) and keeps the strip.
      g.t2 = 1;
      const bool wide = for_kernel_a && g.R == 2 && d == 4 && g.lds_a && g.lds_b && env_int("CG_COLS_BLOCK_A", 512) >= 1024;
      if (wide)
      {
        block = 1024; // the tiled kernel has a 1024-lane variant
        g.cpg = (unsigned)block / g.sub;
        ncols = (nc + g.cpg - 1) / g.cpg;
      }
      if (g.R == 2 && d == 4 && g.lds_a && g.lds_b)
      {
        unsigned t2 = (unsigned)std::max(1, for_kernel_a ? env_int("CG_COLS_T2", big ? 2 : 1) : env_int("CG_COLS_T2_B", 1));
        // exact rows leave an odd number of slots in a workgroup now and then: give one up (96^4: 21 -> 20)
        if (t2 == 2 && g.cpg % 2 != 0 && g.cpg >= 5)
        {
          g.cpg -= 1;
          ncols = (nc + g.cpg - 1) / g.cpg;
        }
        while (t2 > 1 && (g.cpg % t2 != 0 || g.cpg / t2 < 2))
          t2--;
        if (t2 == 1 && wide)
          t2 = 2; // the strip kernel is bounded to 512 lanes
        if (t2 > 1)
        {
          g.t2 = t2;
          const unsigned long long t1 = g.cpg / t2, nblk = g.nb / (unsigned)g.R;
          ncols = ((n[1] + t1 - 1) / t1) * ((nblk + t2 - 1) / t2);
          const unsigned long long nt1 = (n[1] + t1 - 1) / t1;
          // patches of one XCD side by side along axis 1 (128^4, kernel A: 598 -> 587 us, 2.79 -> 2.54 GB fetched)
          g.xcd = (nt1 % 8 == 0 && env_int("CG_COLS_XCD", 1)) ? (unsigned)(nt1 / 8) : 0u;
        }
      }
    }
  }
  // scale the resident-workgroup target by the workgroup size (it is counted in 256-thread units)
  const unsigned long long target = block > 256 ? (unsigned long long)resident_blocks(per_cu) * 256 / block
                                                : (unsigned long long)resident_blocks(per_cu) * (256 / block);
  unsigned long long bpc = choose_segments(ncols, target, g.n1, 4, 2.5); // at least ~4 rows per segment
  bpc = (unsigned long long)env_int("CG_ROWS_BPC", (int)bpc);
  bpc = std::max<unsigned long long>(1, std::min<unsigned long long>(bpc, g.n1));
  const unsigned long long items = ncols * bpc;
  CG_REQUIRE(items < (1ull << 32), "too many work items for the rows engine");
  g.bpc = (unsigned)bpc;
  g.nitems = (unsigned)items;
  if (bpc != 1)
    g.xcd = 0; // the XCD of a work item is only known when items are whole columns
  g.has_lo = has_lo;
  g.has_hi = has_hi;
  // POL_PART (1-D and 2-D): the rows whose r the Infinity Cache keeps
  g.keep_y = (unsigned)std::min<unsigned long long>(keep_bytes() / ((unsigned long long)g.s0 * sizeof(float)), g.n1);
  grid = (int)std::min<unsigned long long>(items, std::min<unsigned long long>(target, (unsigned long long)max_blocks()));
}

// tiles x segments decomposition of the planes engine (see cg_planes.hpp)
static void make_planes_geom(const unsigned *n, int has_lo, int has_hi, PlanesGeom &g, int &block, int &grid, int &R,
                             unsigned s0 = 0, int per_cu = 0)
{
  memset(&g, 0, sizeof(g));
  g.n0 = n[0];
  g.s0 = s0 ? s0 : n[0];
  g.n1 = n[1];
  g.n2 = n[2];
  R = env_int("CG_PLANES_R", 2) >= 4 ? 4 : 2;
  int nw = env_int("CG_PLANES_WAVES", 8);
  nw = std::max(1, std::min(nw, CG_PLANES_MAX_WAVES));
  nw = (int)std::min<unsigned>((unsigned)nw, (n[1] + R - 1) / R);
  block = nw * 64;
  const unsigned ty = (unsigned)nw * R;
  g.nxc = (n[0] + 255) / 256;
  g.nyt = (n[1] + ty - 1) / ty;
  const unsigned long long ncols = (unsigned long long)g.nxc * g.nyt;
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
    cus = prop.multiProcessorCount;
  (void)hipGetLastError();
  const unsigned long long target = std::min<unsigned long long>(
      (unsigned long long)cus * (per_cu > 0 ? per_cu : std::max(1, env_int("CG_PLANES_BPCU", 2))), max_blocks());
  // keep segments >= 8 planes; a segment costs more here than in the rows engine (three planes to start,
  // the LDS hand-off): 640^3 loses 8 % in kernel A with 17 segments instead of 4, 768^3 gains 4 % with 7
  // instead of 3
  unsigned long long nzs = choose_segments(ncols, target, n[2], 8, 6.0);
  nzs = (unsigned long long)env_int("CG_PLANES_NZS", (int)nzs);
  nzs = std::max<unsigned long long>(1, std::min<unsigned long long>(nzs, n[2]));
  g.nzs = (unsigned)nzs;
  const unsigned long long items = ncols * nzs;
  CG_REQUIRE(items < (1ull << 32), "too many work items for the planes engine");
  g.nitems = (unsigned)items;
  g.has_lo = has_lo;
  g.has_hi = has_hi;
  // Kernel A marches up, kernel B down: each starts on the planes the other touched last, part of which the
  // 256 MiB Infinity Cache still holds.  Measured (kernel A us, B unchanged): 1024 x 1024 x 128 320 -> 303,
  // 512^3 331 -> 319, 256^3 39.7 -> 38.3; nothing to gain once a vector is many times the cache (1024^3,
  // 1000^3: +-1 %), so large grids keep both marches ascending.
  const unsigned long long npts = (unsigned long long)n[0] * n[1] * n[2];
  g.dir_a = env_int("CG_PLANES_DIR_A", 1) < 0 ? -1 : 1;
  g.dir_b = env_int("CG_PLANES_DIR_B", npts <= (1ull << 28) ? -1 : 1) < 0 ? -1 : 1;
  grid = (int)std::min<unsigned long long>(items, target);
  // Tiles dealt to the XCDs in contiguous runs (PlanesGeom::xcd_run); the rule needs item % 8 to be the XCD of
  // the workgroup that takes the item.  Both kernels or neither: 1024 x 1024 x 128 gains only when kernel B leaves
  // its lines in the L2 kernel A will read them from (A 302 -> 296 us and 1.32 -> 1.12 GB fetched, 1.07 needed; B
  // 426 -> 418 us, 1.73 -> 1.64 GB; with kernel A alone nothing changes); 384^3 2650 -> 2720 it/s, 640^3 606 ->
  // 613, 2048 x 1024 x 128 720 -> 730.  Only when every item is resident at once: with more items than
  // workgroups kernel B loses what kernel A gains or more (768^3: B 1581 -> 1615 us).
  if (items % 8 == 0 && grid % 8 == 0 && env_int("CG_PLANES_XCD", items <= (unsigned long long)grid))
    g.xcd_run = (unsigned)(items / 8);
}

// contiguous Z-curve ranges, one per workgroup
static void make_morton_geom(int d, unsigned long long N, MortonGeom &g, int &block, int &grid)
{
  memset(&g, 0, sizeof(g));
  g.d = d;
  g.N = N;
  for (int k = 0; k < d; k++)
    g.mmask[k] = morton_mask(d, k);
  g.interleave = env_int("CG_MORTON_INTERLEAVE", 1);
  block = 256;
  const unsigned long long quads = N / 4;
  const unsigned long long want = (quads + block - 1) / block;
  grid = (int)std::max<unsigned long long>(1, std::min<unsigned long long>(want, (unsigned long long)std::min(env_int("CG_MORTON_BLOCKS", CG_MAX_BLOCKS), 2 * CG_MAX_BLOCKS)));
  // with at least one LDS tile per workgroup the tiled kernel is used: never more workgroups than tiles
  if (N % CG_MORTON_TILE == 0 && N / CG_MORTON_TILE >= 64)
    grid = (int)std::min<unsigned long long>((unsigned long long)grid, N / CG_MORTON_TILE);
}

size_t Solver::plane_of(const SolverDesc &d)
{
  size_t p = 1;
  for (int k = 0; k + 1 < d.d; k++)
    p *= d.extents[k];
  return p;
}

size_t Solver::ghost_offset_of(const SolverDesc &d)
{
  // first owned element sits on a 256-byte boundary, the lo ghost plane directly below it
  const size_t p = plane_of(d);
  return (p + 63) / 64 * 64;
}

size_t Solver::buffer_len_of(const SolverDesc &d)
{
  size_t n = 1;
  for (int k = 0; k < d.d; k++)
    n *= d.extents[k];
  return ghost_offset_of(d) + n + plane_of(d) + 64;
}

Solver::Solver(const SolverDesc &desc) : desc_(desc)
{
  CG_REQUIRE(desc.d >= 1 && desc.d <= CG_MAXD, "dimension must be in [1,32]");
  N_ = 1;
  for (int k = 0; k < desc.d; k++)
  {
    CG_REQUIRE(desc.extents[k] >= 1, "grid extents must be positive");
    N_ *= desc.extents[k];
  }
  CG_REQUIRE(N_ < (1ull << 32), "grid must have fewer than 2^32 points");
  layout_ = desc.index_mode == 2 ? LAYOUT_MORTON : LAYOUT_LEX;
  CG_REQUIRE(desc.index_mode >= 0 && desc.index_mode <= 2, "index_mode must be 0, 1 (lexicographic) or 2 (Morton)");
  // A hypercube of two points per axis has the same index in both layouts (bit k = coordinate k): the
  // quad-per-lane Morton kernels serve it (2^24: kernels A / B 485 / 404 -> 205 / 154 us)
  if (layout_ == LAYOUT_LEX && desc.d >= 2 && !desc.has_lo && !desc.has_hi && desc.engine == ENGINE_AUTO &&
      env_int("CG_ENGINE", 0) == 0)
  {
    bool twos = true;
    for (int k = 0; k < desc.d; k++)
      twos = twos && desc.extents[k] == 2;
    if (twos)
      layout_ = LAYOUT_MORTON;
  }
  CG_REQUIRE(!(desc.has_lo || desc.has_hi) || (layout_ == LAYOUT_LEX && desc.d >= 2),
             "slab neighbours need the lexicographic layout and d >= 2");
  // padded rows (see cg_solver.hpp): only for stand-alone grids with the automatic engine choice,
  // large enough to be bandwidth-bound, and not when the caller owns the buffers
  s0_ = desc.extents[0];
  Np_ = N_;
  // Row stride of the internal vectors: a multiple of 4 floats for the 16-byte engines; in 3-D with rows
  // of at least 256 points (the planes engine) a multiple of 32 floats, so that every row starts on a
  // 128-byte line, where that costs at most 4 % more memory (1000^3: kernels A / B 2489 / 4337 us, with
  // rows of 1024 floats 2295 / 3705 us although 2.4 % more bytes move; 500^3: 334 / 506 -> 317 / 461).
  unsigned stride0 = desc.extents[0];
  // (packed short rows: from 12 points on; 7^7 padded to rows of 8 lost a third to the scalar engine)
  if (stride0 % 4 != 0 && (stride0 >= 128 || (rows_packable(desc.d, stride0) && stride0 >= 12)))
    stride0 = (stride0 + 3u) & ~3u;
  if (desc.d == 3 && desc.extents[0] >= 256 && desc.extents[0] % 32 != 0 &&
      ((desc.extents[0] + 31u) & ~31u) - desc.extents[0] <= desc.extents[0] / 25 && env_int("CG_PAD_ROWS", 1) != 4)
    stride0 = (desc.extents[0] + 31u) & ~31u;
  pad_ = layout_ == LAYOUT_LEX && desc.engine == ENGINE_AUTO && !desc.has_lo && !desc.has_hi && !desc.r_buf &&
         !desc.p0_buf && !desc.p1_buf && stride0 != desc.extents[0] && N_ >= (1ull << 19) &&
         desc.d <= CG_ROWS_MAXD + 2 && env_int("CG_ENGINE", 0) == 0 && env_int("CG_PAD_ROWS", 1);
  // A Morton-ordered grid is solved in lexicographic order inside the solver: b and x are permuted in
  // at begin_local() and x is permuted back by finish(), as for padded rows.  The lexicographic engines
  // move fewer bytes per iteration (128^4: 619 against 470 iterations/s) and the stencil's bits do not
  // depend on the layout, so only the order of summation inside the dot products differs.
  relay_ = false;
  if (layout_ == LAYOUT_MORTON && desc.engine == ENGINE_AUTO && !desc.r_buf && !desc.p0_buf && !desc.p1_buf &&
      desc.d >= 2 && N_ >= (1ull << 19) && env_int("CG_ENGINE", 0) == 0 && env_int("CG_MORTON_RELAYOUT", 1))
  {
    bool ok = is_pow2(desc.extents[0]) && desc.extents[0] >= 4;
    int bits = 0;
    for (unsigned v = desc.extents[0]; v > 1; v >>= 1)
      bits++;
    for (int k = 1; k < desc.d; k++)
      ok = ok && desc.extents[k] == desc.extents[0];
    if (ok && bits * desc.d <= 32)
    {
      relay_ = true;
      layout_ = LAYOUT_LEX;
    }
  }
  SolverDesc sized = desc;
  if (pad_)
  {
    s0_ = stride0;
    Np_ = N_ / desc.extents[0] * s0_;
    CG_REQUIRE(Np_ < (1ull << 32), "padded grid must have fewer than 2^32 points");
    sized.extents[0] = s0_;
  }
  plane_ = plane_of(sized);
  goff_ = ghost_offset_of(sized);
  buflen_ = buffer_len_of(sized);

  if (desc.stream)
    stream_ = desc.stream;
  else
  {
    CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    own_stream_ = true;
  }
  auto get = [&](float *given, float *&base, bool &own) {
    if (given)
      base = given;
    else
    {
      CHECK(hipMalloc((void **)&base, buflen_ * sizeof(float)));
      own = true;
    }
  };
  get(desc.r_buf, r_base_, own_r_);
  get(desc.p0_buf, p0_base_, own_p0_);
  get(desc.p1_buf, p1_base_, own_p1_);
  r_ = r_base_ + goff_;
  p0_ = p0_base_ + goff_;
  p1_ = p1_base_ + goff_;
  CG_REQUIRE(((uintptr_t)r_ % 16) == 0 && ((uintptr_t)p0_ % 16) == 0 && ((uintptr_t)p1_ % 16) == 0,
             "solver buffers must be 16-byte aligned");
  const size_t np = 2 * CG_MAX_BLOCKS;
  CHECK(hipMalloc((void **)&partA_own_, np * sizeof(double)));
  CHECK(hipMalloc((void **)&partB_own_, np * sizeof(double)));
  CHECK(hipMemset(partA_own_, 0, np * sizeof(double)));
  CHECK(hipMemset(partB_own_, 0, np * sizeof(double)));
  partA_ = partA_own_;
  partB_ = partB_own_;
  if (pad_ || relay_)
  {
    CHECK(hipMalloc((void **)&xpad_, Np_ * sizeof(float)));
    CHECK(hipMemsetAsync(xpad_, 0, Np_ * sizeof(float), stream_));
  }
  CHECK(hipMalloc((void **)&st_, sizeof(CgState)));
  CHECK(hipHostMalloc((void **)&st_host_, sizeof(CgState)));
  CHECK(hipMemsetAsync(st_, 0, sizeof(CgState), stream_));
  // all ghost planes start (and, on global faces, stay) zero: the Dirichlet boundary
  CHECK(hipMemsetAsync(r_base_, 0, buflen_ * sizeof(float), stream_));
  CHECK(hipMemsetAsync(p0_base_, 0, buflen_ * sizeof(float), stream_));
  CHECK(hipMemsetAsync(p1_base_, 0, buflen_ * sizeof(float), stream_));
  build_geometry();
  // Streaming hints: r stays in the 256 MiB Infinity Cache -- all of it if it fits, a cache-sized part of it
  // otherwise (POL_PART; 1-D and 2-D.  The 3-D engine gets the same effect from marching up in kernel A and down
  // in kernel B: the half of r it touched last is the half it touches first, and keeping a fixed half instead
  // measured no better).  The rows engine in d >= 3 re-reads p's far neighbours, which non-temporal loads would
  // push back to HBM, so it uses plain accesses
  const bool keep_r = engine_ == ENGINE_PLANES || engine_ == ENGINE_MORTON4 || desc_.d <= 2;
  // paired x update (every engine's kernel B implements it)
  xdefer_ = env_int("CG_XDEFER", 1) != 0;
  // Grids whose four vectors (nearly) fit the Infinity Cache together need no policy: every non-temporal access
  // would send a line past a cache that has room for it.  Plain accesses against keeping r: 4096 x 2048 18420 ->
  // 20910 it/s, 4096^2 10870 -> 11680, 5120 x 4096 8530 -> 9120, 1-D 2^24 10740 -> 12070, 256^3 10770 -> 11340;
  // from 6144 x 4096 on (25 M points) keeping r wins (7470 against 6990; 320^3 4540 against 4010).
  const bool all_fit = N_ <= (21ull << 20) && engine_ != ENGINE_MORTON4;
  const bool part = engine_ == ENGINE_ROWS && desc_.d <= 2 && N_ * sizeof(float) > keep_bytes();
  policy_ = env_int("CG_CACHE_POLICY", all_fit ? POL_PLAIN
                                       : engine_ == ENGINE_PLANES ? POL_KEEP_RP
                                       : part ? POL_PART
                                       : keep_r ? POL_KEEP_R : POL_PLAIN);
  // not every engine has every policy: the third exists for the planes engine, the fourth and fifth for the
  // 1-D / 2-D rows
  auto valid = [&](int pol) {
    if (pol < POL_PLAIN || pol > POL_PART)
      return (int)POL_KEEP_R;
    if (pol == POL_KEEP_RP && engine_ != ENGINE_PLANES)
      return (int)POL_KEEP_R;
    if ((pol == POL_STREAM || pol == POL_PART) && !(engine_ == ENGINE_ROWS && desc_.d <= 2))
      return (int)POL_KEEP_R;
    return pol;
  };
  policy_ = valid(policy_);
  // kernel A may take a policy of its own (CG_CACHE_POLICY_A); by default both kernels keep the same part of r
  policy_a_ = valid(env_int("CG_CACHE_POLICY_A", policy_));
  sumA_ = partA_;
  sumB_ = partB_;
  nA_ = nB_ = grid_;
}

Solver::~Solver()
{
  (void)hipStreamSynchronize(stream_);
  if (side_)
    (void)hipStreamSynchronize(side_);
  if (graph_)
    (void)hipGraphExecDestroy(graph_);
  if (tail_graph_)
    (void)hipGraphExecDestroy(tail_graph_);
  if (own_r_)
    (void)hipFree(r_base_);
  if (own_p0_)
    (void)hipFree(p0_base_);
  if (own_p1_)
    (void)hipFree(p1_base_);
  (void)hipFree(partA_own_);
  (void)hipFree(partB_own_);
  (void)hipFree(st_);
  rccl_release();
  if (rank_sums_)
    (void)hipFree(rank_sums_);
  if (edge_)
    (void)hipFree(edge_);
  for (hipEvent_t e : {ev_fork_, ev_copy_, ev_alpha_, ev_join_, ev_cap_begin_, ev_cap_end_})
    if (e)
      (void)hipEventDestroy(e);
  if (side_)
  {
    (void)hipStreamSynchronize(side_);
    (void)hipStreamDestroy(side_);
  }
  if (xpad_)
    (void)hipFree(xpad_);
  if (link_ctl_)
    (void)hipFree(link_ctl_);
  (void)hipHostFree(st_host_);
  if (own_stream_)
    (void)hipStreamDestroy(stream_);
}


// Which engine an automatic choice takes.  The rows and planes engines give a lane four points of
// axis 0 and a wave 256 of them, so a short axis 0 leaves lanes idle unless the rows engine can pack
// several rows into a workgroup (d >= 3 and 4 <= n0 <= 256).  Measured on MI355X, kernels A / B in
// microseconds, packed rows against the one-point-per-lane scalar engine: d = 4, L = 64 56 / 71
// against 125 / 104; d = 5, L = 32 179 / 164 against 354 / 305; d = 6, L = 16 81 / 83 against 175 / 146;
// unpacked rows of 64 points lost 194 / 187 to the scalar engine's 127 / 109.  The planes engine needs
// 256 points per row (d = 3, L = 128: packed rows 10.9 / 12.6, planes 16.2 / 23.4).  Grids below 2^19
// points are launch-bound and take the scalar engine's shorter prologue.
// CG_ENGINE = 1 (rows) | 2 (scalar) | 3 (planes) overrides the choice wherever that engine can run.
static int choose_engine(bool rows_ok, bool planes_ok, bool morton4_ok, int d, unsigned n0, unsigned long long N)
{
  if (morton4_ok)
    return ENGINE_MORTON4;
  const int forced = env_int("CG_ENGINE", 0);
  if (forced == ENGINE_PLANES && planes_ok)
    return ENGINE_PLANES;
  if (forced == ENGINE_ROWS && rows_ok)
    return ENGINE_ROWS;
  if (forced == ENGINE_SCALAR)
    return ENGINE_SCALAR;
  if (planes_ok && n0 >= 256)
    return ENGINE_PLANES;
  if (rows_ok && (n0 >= 128 || rows_packable(d, n0)) && N >= (1ull << 19))
    return ENGINE_ROWS;
  return ENGINE_SCALAR;
}

void Solver::build_geometry()
{
  const int d = desc_.d;
  const unsigned *n = desc_.extents;
  // scalar geometry is always filled (it also serves the Morton layout)
  memset(&sg_, 0, sizeof(sg_));
  sg_.d = d;
  sg_.N = N_;
  unsigned long long s = 1;
  for (int k = 0; k < d; k++)
  {
    sg_.n[k] = n[k];
    sg_.stride[k] = s;
    s *= n[k];
    sg_.mmask[k] = morton_mask(d, k);
  }
  sg_.has_lo = desc_.has_lo;
  sg_.has_hi = desc_.has_hi;
  if (layout_ == LAYOUT_MORTON)
  {
    int bits = 0;
    for (int k = 0; k < d; k++)
    {
      CG_REQUIRE(is_pow2(n[k]) && n[k] == n[0], "the Morton layout needs equal power-of-two extents");
    }
    for (unsigned v = n[0]; v > 1; v >>= 1)
      bits++;
    CG_REQUIRE(bits * d <= 32, "the Morton index must fit 32 bits");
  }

  // small grids: whole iterations inside one workgroup (cg_small.hpp); automatic engine choice only
  small_ = !pad_ && desc_.engine == ENGINE_AUTO && !desc_.has_lo && !desc_.has_hi && N_ <= (size_t)CG_SMALL_MAX &&
           d <= CG_SMALL_MAXD && env_int("CG_SMALL", 1) && env_int("CG_ENGINE", 0) == 0; // a forced engine is obeyed
  if (small_)
  {
    memset(&smg_, 0, sizeof(smg_));
    smg_.d = d;
    smg_.N = (unsigned)N_;
    for (int k = 0; k < d; k++)
    {
      smg_.n[k] = n[k];
      smg_.stride[k] = (unsigned)sg_.stride[k];
      smg_.mmask[k] = sg_.mmask[k];
    }
  }

  const bool rows_ok = layout_ == LAYOUT_LEX && (n[0] % 4 == 0 || pad_) && d <= CG_ROWS_MAXD + 2;
  const bool planes_ok = rows_ok && d == 3;
  const bool morton4_ok = layout_ == LAYOUT_MORTON && d >= 2 && N_ >= 4 && N_ % 4 == 0;
  engine_ = desc_.engine == ENGINE_AUTO ? choose_engine(rows_ok, planes_ok, morton4_ok, d, n[0], N_) : desc_.engine;
  if (engine_ == ENGINE_MORTON4)
    CG_REQUIRE(morton4_ok, "the 4-per-lane Morton engine needs the Morton layout and d >= 2");
  if (engine_ == ENGINE_ROWS)
    CG_REQUIRE(rows_ok, "the rows engine needs the lexicographic layout, n0 % 4 == 0 and d <= 16");
  if (engine_ == ENGINE_PLANES)
    CG_REQUIRE(planes_ok, "the planes engine needs a 3-D lexicographic grid with n0 % 4 == 0");

  // Kernel A and kernel B may be cut into work items differently (the number of resident workgroups each
  // runs best with differs); gridB_ also launches the plain stencil and the dot product of begin_local().
  // grid_ = the larger of the two is the number of partial-sum addends: the tail a kernel does not write
  // stays zero.
  if (engine_ == ENGINE_ROWS)
  {
    make_rows_geom(d, n, desc_.has_lo, desc_.has_hi, rg_, block_, gridB_, s0_);
    // 2-D grids whose vectors exceed the Infinity Cache: kernel A (two reads per write, to a third array)
    // streams best with 2 workgroups per CU, kernel B with 4 (16384 x 8192: A 356 -> 326 us with 2, B 486 ->
    // 506).  Not the folded 1-D march (2^27 points: A 375 -> 448) and not grids the cache helps (8192^2: A
    // 133 -> 159).
    // (rows of at least 16384 points: 32768 x 4096 341 -> 320, 16384^2 653 -> 606; 8192 x 16384 loses, 365 -> 388)
    const bool beyond = d == 2 && N_ >= (1ull << 27) && n[0] >= 16384u;
    const int per_cu_a = env_int("CG_BLOCKS_PER_CU_A", beyond ? 2 : 0);
    make_rows_geom(d, n, desc_.has_lo, desc_.has_hi, rgA_, blockA_, gridA_, s0_, per_cu_a, true);
  }
  else if (engine_ == ENGINE_PLANES)
  {
    const int per_cu_b = env_int("CG_PLANES_BPCU_B", 0);
    make_planes_geom(n, desc_.has_lo, desc_.has_hi, pg_, block_, gridB_, planes_r_, s0_, per_cu_b);
    pgA_ = pg_;
    blockA_ = block_;
    gridA_ = gridB_;
    if (per_cu_b > 0)
      make_planes_geom(n, desc_.has_lo, desc_.has_hi, pgA_, blockA_, gridA_, planes_r_, s0_);
  }
  else if (engine_ == ENGINE_MORTON4)
  {
    make_morton_geom(d, N_, mg_, block_, gridB_);
    gridA_ = gridB_;
    blockA_ = block_;
  }
  else
  {
    block_ = 256;
    const unsigned long long nb = (N_ + block_ - 1) / block_;
    gridB_ = (int)std::min<unsigned long long>(nb, (unsigned long long)max_blocks());
    gridA_ = gridB_;
    blockA_ = block_;
  }
  grid_ = std::max(gridA_, gridB_);
}

size_t Solver::device_bytes() const
{
  size_t n = 2 * (size_t)(2 * CG_MAX_BLOCKS) * sizeof(double) + sizeof(CgState);
  n += ((own_r_ ? 1 : 0) + (own_p0_ ? 1 : 0) + (own_p1_ ? 1 : 0)) * buflen_ * sizeof(float);
  if (xpad_)
    n += Np_ * sizeof(float);
  if (rank_sums_)
    n += (2 + 2 * (size_t)CG_RCCL_MAX_RANKS) * sizeof(double) + 16;
  if (edge_)
    n += 2 * edge_pad_ * sizeof(float);
  if (link_ctl_)
    n += sizeof(LinkCtl);
  return n;
}

const char *Solver::kernel_name() const
{
  if (small_)
    return "k_small";
  switch (engine_)
  {
  case ENGINE_ROWS: return rg_.R ? "k_cols" : "k_rows";
  case ENGINE_PLANES: return "k_planes";
  case ENGINE_MORTON4: // the same rule as launch_morton4
    return (mg_.N >= (unsigned long long)CG_MORTON_TILE * gridB_ && mg_.N % CG_MORTON_TILE == 0 && env_int("CG_MORTON_TILED", 1))
               ? "k_morton_tile"
               : "k_morton4";
  default: return "k_scalar";
  }
}

VecArgs Solver::args() const
{
  VecArgs a{};
  a.x = x_;
  a.r = r_;
  a.p0 = p0_;
  a.p1 = p1_;
  a.st = st_;
  a.sumA = sumA_;
  a.nA = nA_;
  a.sumB = sumB_;
  a.nB = nB_;
  return a;
}

template <int MODE, bool GHOST, int POL>
static void launch_rows_pol(const RowsGeom &g, const VecArgs &a, int grid, int block, hipStream_t s)
{
  if (g.sub && g.R)
  {
    // blocked columns: the streaming hint is the geometry's choice (grid size) or a forced cache policy
    const bool stream = g.stream || POL == POL_KEEP_R || POL == POL_STREAM;
#define CG_COLS_LAUNCH_L(DD, RR, LL)                                                                        \
  do                                                                                                        \
  {                                                                                                         \
    if (stream)                                                                                             \
      hipLaunchKernelGGL((k_cols<MODE, GHOST, DD, RR, true, LL>), dim3(grid), dim3(block), 0, s, g, a); \
    else                                                                                                    \
      hipLaunchKernelGGL((k_cols<MODE, GHOST, DD, RR, false, LL>), dim3(grid), dim3(block), 0, s, g, a);    \
  } while (0)
#define CG_COLS_LAUNCH(DD, RR)                                                                              \
  do                                                                                                        \
  {                                                                                                         \
    if (MODE == MODE_B ? g.lds_b : g.lds_a)                                                                 \
      CG_COLS_LAUNCH_L(DD, RR, true);                                                                       \
    else                                                                                                    \
      CG_COLS_LAUNCH_L(DD, RR, false);                                                                      \
  } while (0)
    if (g.d == 3 && g.R == 2)
      CG_COLS_LAUNCH(3, 2);
    else if (g.d == 3)
      CG_COLS_LAUNCH(3, 1);
    else if (g.d == 4 && g.R == 2 && g.t2 > 1)
    {
      // tiled slots (the geometry guarantees the LDS hand-off for every mode)
      if (block > 512)
      {
        if (stream)
          hipLaunchKernelGGL((k_cols<MODE, GHOST, 4, 2, true, true, 2>), dim3(grid), dim3(block), 0, s, g, a);
        else
          hipLaunchKernelGGL((k_cols<MODE, GHOST, 4, 2, false, true, 2>), dim3(grid), dim3(block), 0, s, g, a);
      }
      else if (stream)
        hipLaunchKernelGGL((k_cols<MODE, GHOST, 4, 2, true, true, 1>), dim3(grid), dim3(block), 0, s, g, a);
      else
        hipLaunchKernelGGL((k_cols<MODE, GHOST, 4, 2, false, true, 1>), dim3(grid), dim3(block), 0, s, g, a);
    }
    else if (g.d == 4 && g.R == 2)
      CG_COLS_LAUNCH(4, 2);
    else if (g.d == 4)
      CG_COLS_LAUNCH(4, 1);
    else if (g.R == 2)
      CG_COLS_LAUNCH(5, 2);
    else
      CG_COLS_LAUNCH(5, 1);
#undef CG_COLS_LAUNCH
#undef CG_COLS_LAUNCH_L
    return;
  }
  if (g.sub)
  {
    if (g.d == 3)
      hipLaunchKernelGGL((k_rows<MODE, GHOST, 3, POL, true>), dim3(grid), dim3(block), 0, s, g, a);
    else
      hipLaunchKernelGGL((k_rows<MODE, GHOST, 0, POL, true>), dim3(grid), dim3(block), 0, s, g, a);
    return;
  }
  switch (g.d)
  {
  case 1: hipLaunchKernelGGL((k_rows<MODE, GHOST, 1, POL>), dim3(grid), dim3(block), 0, s, g, a); break;
  case 2: hipLaunchKernelGGL((k_rows<MODE, GHOST, 2, POL>), dim3(grid), dim3(block), 0, s, g, a); break;
  case 3: hipLaunchKernelGGL((k_rows<MODE, GHOST, 3, POL>), dim3(grid), dim3(block), 0, s, g, a); break;
  default: hipLaunchKernelGGL((k_rows<MODE, GHOST, 0, POL>), dim3(grid), dim3(block), 0, s, g, a);
  }
}

template <int MODE, bool GHOST>
static void launch_rows(const RowsGeom &g, const VecArgs &a, int grid, int block, hipStream_t s, int pol = POL_PLAIN)
{
  if (pol == POL_KEEP_R && MODE != MODE_STENCIL)
    launch_rows_pol<MODE, GHOST, POL_KEEP_R>(g, a, grid, block, s);
  else if (pol == POL_STREAM && MODE != MODE_STENCIL && !g.sub && g.d <= 2)
    launch_rows_pol<MODE, GHOST, POL_STREAM>(g, a, grid, block, s);
  else if (pol == POL_STREAM && MODE != MODE_STENCIL)
    launch_rows_pol<MODE, GHOST, POL_KEEP_R>(g, a, grid, block, s);
  else if (pol == POL_PART && MODE != MODE_STENCIL && !g.sub && g.d == 1)
    hipLaunchKernelGGL((k_rows<MODE, GHOST, 1, POL_PART>), dim3(grid), dim3(block), 0, s, g, a);
  else if (pol == POL_PART && MODE != MODE_STENCIL && !g.sub && g.d == 2)
    hipLaunchKernelGGL((k_rows<MODE, GHOST, 2, POL_PART>), dim3(grid), dim3(block), 0, s, g, a);
  else if (pol == POL_PART && MODE != MODE_STENCIL)
    launch_rows_pol<MODE, GHOST, POL_KEEP_R>(g, a, grid, block, s);
  else
    launch_rows_pol<MODE, GHOST, POL_PLAIN>(g, a, grid, block, s);
}

template <int MODE, bool GHOST, int POL>
static void launch_planes_pol(const PlanesGeom &g, const VecArgs &a, int grid, int block, int R, hipStream_t s)
{
  // the descending march exists for the two CG kernels on ghosted vectors only
  if constexpr (MODE != MODE_STENCIL && GHOST)
  {
    if ((MODE == MODE_A ? g.dir_a : g.dir_b) < 0)
    {
      if (R >= 4)
        hipLaunchKernelGGL((k_planes<MODE, GHOST, POL, 4, false>), dim3(grid), dim3(block), 0, s, g, a);
      else
        hipLaunchKernelGGL((k_planes<MODE, GHOST, POL, 2, false>), dim3(grid), dim3(block), 0, s, g, a);
      return;
    }
  }
  if (R >= 4)
    hipLaunchKernelGGL((k_planes<MODE, GHOST, POL, 4>), dim3(grid), dim3(block), 0, s, g, a);
  else
    hipLaunchKernelGGL((k_planes<MODE, GHOST, POL, 2>), dim3(grid), dim3(block), 0, s, g, a);
}

template <int MODE, bool GHOST>
static void launch_planes(const PlanesGeom &g, const VecArgs &a, int grid, int block, int R, hipStream_t s,
                          int pol = POL_PLAIN)
{
  if (pol == POL_KEEP_RP && MODE != MODE_STENCIL)
    launch_planes_pol<MODE, GHOST, POL_KEEP_RP>(g, a, grid, block, R, s);
  else if (pol == POL_KEEP_R && MODE != MODE_STENCIL)
    launch_planes_pol<MODE, GHOST, POL_KEEP_R>(g, a, grid, block, R, s);
  else
    launch_planes_pol<MODE, GHOST, POL_PLAIN>(g, a, grid, block, R, s);
}

template <int MODE>
static void launch_morton4(const MortonGeom &g, const VecArgs &a, int grid, int block, hipStream_t s, int pol = POL_PLAIN)
{
  // grids of at least one tile per workgroup use the LDS-tiled kernel
  const bool tiled = g.N >= (unsigned long long)CG_MORTON_TILE * grid && g.N % CG_MORTON_TILE == 0 &&
                     env_int("CG_MORTON_TILED", 1);
  if (tiled)
  {
    if (pol == POL_KEEP_R && MODE != MODE_STENCIL)
      hipLaunchKernelGGL((k_morton_tile<MODE, POL_KEEP_R>), dim3(grid), dim3(CG_MORTON_BT), 0, s, g, a);
    else
      hipLaunchKernelGGL((k_morton_tile<MODE, POL_PLAIN>), dim3(grid), dim3(CG_MORTON_BT), 0, s, g, a);
  }
  else if (pol == POL_KEEP_R && MODE != MODE_STENCIL)
    hipLaunchKernelGGL((k_morton4<MODE, POL_KEEP_R>), dim3(grid), dim3(block), 0, s, g, a);
  else
    hipLaunchKernelGGL((k_morton4<MODE, POL_PLAIN>), dim3(grid), dim3(block), 0, s, g, a);
}

template <int MODE, bool GHOST>
static void launch_scalar(const ScalarGeom &g, const VecArgs &a, int layout, int grid, int block, hipStream_t s)
{
  if (layout == LAYOUT_MORTON)
    hipLaunchKernelGGL((k_scalar<MODE, GHOST, LAYOUT_MORTON>), dim3(grid), dim3(block), 0, s, g, a);
  else
    hipLaunchKernelGGL((k_scalar<MODE, GHOST, LAYOUT_LEX>), dim3(grid), dim3(block), 0, s, g, a);
}

void Solver::launch_mode(int mode, bool ghost, const VecArgs &a)
{
  if (engine_ == ENGINE_MORTON4)
  {
    switch (mode)
    {
    case MODE_A: launch_morton4<MODE_A>(mg_, a, gridA_, blockA_, stream_, policy_a_); break;
    case MODE_B: launch_morton4<MODE_B>(mg_, a, gridB_, block_, stream_, policy_); break;
    default: launch_morton4<MODE_STENCIL>(mg_, a, gridB_, block_, stream_);
    }
  }
  else if (engine_ == ENGINE_PLANES)
  {
    switch (mode)
    {
    case MODE_A: launch_planes<MODE_A, true>(pgA_, a, gridA_, blockA_, planes_r_, stream_, policy_a_); break;
    case MODE_B: launch_planes<MODE_B, true>(pg_, a, gridB_, block_, planes_r_, stream_, policy_); break;
    default:
      if (ghost)
        launch_planes<MODE_STENCIL, true>(pg_, a, gridB_, block_, planes_r_, stream_);
      else
        launch_planes<MODE_STENCIL, false>(pg_, a, gridB_, block_, planes_r_, stream_);
    }
  }
  else if (engine_ == ENGINE_ROWS)
  {
    switch (mode)
    {
    case MODE_A: launch_rows<MODE_A, true>(rgA_, a, gridA_, blockA_, stream_, policy_a_); break;
    case MODE_B: launch_rows<MODE_B, true>(rg_, a, gridB_, block_, stream_, policy_); break;
    default:
      if (ghost)
        launch_rows<MODE_STENCIL, true>(rg_, a, gridB_, block_, stream_);
      else
        launch_rows<MODE_STENCIL, false>(rg_, a, gridB_, block_, stream_);
    }
  }
  else
  {
    // Morton vectors have no ghost planes: out-of-grid is index >= N
    const bool gh = ghost && layout_ == LAYOUT_LEX;
    switch (mode)
    {
    case MODE_A:
      if (gh)
        launch_scalar<MODE_A, true>(sg_, a, layout_, gridB_, block_, stream_);
      else
        launch_scalar<MODE_A, false>(sg_, a, layout_, gridB_, block_, stream_);
      break;
    case MODE_B:
      if (gh)
        launch_scalar<MODE_B, true>(sg_, a, layout_, gridB_, block_, stream_);
      else
        launch_scalar<MODE_B, false>(sg_, a, layout_, gridB_, block_, stream_);
      break;
    default:
      if (gh)
        launch_scalar<MODE_STENCIL, true>(sg_, a, layout_, gridB_, block_, stream_);
      else
        launch_scalar<MODE_STENCIL, false>(sg_, a, layout_, gridB_, block_, stream_);
    }
  }
  CHECK(hipGetLastError());
}

void Solver::launch_A()
{
  VecArgs a = args();
  a.part_out = partA_;
  if (rccl_)
  {
    a.rank_sum = &rank_sums_[0];
    a.ticket = ticket(0);
  }
  launch_mode(MODE_A, true, a);
}

void Solver::flush_x()
{
  // nothing to do unless a kernel B that may leave x alone was enqueued since the last flush (the
  // single-workgroup kernel and the step interface update x every iteration); the flush kernel itself
  // decides on the device whether an update is pending
  if (!xdefer_ || !may_be_pending_)
    return;
  may_be_pending_ = false;
  last_deferred_ = false;
  const int vec4 = (((uintptr_t)x_ | (uintptr_t)p0_ | (uintptr_t)p1_) % 16 == 0) ? 1 : 0;
  const int grid = (int)std::max<unsigned long long>(1, std::min<unsigned long long>((Np_ / 4 + 255) / 256, 4096));
  hipLaunchKernelGGL(k_flush_x, dim3(grid), dim3(256), 0, stream_, st_, x_, p0_, p1_, (unsigned long long)Np_, vec4);
  hipLaunchKernelGGL(k_flush_mark, dim3(1), dim3(64), 0, stream_, st_);
  CHECK(hipGetLastError());
}

void Solver::launch_B(bool may_defer)
{
  VecArgs a = args();
  a.part_out = partB_;
  // never two launches in a row that leave x alone: the second would lose the first one's update
  a.xdefer = (xdefer_ && may_defer && !last_deferred_) ? 1 : 0;
  last_deferred_ = a.xdefer != 0;
  if (a.xdefer)
    may_be_pending_ = true;
  if (rccl_)
  {
    a.rank_sum = &rank_sums_[1];
    a.ticket = ticket(1);
  }
  if (linked_ && push_in_b_ && (desc_.has_lo || desc_.has_hi))
  {
    // r's boundary planes travel while kernel B still runs: it stores them into the neighbours' staging
    // areas as it produces them (parity 0 here; the kernel adds the parity of the coming exchange)
    a.push_seq = &link_ctl_->seq[1];
    a.push_stride = 2ull * links_.plane_pad;
    if (desc_.has_lo)
      a.push_lo = links_staging(links_.box[links_.rank - 1], 0, LINK_FROM_HI, links_.plane_pad);
    if (desc_.has_hi)
      a.push_hi = links_staging(links_.box[links_.rank + 1], 0, LINK_FROM_LO, links_.plane_pad);
  }
  launch_mode(MODE_B, true, a);
}

void Solver::launch_stencil(const float *u, float *out, bool ghost)
{
  VecArgs a{};
  a.u = u;
  a.out = out;
  launch_mode(MODE_STENCIL, ghost, a);
}

static void launch_dot(const float *v, const float *w, unsigned long long n, double *part, int grid, hipStream_t s)
{
  const int vec4 = (((uintptr_t)v | (uintptr_t)w) % 16 == 0) ? 1 : 0;
  hipLaunchKernelGGL(k_dot, dim3(grid), dim3(256), 0, s, v, w, n, vec4, part);
  CHECK(hipGetLastError());
}

// Morton <-> lexicographic order of a whole vector.  A workgroup-sized run of threads covers one
// block of 2^kbits consecutive Morton indices -- a box of the grid -- in the box's lexicographic
// order: the lexicographic side is touched in runs of consecutive points, the Morton side inside one
// contiguous block that the caches hold.
__global__ void __launch_bounds__(256) k_relayout(float *__restrict__ dst, const float *__restrict__ src, int d, unsigned L,
                                                  unsigned long long N, int kbits, int to_lex)
{
  // kbits >= 8: the block of Morton indices and everything derived from it is uniform in a workgroup
  const unsigned wpb = 1u << (kbits - 8); // workgroups per block
  const unsigned mhi = (blockIdx.x / wpb) << kbits;
  unsigned tl = ((blockIdx.x % wpb) << 8) | threadIdx.x;
  const unsigned low = (1u << kbits) - 1u;
  unsigned m = mhi;
  unsigned long long i = 0, s = 1;
  for (int j = 0; j < d; j++)
  {
    const unsigned mask = morton_mask(d, j);
    const int cj = __popc(mask & low); // bits of coordinate j inside a block
    const unsigned lo = tl & ((1u << cj) - 1u);
    tl >>= cj;
    m |= pdep32(lo, mask & low);
    i += (unsigned long long)(pext32(mhi, mask) | lo) * s;
    s *= L;
  }
  if (m >= N)
    return;
  if (to_lex)
    dst[i] = src[m];
  else
    dst[m] = src[i];
}

// The same for grids of at least one block of 8192 indices, staged through LDS: the Morton side is
// read or written as whole consecutive float4s, the lexicographic side in float4s that form runs of
// 2^ceil(13/d) consecutive points (128^4: 4.2 ms per pass with the kernel above, the streams' time with this).
template <bool TO_LEX>
__global__ void __launch_bounds__(256) k_relayout_tile(float *__restrict__ dst, const float *__restrict__ src, int d,
                                                       unsigned L)
{
  constexpr int KB = 13, QUADS = (1 << KB) / 4;
  __shared__ float tile[1 << KB];
  const unsigned mhi = blockIdx.x << KB;
  const unsigned low = (1u << KB) - 1u;
  if constexpr (TO_LEX)
  {
    for (int q = threadIdx.x; q < QUADS; q += 256)
      reinterpret_cast<float4 *>(tile)[q] = load4<true>(src + mhi + 4u * q);
    __syncthreads();
  }
  for (int q = threadIdx.x; q < QUADS; q += 256)
  {
    unsigned tl = 4u * q; // first of four consecutive points along axis 0, in the box's lexicographic order
    unsigned m = 0;
    unsigned long long i = 0, s = 1;
    for (int j = 0; j < d; j++)
    {
      const unsigned mask = morton_mask(d, j);
      const int cj = __popc(mask & low); // bits of coordinate j inside a block (axis 0: at least 2)
      const unsigned lo = tl & ((1u << cj) - 1u);
      tl >>= cj;
      m |= pdep32(lo, mask & low);
      i += (unsigned long long)(pext32(mhi, mask) | lo) * s;
      s *= L;
    }
    // the two low bits of the axis-0 coordinate are bits 0 and d of the Morton index
    const unsigned m1 = m | 1u, m2 = m | (1u << d), m3 = m2 | 1u;
    if constexpr (TO_LEX)
      store4<true>(dst + i, make_float4(tile[m], tile[m1], tile[m2], tile[m3]));
    else
    {
      const float4 v = load4<true>(src + i);
      tile[m] = v.x;
      tile[m1] = v.y;
      tile[m2] = v.z;
      tile[m3] = v.w;
    }
  }
  if constexpr (!TO_LEX)
  {
    __syncthreads();
    for (int q = threadIdx.x; q < QUADS; q += 256)
      store4<true>(dst + mhi + 4u * q, reinterpret_cast<const float4 *>(tile)[q]);
  }
}

void Solver::relayout(float *dst, const float *src, bool to_lex)
{
  int total = 0;
  for (unsigned long long v = N_; v > 1; v >>= 1)
    total++;
  if (total >= 13 && desc_.d <= 6 && desc_.extents[0] >= 4)
  {
    const unsigned nblk = (unsigned)(N_ >> 13);
    if (to_lex)
      hipLaunchKernelGGL(k_relayout_tile<true>, dim3(nblk), dim3(256), 0, stream_, dst, src, desc_.d, desc_.extents[0]);
    else
      hipLaunchKernelGGL(k_relayout_tile<false>, dim3(nblk), dim3(256), 0, stream_, dst, src, desc_.d, desc_.extents[0]);
    CHECK(hipGetLastError());
    return;
  }
  const int kbits = std::max(8, std::min(13, total));
  hipLaunchKernelGGL(k_relayout, dim3((unsigned)((N_ + 255) / 256)), dim3(256), 0, stream_, dst, src, desc_.d,
                     desc_.extents[0], (unsigned long long)N_, kbits, to_lex ? 1 : 0);
  CHECK(hipGetLastError());
}

void Solver::copy_rows(float *dst, size_t dpitch, const float *src, size_t spitch)
{
  const size_t n0 = desc_.extents[0];
  CHECK(hipMemcpy2DAsync(dst, dpitch * sizeof(float), src, spitch * sizeof(float), n0 * sizeof(float), N_ / n0,
                         hipMemcpyDeviceToDevice, stream_));
}

// The reference does everything on the legacy default stream, so whatever produced b and x there (add_jl,
// a copy, another library's kernels on stream 0) is implicitly ordered before its solve.  A private
// non-blocking stream has no such ordering: make it wait for what the default stream holds right now.
void order_after_default_stream(hipStream_t s)
{
  if (!s)
    return;
  constexpr int MAXDEV = 64;
  static hipEvent_t ev_of[MAXDEV] = {nullptr};
  static std::mutex guard; // one event per device, shared by every solver: record and wait as one step
  int dev = 0;
  CHECK(hipGetDevice(&dev));
  CG_REQUIRE(dev >= 0 && dev < MAXDEV, "device ordinal out of range");
  std::lock_guard<std::mutex> lock(guard);
  if (!ev_of[dev])
    CHECK(hipEventCreateWithFlags(&ev_of[dev], hipEventDisableTiming));
  CHECK(hipEventRecord(ev_of[dev], nullptr));
  CHECK(hipStreamWaitEvent(s, ev_of[dev], 0));
}

void Solver::begin_local(const float *b_dev, float *x_dev)
{
  if (own_stream_)
    order_after_default_stream(stream_);
  hist_all_.clear();
  hist_seen_ = 0;
  x_ = x_dev;
  if ((graph_ || tail_graph_) && !pad_ && !relay_ && graph_x_ != x_dev)
    drop_graphs();
  if (pad_)
  {
    x_user_ = x_dev;
    x_ = xpad_;
    copy_rows(r_, s0_, b_dev, desc_.extents[0]);
    copy_rows(xpad_, s0_, x_dev, desc_.extents[0]);
  }
  else if (relay_)
  {
    x_user_ = x_dev;
    x_ = xpad_;
    relayout(r_, b_dev, true);
    relayout(xpad_, x_dev, true);
  }
  else
    CHECK(hipMemcpyAsync(r_, b_dev, N_ * sizeof(float), hipMemcpyDeviceToDevice, stream_));
  // p_old of iteration 0 must be finite everywhere it is read: beta = 0 then gives p = r exactly
  CHECK(hipMemsetAsync(p0_base_, 0, buflen_ * sizeof(float), stream_));
  launch_dot(r_, r_, Np_, partB_, gridB_, stream_); // padding holds zeros; as many partials as kernel B writes
}

void Solver::reduce_partials(int which, double *out_dev)
{
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(64), 0, stream_, which ? partB_ : partA_, grid_, out_dev);
  CHECK(hipGetLastError());
}

void Solver::bind_sums(const double *sumA, int nA, const double *sumB, int nB)
{
  sumA_ = sumA ? sumA : partA_;
  nA_ = sumA ? nA : grid_;
  sumB_ = sumB ? sumB : partB_;
  nB_ = sumB ? nB : grid_;
  drop_graphs();
}

void Solver::bind_partials(double *partA, double *partB)
{
  partA_ = partA ? partA : partA_own_;
  partB_ = partB ? partB : partB_own_;
  bind_sums(nullptr, 0, nullptr, 0);
}

void Solver::init_state(double rel, int lagged, int maxit)
{
  hipLaunchKernelGGL(k_init_state, dim3(1), dim3(64), 0, stream_, st_, sumB_, nB_, rel, lagged, maxit);
  CHECK(hipGetLastError());
}

void Solver::add_iterations(int n)
{
  hipLaunchKernelGGL(k_set_maxit, dim3(1), dim3(64), 0, stream_, st_, n);
  CHECK(hipGetLastError());
}

void Solver::begin(const float *b_dev, float *x_dev, double rel, int lagged)
{
  begin_local(b_dev, x_dev);
  if (linked_)
    launch_exchange(1, true);
  else if (rccl_)
  {
    reduce_partials(1, &rank_sums_[1]);
    rccl_allgather(1);
    rccl_exchange_from_r();
  }
  init_state(rel, lagged, 0);
}

// ---- r's boundary planes ahead of kernel B ---------------------------------------------------------------

void Solver::ensure_edge()
{
  if (edge_)
    return;
  edge_pad_ = (plane_ + 63) / 64 * 64;
  CHECK(hipMalloc((void **)&edge_, 2 * edge_pad_ * sizeof(float)));
  CHECK(hipMemset(edge_, 0, 2 * edge_pad_ * sizeof(float)));
}

void Solver::ensure_side()
{
  if (side_)
    return;
  CHECK(hipStreamCreateWithFlags(&side_, hipStreamNonBlocking));
  for (hipEvent_t *e : {&ev_fork_, &ev_copy_, &ev_alpha_, &ev_join_, &ev_cap_begin_, &ev_cap_end_})
    CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
}

float *Solver::edge_plane(int side)
{
  ensure_edge();
  return edge_ + (side ? edge_pad_ : 0);
}

void Solver::edge_copy(hipStream_t s)
{
  CG_REQUIRE(layout_ == LAYOUT_LEX && !pad_, "boundary planes exist for lexicographic slabs only");
  ensure_edge();
  if (desc_.has_lo)
    CHECK(hipMemcpyAsync(edge_, r_, plane_ * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (desc_.has_hi)
    CHECK(hipMemcpyAsync(edge_ + edge_pad_, r_ + N_ - plane_, plane_ * sizeof(float), hipMemcpyDeviceToDevice, s));
}

void Solver::edge_update(hipStream_t s)
{
  if (!desc_.has_lo && !desc_.has_hi)
    return;
  ensure_edge();
  EdgeArgs e{};
  e.p0 = p0_;
  e.p1 = p1_;
  e.st = st_;
  e.sumA = sumA_;
  e.nA = nA_;
  e.edge = edge_;
  e.plane = plane_;
  e.pad = edge_pad_;
  // one point per lane and side where the plane allows it: every point costs a chain of dependent neighbour
  // loads, so few lanes with many points each would be bound by latency (a 1024 x 1024 plane on 256 workgroups: 90 us)
  const int blocks = (int)std::max<size_t>(1, std::min<size_t>((plane_ + 255) / 256, 8192));
  hipLaunchKernelGGL(k_edge_r, dim3(blocks), dim3(256), 0, s, sg_, e);
  CHECK(hipGetLastError());
}

void Solver::bind_links(int rank, int world, void *const *boxes)
{
  if (world == 0)
  {
    // back to the caller-driven step interface with the solver's own partials
    CHECK(hipStreamSynchronize(stream_));
    linked_ = false;
    partA_ = partA_own_;
    partB_ = partB_own_;
    bind_sums(nullptr, 0, nullptr, 0);
    return;
  }
  CG_REQUIRE(!rccl_, "a communicator is bound: unbind it before binding boxes");
  CG_REQUIRE(world >= 1 && world <= CG_LINKS_MAX_RANKS, "direct links support 1..64 ranks");
  CG_REQUIRE(rank >= 0 && rank < world, "rank out of range");
  CG_REQUIRE((desc_.has_lo != 0) == (rank > 0) && (desc_.has_hi != 0) == (rank + 1 < world),
             "has_lo / has_hi of the solver do not match its rank");
  memset(&links_, 0, sizeof(links_));
  for (int j = 0; j < world; j++)
  {
    CG_REQUIRE(boxes[j] != nullptr, "null box");
    links_.box[j] = static_cast<LinkBox *>(boxes[j]);
  }
  if (!link_ctl_)
    CHECK(hipMalloc((void **)&link_ctl_, sizeof(LinkCtl)));
  CHECK(hipMemsetAsync(link_ctl_, 0, sizeof(LinkCtl), stream_));
  links_.ctl = link_ctl_;
  links_.st = st_;
  links_.rank = rank;
  links_.world = world;
  links_.plane = plane_;
  links_.plane_pad = links_plane_pad(plane_);
  links_.budget = (long long)std::max(1, env_int("CG_LINKS_TIMEOUT_MS", 10000)) * 100000ll; // 100 MHz ticks
  partA_ = partA_own_;
  partB_ = partB_own_;
  bind_sums(&link_ctl_->sum[0], 1, &link_ctl_->sum[1], 1);
  push_in_b_ = env_int("CG_LINKS_PUSH_IN_B", 1) != 0;
  linked_ = true;
}

void Solver::launch_exchange(int kind, bool force)
{
  CG_REQUIRE(linked_, "no boxes bound");
  LinkArgs L = links_;
  L.kind = kind;
  L.force = force ? 1 : 0;
  L.pushed = (!force && push_in_b_) ? 1 : 0;
  L.part = kind ? partB_ : partA_;
  L.nparts = grid_;
  int blocks = 1;
  if (kind == 1 && (desc_.has_lo || desc_.has_hi))
  {
    L.r_first = r_;
    L.r_last = r_ + N_ - plane_;
    L.ghost_lo = r_ - plane_;
    L.ghost_hi = r_ + N_;
    // one workgroup per 64 KiB of plane, at most 64: enough to fill the two links it writes to
    const size_t want = (plane_ * sizeof(float) + 65535) / 65536;
    blocks = (int)std::max<size_t>(1, std::min<size_t>(want, 64));
    // every workgroup but the last one to arrive waits inside the kernel for that last one, so all of them
    // must be resident at once: never more than one workgroup per CU
    static int cus = 0;
    if (!cus)
    {
      int dev = 0;
      hipDeviceProp_t prop;
      cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 64;
      (void)hipGetLastError();
    }
    blocks = std::max(1, std::min(env_int("CG_LINKS_BLOCKS", blocks), std::min(256, cus)));
  }
  hipLaunchKernelGGL(k_exchange, dim3(blocks), dim3(256), 0, stream_, L);
  CHECK(hipGetLastError());
}

int Solver::links_error()
{
  if (!linked_)
    return 0;
  int e = 0;
  CHECK(hipMemcpyAsync(&e, &link_ctl_->error, sizeof(int), hipMemcpyDeviceToHost, stream_));
  CHECK(hipStreamSynchronize(stream_));
  return e;
}

void Solver::links_times(float *ms_xA, float *ms_xB) const
{
  if (ms_xA)
    *ms_xA = ms_xA_;
  if (ms_xB)
    *ms_xB = ms_xB_;
}

void Solver::launch_iteration(bool may_defer, hipEvent_t *ev)
{
  // with a communicator bound and slab neighbours, r's boundary planes travel on the side stream while kernel
  // B runs: [side] copy of the planes (beside kernel A) -> [after the p.Ap exchange] edge_update -> send / recv
  // into the neighbours' ghost planes of r, which nothing reads before the next kernel A
  // Small planes (2-D slabs: one row) are latency, not bytes: they go after kernel B on the same stream, and
  // the iteration stays a chain of nodes.  (Every dependency between the two streams costs the graph about
  // 13 us even when it is long satisfied, measured with one rank as its own neighbour: the overlapped form is
  // 27 us per iteration at 8192^2 and 39 us on a 1024 x 1024 x 128 slab above the plain solver, two 4.5-us
  // all-gathers included.  That pays where a plane takes longer than that to cross a link.)
  const bool halo = rccl_ && halo_overlap_ && (desc_.has_lo || desc_.has_hi);
  const bool halo_serial = rccl_ && !halo_overlap_ && (desc_.has_lo || desc_.has_hi);
  if (ev)
    CHECK(hipEventRecord(ev[0], stream_));
  if (halo)
  {
    CHECK(hipEventRecord(ev_fork_, stream_)); // r is final: the previous kernel B is behind this point
    CHECK(hipStreamWaitEvent(side_, ev_fork_, 0));
    edge_copy(side_);
    CHECK(hipEventRecord(ev_copy_, side_));
  }
  launch_A();
  if (ev)
    CHECK(hipEventRecord(ev[1], stream_));
  if (linked_)
    launch_exchange(0, false);
  else if (rccl_)
    rccl_allgather(0);
  if (ev)
    CHECK(hipEventRecord(ev[2], stream_));
  if (halo)
  {
    CHECK(hipEventRecord(ev_alpha_, stream_)); // every rank's p.Ap is in: alpha can be formed
    CHECK(hipStreamWaitEvent(side_, ev_alpha_, 0));
    edge_update(side_);
    rccl_send_recv(edge_, edge_ + edge_pad_, side_);
    CHECK(hipEventRecord(ev_join_, side_));
    CHECK(hipStreamWaitEvent(stream_, ev_copy_, 0)); // kernel B overwrites what the copy reads
  }
  launch_B(may_defer);
  if (ev)
    CHECK(hipEventRecord(ev[3], stream_));
  if (linked_)
    launch_exchange(1, false);
  else if (rccl_)
    rccl_allgather(1);
  if (halo_serial)
    rccl_send_recv(r_, r_ + N_ - plane_, stream_);
  if (halo)
    CHECK(hipStreamWaitEvent(stream_, ev_join_, 0)); // the next kernel A reads r's ghost planes
  if (ev)
    CHECK(hipEventRecord(ev[4], stream_));
}

template <int LAYOUT, int PPT>
static void launch_small_dp(const SmallGeom &g, const VecArgs &a, double *partB, int nB, hipStream_t s)
{
  const dim3 G(1), B(CG_SMALL_THREADS);
  switch (g.d)
  {
  case 1: hipLaunchKernelGGL((k_small<LAYOUT, 1, PPT>), G, B, 0, s, g, a, partB, nB); break;
  case 2: hipLaunchKernelGGL((k_small<LAYOUT, 2, PPT>), G, B, 0, s, g, a, partB, nB); break;
  case 3: hipLaunchKernelGGL((k_small<LAYOUT, 3, PPT>), G, B, 0, s, g, a, partB, nB); break;
  case 4: hipLaunchKernelGGL((k_small<LAYOUT, 4, PPT>), G, B, 0, s, g, a, partB, nB); break;
  default: hipLaunchKernelGGL((k_small<LAYOUT, 0, PPT>), G, B, 0, s, g, a, partB, nB);
  }
}

template <int LAYOUT>
static void launch_small_d(const SmallGeom &g, const VecArgs &a, double *partB, int nB, hipStream_t s)
{
  if (g.N <= 8u * CG_SMALL_THREADS)
    launch_small_dp<LAYOUT, 8>(g, a, partB, nB, s);
  else
    launch_small_dp<LAYOUT, 16>(g, a, partB, nB, s);
}

void Solver::launch_small()
{
  if (layout_ == LAYOUT_MORTON)
    launch_small_d<LAYOUT_MORTON>(smg_, args(), partB_, grid_, stream_);
  else
    launch_small_d<LAYOUT_LEX>(smg_, args(), partB_, grid_, stream_);
}

bool Solver::small_path() const { return small_ && !linked_ && !rccl_ && sumA_ == partA_ && sumB_ == partB_; }

void Solver::iterate(int n, bool use_graph)
{
  if (n <= 0)
    return;
  add_iterations(n);
  if (small_path())
  {
    // one launch runs all n iterations (or stops at convergence) and leaves the state as n rounds
    // of kernels A and B would
    launch_small();
    CHECK(hipGetLastError());
    return;
  }
  const int chunk = std::max(1, env_int("CG_GRAPH_ITERS", 20));
  if (!use_graph)
  {
    for (int i = 0; i < n; i++)
      launch_iteration(i + 1 < n); // the launches of one request pair up; the last one never leaves x alone
    flush_x();
    return;
  }
  // whole chunks replay one captured graph; what is left over replays a second, shorter graph, so that
  // exactly n iterations are launched (iterations beyond maxit would be device-side no-ops, but each
  // still costs a launch)
  auto capture = [&](int iters) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
    // With boundary planes travelling on the second stream the two streams trade places for the capture: the
    // stream the capture starts on takes the plane work and the forked one the kernels and all-gathers.  (A
    // grouped ncclSend / ncclRecv captured on a forked stream sends the HIP runtime into an endless recursion
    // when the graph is built -- RCCL 2.26 / ROCm 7.0 -- while on the capture's own stream it is fine, and so
    // are collectives on a forked stream.)  A graph is a DAG either way: replay does not care which stream a
    // node was recorded on.
    const bool halo = rccl_ && halo_overlap_ && (desc_.has_lo || desc_.has_hi);
    if (halo)
    {
      CHECK(hipEventRecord(ev_cap_begin_, stream_));
      CHECK(hipStreamWaitEvent(side_, ev_cap_begin_, 0));
      std::swap(stream_, side_);
    }
    last_deferred_ = false; // every graph pairs its own launches and ends with x complete (unless a stop falls inside a pair)
    for (int i = 0; i < iters; i++)
      launch_iteration(i + 1 < iters);
    last_deferred_ = false;
    if (halo)
    {
      std::swap(stream_, side_);
      CHECK(hipEventRecord(ev_cap_end_, side_)); // the kernels' stream during the capture
      CHECK(hipStreamWaitEvent(stream_, ev_cap_end_, 0));
    }
    CHECK(hipStreamEndCapture(stream_, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHECK(hipGraphDestroy(g));
    return ge;
  };
  if (n >= chunk && (!graph_ || graph_iters_ != chunk))
  {
    drop_graphs();
    graph_ = capture(chunk);
    graph_iters_ = chunk;
    graph_x_ = x_;
  }
  int left = n;
  for (; left >= chunk; left -= chunk)
    CHECK(hipGraphLaunch(graph_, stream_));
  if (left > 0)
  {
    if (!tail_graph_ || tail_iters_ != left || graph_x_ != x_)
    {
      if (tail_graph_)
        CHECK(hipGraphExecDestroy(tail_graph_));
      tail_graph_ = capture(left);
      tail_iters_ = left;
      graph_x_ = x_;
    }
    CHECK(hipGraphLaunch(tail_graph_, stream_));
  }
  may_be_pending_ = true; // the replayed kernels B pair their x updates
  flush_x(); // x is complete whenever the host gets control back
}

void Solver::drop_graphs()
{
  if (graph_)
    CHECK(hipGraphExecDestroy(graph_));
  if (tail_graph_)
    CHECK(hipGraphExecDestroy(tail_graph_));
  graph_ = nullptr;
  tail_graph_ = nullptr;
  graph_iters_ = tail_iters_ = 0;
}

void Solver::run(int maxit, bool use_graph, int poll)
{
  long long left = maxit < 0 ? (1ll << 40) : maxit;
  CgState s;
  if (small_path())
    poll = 1 << 20; // the single-workgroup kernel stops by itself: no need to come back to the host
  while (left > 0)
  {
    const int n = (int)std::min<long long>(left, poll);
    iterate(n, use_graph);
    left -= n;
    read_state(&s);
    if (s.doneA)
      break;
  }
}

void Solver::read_state(CgState *host_copy)
{
  CHECK(hipMemcpyAsync(st_host_, st_, sizeof(CgState), hipMemcpyDeviceToHost, stream_));
  CHECK(hipStreamSynchronize(stream_));
  *host_copy = *st_host_;
  // keep every r.r the device ring still holds: a solve longer than the ring is polled (run()) more
  // often than the ring wraps, so the host copy ends up complete
  const int n = st_host_->itA;
  if ((int)hist_all_.size() < n)
    hist_all_.resize((size_t)n, NAN);
  for (int i = std::max(hist_seen_, n - CG_HIST + 1); i < n; i++)
    hist_all_[(size_t)i] = st_host_->rr_hist[i % CG_HIST];
  hist_seen_ = std::max(hist_seen_, n);
}

int Solver::finish(float *residue, float *hist, int hist_len)
{
  flush_x();
  if (pad_ && x_user_)
    copy_rows(x_user_, desc_.extents[0], xpad_, s0_); // the read-back below waits for it
  if (relay_ && x_user_)
    relayout(x_user_, xpad_, false);
  CgState s;
  read_state(&s);
  const int n = s.itA;
  // the reference returns the lagged value sqrt(rr at the start of the last iteration),
  // conjugate-gradient_gpu.cu:366,375; with no iteration it returns |r0|
  const int last = n > 0 ? n - 1 : 0;
  if (residue)
    *residue = (float)sqrt((double)s.rr_hist[last % CG_HIST]);
  if (hist)
  {
    // hist[i] = value printed after iteration i+1 (:369); NaN where the device ring was overwritten
    // before the host saw it (more than CG_HIST iterations enqueued without a run() poll or finish())
    for (int i = 0; i < n && i < hist_len; i++)
      hist[i] = (float)sqrt((double)hist_all_[(size_t)i]);
  }
  return n;
}

void Solver::time_kernels(int n, float *ms_A, float *ms_B)
{
  std::vector<hipEvent_t> ev(5 * (size_t)n);
  for (auto &e : ev)
    CHECK(hipEventCreate(&e));
  add_iterations(n);
  for (int i = 0; i < n; i++)
    launch_iteration(i + 1 < n, &ev[5 * (size_t)i]);
  flush_x();
  CHECK(hipStreamSynchronize(stream_));
  double t[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; i++)
    for (int j = 0; j < 4; j++)
    {
      float m;
      CHECK(hipEventElapsedTime(&m, ev[5 * i + j], ev[5 * i + j + 1]));
      t[j] += m;
    }
  for (auto &e : ev)
    CHECK(hipEventDestroy(e));
  if (ms_A)
    *ms_A = (float)(t[0] / n);
  if (ms_B)
    *ms_B = (float)(t[2] / n);
  ms_xA_ = (float)(t[1] / n);
  ms_xB_ = (float)(t[3] / n);
}

// ---- free functions ---------------------------------------------------------------------

void laplace_apply(float *ddf, const float *u, int d, unsigned L, int index_mode, hipStream_t stream)
{
  CG_REQUIRE(d >= 1 && d <= CG_MAXD, "dimension must be in [1,32]");
  // two points per axis: bit k of the lexicographic index is coordinate k, which is the Z-curve index too,
  // so such a hypercube runs on the quad-per-lane Morton kernels (d = 24: 412 -> 129 us)
  const int layout = (index_mode == 2 || (L == 2 && d >= 2)) ? LAYOUT_MORTON : LAYOUT_LEX;
  ScalarGeom sg{};
  sg.d = d;
  unsigned long long s = 1;
  for (int k = 0; k < d; k++)
  {
    sg.n[k] = L;
    sg.stride[k] = s;
    s *= L;
    sg.mmask[k] = morton_mask(d, k);
  }
  sg.N = s;
  VecArgs a{};
  a.u = u;
  a.out = ddf;
  const bool aligned = (((uintptr_t)u | (uintptr_t)ddf) % 16) == 0;
  const bool rows_ok = layout == LAYOUT_LEX && L % 4 == 0 && d <= CG_ROWS_MAXD + 2 && aligned;
  const int engine = layout == LAYOUT_LEX ? choose_engine(rows_ok, rows_ok && d == 3, false, d, L, sg.N) : ENGINE_SCALAR;
  if (engine == ENGINE_PLANES)
  {
    PlanesGeom g;
    unsigned ext[3] = {L, L, L};
    int block = 256, grid = 1, R = 4;
    make_planes_geom(ext, 0, 0, g, block, grid, R);
    launch_planes<MODE_STENCIL, false>(g, a, grid, block, R, stream);
  }
  else if (engine == ENGINE_ROWS)
  {
    RowsGeom g;
    unsigned ext[CG_MAXD];
    for (int k = 0; k < d; k++)
      ext[k] = L;
    int block = 256, grid = 1;
    make_rows_geom(d, ext, 0, 0, g, block, grid);
    launch_rows<MODE_STENCIL, false>(g, a, grid, block, stream);
  }
  else
  {
    if (layout == LAYOUT_MORTON)
    {
      int bits = 0;
      for (unsigned v = L; v > 1; v >>= 1)
        bits++;
      CG_REQUIRE(is_pow2(L) && bits * d <= 32 && sg.N < (1ull << 32), "the Morton layout needs L = 2^m with m*d < 32");
      if (d >= 2 && aligned && sg.N >= 4 && sg.N % 4 == 0)
      {
        MortonGeom g;
        int block = 256, grid = 1;
        make_morton_geom(d, sg.N, g, block, grid);
        launch_morton4<MODE_STENCIL>(g, a, grid, block, stream);
        CHECK(hipGetLastError());
        return;
      }
    }
    const unsigned long long nb = (sg.N + 255) / 256;
    const int grid = (int)std::min<unsigned long long>(nb, (unsigned long long)max_blocks());
    launch_scalar<MODE_STENCIL, false>(sg, a, layout, grid, 256, stream);
  }
  CHECK(hipGetLastError());
}

// launch shape of the stand-alone dot product: a function of n only, so that two calls on the same
// vectors add in the same order
static int dot_grid(unsigned long long n)
{
  const unsigned long long nb = (n / 4 + 255) / 256 + 1;
  return (int)std::min<unsigned long long>(nb, (unsigned long long)max_blocks());
}

// out[0] = (float) sum(part[0..n)), the value inner_product_gpu returns
__global__ void __launch_bounds__(64) k_sum_partials_f32(const double *__restrict__ part, int n, float *__restrict__ out)
{
  __shared__ double sh[8];
  const double s = block_read_sum(part, n, sh);
  if (threadIdx.x == 0)
    out[0] = (float)s;
}

void dot_device_into(const float *v, const float *w, unsigned long long n, double *part, float *out_dev, hipStream_t stream)
{
  const int grid = dot_grid(n);
  launch_dot(v, w, n, part, grid, stream);
  hipLaunchKernelGGL(k_sum_partials_f32, dim3(1), dim3(64), 0, stream, part, grid, out_dev);
  CHECK(hipGetLastError());
}

double dot_device(const float *v, const float *w, unsigned long long n, hipStream_t stream)
{
  // scratch for the partials, one set per device, kept for the life of the process
  constexpr int MAXDEV = 64;
  static double *part_of[MAXDEV] = {nullptr};
  static double *host_of[MAXDEV] = {nullptr};
  static std::mutex guard; // the scratch is shared: one call per device at a time, first use included
  int dev = 0;
  CHECK(hipGetDevice(&dev));
  CG_REQUIRE(dev >= 0 && dev < MAXDEV, "device ordinal out of range");
  std::lock_guard<std::mutex> lock(guard);
  if (!part_of[dev])
  {
    CHECK(hipMalloc((void **)&part_of[dev], (2 * CG_MAX_BLOCKS + 1) * sizeof(double)));
    CHECK(hipHostMalloc((void **)&host_of[dev], sizeof(double)));
  }
  double *part = part_of[dev], *host = host_of[dev];
  const int grid = dot_grid(n);
  launch_dot(v, w, n, part, grid, stream);
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(64), 0, stream, part, grid, part + 2 * CG_MAX_BLOCKS);
  CHECK(hipGetLastError());
  CHECK(hipMemcpyAsync(host, part + 2 * CG_MAX_BLOCKS, sizeof(double), hipMemcpyDeviceToHost, stream));
  CHECK(hipStreamSynchronize(stream));
  return *host;
}

void fill_uniform(float *dst, unsigned long long n, unsigned int seed, hipStream_t stream)
{
  const unsigned long long nb = (n + 255) / 256;
  const int grid = (int)std::min<unsigned long long>(nb, 4096ull);
  hipLaunchKernelGGL(k_fill_uniform, dim3(grid), dim3(256), 0, stream, dst, n, seed);
  CHECK(hipGetLastError());
}

} // namespace cgamd

#if CG_DBG & 64
// the time stamps of a -DCG_DBG=64 build (cg_device.hpp, CG_T)
extern "C" void cg_dbg_times(unsigned long long *out)
{
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(cgamd::cg_dbg_t), 64 * sizeof(unsigned long long));
}
#endif
