// twogrid.hip -- the two-grid correction scheme for the Laplace operator with damped Jacobi as the
// smoother and the float GPU CG as the coarse-grid solver: the "Multigrid" item of the assignment the
// reference implements (tasks.md:32-36; the reference itself stops before it).  Everything runs on the
// device; the host sees one residual norm per cycle.
//
// Grids: L = 2 Lc + 1 interior points per axis (spacing h) over Lc (spacing 2h); coarse point I sits on
// fine point 2I + 1.  The library's operator is the unscaled stencil A = h^2 (-Laplace_h), so the coarse
// correction equation (-Laplace_2h) e = R (r / h^2) reads A_c e_c = 4 R r.
//   smoothing     u <- u + omega / (2d) (f - A u)                      nu1 times before, nu2 times after
//   restriction   full weighting: the tensor product of (1/4, 1/2, 1/4)
//   prolongation  d-linear interpolation: the tensor product of (1/2, 1, 1/2)
#include <math.h>
#include <stdio.h>

#include <algorithm>

#include "../../include/conjugate_gradient_gpu.h"
#include "cg_solver.hpp"
#include "hip_check.hpp"

namespace cgamd
{

// r = f - Au
__global__ void __launch_bounds__(256) k_tg_residual(float *__restrict__ r, const float *__restrict__ f,
                                                     const float *__restrict__ Au, unsigned long long n)
{
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step)
    r[i] = f[i] - Au[i];
}

// u += w (f - Au)
__global__ void __launch_bounds__(256) k_tg_jacobi(float *__restrict__ u, const float *__restrict__ f,
                                                   const float *__restrict__ Au, float w, unsigned long long n)
{
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step)
    u[i] = fmaf(w, f[i] - Au[i], u[i]);
}

// rc = 4 R r: one thread per coarse point; all 3^d fine points around 2I+1 lie inside the grid
__global__ void __launch_bounds__(256) k_tg_restrict(float *__restrict__ rc, const float *__restrict__ r, int d, unsigned L,
                                                     unsigned Lc, unsigned long long Nc)
{
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long I = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; I < Nc; I += step)
  {
    unsigned long long centre = 0, s = 1, rem = I;
    for (int k = 0; k < d; k++)
    {
      centre += (2 * (rem % Lc) + 1) * s;
      rem /= Lc;
      s *= L;
    }
    unsigned combos = 1;
    for (int k = 0; k < d; k++)
      combos *= 3;
    float acc = 0;
    for (unsigned c = 0; c < combos; c++)
    {
      long long off = 0;
      float w = 1.0f;
      unsigned t = c;
      long long sk = 1;
      for (int k = 0; k < d; k++)
      {
        const int delta = (int)(t % 3) - 1;
        t /= 3;
        off += delta * sk;
        w *= delta == 0 ? 0.5f : 0.25f;
        sk *= L;
      }
      acc = fmaf(w, r[(long long)centre + off], acc);
    }
    rc[I] = 4.0f * acc;
  }
}

// u += P ec: one thread per fine point; an odd coordinate sits on a coarse point, an even one between two
// (the one outside the grid is the zero boundary)
__global__ void __launch_bounds__(256) k_tg_prolong_add(float *__restrict__ u, const float *__restrict__ ec, int d, unsigned L,
                                                        unsigned Lc, unsigned long long N)
{
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += step)
  {
    unsigned long long rem = i, sc = 1, base = 0;
    unsigned even = 0;       // axes on which the point lies between two coarse points
    unsigned lo_ok = 0, hi_ok = 0;
    unsigned long long stride[CG_MAXD];
    for (int k = 0; k < d; k++)
    {
      const unsigned c = (unsigned)(rem % L);
      rem /= L;
      stride[k] = sc;
      if (c & 1u)
        base += (unsigned long long)((c - 1) / 2) * sc;
      else
      {
        even |= 1u << k;
        // coarse neighbours c/2 - 1 and c/2
        if (c > 0)
        {
          lo_ok |= 1u << k;
          base += (unsigned long long)(c / 2 - 1) * sc;
        }
        if (c / 2 < Lc)
          hi_ok |= 1u << k;
      }
      sc *= Lc;
    }
    const int ne = __popc(even);
    float acc = 0;
    for (unsigned m = 0; m < (1u << ne); m++)
    {
      // bit j of m picks the upper coarse neighbour on the j-th even axis
      unsigned long long idx = base;
      bool ok = true;
      unsigned bit = 0;
      for (int k = 0; k < d; k++)
      {
        if (!((even >> k) & 1u))
          continue;
        const bool up = (m >> bit) & 1u;
        bit++;
        const bool has_lo = (lo_ok >> k) & 1u;
        if (up)
        {
          ok = ok && ((hi_ok >> k) & 1u);
          idx += has_lo ? stride[k] : 0; // without a lower neighbour `base` already holds index 0 of this axis
        }
        else
          ok = ok && has_lo;
      }
      if (ok)
        acc += ec[idx];
    }
    u[i] += ldexpf(acc, -ne);
  }
}

} // namespace cgamd

using namespace cgamd;

namespace
{

// x += a p ; r -= a q
__global__ void __launch_bounds__(256) k_tg_update(float *__restrict__ x, float *__restrict__ r, const float *__restrict__ p,
                                                   const float *__restrict__ q, float a, unsigned long long n)
{
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step)
  {
    x[i] = fmaf(a, p[i], x[i]);
    r[i] = fmaf(-a, q[i], r[i]);
  }
}

// p = z + b p
__global__ void __launch_bounds__(256) k_tg_xpay(float *__restrict__ p, const float *__restrict__ z, float b,
                                                 unsigned long long n)
{
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step)
    p[i] = fmaf(b, p[i], z[i]);
}

// the two grids, their work vectors and one cycle
struct TwoGrid
{
  int d, L, nu1, nu2;
  unsigned Lc;
  unsigned long long N = 1, Nc = 1;
  float w;
  double coarse_rel;
  Solver *coarse = nullptr;  // the coarse-grid solver of the last level ...
  TwoGrid *child = nullptr;  // ... or the same scheme on the coarse grid (a V-cycle over several levels)
  hipStream_t s = nullptr;
  float *Au = nullptr, *r = nullptr, *rc = nullptr, *ec = nullptr;
  dim3 G, Gc, B{256};
  int inner = 0;

  // levels: grids in the hierarchy, this one included (2: the two-grid scheme); coarsening stops early
  // at a grid whose coarse grid would have an even number of points per axis or fewer than 3
  TwoGrid(int L_, int d_, int nu1_, int nu2_, double omega, double coarse_rel_, int levels = 2)
      : d(d_), L(L_), nu1(nu1_), nu2(nu2_), coarse_rel(coarse_rel_)
  {
    CG_REQUIRE(d >= 1 && d <= 8, "the two-grid scheme is built for 1 <= d <= 8");
    CG_REQUIRE(L >= 3 && (L & 1), "the two-grid scheme needs an odd number of points per axis, L = 2 Lc + 1 >= 3");
    CG_REQUIRE(nu1 >= 0 && nu2 >= 0, "smoothing steps must not be negative");
    Lc = (unsigned)(L - 1) / 2;
    for (int k = 0; k < d; k++)
    {
      N *= (unsigned)L;
      Nc *= Lc;
    }
    CG_REQUIRE(N < (1ull << 31), "grid must fit an int, as in the reference");
    if (omega <= 0)
      omega = 2.0 * d / (2.0 * d + 1.0); // the damping that smooths the upper half of the spectrum best
    w = (float)(omega / (2.0 * d));
    if (levels > 2 && (Lc & 1u) && Lc >= 7)
    {
      child = new TwoGrid((int)Lc, d, nu1, nu2, omega, coarse_rel, levels - 1);
      s = child->s;
    }
    else
    {
      SolverDesc desc;
      desc.d = d;
      for (int k = 0; k < d; k++)
        desc.extents[k] = Lc;
      coarse = new Solver(desc);
      s = coarse->stream();
    }
    CHECK(hipMalloc((void **)&Au, N * sizeof(float)));
    CHECK(hipMalloc((void **)&r, N * sizeof(float)));
    CHECK(hipMalloc((void **)&rc, Nc * sizeof(float)));
    CHECK(hipMalloc((void **)&ec, Nc * sizeof(float)));
    G = dim3((unsigned)std::min<unsigned long long>((N + 255) / 256, 8192ull));
    Gc = dim3((unsigned)std::min<unsigned long long>((Nc + 255) / 256, 8192ull));
  }
  ~TwoGrid()
  {
    (void)hipStreamSynchronize(s);
    (void)hipFree(Au);
    (void)hipFree(r);
    (void)hipFree(rc);
    (void)hipFree(ec);
    delete child;
    delete coarse;
  }
  int levels() const { return child ? 1 + child->levels() : 2; }
  int coarse_iterations() const { return child ? child->coarse_iterations() : inner; }
  TwoGrid(const TwoGrid &) = delete;
  TwoGrid &operator=(const TwoGrid &) = delete;

  void smooth(float *u, const float *f, int steps)
  {
    for (int k = 0; k < steps; k++)
    {
      laplace_apply(Au, u, d, (unsigned)L, 0, s);
      hipLaunchKernelGGL(k_tg_jacobi, G, B, 0, s, u, f, Au, w, N);
    }
  }
  // r = f - A u
  void residual(const float *u, const float *f)
  {
    laplace_apply(Au, u, d, (unsigned)L, 0, s);
    hipLaunchKernelGGL(k_tg_residual, G, B, 0, s, r, f, Au, N);
    CHECK(hipGetLastError());
  }
  double residual_norm(const float *u, const float *f)
  {
    residual(u, f);
    return sqrt(dot_device(r, r, N, s));
  }
  // u <- TG(u) for the right-hand side f
  void cycle(float *u, const float *f)
  {
    smooth(u, f, nu1);
    residual(u, f);
    hipLaunchKernelGGL(k_tg_restrict, Gc, B, 0, s, rc, r, d, (unsigned)L, Lc, Nc);
    CHECK(hipMemsetAsync(ec, 0, Nc * sizeof(float), s));
    if (child)
      child->cycle(ec, rc);
    else
    {
      coarse->begin(rc, ec, coarse_rel, 0);
      coarse->run(100000, true);
      inner += coarse->finish(nullptr, nullptr, 0);
    }
    hipLaunchKernelGGL(k_tg_prolong_add, G, B, 0, s, u, ec, d, (unsigned)L, Lc, N);
    smooth(u, f, nu2);
    CHECK(hipGetLastError());
  }
};

} // namespace

extern "C" int cg_two_grid_gpu(const float *f, float *u, int L, int d, int nu1, int nu2, double omega, int max_cycles,
                               double rel, double coarse_rel, float *hist, int hist_len, int *coarse_iterations)
{
  CG_REQUIRE(max_cycles >= 0, "cycles must not be negative");
  TwoGrid tg(L, d, nu1, nu2, omega, coarse_rel);
  order_after_default_stream(tg.s);
  const double tol = rel * sqrt(dot_device(f, f, tg.N, tg.s));
  int cycles = 0;
  double res = tg.residual_norm(u, f);
  if (hist && hist_len > 0)
    hist[0] = (float)res;
  while (cycles < max_cycles && res > tol)
  {
    tg.cycle(u, f);
    res = tg.residual_norm(u, f);
    cycles++;
    if (hist && cycles < hist_len)
      hist[cycles] = (float)res;
  }
  CHECK(hipStreamSynchronize(tg.s));
  if (coarse_iterations)
    *coarse_iterations = tg.inner;
  return cycles;
}

// The same with the coarse problem handed to the scheme itself: V-cycles over `levels` grids (0: as many
// as the grid allows), CG only on the coarsest.
extern "C" int cg_multigrid_gpu(const float *f, float *u, int L, int d, int levels, int nu1, int nu2, double omega,
                                int max_cycles, double rel, double coarse_rel, float *hist, int hist_len, int *levels_used,
                                int *coarse_iterations)
{
  CG_REQUIRE(max_cycles >= 0 && levels >= 0 && levels != 1, "cycles must not be negative; levels is 0 or at least 2");
  TwoGrid mg(L, d, nu1, nu2, omega, coarse_rel, levels == 0 ? 64 : levels);
  order_after_default_stream(mg.s);
  const double tol = rel * sqrt(dot_device(f, f, mg.N, mg.s));
  int cycles = 0;
  double res = mg.residual_norm(u, f);
  if (hist && hist_len > 0)
    hist[0] = (float)res;
  while (cycles < max_cycles && res > tol)
  {
    mg.cycle(u, f);
    res = mg.residual_norm(u, f);
    cycles++;
    if (hist && cycles < hist_len)
      hist[cycles] = (float)res;
  }
  CHECK(hipStreamSynchronize(mg.s));
  if (levels_used)
    *levels_used = mg.levels();
  if (coarse_iterations)
    *coarse_iterations = mg.coarse_iterations();
  return cycles;
}

// CG preconditioned by one two-grid cycle from a zero guess, z = TG_f=r(0) (tasks.md:36).  With nu1 = nu2
// and an accurate coarse solve the cycle is a symmetric positive definite operator; scalars on the host in
// double, as in the mixed-precision driver.
extern "C" int cg_two_grid_pcg_gpu(const float *f, float *u, int L, int d, int nu, double omega, int max_iters, double rel,
                                   double coarse_rel, float *hist, int hist_len, int *coarse_iterations)
{
  CG_REQUIRE(max_iters >= 0, "iterations must not be negative");
  TwoGrid tg(L, d, nu, nu, omega, coarse_rel);
  const unsigned long long N = tg.N;
  hipStream_t s = tg.s;
  order_after_default_stream(s);
  float *r, *z, *p, *q;
  CHECK(hipMalloc((void **)&r, N * sizeof(float)));
  CHECK(hipMalloc((void **)&z, N * sizeof(float)));
  CHECK(hipMalloc((void **)&p, N * sizeof(float)));
  CHECK(hipMalloc((void **)&q, N * sizeof(float)));
  auto precondition = [&]() { // z = one cycle on A z = r from z = 0
    CHECK(hipMemsetAsync(z, 0, N * sizeof(float), s));
    tg.cycle(z, r);
  };
  const double tol = rel * sqrt(dot_device(f, f, N, s));
  laplace_apply(q, u, d, (unsigned)L, 0, s);
  hipLaunchKernelGGL(k_tg_residual, tg.G, tg.B, 0, s, r, f, q, N);
  double res = sqrt(dot_device(r, r, N, s));
  if (hist && hist_len > 0)
    hist[0] = (float)res;
  int it = 0;
  if (res > tol && max_iters > 0)
  {
    precondition();
    CHECK(hipMemcpyAsync(p, z, N * sizeof(float), hipMemcpyDeviceToDevice, s));
    double rz = dot_device(r, z, N, s);
    while (it < max_iters)
    {
      it++;
      laplace_apply(q, p, d, (unsigned)L, 0, s);
      const double alpha = rz / dot_device(p, q, N, s);
      hipLaunchKernelGGL(k_tg_update, tg.G, tg.B, 0, s, u, r, p, q, (float)alpha, N);
      res = sqrt(dot_device(r, r, N, s));
      if (hist && it < hist_len)
        hist[it] = (float)res;
      if (!(res > tol))
        break;
      precondition();
      const double rz_new = dot_device(r, z, N, s);
      hipLaunchKernelGGL(k_tg_xpay, tg.G, tg.B, 0, s, p, z, (float)(rz_new / rz), N);
      rz = rz_new;
    }
  }
  CHECK(hipGetLastError());
  CHECK(hipStreamSynchronize(s));
  CHECK(hipFree(r));
  CHECK(hipFree(z));
  CHECK(hipFree(p));
  CHECK(hipFree(q));
  if (coarse_iterations)
    *coarse_iterations = tg.inner;
  return it;
}
