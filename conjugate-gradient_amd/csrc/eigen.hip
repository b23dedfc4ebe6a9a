// eigen.hip -- smallest and largest eigenvalue of the discrete Laplace operator by power iteration
// "together with CG", the third item of the assignment the reference implements (tasks.md:10-24; the
// reference stops after items 1 and 2).  lambda_max: q <- A q / |A q| with the matrix-free stencil;
// lambda_min: inverse iteration, q <- A^-1 q / |A^-1 q| with the float CG as A^-1.  Everything stays
// on the device; the host sees two dot products per step.
#include <math.h>
#include <stdio.h>

#include <algorithm>

#include "../../include/conjugate_gradient_gpu.h"
#include "cg_solver.hpp"
#include "hip_check.hpp"

namespace cgamd
{

__global__ void __launch_bounds__(256) k_scale(float *__restrict__ dst, const float *__restrict__ src, float a,
                                               unsigned long long n)
{
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step)
    dst[i] = a * src[i];
}

} // namespace cgamd

using namespace cgamd;

extern "C" int cg_eigen_extremes_gpu(int L, int d, int max_steps, double tol, unsigned seed, double *lambda_max,
                                     int *steps_max, double *lambda_min, int *steps_min, int *cg_iterations)
{
  CG_REQUIRE(L >= 1 && d >= 1 && d <= CG_MAXD, "need L >= 1 and 1 <= d <= 32");
  CG_REQUIRE(max_steps >= 1 && tol >= 0, "need max_steps >= 1 and tol >= 0");
  unsigned long long N = 1;
  for (int k = 0; k < d; k++)
    N *= (unsigned)L;
  CG_REQUIRE(N < (1ull << 31), "grid must fit an int, as in the reference");

  SolverDesc desc;
  desc.d = d;
  for (int k = 0; k < d; k++)
    desc.extents[k] = (unsigned)L;
  Solver cg(desc);
  hipStream_t s = cg.stream();
  order_after_default_stream(s);
  float *q, *z;
  CHECK(hipMalloc((void **)&q, N * sizeof(float)));
  CHECK(hipMalloc((void **)&z, N * sizeof(float)));
  const dim3 G((unsigned)std::min<unsigned long long>((N + 255) / 256, 4096ull)), B(256);
  auto normalise = [&](float *dst, const float *src) -> double {
    const double nn = dot_device(src, src, N, s);
    hipLaunchKernelGGL(k_scale, G, B, 0, s, dst, src, (float)(1.0 / sqrt(nn)), N);
    CHECK(hipGetLastError());
    return nn;
  };

  // ---- largest eigenvalue: z = A q, lambda = q.z with |q| = 1, q = z / |z| --------------------
  int done_max = 0;
  if (lambda_max)
  {
    fill_uniform(z, N, seed, s);
    normalise(q, z);
    double lam = 0, prev = 0;
    for (int k = 1; k <= max_steps; k++)
    {
      laplace_apply(z, q, d, (unsigned)L, 0, s);
      lam = dot_device(q, z, N, s);
      normalise(q, z);
      done_max = k;
      if (k > 1 && fabs(lam - prev) <= tol * fabs(lam))
        break;
      prev = lam;
    }
    *lambda_max = lam;
  }

  // ---- smallest eigenvalue: A z = q by CG, lambda = (z.A z)/(z.z) = (z.q)/(z.z), q = z / |z| ----
  int done_min = 0, cg_total = 0;
  if (lambda_min)
  {
    fill_uniform(z, N, seed + 1u, s);
    normalise(q, z);
    double lam = 0, prev = 0;
    for (int k = 1; k <= max_steps; k++)
    {
      CHECK(hipMemsetAsync(z, 0, N * sizeof(float), s));
      // |q| = 1: the relative tolerance is the absolute one; capped, since float CG stalls at large kappa
      cg.begin(q, z, 1e-6, 0);
      cg.run(20000, true);
      cg_total += cg.finish(nullptr, nullptr, 0);
      const double zq = dot_device(z, q, N, s);
      const double zz = normalise(q, z);
      lam = zq / zz;
      done_min = k;
      if (k > 1 && fabs(lam - prev) <= tol * fabs(lam))
        break;
      prev = lam;
    }
    *lambda_min = lam;
  }
  CHECK(hipStreamSynchronize(s));
  CHECK(hipFree(q));
  CHECK(hipFree(z));
  if (steps_max)
    *steps_max = done_max;
  if (steps_min)
    *steps_min = done_min;
  if (cg_iterations)
    *cg_iterations = cg_total;
  return 0;
}
