// compat_kernels.hip -- the reference's kernel names with the reference's parameters
// (conjugate-gradient_gpu.cuh:20-24,49; conjugate-gradient_gpu.cu:293-308), for callers
// that launch them directly with their own 1-D shapes (tests.cu:354,393,421,460,477;
// julia.cu:25,58).  The CG solver itself does not use these: its iteration runs on the
// fused kernels of cg_device.hpp.
#include <hip/hip_runtime.h>

#include "../../include/conjugate_gradient_gpu.hpp"
#include "index_map.hpp"

using namespace cgamd;

__global__ void fillArray(float *arr, float value, int size)
{
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < size)
    arr[idx] = value;
}

// One thread per point.  The coordinate along each axis comes from one running
// division (the reference re-divides from scratch per neighbour, O(d^2)); the boundary
// is a literal zero, so u needs no slot N.
__global__ void laplace_gpu(float *ddf, float *u, unsigned int d, unsigned int L, unsigned int N,
                            unsigned int index_mode)
{
  const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= N)
    return;
  const unsigned int ind = (unsigned int)gid;
  const float c = u[ind];
  float laplace_value = 0.0f;
  if (index_mode == 2)
  {
    const unsigned int base = morton_base_mask((int)d);
    for (unsigned int k = 0; k < d; k++)
    {
      const unsigned int m = base << k;
      const unsigned int ip = morton_up(ind, m, N), im = morton_dn(ind, m, N);
      const float up = ip < N ? u[ip] : 0.0f;
      const float um = im < N ? u[im] : 0.0f;
      laplace_value = laplace_value + ((-up + 2.0f * c) - um);
    }
  }
  else
  {
    unsigned int rem = ind;
    unsigned long long stride = 1;
    for (unsigned int k = 0; k < d; k++)
    {
      const unsigned int ck = rem % L;
      rem /= L;
      const float up = (ck + 1 < L) ? u[ind + stride] : 0.0f;
      const float um = (ck > 0) ? u[ind - stride] : 0.0f;
      laplace_value = laplace_value + ((-up + 2.0f * c) - um);
      stride *= L;
    }
  }
  ddf[ind] = laplace_value;
}

// Same contract as the reference kernel: adds the dot product of v and w to *g_odata with
// one float atomic per workgroup (so, like the reference's, the result depends on the
// order in which workgroups arrive; inner_product_gpu does not use this kernel).  Reads
// are bounds-checked (the reference reads v[i + blockDim.x] unguarded, gpu.cu:207).
__global__ void reduceMulAddComplete(float *v, float *w, float *g_odata, unsigned int n, const unsigned int nthreads)
{
  (void)nthreads;
  __shared__ float wsum[16];
  const unsigned long long gridSize = (unsigned long long)blockDim.x * 2ull * gridDim.x;
  float sum = 0.0f;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x * 2ull + threadIdx.x; i < n; i += gridSize)
  {
    float t = v[i] * w[i];
    const unsigned long long j = i + blockDim.x;
    if (j < n)
      t += v[j] * w[j];
    sum += t;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    sum += __shfl_xor(sum, o);
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0)
    wsum[wave] = sum;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    float s = 0.0f;
    for (int i = 0; i < nw; i++)
      s += wsum[i];
    atomicAdd(g_odata, s);
  }
}

__global__ void muladd(float *A, float b, float *B, int N)
{
  const long long ind = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (ind >= N)
    return;
  A[ind] = fmaf(b, B[ind], A[ind]);
}

__global__ void muladd3(float *A, float *C, float b, float *B, int N)
{
  const long long ind = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (ind >= N)
    return;
  A[ind] = fmaf(b, B[ind], C[ind]);
}

// julia.cu:44-49
__global__ void add(float *A, float *B)
{
  const long long idx = threadIdx.x + (long long)blockIdx.x * blockDim.x;
  A[idx] += B[idx];
}

void cgamd_launch_add(float *A, float *B, int blocks, int threads)
{
  hipLaunchKernelGGL(add, dim3(blocks), dim3(threads), 0, 0, A, B);
}

void cgamd_launch_laplace_compat(float *ddf, float *u, unsigned d, unsigned L, unsigned N, unsigned index_mode,
                                 unsigned blocks, unsigned threads)
{
  hipLaunchKernelGGL(laplace_gpu, dim3(blocks), dim3(threads), 0, 0, ddf, u, d, L, N, index_mode);
}
