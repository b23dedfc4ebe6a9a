// conjugate_gradient_gpu.hpp -- C++ (hipcc) host API of libconjugate_gradient_gpu.so: the
// declarations a HIP translation unit needs to be a caller in the way the reference's
// main.cu / tests.cu are callers of conjugate-gradient_gpu.cuh:10-50.  The kernels keep
// the reference's names and parameters and may be launched with any 1-D shape.
#pragma once
#include <hip/hip_runtime.h>

#include "conjugate_gradient_gpu.h"

// conjugate-gradient_gpu.cuh:10 -- N+1 zeroed floats, host- and device-dereferenceable
float *cuda_allocate_field(int N);

// conjugate-gradient_gpu.cuh:20-21 -- one grid point per thread; index_mode 0/1
// lexicographic, 2 Morton; out-of-grid neighbours are zero (slot N is never read)
__global__ void laplace_gpu(float *ddf, float *u, unsigned int d, unsigned int L, unsigned int N,
                            unsigned int index_mode);

// conjugate-gradient_gpu.cuh:23-24 -- *g_odata += sum v[i]*w[i]; g_odata must be zeroed by
// the caller; dynamic LDS of nthreads floats is accepted and unused
__global__ void reduceMulAddComplete(float *v, float *w, float *g_odata, unsigned int n, const unsigned int nthreads);

// conjugate-gradient_gpu.cu:293-308
__global__ void muladd(float *A, float b, float *B, int N);
__global__ void muladd3(float *A, float *C, float b, float *B, int N);

// conjugate-gradient_gpu.cuh:49
__global__ void fillArray(float *arr, float value, int size);

// conjugate-gradient_gpu.cuh:28
float norm(float *v, int N);

// conjugate-gradient_gpu.cuh:50 -- glibc rand()/RAND_MAX fill through a host-visible pointer
float *random_array(float *r, int L, int d, int N);

// conjugate-gradient_gpu.cuh:16-18 declares neighbour_index_gpu __host__ __device__.  The host side is the
// library's exported symbol (conjugate_gradient_gpu.h); this is its device-side twin for a caller's own
// kernels: the same rule, inline (device code cannot link against a shared object).  Lexicographic
// (index_mode 0, 1): the step leaves the grid when the coordinate, kept in an unsigned, wraps to -1 or
// reaches L (conjugate-gradient_gpu.cu:131-146).  Morton (index_mode 2): masked increment / decrement of
// the coordinate's bits (interleave.cuh:43-67), N for a step out of the 2^(m d)-point grid.
extern "C" __device__ inline unsigned int neighbour_index_gpu(unsigned int ind, int direction, int amount, unsigned int L,
                                                              unsigned int d, unsigned int N, int index_mode)
{
  if (index_mode == 2)
  {
    if (amount == 0)
      return ind;
    unsigned int base = 0;
    for (unsigned int i = 0; i < 32; i += d)
      base = (base << d) + 1u;
    const unsigned int mask = base << direction, cm = mask & (N - 1u);
    if (amount > 0)
      return (ind & cm) == cm ? N : ((ind & ~mask) | ((((ind & mask) | ~mask) + 1u) & mask));
    return (ind & cm) == 0u ? N : ((ind & ~mask) | (((ind & mask) - 1u) & mask));
  }
  unsigned int cord = ind;
  int n = 1;
  for (int i = 0; i < direction; i++)
  {
    cord /= L;
    n *= (int)L;
  }
  cord = cord % L + (unsigned int)amount;
  if (cord == 0xffffffffu || cord == L)
    return N;
  return ind + (unsigned int)(amount * n);
}

// conjugate-gradient_gpu.cuh:12-14 declare two more __device__ helpers that nothing in the reference calls
// (conjugate-gradient_gpu.cu:57-93); they are kept so that a caller's kernel written against that header
// still compiles.  index_to_cords_cu: cords[i] = coordinate of `index` along axis i, axis 0 fastest (the layout
// of every vector here).  get_index_gpu: Horner's rule over cords[0], cords[1], ... -- cords[0] ends up as the
// MOST significant digit, the opposite of index_to_cords_cu (the reference's own inconsistency; the two are
// inverses of each other only for palindromic coordinates).  N is unused, as in the reference.
__device__ inline int *index_to_cords_cu(int *cords, int index, int L, int d)
{
  for (int i = 0; i < d; i++)
  {
    cords[i] = index % L;
    index /= L;
  }
  return cords;
}

__device__ inline int get_index_gpu(int *cords, int L, int d, int N)
{
  (void)N;
  int index = 0;
  for (int i = 0; i < d; i++)
  {
    index += cords[i];
    if (i < d - 1)
      index *= L;
  }
  return index;
}
