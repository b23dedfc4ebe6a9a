// barrierbench.hip -- measurement tool (not part of the library): what a grid-wide barrier costs on this GPU,
// to decide whether a single persistent kernel with barriers could beat two launches per CG iteration on
// grids that are bound by launch latency.  All workgroups are resident (grid <= CUs x per-CU share); the
// barrier is a monotonically increasing counter in device memory: release fence, atomic add, spin until the
// counter reaches epoch x grid, acquire fence.  Every spin is bounded, so a mistake ends the kernel.
// usage: barrierbench [workgroups, default 256] [lanes, default 256] [barriers, default 2000] [floats of work per lane and step, default 0]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_barriers(unsigned *bar, int *err, int K, float *data, int work, int fence)
{
  const unsigned G = gridDim.x;
  const size_t me = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)G * blockDim.x;
  float acc = 0;
  for (int e = 1; e <= K; e++)
  {
    // "work": every lane writes its own words and reads those of the lane one workgroup away, as a stencil
    // across workgroup boundaries would
    for (int w = 0; w < work; w++)
    {
      data[(size_t)w * total + me] = acc + (float)e;
      acc += data[(size_t)w * total + (me + blockDim.x) % total] * 1e-9f;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
      if (fence)
        __threadfence();
      atomicAdd(bar, 1u);
      const unsigned target = (unsigned)e * G;
      long spins = 0;
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
      {
        if (++spins > (1l << 22))
        {
          *err = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (fence)
        __threadfence();
    }
    __syncthreads();
    if (*err)
      break;
  }
  if (acc == 123.0f)
    data[me] = acc;
}

int main(int argc, char **argv)
{
  const int G = argc > 1 ? atoi(argv[1]) : 256;
  const int T = argc > 2 ? atoi(argv[2]) : 256;
  const int K = argc > 3 ? atoi(argv[3]) : 2000;
  const int work = argc > 4 ? atoi(argv[4]) : 0;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int per_cu = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_barriers, T, 0));
  if (G > prop.multiProcessorCount * per_cu)
  {
    printf("grid %d exceeds what is resident at once (%d x %d)\n", G, prop.multiProcessorCount, per_cu);
    return 1;
  }
  unsigned *bar;
  int *err;
  float *data;
  CK(hipMalloc(&bar, 4));
  CK(hipMalloc(&err, 4));
  CK(hipMalloc(&data, (size_t)(work > 0 ? work : 1) * G * T * 4));
  CK(hipMemset(data, 0, (size_t)(work > 0 ? work : 1) * G * T * 4));
  for (int fence = 0; fence <= 1; fence++)
  {
    for (int rep = 0; rep < 2; rep++)
    {
      CK(hipMemset(bar, 0, 4));
      CK(hipMemset(err, 0, 4));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_barriers, dim3(G), dim3(T), 0, 0, bar, err, K, data, work, fence);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      int herr = 0;
      CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      if (rep)
        printf("grid %d x %d lanes, %d barriers, %d floats of work, fences %d: %.3f us per step%s\n", G, T, K, work, fence,
               ms * 1e3 / K, herr ? "  (TIMED OUT)" : "");
    }
  }
  return 0;
}
