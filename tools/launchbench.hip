// launchbench.hip -- what one dependent kernel costs inside a replayed hipGraph on this GPU: chains of 200 kernel
// nodes on one stream, (a) empty kernels, (b) kernels that read one word written by their predecessor and
// write one, (c) the same with a 64-workgroup block reduction of 64 doubles in between, as the prologue of a
// CG kernel does.  Build: hipcc -O3 --offload-arch=gfx950 tools/launchbench.hip -o tools/bin/launchbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

#define CK(x)                                                                                  \
  do                                                                                           \
  {                                                                                            \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess)                                                                      \
    {                                                                                          \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                \
      exit(1);                                                                                 \
    }                                                                                          \
  } while (0)

__global__ void k_empty() {}

__global__ void k_chain(const double *in, double *out)
{
  if (threadIdx.x == 0)
    out[blockIdx.x] = in[blockIdx.x] + 1.0;
}

__global__ void k_reduce(const double *in, double *out, int n)
{
  __shared__ double sh[4];
  double v = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    v += in[i];
  for (int o = 32; o > 0; o >>= 1)
    v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0)
    sh[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0)
    out[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3] + 1e-9;
}

template <class F>
static double time_graph(hipStream_t s, int nodes, F enqueue)
{
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < nodes; i++)
    enqueue(i);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int w = 0; w < 3; w++)
    CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  const int reps = 20;
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; r++)
    CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  return us / reps / nodes;
}

int main()
{
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  double *a, *b;
  CK(hipMalloc(&a, 4096 * sizeof(double)));
  CK(hipMalloc(&b, 4096 * sizeof(double)));
  CK(hipMemset(a, 0, 4096 * sizeof(double)));
  CK(hipMemset(b, 0, 4096 * sizeof(double)));
  const int nodes = 200;
  for (int grid : {1, 64, 256, 1024})
  {
    const double e = time_graph(s, nodes, [&](int) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, s); });
    const double c = time_graph(s, nodes, [&](int i) {
      hipLaunchKernelGGL(k_chain, dim3(grid), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b);
    });
    const double r = time_graph(s, nodes, [&](int i) {
      hipLaunchKernelGGL(k_reduce, dim3(grid), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, grid);
    });
    printf("grid %5d x 256: empty %.2f us/node, one dependent word %.2f, reduction of the predecessor's %d doubles %.2f\n",
           grid, e, c, grid, r);
  }
  return 0;
}
