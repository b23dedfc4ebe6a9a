// rowsbench.hip -- measurement tool: times the library's rows kernels A and B at 8192^2 with
// parts compiled out (-DCG_DBG=bits: 1 edge loads, 2 lane shifts, 4 prologue reduction,
// 16 halo rows) to price each part.  Results are wrong by construction when bits are set.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../conjugate-gradient_amd/csrc/cg_device.hpp"
using namespace cgamd;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k_init(float *p, size_t n, unsigned seed)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = u01_hash(seed, i);
}

int main(int argc, char **argv)
{
  const unsigned L = 8192; const size_t N = (size_t)L * L; const size_t goff = L, len = N + 4 * L;
  const int bpcu = argc > 1 ? atoi(argv[1]) : 4;
  float *x, *r, *p0, *p1; double *pa, *pb; CgState *st;
  CK(hipMalloc(&x, N * 4)); CK(hipMalloc(&r, len * 4)); CK(hipMalloc(&p0, len * 4)); CK(hipMalloc(&p1, len * 4));
  CK(hipMalloc(&pa, 4096 * 8)); CK(hipMalloc(&pb, 4096 * 8)); CK(hipMalloc(&st, sizeof(CgState)));
  CK(hipMemset(x, 0, N * 4)); CK(hipMemset(r, 0, len * 4)); CK(hipMemset(p0, 0, len * 4)); CK(hipMemset(p1, 0, len * 4));
  k_init<<<2048, 256>>>(x, N, 1); k_init<<<2048, 256>>>(r + goff, N, 2); k_init<<<2048, 256>>>(p0 + goff, N, 3); k_init<<<2048, 256>>>(p1 + goff, N, 4);
  CK(hipMemset(pa, 0, 4096 * 8)); CK(hipMemset(pb, 0, 4096 * 8));
  CgState h; memset(&h, 0, sizeof(h)); h.maxit = 1 << 30; h.lagged = 1; h.reltol = -1.0f;
  for (int i = 0; i < CG_HIST; i++) h.rr_hist[i] = 1.0f;
  CK(hipMemcpy(st, &h, sizeof(h), hipMemcpyHostToDevice));
  RowsGeom g; memset(&g, 0, sizeof(g));
  g.d = 2; g.n0 = L; g.n1 = L; g.nxc = L / 1024; g.bpc = 256 * bpcu / g.nxc; g.nitems = g.nxc * g.bpc;
  VecArgs a; memset(&a, 0, sizeof(a));
  a.x = x; a.r = r + goff; a.p0 = p0 + goff; a.p1 = p1 + goff; a.st = st; a.sumA = pa; a.nA = g.nitems; a.sumB = pb; a.nB = g.nitems;
  hipEvent_t ev[3]; for (auto &v : ev) CK(hipEventCreate(&v));
  float ta = 0, tb = 0; const int reps = 60;
  for (int it = -10; it < reps; it++)
  {
    VecArgs aa = a; aa.part_out = pa; VecArgs bb = a; bb.part_out = pb;
    CK(hipEventRecord(ev[0]));
    hipLaunchKernelGGL((k_rows<MODE_A, true, 2, POL_KEEP_R>), dim3(g.nitems), dim3(256), 0, 0, g, aa);
    CK(hipEventRecord(ev[1]));
    hipLaunchKernelGGL((k_rows<MODE_B, true, 2, POL_KEEP_R>), dim3(g.nitems), dim3(256), 0, 0, g, bb);
    CK(hipEventRecord(ev[2]));
    CK(hipDeviceSynchronize());
    float m; 
    if (it >= 0) { CK(hipEventElapsedTime(&m, ev[0], ev[1])); ta += m; CK(hipEventElapsedTime(&m, ev[1], ev[2])); tb += m; }
  }
  printf("CG_DBG=%d bpcu=%d : A %.4f ms  B %.4f ms  sum %.4f ms\n", CG_DBG, bpcu, ta / reps, tb / reps, (ta + tb) / reps);
  return 0;
}
