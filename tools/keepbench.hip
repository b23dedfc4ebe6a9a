// keepbench.hip -- measurement tool (not part of the library): the two-kernel CG traffic pattern on an
// 8192^2 grid (every vector 256 MiB = the Infinity Cache) with only a FRACTION of r's rows accessed through
// the caches and everything else streamed with the non-temporal hint.  Question: does a smaller resident set
// (that really stays resident) beat asking the cache to keep all of r?
//   A: pn = r + s*po        B: x += s*pn ; r -= s*pn
// usage: keepbench [L, default 8192]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ v4f ld(const v4f *p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }
template <bool NT>
__device__ __forceinline__ void st(v4f *p, v4f v) { if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// rows [ya, yb) of the grid, cut into bpc segments per column; KEEP: r through the caches, everything else
// (and r otherwise) non-temporal.  (The hint must be a compile-time property of the access: a run-time
// choice between the two kinds of load of one address is folded into a plain load.)
template <bool KEEP>
__global__ void __launch_bounds__(256) k_A(const v4f *r, const v4f *po, v4f *pn, unsigned n0_4, unsigned ya, unsigned yb, unsigned bpc, float s)
{
  const unsigned seg = blockIdx.x % bpc, xc = blockIdx.x / bpc, n1 = yb - ya;
  const size_t y0 = ya + (size_t)seg * n1 / bpc, y1 = ya + (size_t)(seg + 1) * n1 / bpc;
  const size_t colx = (size_t)xc * blockDim.x + threadIdx.x;
  v4f vr = ld<!KEEP>(r + y0 * n0_4 + colx), vp = ld<true>(po + y0 * n0_4 + colx);
  for (size_t y = y0; y < y1; y++)
  {
    const size_t yn = y + 1 < y1 ? y + 1 : y;
    v4f nr = ld<!KEEP>(r + yn * n0_4 + colx), np = ld<true>(po + yn * n0_4 + colx);
    st<true>(pn + y * n0_4 + colx, vr + s * vp);
    vr = nr; vp = np;
  }
}
template <bool KEEP, bool KEEP_STORE>
__global__ void __launch_bounds__(256) k_B(v4f *x, v4f *r, const v4f *p, unsigned n0_4, unsigned ya, unsigned yb, unsigned bpc, float s)
{
  const unsigned seg = blockIdx.x % bpc, xc = blockIdx.x / bpc, n1 = yb - ya;
  const size_t y0 = ya + (size_t)seg * n1 / bpc, y1 = ya + (size_t)(seg + 1) * n1 / bpc;
  const size_t colx = (size_t)xc * blockDim.x + threadIdx.x;
  v4f vx = ld<true>(x + y0 * n0_4 + colx), vr = ld<!KEEP>(r + y0 * n0_4 + colx), vp = ld<true>(p + y0 * n0_4 + colx);
  for (size_t y = y0; y < y1; y++)
  {
    const size_t yn = y + 1 < y1 ? y + 1 : y;
    v4f nx = ld<true>(x + yn * n0_4 + colx), nr = ld<!KEEP>(r + yn * n0_4 + colx), np = ld<true>(p + yn * n0_4 + colx);
    st<true>(x + y * n0_4 + colx, vx + s * vp);
    st<!(KEEP && KEEP_STORE)>(r + y * n0_4 + colx, vr - s * vp);
    vx = nx; vr = nr; vp = np;
  }
}
int main(int argc, char **argv)
{
  const unsigned L = argc > 1 ? atoi(argv[1]) : 8192;
  const size_t n = (size_t)L * L;
  v4f *x, *r, *p0, *p1;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&r, n * 4)); CK(hipMalloc(&p0, n * 4)); CK(hipMalloc(&p1, n * 4));
  CK(hipMemset(x, 0, n * 4)); CK(hipMemset(r, 0, n * 4)); CK(hipMemset(p0, 0, n * 4)); CK(hipMemset(p1, 0, n * 4));
  const unsigned bs = 256, nxc = L / 4 / bs, bpc = 1024 / nxc;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rstore = 0; rstore < 2; rstore++)
    for (int f8 = 0; f8 <= 8; f8++)
    {
      // rows [0, ykeep) keep r in the cache (their own launches), the others stream it; segments scale with rows
      const unsigned ykeep = (unsigned)((size_t)L * f8 / 8);
      const unsigned bk = bpc * f8 / 8, bs2 = bpc - bk;
      auto iter = [&](v4f *po, v4f *pn) {
        if (bk) hipLaunchKernelGGL(k_A<true>, dim3(nxc * bk), dim3(bs), 0, 0, r, po, pn, L / 4, 0u, ykeep, bk, 1e-9f);
        if (bs2) hipLaunchKernelGGL(k_A<false>, dim3(nxc * bs2), dim3(bs), 0, 0, r, po, pn, L / 4, ykeep, L, bs2, 1e-9f);
        if (bk && rstore) hipLaunchKernelGGL((k_B<true, true>), dim3(nxc * bk), dim3(bs), 0, 0, x, r, pn, L / 4, 0u, ykeep, bk, 1e-9f);
        if (bk && !rstore) hipLaunchKernelGGL((k_B<true, false>), dim3(nxc * bk), dim3(bs), 0, 0, x, r, pn, L / 4, 0u, ykeep, bk, 1e-9f);
        if (bs2) hipLaunchKernelGGL((k_B<false, false>), dim3(nxc * bs2), dim3(bs), 0, 0, x, r, pn, L / 4, ykeep, L, bs2, 1e-9f);
      };
      for (int i = 0; i < 4; i++) { iter(p0, p1); iter(p1, p0); }
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 0; i < 10; i++) { iter(p0, p1); iter(p1, p0); }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("L %u  r kept in cache for %d/8 of the rows, r store %s : %.4f ms/iteration  %.0f GB/s moved\n", L, f8,
             rstore ? "cached" : "streamed", ms / 20, 32.0 * n / (ms / 20) / 1e6);
    }
  return 0;
}
