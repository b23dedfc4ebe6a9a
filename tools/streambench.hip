// streambench.hip -- measurement tool: plain streaming kernels with the read/write mixes of
// the CG kernels (2R:1W like kernel A, 3R:2W like kernel B, 1R:1W copy), to find what HBM
// delivers for each mix and launch shape on this GPU.  Not part of the library.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
template <int NT>
__device__ __forceinline__ float4 ld(const float4 *p)
{
  if (NT) { v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p)); return make_float4(t.x, t.y, t.z, t.w); }
  return *p;
}
template <int NT>
__device__ __forceinline__ void st(float4 *p, float4 v)
{
  if (NT) { v4f t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(p)); }
  else *p = v;
}

// grid-stride, contiguous across the whole grid per step
template <int MIX, int NT, int UNROLL>
__global__ void __launch_bounds__(256) k_stream(float4 *a, float4 *b, float4 *c, float4 *d, size_t n4, float s)
{
  const size_t step = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (UNROLL - 1) * step < n4; i += UNROLL * step)
  {
    float4 va[UNROLL], vb[UNROLL], vc[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++)
    {
      va[u] = ld<NT>(a + i + u * step);
      if (MIX >= 1) vb[u] = ld<NT>(b + i + u * step);
      if (MIX >= 2) vc[u] = ld<NT>(c + i + u * step);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++)
    {
      float4 o = va[u];
      if (MIX >= 1) { o.x = fmaf(s, vb[u].x, o.x); o.y = fmaf(s, vb[u].y, o.y); o.z = fmaf(s, vb[u].z, o.z); o.w = fmaf(s, vb[u].w, o.w); }
      if (MIX == 0) st<NT>(b + i + u * step, o);          // copy a -> b
      if (MIX == 1) st<NT>(d + i + u * step, o);          // d = a + s b      (2R 1W)
      if (MIX == 2) { st<NT>(a + i + u * step, o); float4 q = vc[u]; q.x = fmaf(-s, vb[u].x, q.x); q.y = fmaf(-s, vb[u].y, q.y); q.z = fmaf(-s, vb[u].z, q.z); q.w = fmaf(-s, vb[u].w, q.w); st<NT>(c + i + u * step, q); } // 3R 2W
    }
  }
}

// column/segment decomposition as in k_rows: each block walks rows of a 4*blockDim-wide column
template <int MIX, int NT>
__global__ void __launch_bounds__(256) k_cols(float4 *a, float4 *b, float4 *c, float4 *d, unsigned n0_4, unsigned n1, unsigned nxc, unsigned bpc, float s)
{
  const unsigned item = blockIdx.x;
  const unsigned seg = item % bpc, xc = item / bpc;
  const size_t y0 = (size_t)seg * n1 / bpc, y1 = (size_t)(seg + 1) * n1 / bpc;
  size_t i = y0 * n0_4 + xc * blockDim.x + threadIdx.x;
  float4 va = ld<NT>(a + i), vb, vc;
  if (MIX >= 1) vb = ld<NT>(b + i);
  if (MIX >= 2) vc = ld<NT>(c + i);
  for (size_t y = y0; y < y1; y++, i += n0_4)
  {
    const size_t in = (y + 1 < y1) ? i + n0_4 : i;
    float4 na = ld<NT>(a + in), nb, nc;
    if (MIX >= 1) nb = ld<NT>(b + in);
    if (MIX >= 2) nc = ld<NT>(c + in);
    float4 o = va;
    if (MIX >= 1) { o.x = fmaf(s, vb.x, o.x); o.y = fmaf(s, vb.y, o.y); o.z = fmaf(s, vb.z, o.z); o.w = fmaf(s, vb.w, o.w); }
    if (MIX == 0) st<NT>(b + i, o);
    if (MIX == 1) st<NT>(d + i, o);
    if (MIX == 2) { st<NT>(a + i, o); float4 q = vc; q.x = fmaf(-s, vb.x, q.x); q.y = fmaf(-s, vb.y, q.y); q.z = fmaf(-s, vb.z, q.z); q.w = fmaf(-s, vb.w, q.w); st<NT>(c + i, q); }
    va = na; vb = nb; vc = nc;
  }
}


// CG-like pair with selectable marching direction per kernel (dir = +1 down, -1 up) and
// nontemporal hints per stream.  A: pn = r + s*po ; B: x += s*pn, r -= s*pn
template <int NTR_, int NTPO, int NTST>
__global__ void __launch_bounds__(256) k_cgA(const float4 *r, const float4 *po, float4 *pn, unsigned n0_4, unsigned n1, unsigned bpc, int dir, float s)
{
  const unsigned item = blockIdx.x;
  const unsigned seg = item % bpc, xc = item / bpc;
  const long long y0 = (long long)seg * n1 / bpc, y1 = (long long)(seg + 1) * n1 / bpc;
  const long long ys = dir > 0 ? y0 : y1 - 1, ye = dir > 0 ? y1 : y0 - 1;
  const size_t colx = xc * blockDim.x + threadIdx.x;
  float4 vr = ld<NTR_>(r + ys * n0_4 + colx), vp = ld<NTPO>(po + ys * n0_4 + colx);
  for (long long y = ys; y != ye; y += dir)
  {
    const long long yn = (y + dir != ye) ? y + dir : y;
    float4 nr = ld<NTR_>(r + yn * n0_4 + colx), np = ld<NTPO>(po + yn * n0_4 + colx);
    float4 o = vr; o.x = fmaf(s, vp.x, o.x); o.y = fmaf(s, vp.y, o.y); o.z = fmaf(s, vp.z, o.z); o.w = fmaf(s, vp.w, o.w);
    st<NTST>(pn + y * n0_4 + colx, o);
    vr = nr; vp = np;
  }
}
template <int NTX, int NTXS, int NTR, int NTRS, int NTP>
__global__ void __launch_bounds__(256) k_cgB(float4 *x, float4 *r, const float4 *p, unsigned n0_4, unsigned n1, unsigned bpc, int dir, float s)
{
  const unsigned item = blockIdx.x;
  const unsigned seg = item % bpc, xc = item / bpc;
  const long long y0 = (long long)seg * n1 / bpc, y1 = (long long)(seg + 1) * n1 / bpc;
  const long long ys = dir > 0 ? y0 : y1 - 1, ye = dir > 0 ? y1 : y0 - 1;
  const size_t colx = xc * blockDim.x + threadIdx.x;
  float4 vx = ld<NTX>(x + ys * n0_4 + colx), vr = ld<NTR>(r + ys * n0_4 + colx), vp = ld<NTP>(p + ys * n0_4 + colx);
  for (long long y = ys; y != ye; y += dir)
  {
    const long long yn = (y + dir != ye) ? y + dir : y;
    float4 nx = ld<NTX>(x + yn * n0_4 + colx), nr = ld<NTR>(r + yn * n0_4 + colx), np = ld<NTP>(p + yn * n0_4 + colx);
    vx.x = fmaf(s, vp.x, vx.x); vx.y = fmaf(s, vp.y, vx.y); vx.z = fmaf(s, vp.z, vx.z); vx.w = fmaf(s, vp.w, vx.w);
    vr.x = fmaf(-s, vp.x, vr.x); vr.y = fmaf(-s, vp.y, vr.y); vr.z = fmaf(-s, vp.z, vr.z); vr.w = fmaf(-s, vp.w, vr.w);
    st<NTXS>(x + y * n0_4 + colx, vx);
    st<NTRS>(r + y * n0_4 + colx, vr);
    vx = nx; vr = nr; vp = np;
  }
}

__global__ void k_rand(float *p, size_t n, unsigned seed)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
  {
    unsigned h = (unsigned)i * 0x9E3779B9u + seed; h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    p[i] = (float)(h >> 8) * (1.0f / 16777216.0f);
  }
}

template <typename F>
static float timeit(F f, int reps)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}


template <int BITS>
static void sweep(float4 *a, float4 *b, float4 *c, float4 *d, unsigned L, unsigned nxc, unsigned bpc, unsigned bs, int dirA, int dirB, size_t n)
{
  if constexpr (BITS < 256)
  {
    constexpr int AR = BITS & 1, AP = (BITS >> 1) & 1, AS = (BITS >> 2) & 1, NX = (BITS >> 3) & 1, NXS = (BITS >> 4) & 1,
                  NR = (BITS >> 5) & 1, NRS = (BITS >> 6) & 1, NP = (BITS >> 7) & 1;
    auto iter = [&](float4 *po, float4 *pn) {
      hipLaunchKernelGGL((k_cgA<AR, AP, AS>), dim3(nxc * bpc), dim3(bs), 0, 0, b, po, pn, L / 4, L, bpc, dirA, 1e-9f);
      hipLaunchKernelGGL((k_cgB<NX, NXS, NR, NRS, NP>), dim3(nxc * bpc), dim3(bs), 0, 0, a, b, pn, L / 4, L, bpc, dirB, 1e-9f);
    };
    float t = timeit([&] { iter(c, d); iter(d, c); }, 8) / 2;
    printf("cgpair bpc %u dirA %+d dirB %+d A(r %d po %d st %d) B(x %d xs %d r %d rs %d p %d) : %.4f ms/iter  %.0f GB/s\n", bpc, dirA, dirB, AR, AP, AS, NX, NXS, NR, NRS, NP, t, 8.0 * n * 4 / t / 1e6);
    sweep<BITS + 1>(a, b, c, d, L, nxc, bpc, bs, dirA, dirB, n);
  }
}

int main()
{
  const unsigned L = 8192; const size_t n = (size_t)L * L, n4 = n / 4;
  float4 *a, *b, *c, *d;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&d, n * 4));
  if (getenv("SB_RANDOM")) { k_rand<<<2048, 256>>>((float *)a, n, 1); k_rand<<<2048, 256>>>((float *)b, n, 2); k_rand<<<2048, 256>>>((float *)c, n, 3); k_rand<<<2048, 256>>>((float *)d, n, 4); CK(hipDeviceSynchronize()); }
  else { CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(c, 0, n * 4)); CK(hipMemset(d, 0, n * 4)); }
  const double words[3] = {2, 3, 5};
  const char *names[3] = {"copy 1R1W", "axpy 2R1W", "upd  3R2W"};
  for (int grid : {1024})
  {
    float t;
#define RUN(MIX, NT, U) t = timeit([&] { hipLaunchKernelGGL((k_stream<MIX, NT, U>), dim3(grid), dim3(256), 0, 0, a, b, c, d, n4, 1e-9f); }, 20); \
    printf("stream grid %6d %s nt=%d unroll=%d : %.4f ms  %.0f GB/s\n", grid, names[MIX], NT, U, t, words[MIX] * n * 4 / t / 1e6);
    RUN(0, 0, 1) RUN(0, 0, 2) RUN(0, 0, 4) RUN(0, 1, 2)
    RUN(1, 0, 1) RUN(1, 0, 2) RUN(1, 0, 4) RUN(1, 1, 2)
    RUN(2, 0, 1) RUN(2, 0, 2) RUN(2, 1, 2)
  }
  for (int bs : {256, 128})
    for (unsigned bpc : {128u})
    {
      const unsigned nxc = L / 4 / bs;
      float t;
#define RUNC(MIX, NT) t = timeit([&] { hipLaunchKernelGGL((k_cols<MIX, NT>), dim3(nxc * bpc), dim3(bs), 0, 0, a, b, c, d, L / 4, L, nxc, bpc, 1e-9f); }, 20); \
      printf("cols bs %d bpc %3u (grid %5u) %s nt=%d : %.4f ms  %.0f GB/s\n", bs, bpc, nxc * bpc, names[MIX], NT, t, words[MIX] * n * 4 / t / 1e6);
      RUNC(0, 0) RUNC(1, 0) RUNC(1, 1) RUNC(2, 0) RUNC(2, 1)
    }

  {
    // a = x, b = r, c = p0, d = p1
    const unsigned bs = 256, nxc = L / 4 / bs;
    for (unsigned bpc : {128u})
      for (int dirB : {1})
        sweep<0>(a, b, c, d, L, nxc, bpc, bs, 1, dirB, n);
  }
  return 0;
}
