// mixbench.hip -- measurement tool (not part of the library): what the memory system delivers for the
// read/write mixes of the two CG kernels on vectors far beyond the Infinity Cache.
//   A-like: d = a + s*b            2 reads, 1 write to a third array
//   A-inpl: a = a + s*b            2 reads, 1 write in place
//   B-like: a += s*c ; b -= s*c    3 reads, 2 writes in place
//   copy, read-only (sum), write-only
// Grid-stride over contiguous float4s, UNROLL independent 16-byte accesses per lane and stream.
// Arrays are carved from one allocation `pad` bytes apart beyond their length (DRAM bank / channel skew);
// `chunk` > 0 replaces the grid-stride sweep by a march: workgroup b walks its own contiguous run of
// `chunk` KiB pieces (the access pattern of the solver's marching kernels).
// usage: mixbench [log2 floats per array, default 27] [grid, default 2048] [pad bytes, default 0] [march 0|1]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
template <int NT>
__device__ __forceinline__ v4f ld(const v4f *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <int NT>
__device__ __forceinline__ void st(v4f *p, v4f v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

enum { COPY, READ2, WRITE1, ALIKE, AINPL, BLIKE, READ3 };

template <int MIX, int NTL, int NTS, int U>
__global__ void __launch_bounds__(256) k_mix(v4f *a, v4f *b, v4f *c, v4f *d, size_t n4, float s, float *sink)
{
  const size_t step = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  v4f acc = {0, 0, 0, 0};
  for (; i + (U - 1) * step < n4; i += U * step)
  {
    v4f va[U], vb[U], vc[U];
#pragma unroll
    for (int u = 0; u < U; u++)
    {
      if (MIX != WRITE1) va[u] = ld<NTL>(a + i + u * step);
      if (MIX == READ2 || MIX == ALIKE || MIX == AINPL || MIX == BLIKE || MIX == READ3) vb[u] = ld<NTL>(b + i + u * step);
      if (MIX == BLIKE || MIX == READ3) vc[u] = ld<NTL>(c + i + u * step);
    }
#pragma unroll
    for (int u = 0; u < U; u++)
    {
      if (MIX == COPY) st<NTS>(b + i + u * step, va[u]);
      if (MIX == READ2) acc += va[u] * vb[u];
      if (MIX == READ3) acc += va[u] * vb[u] + vc[u];
      if (MIX == WRITE1) { v4f o = {s, s, s, s}; st<NTS>(a + i + u * step, o); }
      if (MIX == ALIKE) st<NTS>(d + i + u * step, va[u] + s * vb[u]);
      if (MIX == AINPL) st<NTS>(a + i + u * step, va[u] + s * vb[u]);
      if (MIX == BLIKE) { st<NTS>(a + i + u * step, va[u] + s * vc[u]); st<NTS>(b + i + u * step, vb[u] - s * vc[u]); }
    }
  }
  if (MIX == READ2 || MIX == READ3)
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) *sink = acc.x;
}

// march: workgroup b owns float4s [b*len, (b+1)*len) and walks them 256 float4s (4 KiB) at a time with the
// next piece's loads in flight while the current one is stored
template <int MIX, int NTL, int NTS>
__global__ void __launch_bounds__(256) k_march(v4f *a, v4f *b, v4f *c, v4f *d, size_t n4, float s)
{
  const size_t len = n4 / gridDim.x;
  size_t i = (size_t)blockIdx.x * len + threadIdx.x;
  const size_t end = (size_t)(blockIdx.x + 1) * len;
  v4f va = ld<NTL>(a + i), vb = ld<NTL>(b + i), vc = {0, 0, 0, 0};
  if (MIX == BLIKE) vc = ld<NTL>(c + i);
  for (; i < end; i += 256)
  {
    const size_t in = i + 256 < end ? i + 256 : i;
    v4f na = ld<NTL>(a + in), nb = ld<NTL>(b + in), nc = {0, 0, 0, 0};
    if (MIX == BLIKE) nc = ld<NTL>(c + in);
    if (MIX == ALIKE) st<NTS>(d + i, va + s * vb);
    if (MIX == AINPL) st<NTS>(a + i, va + s * vb);
    if (MIX == BLIKE) { st<NTS>(a + i, va + s * vc); st<NTS>(b + i, vb - s * vc); }
    va = na; vb = nb; vc = nc;
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

int main(int argc, char **argv)
{
  const int lg = argc > 1 ? atoi(argv[1]) : 27;
  const int grid = argc > 2 ? atoi(argv[2]) : 2048;
  const size_t n = (size_t)1 << lg, n4 = n / 4;
  const size_t pad = argc > 3 ? (size_t)atoll(argv[3]) : 0;
  const int march = argc > 4 ? atoi(argv[4]) : 0;
  char *base; float *sink;
  const size_t pitch = n * 4 + pad;
  CK(hipMalloc(&base, 4 * pitch)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(base, 0, 4 * pitch));
  v4f *a = (v4f *)base, *b = (v4f *)(base + pitch), *c = (v4f *)(base + 2 * pitch), *d = (v4f *)(base + 3 * pitch);
  const char *names[] = {"copy   1R1W", "read   2R  ", "write    1W", "A-like 2R1W", "A-inpl 2R1W", "B-like 3R2W", "read   3R  "};
  const double words[] = {2, 2, 1, 3, 3, 5, 3};
  printf("2^%d floats per array (%.0f MiB), grid %d x 256, arrays %zu bytes apart beyond their length%s\n", lg,
         n * 4.0 / 1048576, grid, pad, march ? ", march" : "");
  if (march)
  {
#define RUNM(MIX, NTL, NTS) { float t = timeit([&] { hipLaunchKernelGGL((k_march<MIX, NTL, NTS>), dim3(grid), dim3(256), 0, 0, a, b, c, d, n4, 1e-9f); }, 10); \
    printf("march %s ntload=%d ntstore=%d : %8.1f us  %6.0f GB/s\n", names[MIX], NTL, NTS, t * 1e3, words[MIX] * n * 4 / t / 1e6); }
    RUNM(ALIKE, 0, 0) RUNM(ALIKE, 0, 1) RUNM(ALIKE, 1, 1) RUNM(AINPL, 0, 1) RUNM(AINPL, 1, 1) RUNM(BLIKE, 0, 0) RUNM(BLIKE, 0, 1) RUNM(BLIKE, 1, 1)
    return 0;
  }
#define RUN(MIX, NTL, NTS, U) { float t = timeit([&] { hipLaunchKernelGGL((k_mix<MIX, NTL, NTS, U>), dim3(grid), dim3(256), 0, 0, a, b, c, d, n4, 1e-9f, sink); }, 10); \
    printf("%s ntload=%d ntstore=%d unroll=%d : %8.1f us  %6.0f GB/s\n", names[MIX], NTL, NTS, U, t * 1e3, words[MIX] * n * 4 / t / 1e6); }
#define RUNALL(MIX) RUN(MIX, 0, 0, 2) RUN(MIX, 0, 1, 2) RUN(MIX, 1, 1, 2) RUN(MIX, 1, 1, 4) RUN(MIX, 0, 1, 4)
  RUNALL(COPY) RUNALL(READ2) RUNALL(READ3) RUNALL(WRITE1) RUNALL(ALIKE) RUNALL(AINPL) RUNALL(BLIKE)
  return 0;
}
