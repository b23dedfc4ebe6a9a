// tests.hip -- assert-based GPU test executable in the manner of the reference's
// tests.cu:327-525: the callers launch the library's kernels with their own shapes and
// dereference managed fields on the host.
#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/conjugate_gradient_cpu.h"
#include "../../include/conjugate_gradient_gpu.hpp"

#define OK(call)                                                              \
  {                                                                           \
    hipError_t e_ = (call);                                                   \
    if (e_ != hipSuccess)                                                     \
    {                                                                         \
      fprintf(stderr, "Error: %s:%d, code: %d, reason: %s\n", __FILE__, __LINE__, (int)e_, hipGetErrorString(e_)); \
      exit(1);                                                                \
    }                                                                         \
  }

static bool every_a(const float *x, const float *y, int n)
{
  for (int i = 0; i < n; i++)
    if (!(fabs(x[i] - y[i]) < 1e-3f))
      return false;
  return true;
}

static void test_inner_product_gpu()
{
  int N = 1000;
  float *x = cuda_allocate_field(N);
  fillArray<<<1, 1024>>>(x, 1, N);
  OK(hipDeviceSynchronize());
  float r = inner_product_gpu(x, x, N);
  printf("x*x =%f\n", r);
  assert(r == N);
  cg_free_field(x);
}

static void test_reduce_kernel()
{
  int N = 1000;
  float *x = cuda_allocate_field(N);
  float *acc = cuda_allocate_field(1);
  fillArray<<<4, 256>>>(x, 2, N);
  reduceMulAddComplete<<<2, 128, 128 * sizeof(float)>>>(x, x, acc, N, 128);
  OK(hipDeviceSynchronize());
  assert(acc[0] == 4.0f * N);
  cg_free_field(x);
  cg_free_field(acc);
}

static void test_laplace_1d(bool sine)
{
  int N = 1000;
  float step = (2 * M_PI) / (N - 1);
  float *ddf = cuda_allocate_field(N);
  float *u = cuda_allocate_field(N);
  for (int i = 0; i < N; i++)
  {
    float x = -M_PI + i * step;
    u[i] = sine ? sinf(x) : x * x;
  }
  laplace_gpu<<<(N + 255) / 256, 256>>>(ddf, u, 1, N, N, 0);
  OK(hipDeviceSynchronize());
  for (int i = 1; i < N - 1; i++)
  {
    float a = sine ? ddf[i] / u[i] : ddf[i];
    float b = sine ? ddf[N / 2] / u[N / 2] : ddf[N / 2];
    assert(fabsf(a - b) < 1e-3);
  }
  cg_free_field(ddf);
  cg_free_field(u);
}

// tests.cu:447-464: 100^4 points of zeros through the one-point-per-thread kernel on managed
// fields must not fault (the reference added it after an illegal access seen from Julia)
static void test_laplace_large()
{
  unsigned int L = 100, d = 4, N = 100000000u;
  float *ddf = cuda_allocate_field((int)N);
  float *u = cuda_allocate_field((int)N);
  laplace_gpu<<<(N + 255) / 256, 256>>>(ddf, u, d, L, N, 0);
  OK(hipDeviceSynchronize());
  assert(ddf[0] == 0.0f && ddf[N - 1] == 0.0f);
  cg_free_field(ddf);
  cg_free_field(u);
}

static void test_muladd()
{
  int N = 777;
  float *a = cuda_allocate_field(N), *b = cuda_allocate_field(N), *c = cuda_allocate_field(N);
  for (int i = 0; i < N; i++)
  {
    a[i] = i;
    b[i] = 1;
    c[i] = 2;
  }
  muladd<<<N, 1>>>(a, 0.5f, b, N);
  muladd3<<<4, 256>>>(b, c, 3.0f, b, N);
  OK(hipDeviceSynchronize());
  for (int i = 0; i < N; i++)
    assert(a[i] == i + 0.5f && b[i] == 5.0f);
  cg_free_field(a);
  cg_free_field(b);
  cg_free_field(c);
}

static bool test_cg_gpu(int L, int d, int N)
{
  float *x = cuda_allocate_field(N);
  float *b = cuda_allocate_field(N);
  random_array(x, L, d, N);
  laplace_gpu<<<N, 1>>>(b, x, d, L, N, 1);
  OK(hipDeviceSynchronize());
  float *x0 = cuda_allocate_field(N);
  conjugate_gradient_gpu(b, x0, L, d);
  bool passed = every_a(x, x0, N);
  cg_free_field(x);
  cg_free_field(b);
  cg_free_field(x0);
  printf("result: %d\n", passed);
  return passed;
}

static void test_cpu_side()
{
  int cords[2] = {1, 2};
  assert(get_index(cords, 5, 2, 25) == 11);
  int c2[2] = {0, 2};
  assert(neighbour_index(c2, 1, -1, 5, 2, 25) == 5);
  double x[3] = {1, 2, 3}, y[3] = {4, 5, 6};
  assert(inner_product(x, y, 3) == 32);
}

// neighbour_index_gpu is __host__ __device__ in the reference (conjugate-gradient_gpu.cuh:16-18): a caller's
// kernel calls the device side, the host calls the library's export, and both must agree everywhere
__global__ void neighbours_on_device(unsigned int *out, unsigned int L, unsigned int d, unsigned int N, int index_mode)
{
  const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N)
    return;
  for (unsigned int k = 0; k < d; k++)
    for (int a = -1; a <= 1; a++)
      out[(i * d + k) * 3 + (a + 1)] = neighbour_index_gpu(i, (int)k, a, L, d, N, index_mode);
}

static void test_neighbour_index_device(unsigned int L, unsigned int d, int index_mode)
{
  unsigned int N = 1;
  for (unsigned int k = 0; k < d; k++)
    N *= L;
  unsigned int *out;
  OK(hipMallocManaged((void **)&out, (size_t)N * d * 3 * sizeof(unsigned int)));
  neighbours_on_device<<<(N + 255) / 256, 256>>>(out, L, d, N, index_mode);
  OK(hipDeviceSynchronize());
  for (unsigned int i = 0; i < N; i++)
    for (unsigned int k = 0; k < d; k++)
      for (int a = -1; a <= 1; a++)
        assert(out[(i * d + k) * 3 + (a + 1)] == neighbour_index_gpu(i, (int)k, a, L, d, N, index_mode));
  OK(hipFree(out));
}

// the reference's unused device helpers (conjugate-gradient_gpu.cuh:12-14) keep their meaning
__global__ void cords_on_device(int *out)
{
  int c[3];
  index_to_cords_cu(c, 4 + 5 * (2 + 5 * 3), 5, 3); // coordinates (4, 2, 3), axis 0 fastest
  out[0] = c[0];
  out[1] = c[1];
  out[2] = c[2];
  int rev[3] = {3, 2, 4};
  out[3] = get_index_gpu(rev, 5, 3, 125); // cords[0] is the most significant digit there
}

static void test_dead_device_helpers()
{
  int *out;
  OK(hipMallocManaged((void **)&out, 4 * sizeof(int)));
  cords_on_device<<<1, 1>>>(out);
  OK(hipDeviceSynchronize());
  assert(out[0] == 4 && out[1] == 2 && out[2] == 3);
  assert(out[3] == 4 + 5 * (2 + 5 * 3));
  OK(hipFree(out));
}

int main()
{
  test_cpu_side();
  test_dead_device_helpers();
  test_neighbour_index_device(5, 3, 0);
  test_neighbour_index_device(10, 2, 1);
  test_neighbour_index_device(8, 3, 2);
  test_neighbour_index_device(4, 4, 2);
  assert(neighbour_index_gpu(2, 1, 1, 3, 2, 9, 0) == 5);
  assert(neighbour_index_gpu(2, 0, 1, 3, 2, 9, 0) == 9);
  assert(neighbour_index_gpu(3, 0, -1, 3, 2, 9, 0) == 9);
  test_inner_product_gpu();
  test_reduce_kernel();
  test_laplace_1d(true);
  test_laplace_1d(false);
  test_muladd();
  test_laplace_large();
  assert(test_cg_gpu(10, 2, 100));
  printf("tests passed!\n");
  return 0;
}
