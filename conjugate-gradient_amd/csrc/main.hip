// main.hip -- the benchmark driver of the reference (main.cu:38-53) as a caller of the
// library: 1-D grid of 100000*312 points, random right-hand side, float CG (<= 200 its).
#include <stdio.h>
#include <stdlib.h>

#include "../../include/conjugate_gradient_gpu.hpp"

int main()
{
  int nblocks = 100000;
  int nthreads = 312;
  int N = nblocks * nthreads;
  int L = N;
  int d = 1;
  float *z = cuda_allocate_field(N);
  float *q = cuda_allocate_field(N);
  random_array(q, L, d, N);
  float res = conjugate_gradient_gpu(q, z, L, d);
  fprintf(stderr, "residue %g\n", res);
  cg_free_field(q);
  cg_free_field(z);
  return 0;
}
