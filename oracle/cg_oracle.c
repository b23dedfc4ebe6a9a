/*
 * cg_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, CPU-only restatement of the reference's Laplace / conjugate-gradient
 * algorithms.  It is the checker that the GPU library is compared against and the
 * "port" CPU baseline that bench.py times.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this file's shared object; the product
 * library (conjugate-gradient_amd/) never links, loads or calls it.
 *
 * Parity pin: every double-precision function here is checked bit-for-bit against
 * the reference's own CPU source compiled unchanged (oracle/_ref, built by
 * oracle/Makefile from /root/reference/src/conjugate-gradient_cpu.cu) and against
 * the golden vectors in tests/golden/ that were produced from that build
 * (oracle/make_golden.py).  Both builds use -ffp-contract=off.
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference/src/).  The loop structure, summation order and operation order
 * of the reference are kept literally, including its slow pow()-based index
 * arithmetic, so that results are bit-identical and the CPU timing is the
 * reference's timing.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAXD 32

/* ------------------------------------------------------------------------ */
/* Lexicographic index mapping, CPU flavour (conjugate-gradient_cpu.cu)      */
/* ------------------------------------------------------------------------ */

/* conjugate-gradient_cpu.cu:23-41.  Any coordinate equal to -1 or L selects the
 * boundary slot N; otherwise index = sum_i L^i * cords[i] (axis 0 fastest).  The
 * reference accumulates through a double (pow) into an int; kept. */
int orc_get_index(const int *cords, int L, int d, int N)
{
  for (int c = 0; c < d; c++)
  {
    int cord = cords[c];
    if (cord == -1 || cord == L)
      return N;
  }
  int ind = 0;
  for (int i = 0; i < d; i++)
  {
    ind += pow(L, i) * cords[i];
  }
  return ind;
}

/* conjugate-gradient_cpu.cu:54-61 */
int orc_neighbour_index(const int *cords, int direction, int amount, int L, int d, int N)
{
  int copy_cords[ORC_MAXD];
  memcpy(copy_cords, cords, d * sizeof(int));
  copy_cords[direction] += amount;
  return orc_get_index(copy_cords, L, d, N);
}

/* conjugate-gradient_cpu.cu:76-85 */
int *orc_index_to_cords(int *cords, int index, int L, int d)
{
  for (int i = 0; i < d; i++)
  {
    cords[i] = index % L;
    index /= L;
  }
  return cords;
}

/* ------------------------------------------------------------------------ */
/* Lexicographic index mapping, GPU flavour (conjugate-gradient_gpu.cu)      */
/* ------------------------------------------------------------------------ */

/* conjugate-gradient_gpu.cu:107-114: coordinate of `index` along axis d */
int orc_index_to_cords_gpu(int index, int L, int d)
{
  for (int i = 0; i < d; i++)
  {
    index /= L;
  }
  return index % L;
}

/* conjugate-gradient_gpu.cu:128-148.  Unsigned arithmetic as in the reference:
 * the coordinate is held in an unsigned, so -1 is the wrapped value 0xffffffff.
 * `d` and `index_mode` are accepted and ignored, as in the reference. */
unsigned int orc_neighbour_index_gpu(unsigned int ind, int direction, int amount,
                                     unsigned int L, unsigned int d, unsigned int N,
                                     int index_mode)
{
  (void)d;
  (void)index_mode;
  unsigned int cord = (unsigned int)orc_index_to_cords_gpu((int)ind, (int)L, direction);
  cord += (unsigned int)amount;
  if (cord == (unsigned int)-1 || cord == L)
    return N;
  int n = 1;
  for (int i = 0; i < direction; i++)
  {
    n *= (int)L;
  }
  ind += (unsigned int)(amount * n);
  return ind;
}

/* ------------------------------------------------------------------------ */
/* Double-precision vector operations (conjugate-gradient_cpu.cu)            */
/* ------------------------------------------------------------------------ */

/* conjugate-gradient_cpu.cu:97-113.  u has N+1 slots, u[N] is the boundary value.
 * Per axis the term is evaluated as ((-u_plus) + 2*u_0) - u_minus and the terms are
 * accumulated for axis 0..d-1 in that order. */
double *orc_minus_laplace(double *ddf, const double *u, int d, int L, int N)
{
  for (int ind = 0; ind < N; ind++)
  {
    int cords[ORC_MAXD];
    orc_index_to_cords(cords, ind, L, d);
    double laplace_value = 0;
    for (int i = 0; i < d; i++)
    {
      laplace_value += -u[orc_neighbour_index(cords, i, 1, L, d, N)] +
                       2 * u[orc_neighbour_index(cords, i, 0, L, d, N)] -
                       u[orc_neighbour_index(cords, i, -1, L, d, N)];
    }
    ddf[ind] = laplace_value;
  }
  return ddf;
}

/* conjugate-gradient_cpu.cu:122-140 */
double orc_norm(const double *x, int n)
{
  double norm = 0;
  for (int i = 0; i < n; i++)
  {
    norm += x[i] * x[i];
  }
  return sqrt(norm);
}

/* conjugate-gradient_cpu.cu:150-158 */
double orc_inner_product(const double *x, const double *y, int n)
{
  double product = 0;
  for (int i = 0; i < n; i++)
  {
    product += x[i] * y[i];
  }
  return product;
}

/* conjugate-gradient_cpu.cu:191-201 */
double *orc_allocate_field(int N)
{
  double *r = (double *)calloc((size_t)N + 1, sizeof(double));
  if (r == NULL)
  {
    printf("Memory allocation failed");
    exit(1);
  }
  r[N] = 0;
  return r;
}

/* ------------------------------------------------------------------------ */
/* Solvers (conjugate-gradient_cpu.cu)                                       */
/* ------------------------------------------------------------------------ */

/* conjugate-gradient_cpu.cu:216-261.  Same loop: in-place A*x, r = b - x,
 * tol = 1e-6*|b|, lagged test residue = sqrt(rr_old), no iteration cap.
 * Additions that do not touch the arithmetic: max_iters (< 0 = no cap, the
 * reference behaviour), the iteration count and the history of the lagged
 * residue (one entry per executed iteration, at most hist_len). */
double orc_conjugate_gradient_ex(const double *b, double *x, int L, int d, int max_iters,
                                 int *iters_out, double *hist, int hist_len)
{
  int N = pow(L, d);
  double *r = orc_allocate_field(N);
  orc_minus_laplace(x, x, d, L, N);
  for (int i = 0; i < N; i++)
  {
    r[i] = b[i] - x[i];
  }
  double tol = 1e-6 * orc_norm(b, N);
  double *p = orc_allocate_field(N);
  for (int i = 0; i < N; i++)
  {
    p[i] = r[i];
  }
  double *Ap = orc_allocate_field(N);
  int n = 0;
  double rr = orc_inner_product(r, r, N);
  double residue = orc_norm(r, N);
  while (residue > tol && (max_iters < 0 || n < max_iters))
  {
    orc_minus_laplace(Ap, p, d, L, N);
    double alpha = rr / orc_inner_product(p, Ap, N);
    for (int i = 0; i < N; i++)
    {
      x[i] = x[i] + alpha * p[i];
      r[i] = r[i] - alpha * Ap[i];
    }
    double rr_new = orc_inner_product(r, r, N);
    double beta = rr_new / rr;
    for (int i = 0; i < N; i++)
    {
      p[i] = r[i] + beta * p[i];
    }
    residue = sqrt(rr);
    rr = rr_new;
    if (hist && n < hist_len)
      hist[n] = residue;
    n++;
  }
  free(Ap);
  free(r);
  free(p);
  if (iters_out)
    *iters_out = n;
  return residue;
}

double orc_conjugate_gradient(const double *b, double *x, int L, int d)
{
  return orc_conjugate_gradient_ex(b, x, L, d, -1, NULL, NULL, 0);
}

/* conjugate-gradient_cpu.cu:274-324.  Plain CG from x = 0 with a NON-lagged stop
 * (res = |r| > errtol*|b|) and no cap; rr is recomputed at the top of every
 * iteration.  `verbose` reproduces the reference's printf. */
double orc_preconditioner_ex(const double *b, double *x, int L, int d, double errtol,
                             int *iters_out, int verbose)
{
  int N = pow(L, d);
  double *r = orc_allocate_field(N);
  for (int i = 0; i < N; i++)
  {
    r[i] = b[i];
  }
  double tol = errtol * orc_norm(b, N);
  double *p = orc_allocate_field(N);
  for (int i = 0; i < N; i++)
  {
    p[i] = r[i];
  }
  double *Ap = orc_allocate_field(N);
  int i = 0;
  for (int k = 0; k < N; k++)
  {
    x[k] = 0;
  }
  double res = orc_norm(r, N);
  double rr = NAN;
  while (res > tol)
  {
    i++;
    orc_minus_laplace(Ap, p, d, L, N);
    rr = orc_inner_product(r, r, N);
    double alpha = rr / orc_inner_product(p, Ap, N);
    for (int k = 0; k < N; k++)
    {
      x[k] = x[k] + alpha * p[k];
      r[k] = r[k] - alpha * Ap[k];
    }
    double rr_new = orc_inner_product(r, r, N);
    double beta = rr_new / rr;
    rr = rr_new;
    for (int k = 0; k < N; k++)
    {
      p[k] = r[k] + beta * p[k];
    }
    res = orc_norm(r, N);
  }
  if (verbose)
    printf("number of steps inner %d\n", i);
  free(Ap);
  free(r);
  free(p);
  if (iters_out)
    *iters_out = i;
  return res;
}

/* conjugate-gradient_cpu.cu:334-392.  Outer CG (tol 1e-8*|b|, cap 1000) with the
 * inner CG above (errtol 1e-3) as inexact preconditioner.  The reference aliases
 * Ax / Ap / Minv_r / Minv_r_new to one buffer (:350-352); kept, because p is
 * updated from that buffer after the preconditioner has overwritten it.
 * inner_iters receives the step count of each preconditioner call (the initial
 * one first), at most inner_len entries. */
double *orc_preconditioned_cg_ex(const double *b, double *x, int L, int d, int *outer_out,
                                 int *inner_iters, int inner_len, int verbose)
{
  int N = pow(L, d);
  double *r = orc_allocate_field(N);
  double *Ax = orc_allocate_field(N);
  int ncalls = 0, it_in = 0;
  orc_minus_laplace(Ax, x, d, L, N);
  for (int i = 0; i < N; i++)
  {
    r[i] = b[i] - Ax[i];
  }
  double tol = 1e-8 * orc_norm(b, N);
  double *p = orc_allocate_field(N);
  orc_preconditioner_ex(r, p, L, d, 1e-3, &it_in, verbose);
  if (inner_iters && ncalls < inner_len)
    inner_iters[ncalls] = it_in;
  ncalls++;
  double *Minv_r_new = Ax;
  double *Minv_r = Ax;
  double *Ap = Ax;
  double rMinvr;
  for (int i = 0; i < N; i++)
  {
    Minv_r[i] = p[i];
  }
  rMinvr = orc_inner_product(r, Minv_r, N);
  double r_newMinvr_new;
  int i = 0;
  int maxitter = 1000;
  while (i < maxitter)
  {
    i++;
    orc_minus_laplace(Ap, p, d, L, N);
    double alpha = rMinvr / orc_inner_product(p, Ap, N);
    for (int k = 0; k < N; k++)
    {
      x[k] = x[k] + alpha * p[k];
      r[k] = r[k] - alpha * Ap[k];
    }
    if (orc_norm(r, N) < tol)
      break;
    orc_preconditioner_ex(r, Minv_r_new, L, d, 1e-3, &it_in, verbose);
    if (inner_iters && ncalls < inner_len)
      inner_iters[ncalls] = it_in;
    ncalls++;
    r_newMinvr_new = orc_inner_product(r, Minv_r_new, N);
    double beta = r_newMinvr_new / rMinvr;
    for (int k = 0; k < N; k++)
    {
      p[k] = Minv_r_new[k] + beta * p[k];
    }
    rMinvr = r_newMinvr_new;
  }
  if (verbose)
    printf("number of steps (outer) %i\n", i);
  free(Ax);
  free(r);
  free(p);
  if (outer_out)
    *outer_out = i;
  return x;
}

/* ------------------------------------------------------------------------ */
/* Single-precision restatement of the reference GPU path                    */
/* ------------------------------------------------------------------------ */

/* conjugate-gradient_gpu.cu:164-178 evaluated on the host in fp32 with the
 * reference's operation order.  2*u is exact, so contraction of the per-axis
 * expression cannot change a bit; the result is the unique fp32 value of
 * ((-u_plus) + 2*u_0) - u_minus accumulated over axes 0..d-1.  Out-of-domain
 * neighbours are the boundary slot, whose value is 0 (gpu.cu:37-39), so they are
 * taken as literal 0 here and u needs only N entries. */
void orc_laplace_f32(float *ddf, const float *u, unsigned int d, unsigned int L, unsigned int N)
{
  for (unsigned int ind = 0; ind < N; ind++)
  {
    float laplace_value = 0;
    for (unsigned int i = 0; i < d; i++)
    {
      unsigned int ip = orc_neighbour_index_gpu(ind, (int)i, 1, L, d, N, 0);
      unsigned int i0 = orc_neighbour_index_gpu(ind, (int)i, 0, L, d, N, 0);
      unsigned int im = orc_neighbour_index_gpu(ind, (int)i, -1, L, d, N, 0);
      float up = ip == N ? 0.0f : u[ip];
      float u0 = i0 == N ? 0.0f : u[i0];
      float um = im == N ? 0.0f : u[im];
      float t = -up + 2 * u0;
      t = t - um;
      laplace_value = laplace_value + t;
    }
    ddf[ind] = laplace_value;
  }
}

/* Dot product of fp32 vectors.  The reference (gpu.cu:191-255) sums fp32 partials
 * with atomicAdd in a non-deterministic order, so it has no single fp32 answer;
 * the oracle accumulates the fp32 products in double and rounds once, which is
 * within half an ulp of every possible reference ordering's exact sum. */
float orc_inner_product_f32(const float *v, const float *w, unsigned int N)
{
  double acc = 0;
  for (unsigned int i = 0; i < N; i++)
  {
    acc += (double)(v[i] * w[i]);
  }
  return (float)acc;
}

/* conjugate-gradient_gpu.cu:327-376 on the host in fp32: r = b (x is accumulated
 * into, gpu.cu:336,356), reltol = 1e-6*|b| evaluated in double and rounded to
 * float (:333), cap `maxit` (200 in the reference, :348), lagged test
 * residue = sqrt(rr_old) (:366).  The axpy updates use one fused multiply-add per
 * element, which is what nvcc's default -fmad=true emits for A[i] + b*B[i]
 * (:298,:307).  lagged = 0 switches to the non-lagged stop of the CPU
 * preconditioner (cpu.cu:299,317) with relative tolerance `rel`.
 * hist receives the value the reference prints each iteration (:369). */
float orc_cg_f32(const float *b, float *x, int L, int d, int maxit, double rel, int lagged,
                 int *iters_out, float *hist, int hist_len)
{
  unsigned int N = (unsigned int)pow(L, d);
  float *r = (float *)calloc((size_t)N + 1, sizeof(float));
  float *p = (float *)calloc((size_t)N + 1, sizeof(float));
  float *Ap = (float *)calloc((size_t)N + 1, sizeof(float));
  float reltol = (float)(rel * (double)(float)sqrt((double)orc_inner_product_f32(b, b, N)));
  for (unsigned int i = 0; i < N; i++)
  {
    r[i] = fmaf(1.0f, b[i], r[i]);
    p[i] = fmaf(1.0f, r[i], p[i]);
  }
  float rr = orc_inner_product_f32(r, r, N);
  float rr_new = rr;
  float residue = (float)sqrt((double)rr);
  int i = 0;
  while (residue > reltol && (maxit < 0 || i < maxit))
  {
    orc_laplace_f32(Ap, p, (unsigned int)d, (unsigned int)L, N);
    float alpha = rr / orc_inner_product_f32(p, Ap, N);
    for (unsigned int k = 0; k < N; k++)
    {
      x[k] = fmaf(alpha, p[k], x[k]);
      r[k] = fmaf(-alpha, Ap[k], r[k]);
    }
    rr_new = orc_inner_product_f32(r, r, N);
    float beta = rr_new / rr;
    for (unsigned int k = 0; k < N; k++)
    {
      p[k] = fmaf(beta, p[k], r[k]);
    }
    residue = lagged ? (float)sqrt((double)rr) : (float)sqrt((double)rr_new);
    rr = rr_new;
    if (hist && i < hist_len)
      hist[i] = residue;
    i++;
  }
  free(r);
  free(p);
  free(Ap);
  if (iters_out)
    *iters_out = i;
  return residue;
}

/* ------------------------------------------------------------------------ */
/* Bit-interleaved (Morton) index mapping (interleave.cuh, custom_pdep.cpp)  */
/* ------------------------------------------------------------------------ */

/* custom_pdep.cpp:4-8 (_pdep_u32): scatter the low bits of x to the set bits of
 * mask, lowest first.  Portable restatement of the BMI2 instruction. */
unsigned int orc_pdep(unsigned int x, unsigned int mask)
{
  unsigned int res = 0;
  for (unsigned int bit = 1; mask; bit <<= 1)
  {
    unsigned int lowest = mask & (0u - mask);
    if (x & bit)
      res |= lowest;
    mask &= mask - 1;
  }
  return res;
}

/* custom_pdep.cpp:9-13 (_pext_u32): gather the bits of x selected by mask into
 * the low bits of the result, lowest first. */
unsigned int orc_pext(unsigned int x, unsigned int mask)
{
  unsigned int res = 0;
  for (unsigned int bit = 1; mask; bit <<= 1)
  {
    unsigned int lowest = mask & (0u - mask);
    if (x & lowest)
      res |= bit;
    mask &= mask - 1;
  }
  return res;
}

/* interleave.cuh:14-23: bits 0, d, 2d, ... below 32 */
unsigned int orc_morton_base_mask(int d)
{
  unsigned int m = 0;
  for (int i = 0; i < 32; i += d)
  {
    m = m << d;
    m++;
  }
  return m;
}

/* interleave.cuh:31-36: mask of the bits that hold coordinate `dim` */
unsigned int orc_morton_mask(int d, int dim) { return orc_morton_base_mask(d) << dim; }

/* interleave.cuh:43-54 (NH<-1,N>::calc): coordinate `dim` minus one by a masked
 * subtract; a borrow out of coordinate 0 wraps that coordinate to all ones. */
unsigned int orc_morton_dec(unsigned int arg, int d, int dim)
{
  unsigned int mask = orc_morton_mask(d, dim);
  unsigned int iy = arg & ~mask;
  unsigned int ix = arg & mask;
  unsigned int adj = ix - 1;
  return iy | (adj & mask);
}

/* interleave.cuh:56-67 (NH<1,N>::calc): coordinate `dim` plus one; the carry
 * ripples through the foreign bits because they are filled with ones. */
unsigned int orc_morton_inc(unsigned int arg, int d, int dim)
{
  unsigned int mask = orc_morton_mask(d, dim);
  unsigned int iy = arg & ~mask;
  unsigned int ix = (arg & mask) | ~mask;
  unsigned int adj = ix + 1;
  return iy | (adj & mask);
}

/* interleave.cuh:110-118 */
unsigned int orc_morton_from_coords(const int *coords, int d)
{
  unsigned int res = 0;
  for (int i = 0; i < d; i++)
    res |= orc_pdep((unsigned int)coords[i], orc_morton_mask(d, i));
  return res;
}

/* interleave.cuh:83-91 */
void orc_morton_to_coords(unsigned int idx, int d, int *coords)
{
  for (int j = 0; j < d; j++)
    coords[j] = (int)orc_pext(idx, orc_morton_mask(d, j));
}

/* interleave.cuh:99 */
int orc_morton_is_outside(unsigned int idx, unsigned int size) { return idx >= size; }

/* The stencil the reference sketches in interleave.cuh:160-175, completed with the
 * operation order of conjugate-gradient_gpu.cu:173: u and ddf are stored in Morton
 * order, a neighbour whose index is outside [0,N) is the zero boundary.  Needs
 * L = 2^m, for which idx < L^d <=> every coordinate < L. */
void orc_laplace_morton_f32(float *ddf, const float *u, unsigned int d, unsigned int N)
{
  for (unsigned int ind = 0; ind < N; ind++)
  {
    float laplace_value = 0;
    for (unsigned int i = 0; i < d; i++)
    {
      unsigned int ip = orc_morton_inc(ind, (int)d, (int)i);
      unsigned int im = orc_morton_dec(ind, (int)d, (int)i);
      float up = ip >= N ? 0.0f : u[ip];
      float um = im >= N ? 0.0f : u[im];
      float t = -up + 2 * u[ind];
      t = t - um;
      laplace_value = laplace_value + t;
    }
    ddf[ind] = laplace_value;
  }
}

/* permutation helper for tests: lexicographic index -> Morton index */
unsigned int orc_lex_to_morton(unsigned int lex, unsigned int L, int d)
{
  int c[ORC_MAXD];
  orc_index_to_cords(c, (int)lex, (int)L, d);
  return orc_morton_from_coords(c, d);
}
