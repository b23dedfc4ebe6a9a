// conjugate_gradient_cpu.cpp -- the library's host-side double-precision solver: the CPU
// half of the reference API (conjugate-gradient_cpu.cu).  Written for speed with integer
// index arithmetic (the reference calls pow() per coordinate, cpu.cu:38,78) while keeping
// every floating-point operation and its order, so results are bit-identical.
#include "../../include/conjugate_gradient_cpu.h"

#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

static int ipow_int(int L, int d)
{
  // the reference evaluates int N = pow(L, d) (cpu.cu:218); exact for every grid that fits an int
  return (int)pow((double)L, (double)d);
}

int get_index(int *cords, int L, int d, int N)
{
  for (int c = 0; c < d; c++)
  {
    assert(cords[c] < L + 1 && cords[c] > -2);
    if (cords[c] == -1 || cords[c] == L)
      return N;
  }
  long long ind = 0, s = 1;
  for (int i = 0; i < d; i++)
  {
    ind += s * cords[i];
    s *= L;
  }
  return (int)ind;
}

int neighbour_index(int *cords, int direction, int amount, int L, int d, int N)
{
  const int keep = cords[direction];
  cords[direction] = keep + amount;
  const int r = get_index(cords, L, d, N);
  cords[direction] = keep;
  return r;
}

int *index_to_cords(int *cords, int index, int L, int d)
{
  for (int i = 0; i < d; i++)
  {
    cords[i] = index % L;
    index /= L;
  }
  return cords;
}

// ddf[i] = sum_k ((-u[i+L^k]) + 2 u[i]) - u[i-L^k], out-of-grid neighbours read slot N.
// Coordinates are carried along instead of being re-derived per point.
double *minus_laplace(double *ddf, double *u, int d, int L, int N)
{
  std::vector<int> c(d, 0);
  std::vector<long long> stride(d, 1);
  for (int k = 1; k < d; k++)
    stride[k] = stride[k - 1] * L;
  for (int ind = 0; ind < N; ind++)
  {
    double laplace_value = 0;
    const double two_u = 2 * u[ind];
    for (int k = 0; k < d; k++)
    {
      const double up = (c[k] + 1 == L) ? u[N] : u[ind + stride[k]];
      const double um = (c[k] == 0) ? u[N] : u[ind - stride[k]];
      laplace_value += -up + two_u - um;
    }
    ddf[ind] = laplace_value;
    for (int k = 0; k < d; k++)
    {
      if (++c[k] < L)
        break;
      c[k] = 0;
    }
  }
  return ddf;
}

double norm(const double *x, int n)
{
  double s = 0;
  for (int i = 0; i < n; i++)
    s += x[i] * x[i];
  return sqrt(s);
}

double norm(double *x, int n) { return norm((const double *)x, n); }

double inner_product(const double *x, const double *y, int n)
{
  double s = 0;
  for (int i = 0; i < n; i++)
    s += x[i] * y[i];
  return s;
}

double *allocate_field(int N)
{
  double *r = (double *)calloc((size_t)N + 1, sizeof(double));
  if (r == NULL)
  {
    printf("Memory allocation failed");
    exit(1);
  }
  return r;
}

// cpu.cu:216-261: A x is formed in place (x must be 0 on entry to be meaningful),
// tol = 1e-6 |b|, the loop tests the residual of the previous iteration, no cap.
double conjugate_gradient(const double *b, double *x, int L, int d)
{
  const int N = ipow_int(L, d);
  double *r = allocate_field(N);
  // the reference forms A x in place, reading entries it has already overwritten
  // (cpu.cu:220); the same aliased call keeps the result identical for any x
  minus_laplace(x, x, d, L, N);
  for (int i = 0; i < N; i++)
    r[i] = b[i] - x[i];
  const double tol = 1e-6 * norm(b, N);
  double *p = allocate_field(N);
  memcpy(p, r, (size_t)N * sizeof(double));
  double *Ap = allocate_field(N);
  double rr = inner_product(r, r, N);
  double residue = norm(r, N);
  while (residue > tol)
  {
    minus_laplace(Ap, p, d, L, N);
    const double alpha = rr / inner_product(p, Ap, N);
    for (int i = 0; i < N; i++)
    {
      x[i] = x[i] + alpha * p[i];
      r[i] = r[i] - alpha * Ap[i];
    }
    const double rr_new = inner_product(r, r, N);
    const double beta = rr_new / rr;
    for (int i = 0; i < N; i++)
      p[i] = r[i] + beta * p[i];
    residue = sqrt(rr);
    rr = rr_new;
  }
  free(Ap);
  free(r);
  free(p);
  return residue;
}

// cpu.cu:274-324: CG from x = 0 to |r| <= errtol |b| (test on the fresh residual)
double preconditioner(double *b, double *x, int L, int d, double errtol)
{
  const int N = ipow_int(L, d);
  double *r = allocate_field(N);
  memcpy(r, b, (size_t)N * sizeof(double));
  const double tol = errtol * norm(b, N);
  double *p = allocate_field(N);
  memcpy(p, r, (size_t)N * sizeof(double));
  double *Ap = allocate_field(N);
  for (int k = 0; k < N; k++)
    x[k] = 0;
  int steps = 0;
  double res = norm(r, N);
  while (res > tol)
  {
    steps++;
    minus_laplace(Ap, p, d, L, N);
    const double rr = inner_product(r, r, N);
    const double alpha = rr / inner_product(p, Ap, N);
    for (int k = 0; k < N; k++)
    {
      x[k] = x[k] + alpha * p[k];
      r[k] = r[k] - alpha * Ap[k];
    }
    const double rr_new = inner_product(r, r, N);
    const double beta = rr_new / rr;
    for (int k = 0; k < N; k++)
      p[k] = r[k] + beta * p[k];
    res = norm(r, N);
  }
  printf("number of steps inner %d\n", steps);
  free(Ap);
  free(r);
  free(p);
  return res;
}

// cpu.cu:334-392: outer CG to 1e-8 |b| (cap 1000) preconditioned by the inner CG at 1e-3.
// One work vector serves as A x, A p and M^-1 r in turn, as in the reference (:350-352).
double *preconditioned_cg(double *b, double *x, int L, int d)
{
  const int N = ipow_int(L, d);
  assert(N > 0);
  double *r = allocate_field(N);
  double *w = allocate_field(N);
  minus_laplace(w, x, d, L, N);
  for (int i = 0; i < N; i++)
    r[i] = b[i] - w[i];
  const double tol = 1e-8 * norm(b, N);
  double *p = allocate_field(N);
  preconditioner(r, p, L, d, 1e-3);
  memcpy(w, p, (size_t)N * sizeof(double));
  double rMinvr = inner_product(r, w, N);
  int it = 0;
  const int maxitter = 1000;
  while (it < maxitter)
  {
    it++;
    minus_laplace(w, p, d, L, N);
    const double alpha = rMinvr / inner_product(p, w, N);
    for (int i = 0; i < N; i++)
    {
      x[i] = x[i] + alpha * p[i];
      r[i] = r[i] - alpha * w[i];
    }
    if (norm(r, N) < tol)
      break;
    preconditioner(r, w, L, d, 1e-3);
    const double rMinvr_new = inner_product(r, w, N);
    const double beta = rMinvr_new / rMinvr;
    for (int i = 0; i < N; i++)
      p[i] = w[i] + beta * p[i];
    rMinvr = rMinvr_new;
  }
  printf("number of steps (outer) %i\n", it);
  free(w);
  free(r);
  free(p);
  return x;
}

double *random_array(double *r, int L, int d, int N)
{
  (void)L;
  (void)d;
  for (int i = 0; i < N; i++)
    r[i] = (double)rand() / (double)RAND_MAX;
  return r;
}
