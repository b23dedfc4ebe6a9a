/* cg_oracle_omp.c -- TEST / BENCHMARK INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The loops of the CPU solver (/root/reference/src/conjugate-gradient_cpu.cu:97-113,150-158,
 * 216-261, restated in cg_oracle.c) spread over all cores with OpenMP, for the benchmark's
 * "all cores" CPU figure.  The reference has no threading: this is NOT the reference and is never
 * reported as such.  The dot products are OpenMP reductions, so sums are re-associated and results
 * differ from the serial code in the last bits; the index arithmetic is the reference's (pow()-based
 * get_index and a coordinate copy per neighbour), because that is where the reference's time goes. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define OMP_MAXD 32

static int omp_get_index(const int *cords, int L, int d, int N)
{
  for (int c = 0; c < d; c++)
  {
    int cord = cords[c];
    if (cord == L || cord == -1)
      return N;
  }
  int ind = 0;
  for (int i = 0; i < d; i++)
    ind += (int)(pow(L, i) * cords[i]); /* conjugate-gradient_cpu.cu:38 */
  return ind;
}

static int omp_neighbour_index(const int *cords, int direction, int amount, int L, int d, int N)
{
  int copy[OMP_MAXD];
  memcpy(copy, cords, (size_t)d * sizeof(int)); /* conjugate-gradient_cpu.cu:57-58 */
  copy[direction] += amount;
  return omp_get_index(copy, L, d, N);
}

static void omp_minus_laplace(double *ddf, const double *u, int d, int L, int N)
{
#pragma omp parallel for schedule(static)
  for (int ind = 0; ind < N; ind++)
  {
    int cords[OMP_MAXD];
    int index = ind;
    for (int i = 0; i < d; i++)
    {
      cords[i] = index % L;
      index /= L;
    }
    double v = 0;
    for (int i = 0; i < d; i++)
      v += -u[omp_neighbour_index(cords, i, 1, L, d, N)] + 2 * u[omp_neighbour_index(cords, i, 0, L, d, N)] -
           u[omp_neighbour_index(cords, i, -1, L, d, N)];
    ddf[ind] = v;
  }
}

static double omp_dot(const double *x, const double *y, int n)
{
  double s = 0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int i = 0; i < n; i++)
    s += x[i] * y[i];
  return s;
}

/* x = 0 on entry (as the reference requires); at most max_iters iterations (< 0: until converged).
 * Returns the lagged residue; *iters_out = executed iterations; *threads_out = OpenMP threads used. */
double orc_cg_omp(const double *b, double *x, int L, int d, int max_iters, int *iters_out, int *threads_out)
{
  int N = (int)pow(L, d);
  double *r = (double *)calloc((size_t)N + 1, sizeof(double));
  double *p = (double *)calloc((size_t)N + 1, sizeof(double));
  double *Ap = (double *)calloc((size_t)N + 1, sizeof(double));
  if (threads_out)
    *threads_out = omp_get_max_threads();
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; i++)
  {
    r[i] = b[i];
    p[i] = b[i];
  }
  const double tol = 1e-6 * sqrt(omp_dot(b, b, N));
  double rr = omp_dot(r, r, N);
  double residue = sqrt(rr);
  int n = 0;
  while (residue > tol && (max_iters < 0 || n < max_iters))
  {
    omp_minus_laplace(Ap, p, d, L, N);
    const double alpha = rr / omp_dot(p, Ap, N);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++)
    {
      x[i] = x[i] + alpha * p[i];
      r[i] = r[i] - alpha * Ap[i];
    }
    const double rr_new = omp_dot(r, r, N);
    const double beta = rr_new / rr;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++)
      p[i] = r[i] + beta * p[i];
    residue = sqrt(rr);
    rr = rr_new;
    n++;
  }
  free(Ap);
  free(r);
  free(p);
  if (iters_out)
    *iters_out = n;
  return residue;
}
