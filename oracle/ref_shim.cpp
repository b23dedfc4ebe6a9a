// ref_shim.cpp -- TEST INFRASTRUCTURE.  C-linkage doors into the reference's own CPU
// solver so that Python (ctypes) can call it.  This file contains no algorithm: it
// only declares the prototypes of /root/reference/src/conjugate-gradient_cpu.h:5-16
// by hand (that header carries CUDA qualifiers on line 9 and cannot be included by
// g++) and forwards to them.  It is compiled together with the reference source
// where it lies; see oracle/Makefile (target _ref).
#include <cstdlib>

double *allocate_field(int N);
double *minus_laplace(double *ddf, double *u, int d, int L, int N);
double conjugate_gradient(const double *b, double *x, int L, int d);
double preconditioner(double *b, double *x, int L, int d, double errtol);
double inner_product(const double *x, const double *y, int n);
double *preconditioned_cg(double *b, double *x, int L, int d);
int *index_to_cords(int *cords, int index, int L, int d);
double norm(double *x, int n);
int get_index(int *cords, int L, int d, int N);
int neighbour_index(int *cords, int direction, int amount, int L, int d, int N);
double *random_array(double *r, int L, int d, int N);

extern "C"
{
  double *ref_minus_laplace(double *ddf, double *u, int d, int L, int N) { return minus_laplace(ddf, u, d, L, N); }
  double ref_conjugate_gradient(const double *b, double *x, int L, int d) { return conjugate_gradient(b, x, L, d); }
  double ref_preconditioner(double *b, double *x, int L, int d, double errtol) { return preconditioner(b, x, L, d, errtol); }
  double ref_inner_product(const double *x, const double *y, int n) { return inner_product(x, y, n); }
  double *ref_preconditioned_cg(double *b, double *x, int L, int d) { return preconditioned_cg(b, x, L, d); }
  int *ref_index_to_cords(int *cords, int index, int L, int d) { return index_to_cords(cords, index, L, d); }
  double ref_norm(double *x, int n) { return norm(x, n); }
  int ref_get_index(int *cords, int L, int d, int N) { return get_index(cords, L, d, N); }
  int ref_neighbour_index(int *cords, int direction, int amount, int L, int d, int N) { return neighbour_index(cords, direction, amount, L, d, N); }
  // glibc rand()-based fill used by the reference's tests (cpu.cu:394-402)
  double *ref_random_array(double *r, int L, int d, int N) { return random_array(r, L, d, N); }
  void ref_srand(unsigned int seed) { srand(seed); }
}
