/* conjugate_gradient_cpu.h -- host (double precision) API of the library, C++ linkage.
 *
 * Same names, argument meaning and results as the reference's CPU solver
 * (/root/reference/src/conjugate-gradient_cpu.h:5-16, conjugate-gradient_cpu.cu:23-402):
 * fields carry N+1 doubles, slot N is the zero Dirichlet boundary value.  The results
 * are bit-identical to the reference's (same summation and operation order); the index
 * arithmetic is integer-only instead of pow()-based.
 */
#pragma once

double *allocate_field(int N);                                            /* cpu.cu:191-201 */
double *minus_laplace(double *ddf, double *u, int d, int L, int N);       /* cpu.cu:97-113  */
double conjugate_gradient(const double *b, double *x, int L, int d);      /* cpu.cu:216-261 */
double preconditioner(double *b, double *x, int L, int d, double errtol); /* cpu.cu:274-324 */
double inner_product(const double *x, const double *y, int n);            /* cpu.cu:150-158 */
double *preconditioned_cg(double *b, double *x, int L, int d);            /* cpu.cu:334-392 */
int *index_to_cords(int *cords, int index, int L, int d);                 /* cpu.cu:76-85   */
double norm(double *x, int n);                                            /* cpu.cu:122-130 */
double norm(const double *x, int n);                                      /* cpu.cu:132-140 */
int get_index(int *cords, int L, int d, int N);                           /* cpu.cu:23-41   */
int neighbour_index(int *cords, int direction, int amount, int L, int d, int N); /* cpu.cu:54-61 */
double *random_array(double *r, int L, int d, int N);                     /* cpu.cu:394-402 */
