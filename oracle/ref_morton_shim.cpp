// ref_morton_shim.cpp -- TEST INFRASTRUCTURE.  C-linkage doors into the reference's
// bit-interleaved index type (/root/reference/src/interleave.cuh) for dimensions
// 2, 3 and 4, compiled on the host together with custom_pdep.cpp (BMI2).  The
// CUDA function qualifiers are blanked on the compiler command line
// (-D__host__= -D__device__=, see oracle/Makefile); no reference text is copied.
#include <cstdlib>
#include "interleave.cuh"

template <int N>
static unsigned int enc(const int *c)
{
  Coords<N> co;
  for (int i = 0; i < N; i++)
    co.coords[i] = c[i];
  return from_coords<N>(co).i;
}
template <int N>
static void dec(unsigned int idx, int *c)
{
  Index<N> ix{idx};
  Coords<N> co = ix.to_coords();
  for (int i = 0; i < N; i++)
    c[i] = co.coords[i];
}
template <int N>
static unsigned int nb(unsigned int idx, int dim, int dir)
{
  Index<N> ix{idx};
  return dir > 0 ? ix.template neighbour<1>(dim).i : ix.template neighbour<-1>(dim).i;
}

extern "C"
{
  unsigned int ref_morton_from_coords(const int *c, int d)
  {
    switch (d)
    {
    case 2: return enc<2>(c);
    case 3: return enc<3>(c);
    case 4: return enc<4>(c);
    }
    abort();
  }
  void ref_morton_to_coords(unsigned int idx, int d, int *c)
  {
    switch (d)
    {
    case 2: dec<2>(idx, c); return;
    case 3: dec<3>(idx, c); return;
    case 4: dec<4>(idx, c); return;
    }
    abort();
  }
  unsigned int ref_morton_neighbour(unsigned int idx, int d, int dim, int dir)
  {
    switch (d)
    {
    case 2: return nb<2>(idx, dim, dir);
    case 3: return nb<3>(idx, dim, dir);
    case 4: return nb<4>(idx, dim, dir);
    }
    abort();
  }
  unsigned int ref_morton_mask(int d, int dim)
  {
    switch (d)
    {
    case 2: return Detail::generate_mask<2>(dim);
    case 3: return Detail::generate_mask<3>(dim);
    case 4: return Detail::generate_mask<4>(dim);
    case 5: return Detail::generate_mask<5>(dim);
    case 6: return Detail::generate_mask<6>(dim);
    }
    abort();
  }
  unsigned int ref_pdep(unsigned int x, unsigned int m) { return pdep_custom(x, m); }
  unsigned int ref_pext(unsigned int x, unsigned int m) { return pext_custom(x, m); }
  int ref_interleave_selftest(void) { return tests_interleaved_index(); }
}
