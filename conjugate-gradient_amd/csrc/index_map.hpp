// index_map.hpp -- index mappings of the Laplace grid, usable on host and device.
//
// Two layouts:
//   * lexicographic: index = sum_k L^k c_k, axis 0 contiguous.  Neighbour rule of the
//     reference GPU code (conjugate-gradient_gpu.cu:107-148): a step that leaves
//     [0,L) along the chosen axis yields the boundary slot N, otherwise ind +- L^k.
//   * bit-interleaved (Morton): coordinate k owns bits k, k+d, k+2d, ... of a 32-bit
//     index (interleave.cuh:14-36); +-1 along an axis is a masked add/subtract
//     (interleave.cuh:43-67); encode/decode are pdep/pext over the axis mask
//     (interleave.cuh:83-91,110-118; custom_pdep.cpp:4-13, re-stated without BMI2,
//     which does not exist on the GPU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cgamd
{

#define CG_HD __host__ __device__ __forceinline__

// ---- lexicographic ---------------------------------------------------------------

// coordinate of `index` along `axis` in an L^d grid (conjugate-gradient_gpu.cu:107-114)
CG_HD int lex_coord(int index, int L, int axis)
{
  for (int i = 0; i < axis; i++)
    index /= L;
  return index % L;
}

// conjugate-gradient_gpu.cu:128-148.  The coordinate lives in an unsigned, so "-1" is
// the wrapped value; `d` and `index_mode` do not enter (0 and 1 are both lexicographic).
CG_HD unsigned int lex_neighbour(unsigned int ind, int direction, int amount, unsigned int L,
                                 unsigned int N)
{
  unsigned int cord = (unsigned int)lex_coord((int)ind, (int)L, direction);
  cord += (unsigned int)amount;
  if (cord == 0xffffffffu || cord == L)
    return N;
  int n = 1;
  for (int i = 0; i < direction; i++)
    n *= (int)L;
  return ind + (unsigned int)(amount * n);
}

// ---- Morton ----------------------------------------------------------------------

// bits 0, d, 2d, ... below 32 (interleave.cuh:14-23)
CG_HD unsigned int morton_base_mask(int d)
{
  unsigned int m = 0;
  for (int i = 0; i < 32; i += d)
    m = (m << d) + 1u;
  return m;
}

// bits of coordinate `dim` (interleave.cuh:31-36)
CG_HD unsigned int morton_mask(int d, int dim) { return morton_base_mask(d) << dim; }

// coordinate - 1 under `mask` (interleave.cuh:43-54)
CG_HD unsigned int morton_dec(unsigned int idx, unsigned int mask)
{
  return (idx & ~mask) | (((idx & mask) - 1u) & mask);
}

// coordinate + 1 under `mask`: foreign bits are set so the carry ripples (interleave.cuh:56-67)
CG_HD unsigned int morton_inc(unsigned int idx, unsigned int mask)
{
  return (idx & ~mask) | ((((idx & mask) | ~mask) + 1u) & mask);
}

// The same two steps for a grid of N = 2^(m d) points, with 0xffffffff for a step that leaves it.  The
// reference detects that by "index >= N" (interleave.cuh:99), which needs a spare mask bit above the m
// coordinate bits for the carry or borrow to land in; for d (m + 1) > 32 some masks have none and the step
// wraps around to a valid-looking index (L = 2, d >= 17).  Looking at the coordinate's own bits is
// equivalent wherever the reference's test works, and right everywhere.
CG_HD unsigned int morton_up(unsigned int idx, unsigned int mask, unsigned long long N)
{
  const unsigned int cm = mask & (unsigned int)(N - 1ull);
  return (idx & cm) == cm ? 0xffffffffu : morton_inc(idx, mask);
}
CG_HD unsigned int morton_dn(unsigned int idx, unsigned int mask, unsigned long long N)
{
  const unsigned int cm = mask & (unsigned int)(N - 1ull);
  return (idx & cm) == 0u ? 0xffffffffu : morton_dec(idx, mask);
}

// _pdep_u32 without BMI2 (custom_pdep.cpp:4-8)
CG_HD unsigned int pdep32(unsigned int x, unsigned int mask)
{
  unsigned int res = 0;
  for (unsigned int bit = 1; mask; bit <<= 1)
  {
    unsigned int lowest = mask & (0u - mask);
    if (x & bit)
      res |= lowest;
    mask &= mask - 1u;
  }
  return res;
}

// _pext_u32 without BMI2 (custom_pdep.cpp:9-13)
CG_HD unsigned int pext32(unsigned int x, unsigned int mask)
{
  unsigned int res = 0;
  for (unsigned int bit = 1; mask; bit <<= 1)
  {
    unsigned int lowest = mask & (0u - mask);
    if (x & lowest)
      res |= bit;
    mask &= mask - 1u;
  }
  return res;
}

// interleave.cuh:110-118
CG_HD unsigned int morton_encode(const int *coords, int d)
{
  unsigned int res = 0;
  for (int i = 0; i < d; i++)
    res |= pdep32((unsigned int)coords[i], morton_mask(d, i));
  return res;
}

// interleave.cuh:83-91
CG_HD void morton_decode(unsigned int idx, int d, int *coords)
{
  for (int j = 0; j < d; j++)
    coords[j] = (int)pext32(idx, morton_mask(d, j));
}

} // namespace cgamd
