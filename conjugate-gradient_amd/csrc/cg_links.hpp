// cg_links.hpp -- slab neighbours talking directly over xGMI: no collective library on the path.
//
// Every rank owns a "box" in its own HBM that the other ranks map (HIP IPC between processes,
// plain pointers inside one process) and write into with ordinary stores:
//
//   * slots: one (value, stamp) pair per kind of dot product, stamp parity and source rank.
//     After kernel A or B a rank sums its own per-workgroup partials in the usual fixed order
//     and stores that one double, tagged with the stamp, into the slot [kind][parity][own rank]
//     of EVERY rank's box; it then waits until all `world` stamps of its own box carry the
//     current sequence number and adds the values in rank order.  All ranks add the same
//     doubles in the same order, so alpha, beta and the stop test agree bit for bit.
//   * staging planes: with the r.r exchange a rank also stores its first / last owned
//     hyperplane of r into the lower / upper neighbour's staging area; the neighbour copies
//     them into its ghost planes once the stamps are in (the solver's vectors are ordinary
//     cached device memory, which a remote writer must not touch while kernels run).
//
// Stamps are a per-kind sequence number that every rank advances identically (all control
// decisions derive from identical sums).  Slots and staging are double-buffered by stamp
// parity: a rank can only start exchange q+2 after it finished q+1, which needed every peer's
// stamp q+1, which they store only after finishing exchange q -- so nobody still reads what
// q+2 overwrites.
//
// The exchange is its own small launch between the two big kernels, so an iteration is four
// launches and no host work: it is captured in the same hipGraph as on one GPU.  Only one
// workgroup ever spins (the others of a multi-workgroup plane push wait for its go flag); all
// waits are bounded by a wall-clock budget, after which the rank poisons its sums with NaN
// (every rank then stops: NaN fails the "residue > tolerance" test) and raises an error flag.
#pragma once
#include "cg_device.hpp"

namespace cgamd
{

constexpr int CG_LINKS_MAX_RANKS = 64;

// A slot is two 8-byte words, each carrying half of the double next to the low 32 bits of the
// stamp.  An aligned 8-byte store arrives whole, so a reader that finds the expected stamp in
// both words has the value: the writer needs no fence between value and flag, and the reader
// needs one load round trip per poll.
struct LinkSlot
{
  unsigned long long w[2]; // w[i] = (stamp32 << 32) | (i ? hi32 : lo32) of the double's bits
};

// shared part of a box; the staging planes follow at byte offset CG_LINKS_HEADER:
// float staging[2 stamp parities][2 sides][plane_pad]
struct LinkBox
{
  LinkSlot slot[2][2][CG_LINKS_MAX_RANKS]; // [kind: 0 p.Ap, 1 r.r][stamp parity][source rank]
};
constexpr size_t CG_LINKS_HEADER = sizeof(LinkBox);
static_assert(CG_LINKS_HEADER == 4096, "box header is one page");

enum
{
  LINK_FROM_LO = 0, // plane received from the rank below (becomes the lo ghost plane)
  LINK_FROM_HI = 1
};

// private to the rank (ordinary device memory)
struct LinkCtl
{
  unsigned long long seq[2]; // exchanges completed so far, per kind
  unsigned long long go[2];  // = seq once the stamps of that exchange are in (multi-workgroup pushes)
  unsigned int ticket[2];
  int error;                 // sticky: a wait ran out of time
  int pad;
  double sum[2];             // what kernels A / B read as their one addend
};

struct LinkArgs
{
  LinkBox *box[CG_LINKS_MAX_RANKS];
  LinkCtl *ctl;
  const CgState *st;
  const double *part; // this rank's per-workgroup partials of the kind
  int nparts;
  int rank, world;
  int kind;
  int force;          // run even if the state says the iteration is inactive (start of a solve)
  int pushed;         // kernel B has already stored the boundary planes into the neighbours' staging areas
  const float *r_first, *r_last; // owned boundary planes (kind 1 with neighbours), else null
  float *ghost_lo, *ghost_hi;
  unsigned long long plane, plane_pad; // floats
  long long budget;   // wall_clock64 ticks (100 MHz) a wait may take
};

// at least 128 floats, so that the staging area always holds 64 eight-byte probe words (links_poke)
__host__ __device__ inline size_t links_plane_pad(size_t plane) { return (plane < 128 ? 128 : (plane + 63) / 64 * 64); }
__host__ __device__ inline size_t links_box_bytes(size_t plane)
{
  return CG_LINKS_HEADER + 4 * links_plane_pad(plane) * sizeof(float);
}
__host__ __device__ __forceinline__ float *links_staging(LinkBox *box, int parity, int side, unsigned long long plane_pad)
{
  return reinterpret_cast<float *>(reinterpret_cast<char *>(box) + CG_LINKS_HEADER) +
         ((size_t)parity * 2 + side) * plane_pad;
}

__global__ void k_exchange(const LinkArgs L);

// Boxes: device memory that bypasses this GPU's caches, zeroed.  All return a hipError_t
// value (0 = success) instead of terminating, so that a caller can fall back to collectives.
int links_alloc(size_t bytes, void **box);
int links_free(void *box);
int links_peer_ok(int peer_device);                          // 1 if this device can write that one's memory
int links_export(void *box, unsigned char handle[64]);     // for another process on this node
int links_open(const unsigned char handle[64], void **box);
int links_close(void *box);
int links_poke(void *box, int slot, unsigned long long value);  // host write / read of a staging word
int links_peek(void *box, int slot, unsigned long long *value);

} // namespace cgamd
