// cg_links.hip -- the exchange kernel of the direct xGMI path and the box plumbing
// (see cg_links.hpp for the protocol).
#include "cg_links.hpp"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "hip_check.hpp"

namespace cgamd
{

// dst[0..n) = src[0..n) spread over the whole grid; 16-byte accesses when both sides allow it
__device__ __forceinline__ void grid_copy(float *__restrict__ dst, const float *__restrict__ src, unsigned long long n)
{
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if ((((uintptr_t)dst | (uintptr_t)src) % 16) == 0)
  {
    const unsigned long long n4 = n >> 2;
    for (unsigned long long i = t; i < n4; i += step)
      reinterpret_cast<float4 *>(dst)[i] = reinterpret_cast<const float4 *>(src)[i];
    for (unsigned long long i = (n4 << 2) + t; i < n; i += step)
      dst[i] = src[i];
  }
  else
  {
    for (unsigned long long i = t; i < n; i += step)
      dst[i] = src[i];
  }
}

__global__ void __launch_bounds__(256) k_exchange(const LinkArgs L)
{
  __shared__ double sh[8];
  __shared__ double shv[CG_LINKS_MAX_RANKS];
  __shared__ int sh_last, sh_bad;
  LinkCtl *ctl = L.ctl;
  const int kind = L.kind, tid = threadIdx.x;
  // independent loads first, so that their latencies overlap
  const int active = L.force | L.st->activeB;
  const unsigned long long seq = ctl->seq[kind] + 1;
  const bool poisoned = ctl->error != 0;
  const bool single = gridDim.x == 1;
  double local = 0;
  if (single)
    local = block_read_sum(L.part, L.nparts, sh);
  if (!active)
    return;
  const int par = (int)(seq & 1);
  const bool planes = L.r_first != nullptr;
  const bool has_lo = L.rank > 0, has_hi = L.rank + 1 < L.world;

  // ---- push this rank's boundary planes of r into the neighbours' staging areas ------------
  // (not after an iteration's kernel B, which has stored them there itself while it ran: the end of that
  // launch has made them visible before this one starts)
  if (planes && !L.pushed)
  {
    if (has_lo)
      grid_copy(links_staging(L.box[L.rank - 1], par, LINK_FROM_HI, L.plane_pad), L.r_first, L.plane);
    if (has_hi)
      grid_copy(links_staging(L.box[L.rank + 1], par, LINK_FROM_LO, L.plane_pad), L.r_last, L.plane);
    __threadfence_system(); // the planes are on their way before any stamp is
  }
  bool last = true;
  if (!single)
  {
    __syncthreads();
    if (tid == 0)
    {
      const unsigned t = __hip_atomic_fetch_add(&ctl->ticket[kind], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      sh_last = (t == gridDim.x - 1);
    }
    __syncthreads();
    last = sh_last != 0;
    if (last)
      local = block_read_sum(L.part, L.nparts, sh);
  }
  else if (planes)
    __syncthreads(); // every wave's plane stores are fenced before wave 0 stamps

  if (last)
  {
    // ---- the last workgroup to arrive publishes this rank's sum and collects everyone's -----
    if (tid == 0)
      sh_bad = 0;
    __syncthreads();
    if (tid < L.world)
    {
      const unsigned long long tag = (seq & 0xffffffffull) << 32;
      const unsigned long long bits = (unsigned long long)__double_as_longlong(poisoned ? (double)NAN : local);
      LinkSlot *s = &L.box[tid]->slot[kind][par][L.rank];
      __hip_atomic_store(&s->w[0], tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&s->w[1], tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      LinkSlot *m = &L.box[L.rank]->slot[kind][par][tid];
      const long long t0 = wall_clock64();
      bool ok = true;
      unsigned long long w0, w1;
      for (;;)
      {
        w0 = __hip_atomic_load(&m->w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        w1 = __hip_atomic_load(&m->w[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((w0 & ~0xffffffffull) == tag && (w1 & ~0xffffffffull) == tag)
          break;
        if (poisoned || wall_clock64() - t0 > L.budget)
        {
          ok = false;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      const unsigned long long got = (w0 & 0xffffffffull) | (w1 << 32);
      shv[tid] = ok ? __longlong_as_double((long long)got) : (double)NAN;
      if (!ok)
        sh_bad = 1;
    }
    __syncthreads();
    if (tid == 0)
    {
      double tot = 0;
      for (int j = 0; j < L.world; j++)
        tot += shv[j];
      ctl->sum[kind] = tot;
      ctl->seq[kind] = seq;
      if (sh_bad)
        ctl->error = 1;
      if (!single)
      {
        ctl->ticket[kind] = 0;
        if (planes)
          __hip_atomic_store(&ctl->go[kind], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  else
  {
    if (!planes)
      return;
    if (tid == 0)
    {
      sh_bad = 0;
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(&ctl->go[kind], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq)
      {
        if (wall_clock64() - t0 > 2 * L.budget)
        {
          // the publishing workgroup never got that far: report it and leave the ghost planes alone
          ctl->error = 1;
          sh_bad = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  if (!planes)
    return;
  // ---- neighbours' planes: staging -> ghost planes of r ------------------------------------
  __syncthreads();
  if (sh_bad)
    return; // a wait ran out of time: the staged planes may be stale
  __threadfence_system();
  if (has_lo)
    grid_copy(L.ghost_lo, links_staging(L.box[L.rank], par, LINK_FROM_LO, L.plane_pad), L.plane);
  if (has_hi)
    grid_copy(L.ghost_hi, links_staging(L.box[L.rank], par, LINK_FROM_HI, L.plane_pad), L.plane);
}

// ---- boxes -----------------------------------------------------------------------------------

static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handles travel as 64 bytes");

int links_alloc(size_t bytes, void **box)
{
  *box = nullptr;
  // memory that no cache of this GPU keeps (remote ranks write it while local kernels poll it) and
  // that can be handed to another process: uncached first, fine-grained if that cannot be exported
  const unsigned flags[2] = {hipDeviceMallocUncached, hipDeviceMallocFinegrained};
  hipError_t e = hipErrorUnknown;
  for (int k = 0; k < 2; k++)
  {
    void *p = nullptr;
    e = hipExtMallocWithFlags(&p, bytes, flags[k]);
    if (e != hipSuccess)
    {
      (void)hipGetLastError();
      continue;
    }
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, p);
    if (e == hipSuccess)
      e = hipMemset(p, 0, bytes);
    if (e == hipSuccess)
      e = hipDeviceSynchronize();
    if (e == hipSuccess)
    {
      *box = p;
      return 0;
    }
    (void)hipGetLastError();
    (void)hipFree(p);
  }
  return (int)e;
}

// can kernels on the current device store into memory of `peer_device`?  Enables the access.
int links_peer_ok(int peer_device)
{
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess)
    return 0;
  if (dev == peer_device)
    return 1;
  int can = 0;
  e = hipDeviceCanAccessPeer(&can, dev, peer_device);
  if (e != hipSuccess || !can)
  {
    (void)hipGetLastError();
    return 0;
  }
  e = hipDeviceEnablePeerAccess(peer_device, 0);
  (void)hipGetLastError(); // "already enabled" is fine
  return (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) ? 1 : 0;
}

int links_free(void *box)
{
  const hipError_t e = hipFree(box);
  (void)hipGetLastError();
  return (int)e;
}

int links_export(void *box, unsigned char handle[64])
{
  hipIpcMemHandle_t h;
  const hipError_t e = hipIpcGetMemHandle(&h, box);
  (void)hipGetLastError();
  if (e == hipSuccess)
    memcpy(handle, &h, 64);
  return (int)e;
}

int links_open(const unsigned char handle[64], void **box)
{
  hipIpcMemHandle_t h;
  memcpy(&h, handle, 64);
  *box = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(box, h, hipIpcMemLazyEnablePeerAccess);
  (void)hipGetLastError();
  return (int)e;
}

// Host-side access to word `slot` (0..63) of a box's staging area, through the copy engines: a
// mapping that is not usable comes back as an error code here instead of as a fault in a kernel.
int links_poke(void *box, int slot, unsigned long long value)
{
  if (slot < 0 || slot >= CG_LINKS_MAX_RANKS)
    return (int)hipErrorInvalidValue;
  char *dst = static_cast<char *>(box) + CG_LINKS_HEADER + (size_t)slot * sizeof(value);
  hipError_t e = hipMemcpy(dst, &value, sizeof(value), hipMemcpyDefault);
  if (e == hipSuccess)
    e = hipDeviceSynchronize();
  (void)hipGetLastError();
  return (int)e;
}

int links_peek(void *box, int slot, unsigned long long *value)
{
  if (slot < 0 || slot >= CG_LINKS_MAX_RANKS)
    return (int)hipErrorInvalidValue;
  const char *src = static_cast<const char *>(box) + CG_LINKS_HEADER + (size_t)slot * sizeof(*value);
  const hipError_t e = hipMemcpy(value, src, sizeof(*value), hipMemcpyDefault);
  (void)hipGetLastError();
  return (int)e;
}

int links_close(void *box)
{
  const hipError_t e = hipIpcCloseMemHandle(box);
  (void)hipGetLastError();
  return (int)e;
}

} // namespace cgamd
