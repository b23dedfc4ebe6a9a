// cg_rccl.hip -- the multi-GPU exchange on RCCL (over xGMI inside a node): the dot-product all-reduces and
// the halo exchange of the slab decomposition, enqueued by the library itself on the solver's stream.
//
// One rank per GPU; rank r owns slab r of the grid cut along its slowest axis.  Per CG iteration:
//   after kernel A   ncclAllReduce(sum) of the per-workgroup partials of p.Ap       (<= 32 KiB of doubles)
//   after kernel B   ncclAllReduce(sum) of the partials of r.r, then one grouped
//                    ncclSend / ncclRecv pair per neighbour: r's first / last owned hyperplane into the
//                    neighbour's ghost plane (only r is exchanged; kernel A rebuilds p on the ghost planes)
// The partials are reduced element-wise, not pre-summed: the consumer kernel adds the reduced array in its
// usual fixed order, so no extra reduction launch exists and every rank derives alpha, beta and the stop
// decision from the same bits (an all-reduce hands every rank the same result).  Ranks may own different
// numbers of workgroups: all of them reduce max-over-ranks entries, the tail of a shorter array stays zero.
// The reference has no multi-GPU code; its loop (conjugate-gradient_gpu.cu:349-370) is what every rank runs.
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>

#include "../../include/conjugate_gradient_gpu.h"
#include "cg_solver.hpp"
#include "hip_check.hpp"

namespace cgamd
{

#define RCCL_TRY(call)                                                                                   \
  {                                                                                                      \
    const ncclResult_t nccl_rc_ = (call);                                                                      \
    if (nccl_rc_ != ncclSuccess)                                                                               \
    {                                                                                                    \
      fprintf(stderr, "RCCL error %s:%d, code: %d, reason: %s\n", __FILE__, __LINE__, (int)nccl_rc_, ncclGetErrorString(nccl_rc_)); \
      return (int)nccl_rc_;                                                                                    \
    }                                                                                                    \
  }

// inside the iteration there is nobody to return a code to: same convention as a failed launch
#define RCCL_CHECK(call)                                                                                 \
  {                                                                                                      \
    const ncclResult_t nccl_rc_ = (call);                                                                      \
    if (nccl_rc_ != ncclSuccess)                                                                               \
    {                                                                                                    \
      fprintf(stderr, "Error: %s:%d, code: %d, reason: %s\n", __FILE__, __LINE__, (int)nccl_rc_, ncclGetErrorString(nccl_rc_)); \
      exit(1);                                                                                           \
    }                                                                                                    \
  }

int Solver::bind_rccl(void *comm, int rank, int world)
{
  CHECK(hipStreamSynchronize(stream_));
  if (!comm)
  {
    rccl_ = false;
    comm_ = nullptr;
    partA_ = partA_own_;
    partB_ = partB_own_;
    bind_sums(nullptr, 0, nullptr, 0);
    return 0;
  }
  CG_REQUIRE(!linked_, "boxes are bound: unbind them before binding a communicator");
  CG_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank out of range");
  CG_REQUIRE((desc_.has_lo != 0) == (rank > 0) && (desc_.has_hi != 0) == (rank + 1 < world),
             "has_lo / has_hi of the solver do not match its rank");
  ncclComm_t c = static_cast<ncclComm_t>(comm);
  int n = 0, r = -1;
  RCCL_TRY(ncclCommCount(c, &n));
  RCCL_TRY(ncclCommUserRank(c, &r));
  CG_REQUIRE(n == world && r == rank, "rank / world do not match the communicator");
  const size_t np = 2 * CG_MAX_BLOCKS;
  if (!sum_bufs_)
    CHECK(hipMalloc((void **)&sum_bufs_, 2 * np * sizeof(double)));
  CHECK(hipMemset(sum_bufs_, 0, 2 * np * sizeof(double)));
  partA_ = partA_own_;
  partB_ = partB_own_;
  // what a kernel does not write must read as zero on every rank
  CHECK(hipMemset(partA_own_, 0, np * sizeof(double)));
  CHECK(hipMemset(partB_own_, 0, np * sizeof(double)));
  // the common length of the reduced arrays; this first collective also sets the connections up outside
  // any stream capture
  int *cnt = reinterpret_cast<int *>(sum_bufs_);
  int mine = grid_;
  CHECK(hipMemcpy(cnt, &mine, sizeof(int), hipMemcpyHostToDevice));
  RCCL_TRY(ncclAllReduce(cnt, cnt + 1, 1, ncclInt, ncclMax, c, stream_));
  CHECK(hipStreamSynchronize(stream_));
  int most = 0;
  CHECK(hipMemcpy(&most, cnt + 1, sizeof(int), hipMemcpyDeviceToHost));
  CG_REQUIRE(most >= grid_ && most <= (int)np, "unexpected number of partial sums across the ranks");
  CHECK(hipMemset(sum_bufs_, 0, 2 * sizeof(int)));
  nparts_all_ = most;
  comm_ = comm;
  comm_rank_ = rank;
  comm_world_ = world;
  rccl_ = true;
  bind_sums(sum_bufs_, most, sum_bufs_ + np, most);
  // the neighbour connections as well (r's ghost planes receive the boundary planes of the zeroed r)
  rccl_exchange_planes();
  CHECK(hipStreamSynchronize(stream_));
  return 0;
}

void Solver::rccl_allreduce(int which)
{
  const size_t np = 2 * CG_MAX_BLOCKS;
  RCCL_CHECK(ncclAllReduce(which ? partB_ : partA_, sum_bufs_ + (which ? np : 0), (size_t)nparts_all_, ncclDouble,
                           ncclSum, static_cast<ncclComm_t>(comm_), stream_));
}

void Solver::rccl_exchange_planes()
{
  if (!desc_.has_lo && !desc_.has_hi)
    return;
  ncclComm_t c = static_cast<ncclComm_t>(comm_);
  RCCL_CHECK(ncclGroupStart());
  if (desc_.has_lo)
  {
    RCCL_CHECK(ncclSend(r_, plane_, ncclFloat, comm_rank_ - 1, c, stream_));
    RCCL_CHECK(ncclRecv(r_ - plane_, plane_, ncclFloat, comm_rank_ - 1, c, stream_));
  }
  if (desc_.has_hi)
  {
    RCCL_CHECK(ncclSend(r_ + N_ - plane_, plane_, ncclFloat, comm_rank_ + 1, c, stream_));
    RCCL_CHECK(ncclRecv(r_ + N_, plane_, ncclFloat, comm_rank_ + 1, c, stream_));
  }
  RCCL_CHECK(ncclGroupEnd());
}

} // namespace cgamd

using namespace cgamd;

static_assert(NCCL_UNIQUE_ID_BYTES == CG_RCCL_ID_BYTES, "unique ids travel as 128 bytes");

extern "C" int cg_rccl_unique_id(unsigned char *id128)
{
  ncclUniqueId id;
  RCCL_TRY(ncclGetUniqueId(&id));
  memcpy(id128, &id, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

extern "C" int cg_rccl_comm_create(const unsigned char *id128, int rank, int world, void **comm)
{
  *comm = nullptr;
  ncclUniqueId id;
  memcpy(&id, id128, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t c = nullptr;
  RCCL_TRY(ncclCommInitRank(&c, world, id, rank));
  *comm = c;
  return 0;
}

extern "C" int cg_rccl_comm_destroy(void *comm)
{
  if (comm)
    RCCL_TRY(ncclCommDestroy(static_cast<ncclComm_t>(comm)));
  return 0;
}

struct cg_solver;
Solver &cg_solver_impl(cg_solver *s);

extern "C" int cg_solver_bind_rccl(cg_solver *s, void *comm, int rank, int world)
{
  return cg_solver_impl(s).bind_rccl(comm, rank, world);
}
