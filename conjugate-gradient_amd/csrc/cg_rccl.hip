// cg_rccl.hip -- the multi-GPU exchange on RCCL (over xGMI inside a node): the dot-product exchanges and
// the halo exchange of the slab decomposition, enqueued by the library itself.
//
// One rank per GPU; rank r owns slab r of the grid cut along its slowest axis.  Per CG iteration:
//   after kernel A   ncclAllGather of ONE double per rank: this rank's p.Ap, added up in the fixed order by
//                    the last workgroup of kernel A to finish (no reduction launch); kernel B's prologue
//                    adds the `world` doubles in a fixed order
//   after kernel B   the same for r.r
//   beside kernel B  on a second stream and a communicator of its own: the two boundary planes r will have
//                    after this iteration are computed ahead of kernel B from a copy of the planes taken
//                    beside kernel A (Solver::edge_update, ~8 planes of traffic), then one grouped
//                    ncclSend / ncclRecv pair per neighbour puts them into the neighbours' ghost planes of r,
//                    which nothing reads before the next kernel A.  Only r is exchanged; kernel A rebuilds p
//                    on the ghost planes.
// Critical path of an iteration: kernel A, all-gather (8 bytes per rank), kernel B, all-gather; the planes
// (4 MiB each way per neighbour on a 1024 x 1024 x 128 slab) are off it.  Every rank adds the same doubles in
// the same order, so alpha, beta and the stop decision agree bit for bit.
// The reference has no multi-GPU code; its loop (conjugate-gradient_gpu.cu:349-370) is what every rank runs.
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
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

double *Solver::gathered(int which) const { return rank_sums_ + 2 + (size_t)which * CG_RCCL_MAX_RANKS; }

unsigned int *Solver::ticket(int which) const
{
  return reinterpret_cast<unsigned int *>(rank_sums_ + 2 + 2 * (size_t)CG_RCCL_MAX_RANKS) + which;
}

void Solver::rccl_release()
{
  if (own_halo_comm_ && comm_halo_)
  {
    const ncclResult_t rc = ncclCommDestroy(static_cast<ncclComm_t>(comm_halo_));
    if (rc != ncclSuccess)
      fprintf(stderr, "RCCL: destroying the halo communicator failed: %s\n", ncclGetErrorString(rc));
  }
  comm_halo_ = nullptr;
  own_halo_comm_ = false;
}

int Solver::rccl_ranks() const
{
  int n = 0;
  if (rccl_ && ncclCommCount(static_cast<ncclComm_t>(comm_), &n) != ncclSuccess)
    n = -1;
  return n;
}

int Solver::bind_rccl(void *comm, int rank, int world, int lo_rank, int hi_rank)
{
  CHECK(hipStreamSynchronize(stream_));
  if (side_)
    CHECK(hipStreamSynchronize(side_));
  if (!comm)
  {
    drop_graphs(); // RCCL keeps a communicator alive while a graph that holds its operations exists
    rccl_release();
    rccl_ = false;
    comm_ = nullptr;
    partA_ = partA_own_;
    partB_ = partB_own_;
    bind_sums(nullptr, 0, nullptr, 0);
    return 0;
  }
  CG_REQUIRE(!linked_, "boxes are bound: unbind them before binding a communicator");
  CG_REQUIRE(world >= 1 && world <= CG_RCCL_MAX_RANKS && rank >= 0 && rank < world, "rank out of range");
  CG_REQUIRE(lo_rank >= -1 && lo_rank < world && hi_rank >= -1 && hi_rank < world, "neighbour rank out of range");
  CG_REQUIRE((desc_.has_lo != 0) == (lo_rank >= 0) && (desc_.has_hi != 0) == (hi_rank >= 0),
             "has_lo / has_hi of the solver do not match its neighbour ranks");
  ncclComm_t c = static_cast<ncclComm_t>(comm);
  int n = 0, r = -1;
  RCCL_TRY(ncclCommCount(c, &n));
  RCCL_TRY(ncclCommUserRank(c, &r));
  CG_REQUIRE(n == world && r == rank, "rank / world do not match the communicator");
  const size_t bytes = (2 + 2 * (size_t)CG_RCCL_MAX_RANKS) * sizeof(double) + 16;
  if (!rank_sums_)
    CHECK(hipMalloc((void **)&rank_sums_, bytes));
  CHECK(hipMemset(rank_sums_, 0, bytes));
  partA_ = partA_own_;
  partB_ = partB_own_;
  // what a kernel does not write must read as zero
  CHECK(hipMemset(partA_own_, 0, 2 * CG_MAX_BLOCKS * sizeof(double)));
  CHECK(hipMemset(partB_own_, 0, 2 * CG_MAX_BLOCKS * sizeof(double)));
  // nothing may be allocated or created once a capture is open: do it all here
  ensure_edge();
  ensure_side();
  // the plane exchange runs beside the dot-product exchanges: a communicator of its own keeps the two
  // independent (collective; if RCCL cannot split, the planes share comm and RCCL orders them itself)
  drop_graphs();
  rccl_release();
  ncclComm_t halo = nullptr;
  const char *shared = getenv("CG_RCCL_SHARED_COMM"); // 1: no second communicator (every rank alike)
  const ncclResult_t sp = (shared && *shared == '1') ? ncclInvalidUsage : ncclCommSplit(c, 0, rank, &halo, nullptr);
  // every rank must end up with the same choice: agree on it (minimum over the ranks of "my split worked")
  int *flag = reinterpret_cast<int *>(rank_sums_);
  int mine = (sp == ncclSuccess && halo) ? 1 : 0, all = 0;
  CHECK(hipMemcpy(flag, &mine, sizeof(int), hipMemcpyHostToDevice));
  RCCL_TRY(ncclAllReduce(flag, flag + 1, 1, ncclInt, ncclMin, c, stream_));
  CHECK(hipStreamSynchronize(stream_));
  CHECK(hipMemcpy(&all, flag + 1, sizeof(int), hipMemcpyDeviceToHost));
  CHECK(hipMemset(rank_sums_, 0, bytes));
  if (all)
  {
    comm_halo_ = halo;
    own_halo_comm_ = true;
  }
  else
  {
    if (mine)
      (void)ncclCommDestroy(halo);
    if (!(shared && *shared == '1'))
      fprintf(stderr, "RCCL: ncclCommSplit failed on some rank (here: %s); boundary planes share the communicator\n",
              ncclGetErrorString(sp));
    comm_halo_ = comm;
  }
  comm_ = comm;
  comm_rank_ = rank;
  comm_world_ = world;
  lo_rank_ = lo_rank;
  hi_rank_ = hi_rank;
  {
    // planes of at least CG_RCCL_OVERLAP_BYTES (default 256 KiB) travel beside kernel B, smaller ones after it
    const char *e = getenv("CG_RCCL_OVERLAP_BYTES");
    const long long least = (e && *e) ? atoll(e) : (256ll << 10);
    halo_overlap_ = (long long)(plane_ * sizeof(float)) >= least;
  }
  rccl_ = true;
  bind_sums(gathered(0), world, gathered(1), world);
  // set every connection up outside any stream capture: one all-gather, one plane exchange (r's ghost
  // planes receive boundary planes of whatever r holds; the next begin() overwrites them)
  rccl_allgather(0);
  rccl_exchange_from_r();
  CHECK(hipStreamSynchronize(stream_));
  CHECK(hipMemset(rank_sums_, 0, bytes));
  return 0;
}

void Solver::rccl_allgather(int which)
{
  RCCL_CHECK(ncclAllGather(&rank_sums_[which], gathered(which), 1, ncclDouble, static_cast<ncclComm_t>(comm_), stream_));
}

// to_lo / to_hi: what the rank below / above receives into its hi / lo ghost plane of r; what they send lands
// in this rank's ghost planes.  A rank that is its own neighbour on both sides (a ring of one, used to test
// this path on one GPU) receives its sends in the order it posted them.
void Solver::rccl_send_recv(const float *to_lo, const float *to_hi, hipStream_t s)
{
  if (lo_rank_ < 0 && hi_rank_ < 0)
    return;
  ncclComm_t c = static_cast<ncclComm_t>(comm_halo_);
  RCCL_CHECK(ncclGroupStart());
  if (lo_rank_ >= 0)
  {
    RCCL_CHECK(ncclSend(to_lo, plane_, ncclFloat, lo_rank_, c, s));
    RCCL_CHECK(ncclRecv(r_ - plane_, plane_, ncclFloat, lo_rank_, c, s));
  }
  if (hi_rank_ >= 0)
  {
    RCCL_CHECK(ncclSend(to_hi, plane_, ncclFloat, hi_rank_, c, s));
    RCCL_CHECK(ncclRecv(r_ + N_, plane_, ncclFloat, hi_rank_, c, s));
  }
  RCCL_CHECK(ncclGroupEnd());
}

void Solver::rccl_exchange_from_r()
{
  if (lo_rank_ < 0 && hi_rank_ < 0)
    return;
  if (!halo_overlap_)
  {
    rccl_send_recv(r_, r_ + N_ - plane_, stream_);
    return;
  }
  CHECK(hipEventRecord(ev_fork_, stream_));
  CHECK(hipStreamWaitEvent(side_, ev_fork_, 0));
  rccl_send_recv(r_, r_ + N_ - plane_, side_);
  CHECK(hipEventRecord(ev_join_, side_));
  CHECK(hipStreamWaitEvent(stream_, ev_join_, 0));
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
  // slab r lies between slabs r - 1 and r + 1
  return cg_solver_impl(s).bind_rccl(comm, rank, world, rank > 0 ? rank - 1 : -1, rank + 1 < world ? rank + 1 : -1);
}

extern "C" int cg_solver_bind_rccl_peers(cg_solver *s, void *comm, int rank, int world, int lo_rank, int hi_rank)
{
  return cg_solver_impl(s).bind_rccl(comm, rank, world, lo_rank, hi_rank);
}

extern "C" int cg_solver_rccl_ranks(cg_solver *s) { return cg_solver_impl(s).rccl_ranks(); }
