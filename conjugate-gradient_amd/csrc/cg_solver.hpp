// cg_solver.hpp -- host control of the fused CG iteration (single precision, one GPU or
// one slab of a multi-GPU grid).  Mirrors the loop of conjugate-gradient_gpu.cu:327-376
// with alpha/beta, the iteration counter and the stop test on the device; the host only
// enqueues launches (optionally replayed from a hipGraph) and reads the state back once.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include <vector>

#include "cg_device.hpp"
#include "cg_links.hpp"
#include "cg_morton.hpp"
#include "cg_planes.hpp"
#include "cg_cols.hpp"
#include "cg_small.hpp"

namespace cgamd
{

enum Engine
{
  ENGINE_AUTO = 0,
  ENGINE_ROWS = 1,
  ENGINE_SCALAR = 2,
  ENGINE_PLANES = 3, // 3-D only: march along axis 2
  ENGINE_MORTON4 = 4 // Morton layout, 4 points per lane
};

struct SolverDesc
{
  int d = 0;
  unsigned extents[CG_MAXD] = {0};
  int index_mode = 0;            // 0,1: lexicographic (identical, as in the reference); 2: Morton
  int has_lo = 0, has_hi = 0;    // this slab has a neighbour below / above along the last axis
  hipStream_t stream = nullptr;  // nullptr: the solver creates its own
  float *r_buf = nullptr;        // optional caller-owned storage, buffer_len() floats each
  float *p0_buf = nullptr;
  float *p1_buf = nullptr;
  int engine = ENGINE_AUTO;
};

class Solver
{
public:
  explicit Solver(const SolverDesc &desc);
  ~Solver();
  Solver(const Solver &) = delete;
  Solver &operator=(const Solver &) = delete;

  // geometry of the ghosted vectors: [pad | lo ghost plane | N owned | hi ghost plane]
  static size_t plane_of(const SolverDesc &d);
  static size_t ghost_offset_of(const SolverDesc &d); // floats before the first owned element
  static size_t buffer_len_of(const SolverDesc &d);
  size_t N() const { return N_; }
  size_t plane() const { return plane_; }
  size_t ghost_offset() const { return goff_; }
  hipStream_t stream() const { return stream_; }
  float *r() const { return r_; }   // first owned element
  float *p(int which) const { return which ? p1_ : p0_; }
  double *partials(int which) const { return which ? partB_ : partA_; }
  int engine() const { return engine_; }
  bool paired_x() const { return xdefer_; }
  int blocks() const { return grid_; }
  int blocks_of(int which) const { return which ? gridB_ : gridA_; } // partials kernel A (0) / kernel B (1) writes
  size_t device_bytes() const; // device memory this solver allocated itself
  const char *kernel_name() const;

  // ---- whole-solve interface (all asynchronous on stream() until finish) -------------
  // r = b, p = 0, r.r; x is the caller's device vector of N floats and is accumulated into
  void begin(const float *b_dev, float *x_dev, double rel, int lagged);
  // enqueue n more iterations (they become no-ops once the stop test has fired)
  void iterate(int n, bool use_graph);
  // wait, return the number of executed iterations; residue = what the reference returns
  int finish(float *residue, float *hist, int hist_len);
  // enqueue up to maxit iterations (maxit < 0: unbounded), polling the device stop flag
  // every `poll` iterations so that a converged solve stops launching; synchronous
  void run(int maxit, bool use_graph, int poll = 40);

  // ---- step interface for the slab driver (scalars all-reduced between the steps) -----
  void begin_local(const float *b_dev, float *x_dev);         // r = b, p0 = 0, partials(1) = local r.r
  void reduce_partials(int which, double *out_dev);           // out[0] = sum of partials(which)
  void bind_sums(const double *sumA, int nA, const double *sumB, int nB);
  // per-workgroup partials go to (and the sums are read from) caller-owned arrays of blocks() doubles,
  // so that the caller can all-reduce them element-wise instead of reducing first
  void bind_partials(double *partA, double *partB);
  void init_state(double rel, int lagged, int maxit);         // from the bound r.r sum
  void add_iterations(int n);                                 // maxit = it + n
  void launch_A();
  // may_defer: this launch may leave x alone (it does unless the previous launch did); the caller ends with flush_x()
  void launch_B(bool may_defer = true);
  void flush_x();
  void launch_stencil(const float *u, float *out, bool ghost); // out = A u with this geometry
  void read_state(CgState *host_copy);                        // synchronises

  // per-kernel device time over n iterations, HIP events on stream(); plain launches
  void time_kernels(int n, float *ms_A, float *ms_B);

  // ---- direct xGMI path (cg_links.hpp): with boxes bound, begin / iterate / finish above run
  // the multi-rank solve by themselves (an exchange launch after every kernel A and B) -------
  // boxes[0..world): every rank's box as mapped in this process, boxes[rank] this rank's own
  void bind_links(int rank, int world, void *const *boxes);
  bool linked() const { return linked_; }
  void launch_exchange(int kind, bool force); // kind 0: p.Ap; 1: r.r and r's boundary planes
  int links_error();                          // synchronises; 1 if a wait ever ran out of time
  void links_times(float *ms_xA, float *ms_xB) const; // exchange launches, from the last time_kernels

  // ---- r's boundary planes ahead of kernel B (what lets the halo exchange run beside kernel B) ----------
  // edge_copy: the edge buffers take r's first / last owned hyperplane (r as the previous kernel B left it);
  // edge_update: they become what kernel B of the coming iteration will leave in those planes, bit for bit
  // (needs the sum of p.Ap bound, i.e. runs after kernel A's dot product has been exchanged).  Both are
  // enqueued on `s`; neither touches r, so they may run while kernel B does.
  void edge_copy(hipStream_t s);
  void edge_update(hipStream_t s);
  float *edge_plane(int side); // 0: the first owned plane (for the rank below), 1: the last (for the rank above)

  // ---- RCCL path: with a communicator bound, begin / iterate / finish run the multi-rank solve by
  // themselves.  After kernel A one ncclAllGather of one double per rank (this rank's p.Ap, added up by the
  // kernel's last workgroup), after kernel B one of r.r; r's boundary planes travel on a second stream while
  // kernel B runs (edge_update, then a grouped ncclSend / ncclRecv on a communicator of its own), joined
  // before the next kernel A.  Everything is captured in the same hipGraph as the kernels (no host work
  // inside the loop).
  // comm: an ncclComm_t whose rank r holds slab r (cg_rccl.hip creates one from a unique id); nullptr unbinds.
  // lo_rank / hi_rank: the ranks that own the slab below / above (-1: none; must agree with has_lo / has_hi).
  // Returns an ncclResult_t value, 0 = success.  Collective: every rank of the communicator calls it.
  int bind_rccl(void *comm, int rank, int world, int lo_rank, int hi_rank);
  bool rccl_bound() const { return rccl_; }
  int rccl_ranks() const; // ranks of the bound communicator as RCCL counts them (0: none bound)

private:
  void build_geometry();
  // A, B and, with boxes or a communicator bound, the exchanges; ev (5 events, may be null) are recorded
  // before A, after A, after A's exchange, after B, after B's exchange
  void launch_iteration(bool may_defer = true, hipEvent_t *ev = nullptr);
  void launch_mode(int mode, bool ghost, const VecArgs &a);
  VecArgs args() const;

  SolverDesc desc_;
  size_t N_ = 0, plane_ = 0, goff_ = 0, buflen_ = 0;
  hipStream_t stream_ = nullptr;
  bool own_stream_ = false, own_r_ = false, own_p0_ = false, own_p1_ = false;
  float *r_base_ = nullptr, *p0_base_ = nullptr, *p1_base_ = nullptr;
  float *r_ = nullptr, *p0_ = nullptr, *p1_ = nullptr, *x_ = nullptr;
  double *partA_ = nullptr, *partB_ = nullptr;
  double *partA_own_ = nullptr, *partB_own_ = nullptr;
  const double *sumA_ = nullptr, *sumB_ = nullptr;
  int nA_ = 0, nB_ = 0;
  CgState *st_ = nullptr;
  CgState *st_host_ = nullptr;
  int engine_ = ENGINE_SCALAR;
  int policy_ = POL_KEEP_R;   // kernel B (and what both kernels share)
  int policy_a_ = POL_KEEP_R; // kernel A
  bool xdefer_ = false; // kernel B of this solver's engine implements the paired x update
  bool last_deferred_ = false;  // the last kernel B enqueued left x alone: the next one must not
  bool may_be_pending_ = false; // a kernel B that may have left x alone was enqueued since the last flush_x()
  int layout_ = LAYOUT_LEX;
  RowsGeom rg_{}, rgA_{};     // kernel B (and the plain stencil) / kernel A
  PlanesGeom pg_{}, pgA_{};
  MortonGeom mg_{};
  int planes_r_ = 4;
  ScalarGeom sg_{};
  int grid_ = 1, block_ = 256;               // grid_: number of partial-sum addends = max(gridA_, gridB_)
  int gridA_ = 1, gridB_ = 1, blockA_ = 256; // launch shapes of kernel A / kernel B
  // padded rows: a lexicographic grid whose axis 0 is not a multiple of 4 points is kept internally
  // with rows of s0_ = n0 rounded up to 4 floats (padding zero), so that the 16-byte engines apply;
  // b and x are copied in at begin_local() and x is copied back by finish()
  bool pad_ = false;
  unsigned s0_ = 0;          // internal row stride (== extents[0] without padding)
  size_t Np_ = 0;            // internal vector length
  float *xpad_ = nullptr;    // internal x
  float *x_user_ = nullptr;  // the caller's x while a padded solve is open
  void copy_rows(float *dst, size_t dpitch, const float *src, size_t spitch); // all rows of the grid
  // a Morton-ordered grid kept in lexicographic order inside the solver (the same begin / finish copies,
  // as permutations)
  bool relay_ = false;
  void relayout(float *dst, const float *src, bool to_lex);
  bool small_ = false;      // grid fits one workgroup: iterate() may run it in a single launch
  SmallGeom smg_{};
  void launch_small();
  bool small_path() const;  // ... and nothing (slab neighbours, external sums, boxes) forbids it
  bool rccl_ = false;
  void *comm_ = nullptr;     // ncclComm_t
  int comm_rank_ = 0, comm_world_ = 1;
  int lo_rank_ = -1, hi_rank_ = -1;
  bool halo_overlap_ = true; // the planes travel beside kernel B (large planes) or after it on stream_ (small ones)
  void *comm_halo_ = nullptr; // ncclComm_t for the plane exchange (split off comm_; comm_ itself if that failed)
  bool own_halo_comm_ = false;
  // rank_sums_[0..1]: this rank's p.Ap / r.r; then [2][CG_RCCL_MAX_RANKS] every rank's, gathered; then two tickets
  double *rank_sums_ = nullptr;
  double *gathered(int which) const;
  unsigned int *ticket(int which) const;
  void rccl_allgather(int which);
  void rccl_send_recv(const float *to_lo, const float *to_hi, hipStream_t s);
  void rccl_exchange_from_r(); // r's boundary planes as they are now (start of a solve), on the side stream
  void rccl_release();
  // second stream and the events that tie it to stream_ (also inside a capture)
  hipStream_t side_ = nullptr;
  hipEvent_t ev_fork_ = nullptr, ev_copy_ = nullptr, ev_alpha_ = nullptr, ev_join_ = nullptr;
  hipEvent_t ev_cap_begin_ = nullptr, ev_cap_end_ = nullptr; // fork / join of the second stream around a capture
  void ensure_side();
  float *edge_ = nullptr;    // [2][edge_pad_]
  size_t edge_pad_ = 0;
  void ensure_edge();
  bool linked_ = false;
  bool push_in_b_ = true;    // kernel B stores r's boundary planes into the neighbours' staging areas itself
  LinkArgs links_{};
  LinkCtl *link_ctl_ = nullptr;
  float ms_xA_ = 0, ms_xB_ = 0;
  hipGraphExec_t graph_ = nullptr, tail_graph_ = nullptr; // a chunk of iterations; the remainder of a request
  int graph_iters_ = 0, tail_iters_ = 0;
  void drop_graphs();
  std::vector<float> hist_all_; // r.r of every iteration the host has seen so far (read_state)
  int hist_seen_ = 0;
  float *graph_x_ = nullptr;
};

// y = A u for caller-owned arrays without ghost planes (u may have N or N+1 entries; the
// boundary slot is never read).  Synchronous on the given stream only if sync is set.
void laplace_apply(float *ddf, const float *u, int d, unsigned L, int index_mode, hipStream_t stream);

// deterministic two-stage dot product; returns the double sum
double dot_device(const float *v, const float *w, unsigned long long n, hipStream_t stream);

// make `s` wait for everything enqueued on the legacy default stream so far (no-op for s == nullptr)
void order_after_default_stream(hipStream_t s);

// the same dot product with nothing allocated and nothing read back: the per-workgroup partials go to
// `part` (device memory, at least 2 * CG_MAX_BLOCKS doubles) and (float) of their sum to out_dev[0];
// asynchronous on `stream`
void dot_device_into(const float *v, const float *w, unsigned long long n, double *part, float *out_dev,
                     hipStream_t stream);

void fill_uniform(float *dst, unsigned long long n, unsigned int seed, hipStream_t stream);

} // namespace cgamd
