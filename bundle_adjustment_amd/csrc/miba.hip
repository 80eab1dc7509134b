// miba.hip — C ABI (include/miba.h), device memory, LM loop and PCG driver of libmiba.
//
// Replaces ceres::Solve as called at /root/reference/src/BundleAdjustment.cpp:108 with the
// options of /root/reference/app/main.cpp:21-28: trust-region Levenberg-Marquardt
//  whose linear step is an implicit-Schur block-Jacobi PCG
// (App. B.3).  All O(observations) and O(points) arithmetic runs in the HIP kernels of
// miba_kernels.hpp; the host keeps only the scalar LM state machine.  No CPU fallback.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/miba.h"
#include "miba_kernels.hpp"
#include "miba_mf.hpp"
#include "miba_mf2.hpp"
#include "miba_pcg.hpp"
#include "miba_dense.hpp"
#include "miba_sparse.hpp"
#include "miba_plan.hpp"
#include "miba_comm.hpp"
#include "miba_sort.hpp"

using namespace miba;

namespace {

thread_local std::string g_last_error;

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct EvPair {
  hipEvent_t a, b;
  int kid;
  int tag;   // > 0: product of PCG iteration `tag`, classified (worked / returned at once) when the solve knows its length
};

// ---- single-process multi-device group ------------------------------------------------------------------------
// One handle drives 1..8 GPUs from one process (SURVEY 8b "Threading"): the group handle owns one rank handle per
// device and runs every entry point on one host thread per rank (rank 0 on the caller's thread).  The ranks are the
// same code as the one-process-per-GPU ranks; only the transport of allreduce() differs:
//   kCommRccl  RCCL communicators of ncclCommInitAll (the default for distinct devices)
//   kCommPeer  the library's own all-reduce: a 2-hop reduce-scatter + all-gather kernel over peer-mapped, fine-grained
//              staging buffers, rank-ordered sums, every rank receives the same bits
//   kCommHost  staged through host memory by the rank threads (no peer access needed; the default when a device id
//              repeats, which RCCL refuses)
enum { kCommRccl = 0, kCommPeer = 1, kCommHost = 2 };
struct RankGroup {
  int n = 0;
  int mode = kCommRccl;
  std::vector<miba_handle> sub;
  std::atomic<int> bar_count{0}, bar_gen{0}, abort{0};
  // kCommHost: per-rank pinned staging (kPeerChunk doubles) and the combined result
  std::vector<double*> hbuf;
  // kCommPeer: per rank, fine-grained device memory: [in | out] staging halves and the signal block
  std::vector<double*> pstage;
  std::vector<PeerSignals*> psig;
  // kCommPeer: per rank, the exchange buffer of the PCG tail kernel [2][n ranks][9 n_cam] (fine-grained; allocated with the problem)
  std::vector<double*> xstage;
  // decisions of rank 0 handed to the other ranks once per LM iteration (callback result, time limit)
  std::atomic<int> ctl_stop[2];
  // the solve in progress runs its iteration callback on a worker thread, from state snapshots (options.async_callbacks)
  std::atomic<int> async_cb{0};
  // spin barrier of the rank threads; fails (instead of hanging) when a rank has left with an error
  int barrier() {
    const int g = bar_gen.load(std::memory_order_acquire);
    if (bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
      bar_count.store(0, std::memory_order_relaxed);
      bar_gen.store(g + 1, std::memory_order_release);
      return 0;
    }
    const double t0 = now_s();
    for (long spins = 0; bar_gen.load(std::memory_order_acquire) == g; ++spins) {
      if (abort.load(std::memory_order_acquire)) return -1;
      if ((spins & 1023) == 1023) {
        if (now_s() - t0 > 120.0) { abort.store(1); return -1; }
        sched_yield();
      }
    }
    return 0;
  }
};

}  // namespace

// ---- static scratch of miba_dense_spd_solve_static (the entry point that allocates nothing) --------------------
// Layout in doubles: padded matrix [(N + 64) x N] at 0 (N <= kStaticMaxN), the diagonal-block factors at kStaticLdiagOff,
// the pivot flag (one int32) at kStaticFlagOff.  Every GPU the library runs on holds its own copy of the array until the
// process ends; in host code the array's NAME is only a stand-in for hipGetSymbolAddress, never a device address.
constexpr int kStaticMaxN = MIBA_STATIC_SOLVE_MAX_N;
static_assert(kStaticMaxN % kNB == 0, "static solve limit must be a multiple of the Cholesky block");
constexpr size_t kStaticLdiagOff = (size_t)(kStaticMaxN + kNB) * kStaticMaxN;
constexpr size_t kStaticFlagOff = kStaticLdiagOff + (size_t)(kStaticMaxN / kNB) * kLdiag;
constexpr size_t kStaticDoubles = kStaticFlagOff + 32;
static __device__ __align__(256) double scratch[kStaticDoubles];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
double* g_scratch_ptr[kMaxDevices] = {};   // device address of `scratch` on each device, looked up once per device

// device address of this device's copy of the scratch array; the device must be current
hipError_t static_scratch_for(int dev, double** out) {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
  if (!g_scratch_ptr[dev]) {
    double* p = nullptr;
    const hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return hipSuccess;
}
}  // namespace

struct miba_handle_s {
  miba_config cfg{};
  int device = 0;
  hipStream_t st = nullptr;
  ncclComm_t comm = nullptr;
  RankGroup* grp = nullptr;  // rank handle of a single-process multi-device group: the group it belongs to
  RankGroup* owned = nullptr;  // group handle (what miba_create_multi returns): the group it owns; it has no device state
  uint32_t peer_epoch = 0;   // kCommPeer: all-reduces issued so far (identical on every rank)
  int pcg_launches = 0, pcg_exchange = 0;   // shape of the last PCG solve's iteration (miba_kernel_times)
  uint32_t x_epoch = 0;      // kCommPeer: exchanges inside the PCG tail kernel issued so far (identical on every rank)
  double* static_scratch = nullptr;   // this device's copy of the file-scope scratch array (miba_dense_spd_solve_static)
  std::string err;
  bool has_problem = false;

  std::shared_ptr<Plan> plan_sp = std::make_shared<Plan>();   // global plan (host); shared by the ranks of a group
  Plan& plan_ref() { return *plan_sp; }
  // local shard extents
  int64_t t0 = 0, t1 = 0;    // tiles
  int64_t slot_base = 0, n_slots = 0, n_obs_local = 0;
  int32_t pt_base = 0, n_pt = 0;   // sorted-point range owned
  int32_t n_cam = 0;
  int64_t n_tiles = 0, n_rows = 0;
  int64_t n_valid_global = 0;

  // device arrays
  Tile* d_tiles = nullptr; Group* d_grps = nullptr;
  int32_t *d_idx = nullptr, *d_ocam = nullptr, *d_tile_cams = nullptr, *d_slab_row = nullptr, *d_cam_row_off = nullptr;
  int2* d_pt_rng = nullptr;
  int2* d_wl = nullptr;   // per 64-slot wavelet: {first point, number of points}
  uint8_t* d_cam_active = nullptr;
  double *d_oxy = nullptr, *d_cam = nullptr, *d_cam_cand = nullptr, *d_ct = nullptr, *d_sc = nullptr;
  double *d_ct_cand = nullptr, *d_cm = nullptr, *d_ptrec = nullptr;   // candidate-state camera table; matrix-free table
  int32_t* d_giant = nullptr;                     // local ids of giant tiles
  // camera-major sub-tile layout of the product (miba_mf2.hpp); k3_cm = 0: the point-major product (MIBA_K3=pm, tests / A-B)
  CmTile* d_cmt = nullptr; CmSub* d_cms = nullptr; uint2* d_cmi = nullptr; int4* d_cam64 = nullptr; uint8_t* d_cmslot = nullptr;
  int k3_cm = 1;
  double2* d_oxy_cm = nullptr;   // measurements in the order of the camera-major entries (K4 over the sub-tiles, k_mf_backsub_cm)
  // camera-major copy of the local observations (K2-mf camera side): segments of one camera each
  int4* d_seg = nullptr; int32_t* d_cmo_pt = nullptr; double2* d_cmo_xy = nullptr; int32_t* d_cam_seg_off = nullptr;
  double* d_slabB = nullptr;
  double* d_cfree = nullptr;                      // [n_cam][9] 1 = free, 0 = entry of a constant parameter block
  int32_t n_seg = 0;
  int32_t n_giant = 0;
  int pcg_nb = 1, red_nb = 1, red_cpb = 4;         // grids of the fused PCG kernels / camera reduction
  int n_cu = 1, dev_share = 1;                    // compute units of the device; ranks of this process that share it
  int tail_nb = 1, tail_n_vw = 0;                 // grid of pcg_tail: resident as a whole (it holds grid barriers); wave records
  int tail_occ = 1;                               // workgroups of pcg_tail one compute unit can hold (occupancy query at create)
  int k4_pm = 0;                                  // MIBA_K4=pm
  int k3_remap = 1;                               // the product's tile blocks in XCD-aware order (MIBA_XCD_REMAP=0: launch order)
  int xcd_remap = 1;                              // MIBA_XCD_REMAP=0: workgroups of the camera-side kernels in launch order
  int tail_split = 0;                             // 1: the PCG tail runs without grid barriers (three launches): MIBA_TAIL_SPLIT=1, or a barrier timed out once
  long long tail_timeout = 200000000ll;           // barrier patience in 100 MHz ticks (MIBA_TAIL_TIMEOUT_TICKS: tests)
  int32_t* d_tail_rec = nullptr;                  // [tail_n_vw][kTailRec] camera ranges of the tail's waves
  long long* d_tail_dbg = nullptr;                // MIBA_TAIL_DEBUG=1: phase time stamps of the tail's workgroups
  unsigned* d_bar = nullptr;                      // barrier counter of pcg_tail
  unsigned long long* d_clk = nullptr;            // [4] clock stamps of one workgroup of the last product launch (measurement)
  int cam_model = 0;                              // MIBA_MODEL_*: camera model of the problem (miba_problem.camera_model)
  std::vector<double> principal;                  // model 1: (ox, oy) per camera that the uploaded observations carry
  Loss loss = {kLossNone, 1.0};                   // robust loss of the current solve (options.loss_type / loss_scale)
  double* d_omega = nullptr;                      // [n_slots] loss weights rho' of the current linearisation (loss only)
  double *d_X = nullptr, *d_Xc = nullptr, *d_sp = nullptr, *d_J = nullptr;
  double* d_gather = nullptr;   // one process per GPU: all points of the problem, for miba_get_state's gather (allocated on first use)
  double *d_gp = nullptr, *d_dgp = nullptr, *d_yp = nullptr;
  double *d_slab9 = nullptr, *d_camacc = nullptr, *d_Minv = nullptr, *d_Dc2 = nullptr, *d_rhs = nullptr;
  double *d_px = nullptr, *d_pr = nullptr, *d_pz = nullptr, *d_pp = nullptr, *d_pq = nullptr, *d_probe = nullptr;
  // exact reduced solve (miba_dense.hpp), allocated on first use: dense S [N + 64][N], factors of its diagonal blocks, pivot flag
  double *d_S = nullptr, *d_Linv = nullptr;
  int32_t* d_dflag = nullptr;
  int dense_N = 0;
  double* d_camacc2 = nullptr;   // SCHUR_JACOBI: per-camera sum G'WG [n_cam][64], allocated on first use
  int4* d_dseg = nullptr; int32_t* d_dpts = nullptr; int32_t n_dseg = 0;   // planned assembly: block segments, their points
  bool dense_planned = false, dense_plan_tried = false;
  // tile-sparse exact solve (miba_sparse.hpp), built on first use: tiles of the lower block triangle of S + the right-hand-side
  // tile row, tile map, ordering (pos: camera group -> tile row, grp: its inverse), per-column and per-row tile lists
  bool k1_fold = true, cand_cost_done = false;   // candidate cost inside the back-substitution (backsub_and_candidate)
  int sparse_state = 0;            // 0 not tried, 1 ready, -1 the pattern does not fit (PCG to eta_exact stands in)
  int sp_nT = 0;
  int64_t sp_n_mat = 0;            // tiles of the matrix; the right-hand-side tiles follow them
  double *d_sp_tiles = nullptr, *d_sp_Ldiag = nullptr, *d_sp_y = nullptr, *d_sp_x = nullptr;
  int32_t *d_sp_map = nullptr, *d_sp_pos = nullptr, *d_sp_grp = nullptr;
  // dissection of the tile graph (sparse_symbolic): the order's groups (arcs, then separator sets by level) of one level touch no
  // column of each other, so a forward step takes one column of EVERY group of the level (two launches for all of them), level after
  // level.  Updates of a group that land outside the group (in columns of higher levels, or their right-hand side) go to private tiles
  // of the group and are added to the matrix in a fixed order in front of the next level (k_tchol_merge).  The workgroup lists of
  // all launches (d_sp_wg) are written once, here: sp_fwd / sp_bwd hold every step's offsets into them.
  struct SpStep { int64_t col0; int32_t n_col; int64_t upd0; int32_t n_upd; };
  int sp_n_arcs = 1;
  std::vector<int32_t> sp_arc;
  int4* d_sp_wg = nullptr;
  std::vector<SpStep> sp_fwd, sp_bwd;   // backward steps use col0 / n_col only
  struct SpMerge { size_t at; int32_t ptr0, n; };   // before forward step `at`: merge destinations [ptr0, ptr0 + n) of the lists
  std::vector<SpMerge> sp_merges;                   // (one per level of the order that left private tiles)
  int32_t *d_sp_merge_ptr = nullptr, *d_sp_merge_src = nullptr, *d_sp_merge_dst = nullptr;
  int64_t sp_n_priv = 0;           // private tiles (behind the right-hand-side tiles)
  int64_t sp_fill_tiles = 0;       // tiles below the diagonal after fill (reported)
  double *d_part_tile = nullptr, *d_part_small = nullptr, *d_part_pcg = nullptr;
  double *d_sum = nullptr, *d_max = nullptr;   // [S_COUNT] each
  Ctl* d_ctl = nullptr;
  std::vector<void*> allocs;

  // State snapshots for iteration callbacks that run beside the solver (options.async_callbacks): after every iteration that
  // changed the state, the cameras and this shard's points are copied on a second stream into one of two pinned buffers;
  // miba_get_state called from inside such a callback reads the snapshot (snap_read >= 0) and touches no device.
  hipStream_t st2 = nullptr;
  hipEvent_t ev_state = nullptr, ev_snap[2] = {nullptr, nullptr};
  double* snap_cam[2] = {nullptr, nullptr};
  double* snap_X[2] = {nullptr, nullptr};
  size_t snap_cam_n = 0, snap_X_n = 0;
  int snap_last = -1;                       // buffer of the newest snapshot
  bool snap_recorded[2] = {false, false};   // ev_snap[b] has been recorded at least once
  std::atomic<int> snap_seq{-1};            // newest iteration whose snapshot has been enqueued (published for the callback worker)
  int snap_buf_of[2] = {-1, -1};            // buffer that holds the state of iteration k, at [k & 1]
  int snap_read = -1;                       // >= 0 while a callback runs on the worker: the buffer miba_get_state reads
  // host pinned
  double* h_stage = nullptr; size_t h_stage_n = 0;   // staging for miba_get_state (points), grown on demand
  double* h_scal = nullptr;   // 2*S_COUNT
  Ctl* h_ctl = nullptr;

  bool jac_valid = false;     // d_J holds the (scaled) Jacobian of the current state
  bool scale_valid = false;
  std::vector<miba_iteration> iters;

  // kernel timing
  std::vector<EvPair> ev_free, ev_busy;
  int time_every = 8;            // the product is timed in one PCG iteration out of this many (MIBA_TIME_EVERY)
  hipEvent_t ev_poll = nullptr;  // marks the device-to-host copy of the PCG control block
  miba_kernel_times kt{};
};

namespace {

int fail(miba_handle h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  if (h) h->err = buf;
  return code;
}

#define HIPCHK(h, call)                                                                          \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) return fail(h, MIBA_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

#define NCCLCHK(h, call)                                                                         \
  do {                                                                                           \
    ncclResult_t r_ = (call);                                                                    \
    if (r_ != ncclSuccess) return fail(h, MIBA_ERR_RCCL, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

template <typename T>
int dev_alloc(miba_handle h, T** out, size_t count) {
  void* p = nullptr;
  const size_t bytes = (count ? count : 1) * sizeof(T);
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) return fail(h, MIBA_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
  h->allocs.push_back(p);
  *out = (T*)p;
  return MIBA_OK;
}

template <typename T>
int upload(miba_handle h, T* dst, const T* src, size_t count) {
  if (!count) return MIBA_OK;
  HIPCHK(h, hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyHostToDevice, h->st));
  HIPCHK(h, hipStreamSynchronize(h->st));
  return MIBA_OK;
}

void free_problem(miba_handle h) {
  for (void* p : h->allocs) (void)hipFree(p);
  h->allocs.clear();
  if (h->grp && (size_t)h->cfg.rank < h->grp->xstage.size()) h->grp->xstage[h->cfg.rank] = nullptr;   // (was one of the allocations)
  h->has_problem = false;
  h->jac_valid = false;
  h->scale_valid = false;
}

// ---- kernel timing -------------------------------------------------------------------------
// Event pairs whose launches have finished are added to the kernel times.  Pairs of PCG products (tag > 0) wait until
// classify_pcg_events has given them their final kernel id; pairs still in flight stay for the next call.
void drain_events(miba_handle h) {
  std::vector<EvPair> keep;
  for (auto& e : h->ev_busy) {
    if (e.tag > 0) { keep.push_back(e); continue; }
    float ms = 0.f;
    const hipError_t r = hipEventElapsedTime(&ms, e.a, e.b);
    if (r == hipErrorNotReady) { keep.push_back(e); continue; }
    if (r == hipSuccess && e.kid >= 0) {
      h->kt.total_ms[e.kid] += ms;
      h->kt.launches[e.kid] += 1;
    }
    h->ev_free.push_back(e);
  }
  (void)hipGetLastError();   // hipErrorNotReady is not an error of the stream
  h->ev_busy.swap(keep);
}

// end of a PCG solve of n_it iterations: the timed products of later iterations were enqueued ahead of the host poll
// and returned at once (ctl->done); they are counted apart from the ones that did the work
void classify_pcg_events(miba_handle h, int n_it) {
  for (auto& e : h->ev_busy)
    if (e.tag > 0) {
      if (e.tag > n_it) e.kid = (e.kid == MIBA_K_SCHUR_MATVEC) ? MIBA_K_SCHUR_MATVEC_SKIPPED : -1;   // -1: not counted
      e.tag = 0;
    }
}

struct Timed {
  miba_handle h;
  EvPair e{};
  bool on = false;
  Timed(miba_handle h_, int kid, int tag = 0, bool sample = true) : h(h_) {
    if (!sample) return;
    if (h->ev_free.empty()) {
      if (h->ev_busy.size() >= 4096) return;  // resolved at the next host sync
      EvPair n{};
      if (hipEventCreate(&n.a) != hipSuccess || hipEventCreate(&n.b) != hipSuccess) return;
      h->ev_free.push_back(n);
    }
    e = h->ev_free.back();
    h->ev_free.pop_back();
    e.kid = kid;
    e.tag = tag;
    on = (hipEventRecord(e.a, h->st) == hipSuccess);
    if (!on) h->ev_free.push_back(e);
  }
  ~Timed() {
    if (!on) return;
    (void)hipEventRecord(e.b, h->st);
    h->ev_busy.push_back(e);
  }
};

int sync_stream(miba_handle h) {
  HIPCHK(h, hipStreamSynchronize(h->st));
  drain_events(h);
  return MIBA_OK;
}

// blocks for n items; never 0 (an empty shard still launches one block whose bounds checks do nothing)
inline int nblk(size_t n, int bs) { return n ? (int)((n + bs - 1) / bs) : 1; }

// kCommHost: every rank thread copies its piece to pinned host memory, the threads meet, every rank combines the pieces
// in rank order (the same bits everywhere) and copies the result back.  Blocking; needs no peer access.
int allreduce_host(miba_handle h, double* buf, size_t n, ncclRedOp_t op) {
  RankGroup& G = *h->grp;
  const int r = h->cfg.rank;
  for (size_t o = 0; o < n; o += kPeerChunk) {
    const size_t m = std::min(kPeerChunk, n - o);
    HIPCHK(h, hipMemcpyAsync(G.hbuf[r], buf + o, m * sizeof(double), hipMemcpyDeviceToHost, h->st));
    HIPCHK(h, hipStreamSynchronize(h->st));
    if (G.barrier()) return fail(h, MIBA_ERR_RCCL, "a rank of the group left a reduction (see its error)");
    double* out = G.hbuf[G.n + r];
    for (size_t i = 0; i < m; ++i) {
      double v = G.hbuf[0][i];
      for (int q = 1; q < G.n; ++q) v = (op == ncclMax) ? std::max(v, G.hbuf[q][i]) : v + G.hbuf[q][i];
      out[i] = v;
    }
    if (G.barrier()) return fail(h, MIBA_ERR_RCCL, "a rank of the group left a reduction (see its error)");
    HIPCHK(h, hipMemcpyAsync(buf + o, out, m * sizeof(double), hipMemcpyHostToDevice, h->st));
    HIPCHK(h, hipStreamSynchronize(h->st));
  }
  return MIBA_OK;
}

// kCommPeer: the 2-hop kernel of miba_comm.hpp, stream-ordered, no host synchronisation
int allreduce_peer(miba_handle h, double* buf, size_t n, ncclRedOp_t op) {
  RankGroup& G = *h->grp;
  PeerArgs A{};
  for (int q = 0; q < G.n; ++q) { A.stage[q] = G.pstage[q]; A.sig[q] = G.psig[q]; }
  A.n_ranks = G.n;
  A.rank = h->cfg.rank;
  A.op_max = (op == ncclMax) ? 1 : 0;
  for (size_t o = 0; o < n; o += kPeerChunk) {
    const size_t m = std::min(kPeerChunk, n - o);
    A.epoch = ++h->peer_epoch;
    hipLaunchKernelGGL(k_peer_allreduce, dim3(kPeerBlocks), dim3(256), 0, h->st, A, buf + o, m);
    HIPCHK(h, hipGetLastError());
  }
  return MIBA_OK;
}

int allreduce(miba_handle h, double* buf, size_t n, ncclRedOp_t op) {
  if (h->grp && h->grp->mode == kCommHost) return allreduce_host(h, buf, n, op);
  if (h->grp && h->grp->mode == kCommPeer) return allreduce_peer(h, buf, n, op);
  if (!h->comm) return MIBA_OK;
  NCCLCHK(h, ncclAllReduce(buf, buf, n, ncclDouble, op, h->comm, h->st));
  return MIBA_OK;
}

// more than one rank contributes to the sums
inline bool multi_rank(miba_handle h) { return h->comm != nullptr || (h->grp && h->grp->n > 1); }

// fn(rank handle, rank) on one host thread per rank, rank 0 on the caller's thread.  A rank that fails raises the
// group's abort flag so that the others leave their barriers instead of waiting for it.
template <typename F>
int group_run(miba_handle g, F fn) {
  RankGroup& G = *g->owned;
  G.abort.store(0);
  std::vector<int> rc(G.n, 0);
  std::vector<std::thread> th;
  for (int r = 1; r < G.n; ++r)
    th.emplace_back([&, r] {
      rc[r] = fn(G.sub[r], r);
      if (rc[r]) G.abort.store(1);
    });
  rc[0] = fn(G.sub[0], 0);
  if (rc[0]) G.abort.store(1);
  for (auto& t : th) t.join();
  for (int r = 0; r < G.n; ++r)
    if (rc[r]) {
      g->err = "rank " + std::to_string(r) + ": " + G.sub[r]->err;
      g_last_error = g->err;
      return rc[r];
    }
  return MIBA_OK;
}

// once per LM iteration: rank 0's verdict (callback result, time limit) reaches every rank of a group
int group_exchange_stop(miba_handle h, int iteration, int* code) {
  if (!h->grp || h->grp->n < 2) return MIBA_OK;
  RankGroup& G = *h->grp;
  if (h->cfg.rank == 0) G.ctl_stop[iteration & 1].store(*code, std::memory_order_release);
  if (G.barrier()) return fail(h, MIBA_ERR_RCCL, "a rank of the group left the solve (see its error)");
  *code = G.ctl_stop[iteration & 1].load(std::memory_order_acquire);
  return MIBA_OK;
}

// ---- building blocks -----------------------------------------------------------------------
int launch_cam_tables(miba_handle h, const double* cam, bool candidate = false) {
  hipLaunchKernelGGL(k_cam_tables, dim3(nblk(h->n_cam, 128)), dim3(128), 0, h->st, h->n_cam, cam,
                     candidate ? h->d_ct_cand : h->d_ct, candidate ? (double*)nullptr : h->d_cm, h->cam_model);
  HIPCHK(h, hipGetLastError());
  return MIBA_OK;
}

// the stored Jacobian (26 doubles per slot) exists only for miba_evaluate, which hands it to the caller
int ensure_J(miba_handle h) {
  if (h->d_J) return MIBA_OK;
  return dev_alloc(h, &h->d_J, (size_t)kJComp * (size_t)(h->n_slots ? h->n_slots : 1));
}

// residual + Jacobian at (cam, X); cost -> d_sum[S_COST] (all-reduced)
int eval_jac(miba_handle h, const double* cam, const double* X, bool scaled) {
  int rc = ensure_J(h);
  if (rc) return rc;
  rc = launch_cam_tables(h, cam);
  if (rc) return rc;
  if (h->n_tiles) {
    Timed t(h, MIBA_K_EVAL_JAC);
    hipLaunchKernelGGL(k_eval<true>, dim3((unsigned)h->n_tiles), dim3(kChunk), 0, h->st, h->d_tiles, h->d_grps, h->d_idx,
                       h->d_ocam, h->d_oxy, h->d_ct, X, scaled ? h->d_sc : nullptr, scaled ? h->d_sp : nullptr, h->d_J,
                       h->d_part_tile);
  }
  HIPCHK(h, hipGetLastError());
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, h->st, h->d_part_tile, (size_t)h->n_tiles, 1, 0, 0, h->d_sum + S_COST);
  HIPCHK(h, hipGetLastError());
  return allreduce(h, h->d_sum + S_COST, 1, ncclSum);
}

int eval_cost(miba_handle h, const double* cam, const double* X, bool reduce = true) {
  // candidate state: its camera table must not disturb the table of the current state (K3-mf reads it)
  int rc = launch_cam_tables(h, cam, true);
  if (rc) return rc;
  if (h->n_tiles) {
    Timed t(h, MIBA_K_EVAL_COST);
#define MIBA_COST_ARGS                                                                                                   \
  dim3((unsigned)(h->n_tiles + h->n_giant)), dim3(kChunk), 0, h->st, (int)h->n_giant, h->d_tiles, h->d_giant, h->d_grps,   \
      h->d_idx, h->d_ocam, h->d_wl, h->d_tile_cams, h->d_oxy, h->d_ct_cand, X, h->d_part_tile, h->loss, 1, 0
    if (h->loss.type != kLossNone) hipLaunchKernelGGL(k_mf_cost<true>, MIBA_COST_ARGS);
    else hipLaunchKernelGGL(k_mf_cost<false>, MIBA_COST_ARGS);
#undef MIBA_COST_ARGS
  }
  HIPCHK(h, hipGetLastError());
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, h->st, h->d_part_tile, (size_t)h->n_tiles, 1, 0, 0, h->d_sum + S_COST);
  HIPCHK(h, hipGetLastError());
  return reduce ? allreduce(h, h->d_sum + S_COST, 1, ncclSum) : MIBA_OK;
}

// K2 + camera reductions + camera blocks; fills Cinv, gp, Minv, Dc2, rhs and the gradient norms
// K2-mf over all tiles of this shard (camera tables of the current state must be in d_ct / d_cm):
// per-point blocks, raw camera slab rows, cost -> d_sum[S_COST] (all-reduced)
int mf_init_launch(miba_handle h, double inv_r, const miba_options* o, bool want_diag, bool reduce_cost = true) {
  // the point side (K2a) and the camera side (K2b) are timed apart: MIBA_K_SCHUR_INIT / MIBA_K_SCHUR_INIT_CAM
  auto camrows = [&]() {
    if (!h->n_seg) return;
    Timed t(h, MIBA_K_SCHUR_INIT_CAM);
    // XCD-aware order (see the kernel): the grid padded to a multiple of eight workgroups
    const int remap = h->xcd_remap;
    const int nb = remap ? (nblk(h->n_seg, 4) + 7) / 8 * 8 : nblk(h->n_seg, 4);
    if (h->loss.type != kLossNone)
      hipLaunchKernelGGL(k_mf_camrows<true>, dim3(nb), dim3(256), 0, h->st, h->n_seg, h->d_seg, h->d_cmo_pt, h->d_cmo_xy,
                         h->d_ct, h->d_cm, h->d_ptrec, h->loss, h->d_slabB, remap);
    else
      hipLaunchKernelGGL(k_mf_camrows<false>, dim3(nb), dim3(256), 0, h->st, h->n_seg, h->d_seg, h->d_cmo_pt, h->d_cmo_xy,
                         h->d_ct, h->d_cm, h->d_ptrec, h->loss, h->d_slabB, remap);
  };
  if (h->n_tiles) {
    Timed t(h, MIBA_K_SCHUR_INIT);
#define MIBA_INIT_ARGS(list)                                                                                              \
  h->d_tiles, list, h->d_grps, h->d_idx, h->d_ocam, h->d_wl, h->d_tile_cams, h->d_slab_row, h->d_oxy, h->d_ct, h->d_cm,   \
      h->d_X, h->d_sp, inv_r, o->min_lm_diagonal, o->max_lm_diagonal, h->d_gp, want_diag ? h->d_dgp : (double*)nullptr,    \
      h->d_ptrec, h->d_part_tile, h->d_ctl, h->loss, h->d_omega
    // point side (wavelets); with a robust loss it leaves rho' per slot.  The camera side follows over the camera-major copy.
    if (h->loss.type != kLossNone) {
      hipLaunchKernelGGL((k_mf_init<false, true>), dim3((unsigned)h->n_tiles), dim3(kChunk), 0, h->st, MIBA_INIT_ARGS((const int32_t*)nullptr));
      if (h->n_giant) hipLaunchKernelGGL((k_mf_init<true, true>), dim3((unsigned)h->n_giant), dim3(kChunk), 0, h->st, MIBA_INIT_ARGS(h->d_giant));
    } else {
      hipLaunchKernelGGL((k_mf_init<false, false>), dim3((unsigned)h->n_tiles), dim3(kChunk), 0, h->st, MIBA_INIT_ARGS((const int32_t*)nullptr));
      if (h->n_giant) hipLaunchKernelGGL((k_mf_init<true, false>), dim3((unsigned)h->n_giant), dim3(kChunk), 0, h->st, MIBA_INIT_ARGS(h->d_giant));
    }
#undef MIBA_INIT_ARGS
  }
  if (h->n_tiles) camrows();
  HIPCHK(h, hipGetLastError());
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, h->st, h->d_part_tile, (size_t)h->n_tiles, 4, 0, 0, h->d_sum + S_COST);
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, h->st, h->d_part_tile, (size_t)h->n_tiles, 4, 1, 1, h->d_max + S_GMAX_P);
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, h->st, h->d_part_tile, (size_t)h->n_tiles, 4, 2, 0, h->d_sum + S_G2_P);
  HIPCHK(h, hipGetLastError());
  return reduce_cost ? allreduce(h, h->d_sum + S_COST, 1, ncclSum) : MIBA_OK;
}

// camera sums of K2: rows of the segment slab the camera pass has left
int cam_reduce64(miba_handle h, const double* sc) {
  hipLaunchKernelGGL(k_cam_reduce64, dim3(nblk(h->n_cam, 4)), dim3(256), 0, h->st, h->n_cam, h->d_cam_seg_off, h->d_slabB, sc, h->d_camacc);
  HIPCHK(h, hipGetLastError());
  return MIBA_OK;
}

// robust loss of this solve: validated, kept in the handle for the kernel launches; the weight array is allocated on
// first use.
int set_loss(miba_handle h, const miba_options* o) {
  if (o->loss_type == MIBA_LOSS_NONE) {
    h->loss = Loss{kLossNone, 1.0};
    return MIBA_OK;
  }
  if (o->loss_type != MIBA_LOSS_HUBER && o->loss_type != MIBA_LOSS_CAUCHY) return fail(h, MIBA_ERR_INVALID, "unknown loss_type %d", (int)o->loss_type);
  if (!(o->loss_scale > 0.0) || !std::isfinite(o->loss_scale)) return fail(h, MIBA_ERR_INVALID, "loss_scale must be positive and finite");
  if (!h->d_omega) {
    int rc = dev_alloc(h, &h->d_omega, h->n_slots ? h->n_slots : 1);
    if (rc) return rc;
  }
  h->loss = Loss{o->loss_type == MIBA_LOSS_HUBER ? kLossHuber : kLossCauchy, o->loss_scale};
  return MIBA_OK;
}

int schur_init(miba_handle h, double radius, const miba_options* o) {
  const double inv_r = 1.0 / radius;
  HIPCHK(h, hipMemsetAsync(&h->d_ctl->fail, 0, sizeof(int32_t), h->st));
  int rc;
  // With several ranks the three scalars of this stage (cost, |g_p|^2, max |g_p|) ride in the unused 64th entry of
  // the camera sums (cameras 0, 1 and 2 + rank: one slot per rank turns the max into a sum), so the stage costs ONE
  // all-reduce instead of four.
  const bool pack = multi_rank(h) && h->n_cam >= 2 + h->cfg.world_size;
  rc = mf_init_launch(h, inv_r, o, false, !pack);
  if (rc) return rc;
  rc = cam_reduce64(h, h->d_sc);
  if (rc) return rc;
  if (pack) hipLaunchKernelGGL(k_pack_scalars, dim3(1), dim3(64), 0, h->st, h->d_camacc, h->d_sum, h->d_max, h->cfg.rank, 0, h->cfg.world_size);
  rc = allreduce(h, h->d_camacc, (size_t)kK2W * h->n_cam, ncclSum);
  if (rc) return rc;
  if (pack) hipLaunchKernelGGL(k_pack_scalars, dim3(1), dim3(64), 0, h->st, h->d_camacc, h->d_sum, h->d_max, h->cfg.rank, 1, h->cfg.world_size);
  HIPCHK(h, hipGetLastError());
  const double* acc2 = nullptr;
  if (o->preconditioner_type == MIBA_PRECOND_SCHUR_JACOBI && o->linear_solver_type == MIBA_ITERATIVE_SCHUR) {
    // block diagonal of S itself: one more camera-major pass for sum G'WG (needs the point records K2 has just written)
    if (!h->d_camacc2) {
      rc = dev_alloc(h, &h->d_camacc2, (size_t)kK2W * h->n_cam);
      if (rc) return rc;
    }
    if (h->n_seg && h->loss.type != kLossNone)
      hipLaunchKernelGGL(k_mf_schurdiag<true>, dim3(nblk(h->n_seg, 4)), dim3(256), 0, h->st, h->n_seg, h->d_seg, h->d_cmo_pt, h->d_cmo_xy,
                         h->d_ct, h->d_cm, h->d_ptrec, h->loss, h->d_slabB);
    else if (h->n_seg)
      hipLaunchKernelGGL(k_mf_schurdiag<false>, dim3(nblk(h->n_seg, 4)), dim3(256), 0, h->st, h->n_seg, h->d_seg, h->d_cmo_pt, h->d_cmo_xy,
                         h->d_ct, h->d_cm, h->d_ptrec, h->loss, h->d_slabB);
    hipLaunchKernelGGL(k_cam_reduce64, dim3(nblk(h->n_cam, 4)), dim3(256), 0, h->st, h->n_cam, h->d_cam_seg_off, h->d_slabB, h->d_sc,
                       h->d_camacc2);
    HIPCHK(h, hipGetLastError());
    rc = allreduce(h, h->d_camacc2, (size_t)kK2W * h->n_cam, ncclSum);
    if (rc) return rc;
    acc2 = h->d_camacc2;
  }
  const int nb = nblk(h->n_cam, 64);
  hipLaunchKernelGGL(k_cam_blocks, dim3(nb), dim3(64), 0, h->st, h->n_cam, h->d_camacc, acc2, h->d_cam_active, h->d_sc, inv_r,
                     o->min_lm_diagonal, o->max_lm_diagonal, h->d_Minv, h->d_Dc2, h->d_rhs, h->d_part_small, h->d_ctl);
  HIPCHK(h, hipGetLastError());
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, h->st, h->d_part_small, (size_t)nb, 2, 0, 1, h->d_max + S_GMAX_C);
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, h->st, h->d_part_small, (size_t)nb, 2, 1, 0, h->d_sum + S_G2_C);
  HIPCHK(h, hipGetLastError());
  // point gradient norms: K2 has left them for the local points; all-reduce
  if (pack) return MIBA_OK;
  rc = allreduce(h, h->d_sum + S_G2_P, 1, ncclSum);
  if (rc) return rc;
  return allreduce(h, h->d_max + S_GMAX_P, 1, ncclMax);
}

// y = S x  (x, y: 9*n_cam device vectors).  ctl: control block whose done flag turns the launch into a no-op;
// part_xy (matrix-free, single GPU only): per-block partials of x.y from the camera reduction, red_nb entries.
// pcg_iter > 0: the product of that PCG iteration.  An event pair around a launch costs two 6 us bubbles on the
// stream, so inside the PCG the product is timed in one iteration out of h->time_every only (the first of every solve
// among them); products outside the PCG (probes, tools) are always timed.
// tile side of the matrix-free product: the raw slab rows of S x per (tile, camera)
void mf_matvec_tiles(miba_handle h, const double* x, Ctl* ctl, int pcg_iter, bool sample) {
  if (!h->n_tiles) return;
  Timed t(h, MIBA_K_SCHUR_MATVEC, pcg_iter, sample);
#define MIBA_MATVEC_ARGS                                                                                                \
  dim3((unsigned)(h->n_tiles + h->n_giant)), dim3(kChunk), 0, h->st, (int)h->n_giant, h->d_tiles, h->d_giant, h->d_grps,  \
      h->d_idx, h->d_ocam, h->d_wl, h->d_tile_cams, h->d_slab_row, h->d_ct, h->d_cm, h->d_sc, h->d_X, h->d_ptrec,        \
      (const double*)h->d_omega, x, h->d_slab9, ctl
  if (h->loss.type != kLossNone) hipLaunchKernelGGL(k_mf_matvec<true>, MIBA_MATVEC_ARGS);
  else if (h->k3_cm)
    hipLaunchKernelGGL(k_mf_matvec_cm, dim3((unsigned)((h->k3_remap ? (h->n_tiles + 7) / 8 * 8 : h->n_tiles) + h->n_giant)), dim3(kChunk), 0, h->st,
                       (int)h->n_giant, h->d_tiles, h->d_giant,
                       h->d_grps, h->d_idx, h->d_ocam, h->d_wl, h->d_tile_cams, h->d_slab_row, h->d_ct, h->d_cm, h->d_sc, h->d_X, h->d_ptrec, x,
                       h->d_slab9, ctl, CmArgs{h->d_cmt, h->d_cms, h->d_cmi, h->d_cmslot, h->d_cam64, h->d_clk}, h->k3_remap ? (int)h->n_tiles : 0);
  else hipLaunchKernelGGL(k_mf_matvec<false>, MIBA_MATVEC_ARGS);
#undef MIBA_MATVEC_ARGS
}

int schur_matvec(miba_handle h, const double* x, double* y, Ctl* ctl, double* part_xy, int pcg_iter = 0) {
  const bool sample = pcg_iter <= 0 || (pcg_iter - 1) % h->time_every == 0;
  mf_matvec_tiles(h, x, ctl, pcg_iter, sample);
  {
    Timed t(h, MIBA_K_CAM_REDUCE, pcg_iter, sample && pcg_iter > 0);
    hipLaunchKernelGGL(k_cam_reduce_mf, dim3(h->red_nb), dim3(256), 0, h->st, h->n_cam, h->red_cpb, h->d_cam_row_off, h->d_slab9,
                       h->d_ct, h->d_cm, h->d_sc, h->d_Dc2, x, h->cfg.rank == 0 ? 1.0 : 0.0, y, part_xy, ctl);
  }
  HIPCHK(h, hipGetLastError());
  // x.y is linear in y, so with several ranks the per-block partials of x.y_local are all-reduced together with
  // y_local (the caller places them right behind y): every rank then sums the same red_nb numbers
  const bool tail = part_xy && part_xy == y + 9 * (size_t)h->n_cam;
  if (multi_rank(h) && part_xy && !tail) return fail(h, MIBA_ERR_INVALID, "x.y partials must follow y for the all-reduce");
  return allreduce(h, y, 9 * (size_t)h->n_cam + (tail ? (size_t)h->red_nb : 0), ncclSum);
}

// constant parameter blocks: a zero scale removes the column from every kernel (F_s = F diag(sc))
int apply_const_mask(miba_handle h) {
  hipLaunchKernelGGL(k_mul, dim3(nblk(9 * (size_t)h->n_cam, 256)), dim3(256), 0, h->st, 9 * (size_t)h->n_cam, h->d_cfree, h->d_sc);
  HIPCHK(h, hipGetLastError());
  return MIBA_OK;
}

int unit_scale(miba_handle h);

// Jacobi scale of the matrix-free path: one K2-mf pass with unit scales gives the squared column norms
// as diag(F'F) per camera and diag(E'E) per point
int make_scale_mf(miba_handle h, const miba_options* o) {
  int rc = unit_scale(h);
  if (rc) return rc;
  rc = mf_init_launch(h, 1.0, o, true);
  if (rc) return rc;
  rc = cam_reduce64(h, nullptr);
  if (rc) return rc;
  rc = allreduce(h, h->d_camacc, (size_t)kK2W * h->n_cam, ncclSum);
  if (rc) return rc;
  hipLaunchKernelGGL(k_scale_from_acc, dim3(nblk(9 * (size_t)h->n_cam, 256)), dim3(256), 0, h->st, h->n_cam, h->d_camacc, h->d_sc);
  hipLaunchKernelGGL(k_make_scale, dim3(nblk(3 * (size_t)h->n_pt, 256)), dim3(256), 0, h->st, 3 * (size_t)h->n_pt, h->d_dgp, h->d_sp);
  HIPCHK(h, hipGetLastError());
  return apply_const_mask(h);
}

int unit_scale(miba_handle h) {
  hipLaunchKernelGGL(k_fill, dim3(nblk(9 * (size_t)h->n_cam, 256)), dim3(256), 0, h->st, 9 * (size_t)h->n_cam, 1.0, h->d_sc);
  hipLaunchKernelGGL(k_fill, dim3(nblk(3 * (size_t)h->n_pt, 256)), dim3(256), 0, h->st, 3 * (size_t)h->n_pt, 1.0, h->d_sp);
  HIPCHK(h, hipGetLastError());
  return MIBA_OK;
}

// unit Jacobi scale of an LM run without jacobi_scaling: still zero on constant blocks
int unit_scale_masked(miba_handle h) {
  int rc = unit_scale(h);
  if (rc) return rc;
  return apply_const_mask(h);
}

// ConjugateGradientsSolver::Solve on S y = rhs with x0 = 0, matrix-free kernels.  Solution in d_px.
// Per iteration two launches: the tiles' side of the product (k_mf_matvec) and pcg_tail (miba_pcg.hpp: camera-side sum, vector
// updates, scalars, next direction); with several ranks the camera-side sum and its all-reduce stand between them.  The control
// block is device-resident and double-buffered (see miba_kernels.hpp).
// reduce: 0 = q is there already, 1 = phase A in the launch, 2 = phase A and the exchange between the ranks in the launch
template <int MODE>
void launch_tail(miba_handle h, const TailArgs& A, int reduce) {
  if (reduce == 2) hipLaunchKernelGGL((pcg_tail<MODE, 2>), dim3(h->tail_nb), dim3(kTailThreads), 0, h->st, A);
  else if (reduce == 1) hipLaunchKernelGGL((pcg_tail<MODE, 1>), dim3(h->tail_nb), dim3(kTailThreads), 0, h->st, A);
  else hipLaunchKernelGGL((pcg_tail<MODE, 0>), dim3(h->tail_nb), dim3(kTailThreads), 0, h->st, A);
}

int pcg_solve(miba_handle h, double q_tol, int min_it, int max_it, int check_every, int* n_it, int* status) {
  const size_t n = 9 * (size_t)h->n_cam;
  const int nb = h->tail_nb;
  double* part_nb = h->d_part_pcg;
  double* part_q = h->d_part_pcg + kPcgMaxBlocks;
  double* part_rho[2] = {h->d_part_pcg + 2 * kPcgMaxBlocks, h->d_part_pcg + 3 * kPcgMaxBlocks};
  classify_pcg_events(h, 1 << 30);   // pairs a solve that ended in an error has left behind are not kept waiting
  const bool multi = multi_rank(h);
  TailArgs A{};
  A.n_cam = h->n_cam;
  A.wave_rec = h->d_tail_rec; A.n_vw = h->tail_n_vw;
  A.slab = h->d_slab9; A.ct = h->d_ct; A.cm = h->d_cm; A.sc = h->d_sc; A.d2 = h->d_Dc2;
  A.diag_w = h->cfg.rank == 0 ? 1.0 : 0.0;
  A.p = h->d_pp; A.q = h->d_pq; A.b = h->d_rhs; A.Minv = h->d_Minv; A.x = h->d_px; A.r = h->d_pr; A.z = h->d_pz;
  A.ctl = h->d_ctl;
  A.n_pq = nb;
  // several ranks: phases B and C walk the cameras in the same order on every rank (the wave records follow the rank's own rows)
  A.uniform = multi ? 1 : 0;
  // ranks of one process under the peer transport: the exchange of q is part of the tail kernel (pcg_tail<MODE, 2>)
  const bool can_fuse = multi && h->grp && h->grp->mode == kCommPeer && !h->grp->xstage.empty() && !getenv("MIBA_TAIL_NO_FUSE") &&
                        std::all_of(h->grp->xstage.begin(), h->grp->xstage.end(), [](const double* q) { return q != nullptr; });
  if (can_fuse) {
    for (int q = 0; q < h->grp->n; ++q) { A.xstage[q] = h->grp->xstage[q]; A.xsig[q] = h->grp->psig[q]; }
    A.n_ranks = h->grp->n;
    A.rank = h->cfg.rank;
  }
  A.part_q = part_q; A.part_rho0 = part_rho[0]; A.part_rho1 = part_rho[1];
  A.q_tol = q_tol; A.min_it = min_it; A.max_it = max_it;
  A.bar = h->d_bar;
  A.bar_timeout = h->tail_timeout;
  A.dbg = h->d_tail_dbg;
  if (check_every < 1) check_every = 1;
  // camera-side sum (phase A of the tail as a launch of its own: this rank's slab rows) and all-reduce (several ranks)
  auto reduce_step = [&](bool on_x, int j, bool sample) -> int {
    {
      Timed t(h, MIBA_K_CAM_REDUCE, j, sample);
      TailArgs R = A;
      R.j = j;
      R.a_on_x = on_x ? 1 : 0;
      hipLaunchKernelGGL((pcg_tail<3, 1>), dim3(h->tail_nb), dim3(kTailThreads), 0, h->st, R);
    }
    HIPCHK(h, hipGetLastError());
    return allreduce(h, h->d_pq, n + (on_x ? 0 : (size_t)nb), ncclSum);
  };
  // The tail after phase A.  One launch with two grid barriers when the device is this solver's own; without barriers
  // (tail_split: MIBA_TAIL_SPLIT=1, or after a barrier of this handle has timed out once) phase B and phase C are launches of
  // their own, which any amount of foreign work on the device can delay but not deadlock.
  auto tail_rest = [&](auto mode_tag, int reduce) {
    constexpr int MODE = decltype(mode_tag)::value;
    if (reduce == 2) A.epoch = ++h->x_epoch;
    if (!h->tail_split || MODE == 1) { launch_tail<MODE>(h, A, reduce); return; }
    TailArgs B = A;
    B.split = 1;
    launch_tail<MODE>(h, B, 0);
    B.split = 2;
    launch_tail<MODE>(h, B, 0);
  };
  for (int attempt = 0;; ++attempt) {
    const bool split = h->tail_split != 0;
    const bool fused = can_fuse && !split;
    const bool own_reduce = (multi && !fused) || split;   // phase A as its own launch
    const int in_tail = own_reduce ? 0 : (fused ? 2 : 1);   // what the tail launch does before phase B
    // one rank (or the exchange inside the tail): the tail's own p.q partials; several ranks with an all-reduce: the camera
    // reduction's, behind q so that one all-reduce takes both
    A.part_pq = (multi && !fused) ? h->d_pq + n : h->d_part_pcg + 4 * kPcgMaxBlocks;
    h->pcg_launches = 1 + (own_reduce ? 1 : 0) + (split ? 2 : 1);
    h->pcg_exchange = !multi ? 0 : (fused ? 2 : 1);
    HIPCHK(h, hipMemsetAsync(h->d_bar, 0, kBarWords * sizeof(unsigned), h->st));
    hipLaunchKernelGGL(pcg_start, dim3(nb), dim3(256), 0, h->st, h->n_cam, h->d_rhs, h->d_Minv, h->d_px, h->d_pr, h->d_pz, h->d_ctl,
                       part_nb, part_rho[1]);
    hipLaunchKernelGGL(pcg_start2, dim3(1), dim3(256), 0, h->st, part_nb, nb, h->d_ctl);
    hipLaunchKernelGGL(pcg_chkdir, dim3(nb), dim3(256), 0, h->st, 1, n, h->d_pz, h->d_pp, h->d_ctl, part_rho[0], part_rho[1], nb, A.part_pq,
                       A.n_pq, part_q, q_tol, min_it, max_it);
    HIPCHK(h, hipGetLastError());
    for (int j = 1;; ++j) {
      Ctl* cur = h->d_ctl + ((j - 1) & 1);   // C_{j-1}: state after the j-1 iterations tested so far
      const int it = j - 1;
      // Host poll: the control block is copied out and marked by an event, iteration j is enqueued behind it, and only
      // then does the host wait for the copy: the device works on iteration j while the host looks at C_{j-1}, instead
      // of idling for a host round trip (34 us per poll at Final-13682).  If C_{j-1} says done, iteration j's kernels
      // return at once, like every launch enqueued past convergence.
      const bool poll = it >= 1 && (it % check_every == 0 || it >= max_it);
      if (poll) {
        HIPCHK(h, hipMemcpyAsync(h->h_ctl, cur, sizeof(Ctl), hipMemcpyDeviceToHost, h->st));
        HIPCHK(h, hipEventRecord(h->ev_poll, h->st));
      }
      const bool sample = (j - 1) % h->time_every == 0;
      const bool reset = (j % 10 == 0);  // residual_reset_period
      A.j = j;
      int rc = MIBA_OK;
      mf_matvec_tiles(h, h->d_pp, cur, j, sample);
      if (own_reduce && (rc = reduce_step(false, j, sample))) return rc;
      if (!reset) {
        Timed t(h, own_reduce ? -1 : MIBA_K_CAM_REDUCE, j, sample && !own_reduce);
        tail_rest(std::integral_constant<int, 0>(), in_tail);
      } else {
        tail_rest(std::integral_constant<int, 1>(), in_tail);
        HIPCHK(h, hipGetLastError());
        mf_matvec_tiles(h, h->d_px, cur, j, sample);   // q = S p is dead once x is updated: reuse it for S x
        if (own_reduce && (rc = reduce_step(true, j, false))) return rc;
        tail_rest(std::integral_constant<int, 2>(), in_tail);
      }
      HIPCHK(h, hipGetLastError());
      if (poll) {
        HIPCHK(h, hipEventSynchronize(h->ev_poll));
        if (h->h_ctl->done) break;
        if (it >= max_it + check_every) return fail(h, MIBA_ERR_INVALID, "PCG did not report termination");
      }
    }
    if (!(h->h_ctl->fail & 16)) break;
    // A grid barrier of the tail kernel was not reached in time: something else holds compute units of this device, so the
    // tail's grid was not resident as a whole.  From here on this handle uses the form without barriers; the solve starts over
    // (x0 = 0, the control block is rebuilt by pcg_start, which keeps the factorisation flags).
    if (attempt > 0 || split)
      return fail(h, MIBA_ERR_HIP, "the PCG tail kernel's grid barrier timed out (its grid was not resident as a whole)");
    HIPCHK(h, hipStreamSynchronize(h->st));
    classify_pcg_events(h, 0);
    h->tail_split = 1;
    if (h->cfg.log_level > 0) fprintf(stderr, "[miba] a grid barrier of the PCG tail timed out: this handle now runs the tail without barriers\n");
    const int32_t flags = h->h_ctl->fail & ~16;
    HIPCHK(h, hipMemcpyAsync(&h->d_ctl->fail, &flags, sizeof(int32_t), hipMemcpyHostToDevice, h->st));
    HIPCHK(h, hipStreamSynchronize(h->st));
  }
  if (A.dbg) {   // phase time stamps of the last tail launch that went through all phases: spread over the workgroups
    std::vector<long long> st(8 * (size_t)h->tail_nb);
    (void)hipMemcpy(st.data(), A.dbg, st.size() * 8, hipMemcpyDeviceToHost);
    long long t0 = st[0];
    for (int b = 0; b < h->tail_nb; ++b) t0 = std::min(t0, st[b * 8]);
    fprintf(stderr, "pcg_tail stamps (10 ns ticks, min..max over %d workgroups): ", h->tail_nb);
    const char* nm[6] = {"start", "A done", "barrier 1", "scalars", "B done", "barrier 2"};
    for (int n = 0; n < 6; ++n) {
      long long lo = 1LL << 60, hi = -(1LL << 60);
      for (int b = 0; b < h->tail_nb; ++b) { lo = std::min(lo, st[b * 8 + n] - t0); hi = std::max(hi, st[b * 8 + n] - t0); }
      fprintf(stderr, "%s %lld..%lld%s", nm[n], lo, hi, n < 5 ? ", " : "\n");
    }
  }
  *n_it = h->h_ctl->n_it;
  *status = h->h_ctl->status;
  classify_pcg_events(h, *n_it);
  return MIBA_OK;
}


// ---- exact reduced solve (non-iterative linear_solver_type) ---------------------------------
// Used when the options ask for a factorisation-based solver and the matrix-free kernels are active:
//   kExactDense   N = 9 n_cam (rounded up to 64) <= kDenseMaxN: dense S, blocked Cholesky (miba_dense.hpp)
//   kExactSparse  beyond that: S as 64 x 64 tiles where cameras share points plus fill (miba_sparse.hpp), if the pattern fits
//   kExactNone    otherwise, or with MIBA_DENSE=0 / MIBA_EXACT=pcg: PCG to eta_exact stands in
// MIBA_EXACT=dense|sparse|pcg overrides the choice by size (tests).
enum { kExactNone = 0, kExactDense = 1, kExactSparse = 2 };
int sparse_symbolic(miba_handle h);
// Device memory -> pageable host memory through two pinned staging buffers: the DMA of chunk i + 1 runs while the worker threads
// copy chunk i out.  (A plain hipMemcpy into pageable memory runs at about 5 GB/s here, the bus at ten times that; pinning
// the destination itself costs as much as the slow copy.)  Synchronous: returns when dst holds the data.
static hipError_t download_staged(void* dst, const void* src, size_t bytes, hipStream_t st) {
  constexpr size_t kStage = (size_t)16 << 20;
  if (bytes < 2 * kStage) {
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
  }
  void* stage[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
  hipError_t e = hipHostMalloc(&stage[0], kStage);
  if (e == hipSuccess) e = hipHostMalloc(&stage[1], kStage);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[0], hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
  const size_t n_chunk = (bytes + kStage - 1) / kStage;
  auto issue = [&](size_t i) {
    const size_t len = std::min(kStage, bytes - i * kStage);
    hipError_t r = hipMemcpyAsync(stage[i & 1], (const char*)src + i * kStage, len, hipMemcpyDeviceToHost, st);
    return r != hipSuccess ? r : hipEventRecord(ev[i & 1], st);
  };
  if (e == hipSuccess) e = issue(0);
  for (size_t i = 0; i < n_chunk && e == hipSuccess; ++i) {
    if (i + 1 < n_chunk) e = issue(i + 1);   // (its buffer was emptied by the host copy of chunk i - 1)
    if (e == hipSuccess) e = hipEventSynchronize(ev[i & 1]);
    if (e != hipSuccess) break;
    const size_t len = std::min(kStage, bytes - i * kStage);
    char* d = (char*)dst + i * kStage;
    const char* sp = (const char*)stage[i & 1];
    plan_parallel_for((int64_t)len, [&](int64_t a, int64_t b) { memcpy(d + a, sp + a, (size_t)(b - a)); }, (int64_t)1 << 20);
  }
  if (e != hipSuccess) (void)hipStreamSynchronize(st);
  if (ev[0]) (void)hipEventDestroy(ev[0]);
  if (ev[1]) (void)hipEventDestroy(ev[1]);
  if (stage[0]) (void)hipHostFree(stage[0]);
  if (stage[1]) (void)hipHostFree(stage[1]);
  return e;
}

int exact_kind(miba_handle h, const miba_options* o, int* kind) {
  *kind = kExactNone;
  if (o->linear_solver_type == MIBA_ITERATIVE_SCHUR) return MIBA_OK;
  const char* e = getenv("MIBA_DENSE");
  if (e && atoi(e) == 0) return MIBA_OK;
  const char* x = getenv("MIBA_EXACT");
  if (x && !strcmp(x, "pcg")) return MIBA_OK;
  // The tiles first, whatever the size (a small problem's Schur complement is dense or nearly so, and the tile kernels are still
  // the faster ones: trafalgar-257 2.3 -> 1.5 ms, dubrovnik-356 4.0 -> 2.5 ms per solve); the dense factorisation where the
  // tiles are not available (more than 15 ranks, not enough memory) and the matrix fits.  MIBA_EXACT=dense|sparse forces one.
  const size_t N = (9 * (size_t)h->n_cam + kNB - 1) / kNB * kNB;
  const bool fits_dense = N <= (size_t)kDenseMaxN;
  if (x && !strcmp(x, "dense")) {
    if (fits_dense) *kind = kExactDense;
    return MIBA_OK;
  }
  if (h->sparse_state == 0) {
    const int rc = sparse_symbolic(h);
    if (rc) return rc;
  }
  if (h->sparse_state <= 0 && fits_dense && !(x && !strcmp(x, "sparse"))) { *kind = kExactDense; return MIBA_OK; }
  if (h->sparse_state > 0) *kind = kExactSparse;
  return MIBA_OK;
}


// Pair plan of the explicit Schur complement: every (observation a, observation b) of one point with
// camera(a) > camera(b), or the same camera (both orders and a == b: duplicates square correctly), is one
// instance of block (camera(a), camera(b)); instances are sorted by block (ascending points inside a block) and cut into
// segments.  Built on the device (dense_plan_device: a radix sort of the instances, up to 2^30 of them, any number of
// cameras); on the host threads' counting sort where that does not apply (a point that sees a camera twice, no memory for
// the sort) up to 20,000 cameras and 2^28 instances.  Skipped (k_dense_pairs' atomics take over) beyond that or with
// MIBA_DENSE_PLAN=0; MIBA_DENSE_PLAN=host keeps the host plan (the reference for the test of the device plan).
constexpr int64_t kDensePlanMax = (int64_t)1 << 28;
constexpr int64_t kDensePlanMaxDevice = (int64_t)1 << 30;

static void dense_plan_segments(const std::vector<uint32_t>& bstart, const std::vector<uint64_t>& bkey, int64_t total, std::vector<int4>& segs) {
  const size_t nb = bstart.size();
  segs.reserve(nb + (size_t)(total / kDenseSeg));
  for (size_t k = 0; k < nb; ++k) {
    const int64_t f = bstart[k], n = (k + 1 < nb ? (int64_t)bstart[k + 1] : total) - f;
    const int ci = (int)(bkey[k] >> 24), cj = (int)(bkey[k] & 0xFFFFFFu);
    for (int64_t o = 0; o < n; o += kDenseSeg)
      segs.push_back(make_int4(ci, (int)(f + o), (int)std::min<int64_t>(kDenseSeg, n - o), cj | ((n > kDenseSeg ? 1 : 0) << 30)));
  }
}

// *done = true: d_dseg / d_dpts hold the plan.  *done = false with MIBA_OK: not applicable, the caller tries the host plan.
static int dense_plan_device(miba_handle h, bool* done) {
  *done = false;
  const Plan& P = h->plan_ref();
  const size_t np = (size_t)h->n_pt;
  if (np == 0 || h->n_slots == 0) return MIBA_OK;
  int64_t total = 0;
  std::vector<int32_t> long_pts;
  for (size_t q = 0; q < np; ++q) {
    const int64_t L = P.pt_count[h->pt_base + q];
    total += L * (L + 1) / 2;
    if (L > 64) long_pts.push_back((int32_t)q);
  }
  if (total <= 0 || total > kDensePlanMaxDevice) return MIBA_OK;
  const size_t n = (size_t)total;
  size_t free_b = 0, total_b = 0;
  HIPCHK(h, hipMemGetInfo(&free_b, &total_b));
  if ((double)n * 40.0 + (double)np * 16.0 > 0.5 * (double)free_b) return MIBA_OK;   // keys twice, values twice, sort scratch
  uint64_t *cnt = nullptr, *off = nullptr, *k0 = nullptr, *k1 = nullptr, *bkey = nullptr;
  uint32_t *v0 = nullptr, *pts = nullptr;
  int32_t *dup = nullptr, *d_long = nullptr;
  auto cleanup = [&]() {
    (void)hipFree(cnt); (void)hipFree(off); (void)hipFree(k0); (void)hipFree(k1); (void)hipFree(bkey); (void)hipFree(v0); (void)hipFree(dup);
    (void)hipFree(d_long); (void)hipFree(pts);
  };
  hipError_t e = hipMalloc(&cnt, np * 8);
  if (e == hipSuccess) e = hipMalloc(&off, np * 8);
  if (e == hipSuccess) e = hipMalloc(&k0, n * 8);
  if (e == hipSuccess) e = hipMalloc(&k1, n * 8);
  if (e == hipSuccess) e = hipMalloc(&v0, n * 4);
  if (e == hipSuccess) e = hipMalloc(&pts, n * 4);
  if (e == hipSuccess) e = hipMalloc(&dup, sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc(&d_long, (long_pts.size() + 1) * sizeof(int32_t));
  if (e != hipSuccess) { (void)hipGetLastError(); cleanup(); return MIBA_OK; }
  int32_t h_dup = 0;
  e = hipMemsetAsync(dup, 0, sizeof(int32_t), h->st);
  if (e == hipSuccess && !long_pts.empty()) e = hipMemcpyAsync(d_long, long_pts.data(), long_pts.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_inst_count, dim3(nblk(np, 256)), dim3(256), 0, h->st, (int32_t)np, (const int2*)h->d_pt_rng, (const int32_t*)h->d_ocam, cnt, dup);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = device_exclusive_scan(cnt, off, np, h->st);
  if (e == hipSuccess) e = hipMemcpy(&h_dup, dup, sizeof(int32_t), hipMemcpyDeviceToHost);
  if (e == hipSuccess && h_dup) { cleanup(); return MIBA_OK; }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_inst_fill, dim3(nblk(np, 256)), dim3(256), 0, h->st, (int32_t)np, (const int2*)h->d_pt_rng, (const int32_t*)h->d_ocam,
                       (const uint64_t*)off, k0, v0);
    if (!long_pts.empty())
      hipLaunchKernelGGL(k_inst_fill_long, dim3((unsigned)long_pts.size()), dim3(256), 0, h->st, (const int32_t*)d_long, (const int2*)h->d_pt_rng,
                         (const int32_t*)h->d_ocam, (const uint64_t*)off, k0, v0);
    e = hipGetLastError();
  }
  int end_bit = 25;
  while (end_bit < 48 && ((int64_t)1 << (end_bit - 24)) < (int64_t)h->n_cam) ++end_bit;
  if (e == hipSuccess) e = device_sort_pairs(k0, k1, v0, pts, n, end_bit, h->st);
  // block starts: flags and their ranks live in the first key buffer, which the sort has left free
  uint32_t* flag = (uint32_t*)k0;
  uint32_t* rank = flag + n;
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_inst_heads, dim3(nblk(n, 256)), dim3(256), 0, h->st, (int64_t)n, (const uint64_t*)k1, flag);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = device_exclusive_scan(flag, rank, n, h->st);
  uint32_t last[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpy(&last[0], rank + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(&last[1], flag + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost);
  const size_t nb = (size_t)last[0] + last[1];
  std::vector<uint32_t> bstart(nb);
  std::vector<uint64_t> bkeys(nb);
  if (e == hipSuccess) e = hipMalloc(&bkey, (nb ? nb : 1) * 8);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_inst_blocks, dim3(nblk(n, 256)), dim3(256), 0, h->st, (int64_t)n, (const uint64_t*)k1, (const uint32_t*)flag, (const uint32_t*)rank,
                       v0, bkey);   // (the unsorted values are not needed any more: v0 takes the block starts)
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->st);
  if (e == hipSuccess && nb) e = hipMemcpy(bstart.data(), v0, nb * sizeof(uint32_t), hipMemcpyDeviceToHost);
  if (e == hipSuccess && nb) e = hipMemcpy(bkeys.data(), bkey, nb * 8, hipMemcpyDeviceToHost);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    cleanup();
    return fail(h, MIBA_ERR_HIP, "assembly plan on the device failed: %s", hipGetErrorString(e));
  }
  std::vector<int4> segs;
  dense_plan_segments(bstart, bkeys, total, segs);
  h->allocs.push_back(pts);   // the sorted points stay: they are the plan
  h->d_dpts = (int32_t*)pts;
  pts = nullptr;
  cleanup();
  int rc = dev_alloc(h, &h->d_dseg, segs.size());
  if (rc) return rc;
  if (!segs.empty()) HIPCHK(h, hipMemcpy(h->d_dseg, segs.data(), segs.size() * sizeof(int4), hipMemcpyHostToDevice));
  h->n_dseg = (int32_t)segs.size();
  h->dense_planned = true;
  *done = true;
  if (getenv("MIBA_DEBUG_PLAN"))
    fprintf(stderr, "[miba plan] assembly plan on the device: %lld instances, %zu camera blocks, %zu segments\n", (long long)total, nb, segs.size());
  return MIBA_OK;
}

static int dense_plan_impl(miba_handle h) {
  h->dense_planned = false;
  h->dense_plan_tried = true;
  const char* pe = getenv("MIBA_DENSE_PLAN");
  if (pe && pe[0] == '0') return MIBA_OK;
  if (!(pe && !strcmp(pe, "host"))) {
    bool done = false;
    const int rc = dense_plan_device(h, &done);
    if (rc || done) return rc;
  }
  const Plan& P = h->plan_ref();
  const int32_t nc = h->n_cam;
  if (nc > 20000) return MIBA_OK;   // the block directory below is dense in camera pairs: 1.6 GB of host memory at this bound
  const size_t nkey = (size_t)nc * (nc + 1) / 2;
  // the host copy of the per-slot camera ids was released after the upload: read the shard's back
  std::vector<int32_t> ocam((size_t)(h->n_slots ? h->n_slots : 1));
  if (h->n_slots) HIPCHK(h, hipMemcpy(ocam.data(), h->d_ocam, (size_t)h->n_slots * sizeof(int32_t), hipMemcpyDeviceToHost));
  std::vector<int64_t> first(nkey + 1, 0);
  auto key_of = [](int32_t ci, int32_t cj) { return (size_t)ci * (ci + 1) / 2 + cj; };
  int64_t total = 0;
  for (int32_t q = 0; q < h->n_pt; ++q) {
    const int32_t* c = ocam.data() + (P.pt_first[h->pt_base + q] - h->slot_base);
    const int32_t L = P.pt_count[h->pt_base + q];
    for (int32_t a = 0; a < L; ++a)
      for (int32_t b = 0; b < L; ++b)
        if (c[a] >= c[b]) { first[key_of(c[a], c[b]) + 1]++; ++total; }
    if (total > kDensePlanMax) return MIBA_OK;
  }
  // first[k + 1] = start of block k while the instances are filled in (it is block k's cursor, and ends at the block's end, which
  // is block k + 1's start): afterwards first[] is the usual directory, without a second array of this size
  {
    int64_t run = 0;
    for (size_t k = 0; k < nkey; ++k) { const int64_t c = first[k + 1]; first[k + 1] = run; run += c; }
  }
  std::vector<int32_t> pts((size_t)(total ? total : 1));
  for (int32_t q = 0; q < h->n_pt; ++q) {
    const int32_t* c = ocam.data() + (P.pt_first[h->pt_base + q] - h->slot_base);
    const int32_t L = P.pt_count[h->pt_base + q];
    for (int32_t a = 0; a < L; ++a)
      for (int32_t b = 0; b < L; ++b)
        if (c[a] >= c[b]) pts[(size_t)first[key_of(c[a], c[b]) + 1]++] = q;
  }
  std::vector<int4> segs;
  for (int32_t ci = 0; ci < nc; ++ci)
    for (int32_t cj = 0; cj <= ci; ++cj) {
      const size_t k = key_of(ci, cj);
      const int64_t n = first[k + 1] - first[k];
      for (int64_t o = 0; o < n; o += kDenseSeg)
        segs.push_back(make_int4(ci, (int)(first[k] + o), (int)std::min<int64_t>(kDenseSeg, n - o), cj | ((n > kDenseSeg ? 1 : 0) << 30)));
    }
  int rc = dev_alloc(h, &h->d_dseg, segs.size());
  if (rc) return rc;
  rc = dev_alloc(h, &h->d_dpts, pts.size());
  if (rc) return rc;
  if (!segs.empty()) HIPCHK(h, hipMemcpy(h->d_dseg, segs.data(), segs.size() * sizeof(int4), hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_dpts, pts.data(), pts.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  h->n_dseg = (int32_t)segs.size();
  h->dense_planned = true;
  return MIBA_OK;
}

int dense_plan(miba_handle h) {
  const int rc = dense_plan_impl(h);
  // Ranks of one process that share a device (tests): the plan frees its scratch, and hipFree waits for everything the device
  // is running, a faster rank's all-reduce kernel included, which in turn waits for this rank.  Nobody goes on to the assembly
  // before every rank is done here.  (One rank per device, the production layout, has nothing to wait for.)
  if (!rc && h->grp && h->grp->n > 1 && h->grp->barrier()) return fail(h, MIBA_ERR_RCCL, "a rank of the group left the assembly plan (see its error)");
  return rc;
}

int ensure_dense(miba_handle h) {
  if (h->d_S) return MIBA_OK;
  const size_t N = (9 * (size_t)h->n_cam + kNB - 1) / kNB * kNB;
  int rc = dev_alloc(h, &h->d_S, (N + kNB) * N);   // 64 extra rows: row N carries the right-hand side
  if (rc) return rc;
  rc = dev_alloc(h, &h->d_Linv, (N / kNB) * (size_t)kLdiag);
  if (rc) return rc;
  rc = dev_alloc(h, &h->d_dflag, 1);
  if (rc) return rc;
  h->dense_N = (int)N;
  return h->dense_plan_tried ? MIBA_OK : dense_plan(h);
}

// S of the current Schur init into the storage D names (dense lower triangle, or tiles): cleared, point pairs, all-reduce over
// the first n_reduce doubles, block diagonal
int reduced_assemble(miba_handle h, const SDst D, size_t n_clear, size_t n_reduce) {
  HIPCHK(h, hipMemsetAsync(D.S, 0, n_clear * sizeof(double), h->st));
  const bool loss = h->loss.type != kLossNone;   // the planned instance lists name no slots, the loss weights are per slot
  if (h->dense_planned && !loss) {
    if (h->n_dseg)
      hipLaunchKernelGGL(k_dense_blocks, dim3(h->xcd_remap ? (h->n_dseg + 7) / 8 * 8 : h->n_dseg), dim3(128), 0, h->st, h->n_dseg,
                         h->d_dseg, h->d_dpts, h->d_ct, h->d_cm, h->d_sc, h->d_ptrec, D, h->xcd_remap);
  } else if (h->n_tiles) {
#define MIBA_PAIRS_ARGS dim3((unsigned)h->n_tiles), dim3(kChunk), 0, h->st, h->d_tiles, h->d_idx, h->d_wl, h->d_tile_cams, h->d_ct, h->d_cm, \
                        h->d_sc, h->d_ptrec, (const double*)h->d_omega, D
#define MIBA_GIANT_ARGS dim3((unsigned)h->n_giant), dim3(kChunk), 0, h->st, h->d_tiles, h->d_giant, h->d_grps, h->d_idx, h->d_ocam, h->d_ct, \
                        h->d_cm, h->d_sc, h->d_ptrec, (const double*)h->d_omega, D
    if (loss) {
      hipLaunchKernelGGL(k_dense_pairs<true>, MIBA_PAIRS_ARGS);
      if (h->n_giant) hipLaunchKernelGGL(k_dense_pairs_giant<true>, MIBA_GIANT_ARGS);
    } else {
      hipLaunchKernelGGL(k_dense_pairs<false>, MIBA_PAIRS_ARGS);
      if (h->n_giant) hipLaunchKernelGGL(k_dense_pairs_giant<false>, MIBA_GIANT_ARGS);
    }
#undef MIBA_PAIRS_ARGS
#undef MIBA_GIANT_ARGS
  }
  HIPCHK(h, hipGetLastError());
  const int rc = allreduce(h, D.S, n_reduce, ncclSum);
  if (rc) return rc;
  hipLaunchKernelGGL(k_dense_diag, dim3(nblk(81 * (size_t)h->n_cam, 256)), dim3(256), 0, h->st, h->n_cam, h->d_camacc, h->d_Dc2, D);
  HIPCHK(h, hipGetLastError());
  return MIBA_OK;
}

// S (lower triangle, row-major [N][N]) of the current Schur init
int dense_assemble(miba_handle h) {
  int rc = ensure_dense(h);
  if (rc) return rc;
  const int N = h->dense_N;
  return reduced_assemble(h, SDst{h->d_S, N, nullptr, nullptr, 0}, ((size_t)N + kNB) * N, (size_t)N * N);
}

// blocked Cholesky of the lower triangle of A ([N + 64][N], N a multiple of 64, rows N.. zero) and the solve
// A x = b; x receives the first n_out entries.  *status: 0 ok, 2 a pivot was not positive (A is not SPD).
// launches only: no allocation, no host synchronisation (shared by the static entry point)
int dense_factor_enqueue(miba_handle h, double* A, int N, double* Ldiag, const double* b, int n_b, double* x, int n_out,
                         int32_t* d_flag) {
  HIPCHK(h, hipMemsetAsync(d_flag, 0, sizeof(int32_t), h->st));
  hipLaunchKernelGGL(k_dense_rhs, dim3(nblk(N, 256)), dim3(256), 0, h->st, n_b, N, b, A);
  const int nb = N / kNB;
  const int NR = N + kNB;
  for (int k = 0; k < nb; ++k) {
    const int k0 = k * kNB;
    const int below = nb - 1 - k;                        // 64-column tiles to the right
    hipLaunchKernelGGL(k_chol_panel, dim3(below + 1), dim3(kNB), 0, h->st, A, N, NR, k0, Ldiag + (size_t)k * kLdiag, (int*)d_flag);
    if (below > 0) hipLaunchKernelGGL(k_chol_update, dim3(below, below + 1), dim3(256), 0, h->st, A, N, k0);
  }
  for (int k = nb - 1; k >= 0; --k) {
    const int k0 = k * kNB;
    hipLaunchKernelGGL(k_chol_bwd, dim3(nblk((size_t)k0, 256)), dim3(256), 0, h->st, A, N, k0, (const double*)(Ldiag + (size_t)k * kLdiag), x,
                       n_out);
  }
  HIPCHK(h, hipGetLastError());
  return MIBA_OK;
}

int dense_factor_solve(miba_handle h, double* A, int N, double* Ldiag, const double* b, int n_b, double* x, int n_out,
                       int32_t* d_flag, int* status) {
  int rc0 = dense_factor_enqueue(h, A, N, Ldiag, b, n_b, x, n_out, d_flag);
  if (rc0) return rc0;
  int32_t flag = 0;
  HIPCHK(h, hipMemcpyAsync(&flag, d_flag, sizeof(int32_t), hipMemcpyDeviceToHost, h->st));
  if (h->d_ctl) HIPCHK(h, hipMemcpyAsync(h->h_ctl, h->d_ctl, sizeof(Ctl), hipMemcpyDeviceToHost, h->st));   // Schur-init failure flag
  int rc = sync_stream(h);
  if (rc) return rc;
  *status = flag ? 2 : 0;
  return MIBA_OK;
}

// exact solution of S y = rhs into d_px
int dense_solve(miba_handle h, int* status) {
  int rc = dense_assemble(h);
  if (rc) return rc;
  return dense_factor_solve(h, h->d_S, h->dense_N, h->d_Linv, h->d_rhs, 9 * h->n_cam, h->d_px, 9 * h->n_cam, h->d_dflag, status);
}

// ---- tile-sparse exact solve ------------------------------------------------------------------
// Symbolic phase, once per problem, on the host: which 64 x 64 tiles of S (tile row = kTileCams consecutive cameras) hold
// an entry, a minimum-degree elimination order of the tile rows, the fill that order produces, and the tile lists the numeric
// kernels walk.  With several ranks every rank sees only its points: the tile graphs are united by a sum all-reduce of the
// adjacency bits, 13 bits to a double, one hexadecimal digit per bit (up to 15 ranks cannot carry).
int sparse_symbolic(miba_handle h) {
  h->sparse_state = -1;
  const Plan& P = h->plan_ref();
  const int nT = (h->n_cam + kTileCams - 1) / kTileCams;
  const size_t nW = ((size_t)nT + 63) / 64;
  if (nT < 1 || nT > 32768 || h->cfg.world_size > 15) return MIBA_OK;
  std::vector<uint64_t> adj((size_t)nT * nW, 0);
  // memory the tiles may take: half of what is free now (the dense assembly plan and the lists need room too); with several
  // ranks the smallest figure, so that every rank takes the same decision
  size_t free_b = 0, total_b = 0;
  HIPCHK(h, hipMemGetInfo(&free_b, &total_b));
  double free_all = (double)free_b;
  {
    std::vector<int32_t> ocam((size_t)(h->n_slots ? h->n_slots : 1));
    // (ranks of one process may share a device in tests: the staged download frees pinned memory, which waits for the whole
    // device, a faster rank's all-reduce kernel included; the plain copy waits for nothing but itself)
    if (h->n_slots && h->grp && h->grp->n > 1) HIPCHK(h, hipMemcpy(ocam.data(), h->d_ocam, (size_t)h->n_slots * sizeof(int32_t), hipMemcpyDeviceToHost));
    else if (h->n_slots) HIPCHK(h, download_staged(ocam.data(), h->d_ocam, (size_t)h->n_slots * sizeof(int32_t), h->st));
    std::vector<int32_t> g;
    std::vector<uint64_t> mask(nW);
    for (int32_t q = 0; q < h->n_pt; ++q) {
      const int32_t* c = ocam.data() + (P.pt_first[h->pt_base + q] - h->slot_base);
      const int32_t L = P.pt_count[h->pt_base + q];
      g.clear();
      for (int32_t a = 0; a < L; ++a) g.push_back(c[a] / kTileCams);
      std::sort(g.begin(), g.end());
      g.erase(std::unique(g.begin(), g.end()), g.end());
      if (g.size() <= 32) {
        for (size_t a = 0; a < g.size(); ++a)
          for (size_t b2 = 0; b2 < g.size(); ++b2) adj[(size_t)g[a] * nW + (g[b2] >> 6)] |= (uint64_t)1 << (g[b2] & 63);
      } else {
        std::fill(mask.begin(), mask.end(), 0);
        for (int32_t v : g) mask[v >> 6] |= (uint64_t)1 << (v & 63);
        for (int32_t v : g)
          for (size_t w = 0; w < nW; ++w) adj[(size_t)v * nW + w] |= mask[w];
      }
    }
  }
  if (multi_rank(h)) {
    const size_t nd = adj.size() * 5;
    std::vector<double> pk(nd);
    for (size_t i = 0; i < adj.size(); ++i)
      for (int k = 0; k < 5; ++k) {
        uint64_t v = 0;
        for (int j = 0; j < 13 && 13 * k + j < 64; ++j) v |= ((adj[i] >> (13 * k + j)) & 1) << (4 * j);
        pk[5 * i + k] = (double)v;
      }
    double* d_pk = nullptr;
    HIPCHK(h, hipMalloc(&d_pk, nd * sizeof(double)));
    hipError_t e = hipMemcpy(d_pk, pk.data(), nd * sizeof(double), hipMemcpyHostToDevice);
    int rc = (e == hipSuccess) ? allreduce(h, d_pk, nd, ncclSum) : fail(h, MIBA_ERR_HIP, "upload of the tile graph failed: %s", hipGetErrorString(e));
    if (!rc) {
      e = hipStreamSynchronize(h->st);
      if (e == hipSuccess) e = hipMemcpy(pk.data(), d_pk, nd * sizeof(double), hipMemcpyDeviceToHost);
      if (e != hipSuccess) rc = fail(h, MIBA_ERR_HIP, "download of the tile graph failed: %s", hipGetErrorString(e));
    }
    if (!rc) {
      const double neg = -free_all;
      e = hipMemcpy(d_pk, &neg, sizeof(double), hipMemcpyHostToDevice);
      if (e == hipSuccess) rc = allreduce(h, d_pk, 1, ncclMax);
      if (!rc && e == hipSuccess) e = hipStreamSynchronize(h->st);
      double got = neg;
      if (!rc && e == hipSuccess) e = hipMemcpy(&got, d_pk, sizeof(double), hipMemcpyDeviceToHost);
      if (!rc && e != hipSuccess) rc = fail(h, MIBA_ERR_HIP, "exchange of the free memory failed: %s", hipGetErrorString(e));
      free_all = -got;
    }
    (void)hipFree(d_pk);
    if (rc) return rc;
    for (size_t i = 0; i < adj.size(); ++i) {
      uint64_t w = 0;
      for (int k = 0; k < 5; ++k) {
        const uint64_t v = (uint64_t)pk[5 * i + k];
        for (int j = 0; j < 13 && 13 * k + j < 64; ++j)
          if ((v >> (4 * j)) & 15) w |= (uint64_t)1 << (13 * k + j);
      }
      adj[i] = w;
    }
  }
  for (int v = 0; v < nT; ++v) adj[(size_t)v * nW + (v >> 6)] &= ~((uint64_t)1 << (v & 63));

  const int64_t max_tiles = (int64_t)(free_all / 2 / (4096 * sizeof(double)));

  // Elimination order (order_tile_graph, miba_plan.hpp): the tile graph dissected into arcs and nested separator sets (groups) whose
  // columns are factored side by side level by level, minimum degree inside.  The factorisation is a chain of steps, each two launches
  // whose latency (about 32 us together) the work of a narrow column does not fill: the number of arcs trades the length of the arcs'
  // chains (tile rows / arcs) against the separators' (levels x rows of a set, about 8 tile rows per cut at Final-13682) and the work
  // a step holds.  The symbolic phase orders the graph for a few arc counts and keeps the cheapest under tile_order_cost's model (every
  // rank sees the same graph and takes the same decision); MIBA_SPARSE_ARCS=1..64 fixes the number instead (tests, measurements).
  TileOrder O;
  {
    std::vector<int> cand;
    if (const char* ea = getenv("MIBA_SPARSE_ARCS")) { const int v = atoi(ea); if (v >= 1 && v <= kMaxArcs) cand.push_back(v); }
    if (cand.empty()) {
      for (int v : {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64})
        if (v == 1 || nT >= 32 * v) cand.push_back(v);
    }
    const bool nested = !(getenv("MIBA_SPARSE_NESTED") && getenv("MIBA_SPARSE_NESTED")[0] == '0');
    // (the candidates are ordered side by side on host threads: at 7,143 tile rows one ordering takes 40 ms)
    struct Cand { std::vector<uint64_t> adj; TileOrder O; bool ok = false; double cost = 0.0; };
    std::vector<Cand> cs(cand.size());
    for (size_t i = 0; i < cand.size(); ++i) {
      if (i + 1 == cand.size()) cs[i].adj = std::move(adj);
      else cs[i].adj = adj;
    }
    auto order_one = [&](size_t i) {
      Cand& c = cs[i];
      try {   // (a host thread: nothing may leave it)
        c.ok = order_tile_graph(c.adj, nT, cand[i], max_tiles, h->cfg.log_level, c.O, nested);
        if (c.ok) c.cost = tile_order_cost(c.O);
      } catch (...) { c.ok = false; }
    };
    {
      std::vector<std::thread> th;
      for (size_t i = 1; i < cand.size(); ++i) th.emplace_back(order_one, i);
      order_one(0);
      for (auto& t : th) t.join();
    }
    double best = 0.0;
    bool have = false;
    for (size_t i = 0; i < cand.size(); ++i) {
      Cand& c = cs[i];
      if (!c.ok) continue;
      if (c.O.n_arcs != cand[i] && cand[i] != 1 && cand.size() > 1) continue;   // not dissected that finely: the plain order is a candidate of its own
      if (h->cfg.log_level > 0 || getenv("MIBA_SPARSE_DEBUG")) {
        int levels = 0;
        for (int32_t l2 : c.O.group_level) levels = std::max(levels, (int)l2);
        fprintf(stderr, "[miba] tile order with %d arc(s): %lld tiles below the diagonal, %d separator rows in %d set(s) on %d level(s), modelled %.2f ms\n",
                c.O.n_arcs, (long long)c.O.n_off, nT - c.O.arc_start[c.O.n_arcs], (int)c.O.group_level.size() - c.O.n_arcs, levels, c.cost * 1e-3);
      }
      if (!have || c.cost < best) {
        have = true; best = c.cost;
        std::swap(O, c.O);
      }
    }
    if (!have) return MIBA_OK;   // does not fit: sparse_state stays -1
  }
  const int n_arcs = O.n_arcs;
  const std::vector<int32_t>&pos = O.pos, &grp = O.grp, &arc_start = O.arc_start;
  const std::vector<std::vector<int32_t>>& col_nodes = O.col_nodes;
  const int64_t n_off = O.n_off;
  const int sep0 = arc_start[n_arcs];          // first separator column (nT: none)
  const int nS = nT - sep0;
  const int nG = (int)O.group_level.size();
  std::vector<int32_t> group_of(nT), group_end(nT);
  for (int g = 0; g < nG; ++g)
    for (int k = O.group_start[g]; k < O.group_start[g + 1]; ++k) { group_of[k] = g; group_end[k] = O.group_start[g + 1]; }

  // slots: the diagonal tile of tile row p is slot p, the tiles below the diagonal follow column by column, then the
  // right-hand-side tile of every column
  const int64_t n_mat = nT + n_off;
  std::vector<int32_t> map((size_t)(nT + 1) * nT, -1);
  std::vector<int32_t> col_rows, col_slots;
  std::vector<int64_t> col_ptr(nT + 1, 0);
  std::vector<std::vector<std::pair<int32_t, int32_t>>> row_list(nT);
  col_rows.reserve((size_t)(n_mat + nT)); col_slots.reserve((size_t)(n_mat + nT));
  int64_t next = nT;
  for (int k = 0; k < nT; ++k) {
    col_ptr[k] = (int64_t)col_rows.size();
    std::vector<int32_t> rows;
    for (int32_t u : col_nodes[k]) rows.push_back(pos[u]);
    std::sort(rows.begin(), rows.end());
    col_rows.push_back(k); col_slots.push_back(k);
    map[(size_t)k * nT + k] = k;
    for (int32_t r : rows) {
      const int32_t slot = (int32_t)next++;
      map[(size_t)r * nT + k] = slot;
      col_rows.push_back(r); col_slots.push_back(slot);
      row_list[r].push_back({k, slot});
    }
    col_rows.push_back(nT); col_slots.push_back((int32_t)(n_mat + k));
    map[(size_t)nT * nT + k] = (int32_t)(n_mat + k);
  }
  col_ptr[nT] = (int64_t)col_rows.size();
  // private tiles of the groups: every (row, column) pair of a group's column whose column lies outside the group (in a set of a
  // higher level: the groups of one level run side by side and may meet there).  They are added to the matrix, per destination in
  // group order, in front of the first step of the next level (merge lists per level).
  const bool split = nG > 1;
  std::vector<std::unordered_map<int64_t, int32_t>> priv(nG);
  std::vector<int32_t> merge_ptr, merge_src, merge_dst;
  std::vector<std::array<int32_t, 3>> merge_seg;   // producing level, first destination, destinations
  int64_t n_priv = 0;
  if (split) {
    int max_level = 0;
    for (int32_t l2 : O.group_level) max_level = std::max(max_level, (int)l2);
    for (int l2 = 0; l2 < max_level; ++l2) {
      std::vector<std::pair<int32_t, int32_t>> contrib;   // (destination slot, private slot)
      for (int g = 0; g < nG; ++g) {
        if (O.group_level[g] != l2) continue;
        const int gend = O.group_start[g + 1];
        for (int k = O.group_start[g]; k < gend; ++k) {
          const int32_t* rows = col_rows.data() + col_ptr[k] + 1;   // below the diagonal, ascending, the right-hand-side row (nT) last
          const int m = (int)(col_ptr[k + 1] - col_ptr[k]) - 1;
          int j0 = 0;
          while (j0 < m && rows[j0] < gend) ++j0;
          for (int J = j0; J < m; ++J) {
            if (rows[J] >= nT) break;
            for (int I = J; I < m; ++I) {
              const int64_t key = (int64_t)rows[I] * (nT + 1) + rows[J];
              if (priv[g].find(key) == priv[g].end()) {
                const int32_t slot = (int32_t)(n_mat + nT + n_priv++);
                priv[g].emplace(key, slot);
                contrib.push_back({map[(size_t)rows[I] * nT + rows[J]], slot});
              }
            }
          }
        }
      }
      if (contrib.empty()) continue;
      std::stable_sort(contrib.begin(), contrib.end(), [](const std::pair<int32_t, int32_t>& x, const std::pair<int32_t, int32_t>& y) { return x.first < y.first; });
      const int32_t d0 = (int32_t)merge_dst.size();
      for (size_t i = 0; i < contrib.size(); ++i) {
        if (i == 0 || contrib[i].first != contrib[i - 1].first) { merge_ptr.push_back((int32_t)merge_src.size()); merge_dst.push_back(contrib[i].first); }
        merge_src.push_back(contrib[i].second);
      }
      merge_ptr.push_back((int32_t)merge_src.size());   // (every level's list has its own end)
      merge_seg.push_back({l2, d0, (int32_t)merge_dst.size() - d0});
      merge_dst.push_back(-1);                          // (keeps merge_ptr and merge_dst index for index)
    }
    if (2 * (int64_t)nT + n_off + n_priv > max_tiles) return MIBA_OK;   // does not fit
  }
  // The workgroup lists of every launch.  Forward: the groups of a level in lock-step (tile_order_steps: step st of a level takes column
  // group_start[g] + st of every group g of the level that has one), level after level.  Backward: the levels from the highest down,
  // row group_start[g + 1] - 1 - st of every group (the rows of a group touch only columns of the group and of lower levels, and two
  // groups of a level share no column).
  std::vector<miba_handle_s::SpStep> fwd, bwd;
  std::vector<std::vector<int32_t>> fwd_cols, bwd_rows;
  std::vector<uint8_t> first_of_level;
  tile_order_steps(O, fwd_cols, first_of_level);
  std::vector<miba_handle_s::SpMerge> merges;
  for (size_t st = 0; st < fwd_cols.size(); ++st) {
    if (!first_of_level[st]) continue;
    const int lv = O.group_level[group_of[fwd_cols[st][0]]];
    // everything the levels below have left (a level without private tiles has no entry)
    for (const auto& sg : merge_seg)
      if (sg[0] < lv && (st == 0 || sg[0] >= O.group_level[group_of[fwd_cols[st - 1][0]]])) merges.push_back({st, sg[1], sg[2]});
  }
  {
    int max_level = 0;
    for (int32_t l2 : O.group_level) max_level = std::max(max_level, (int)l2);
    for (int l2 = max_level; l2 >= 0; --l2) {
      int longest = 0;
      for (int g = 0; g < nG; ++g)
        if (O.group_level[g] == l2) longest = std::max(longest, (int)(O.group_start[g + 1] - O.group_start[g]));
      for (int st = 0; st < longest; ++st) {
        std::vector<int32_t> ks;
        for (int g = 0; g < nG; ++g)
          if (O.group_level[g] == l2 && O.group_start[g + 1] - 1 - st >= O.group_start[g]) ks.push_back(O.group_start[g + 1] - 1 - st);
        bwd_rows.push_back(ks);
      }
    }
  }
  auto chain_start = [&](int k) { return (int)O.group_start[group_of[k]]; };   // first row of the chain of backward steps row k belongs to
  int64_t n_wg = 0;
  for (auto& ks : fwd_cols)
    for (int32_t k : ks) {
      const int64_t m = col_ptr[k + 1] - col_ptr[k] - 1;
      n_wg += m + m * (m - 1) / 2;   // (the right-hand-side row is never a column: m - 1 columns, pairs I >= J)
    }
  n_wg += n_off + 2 * (int64_t)nT;
  if ((double)n_wg * sizeof(int4) > std::min(free_all / 8, 4e9)) return MIBA_OK;   // the lists (built on the host first) do not fit: PCG stands in
  std::vector<int4> wg;
  wg.reserve((size_t)n_wg);
  // pre[c]: the diagonal tile of column c is factored by the update launch of the step before (its workgroup for (c, c), listed
  // first, has the tile's last update: column c - 1 of the same chain, one step earlier, holds row c) and k_tchol_col finds W in memory
  std::vector<int32_t> step_of(nT, -1);
  for (size_t st = 0; st < fwd_cols.size(); ++st)
    for (int32_t k : fwd_cols[st]) step_of[k] = (int32_t)st;
  const bool ahead_on = !(getenv("MIBA_SPARSE_AHEAD") && getenv("MIBA_SPARSE_AHEAD")[0] == '0');
  std::vector<uint8_t> pre(nT + 1, 0);
  for (size_t st = 0; st < fwd_cols.size(); ++st) {
    const std::vector<int32_t>& ks = fwd_cols[st];
    miba_handle_s::SpStep S{(int64_t)wg.size(), 0, 0, 0};
    for (int32_t k : ks) {
      const int64_t c0 = col_ptr[k];
      const int m = (int)(col_ptr[k + 1] - c0) - 1;   // tiles below the diagonal, the right-hand-side tile among them
      for (int b = 0; b < m; ++b) wg.push_back(int4{pre[k] ? -1 : k, col_slots[c0 + 1 + b], k, b});
    }
    S.n_col = (int32_t)(wg.size() - S.col0);
    S.upd0 = (int64_t)wg.size();
    // the updates that complete the next step's diagonal tiles first: they have the longest way to go
    for (int32_t k : ks) {
      const int64_t c0 = col_ptr[k];
      const int m = (int)(col_ptr[k + 1] - c0) - 1;
      const int32_t* rows = col_rows.data() + c0 + 1;
      const int32_t* slots = col_slots.data() + c0 + 1;
      const bool to_private = split && k + 1 >= group_end[k];
      if (ahead_on && m > 0 && rows[0] == k + 1 && k + 1 < nT && step_of[k + 1] == (int32_t)st + 1 && !to_private) {
        pre[k + 1] = 1;
        wg.push_back(int4{slots[0], slots[0], map[(size_t)(k + 1) * nT + (k + 1)], k + 2});
      }
    }
    // the other diagonal tiles of the next step were complete before this launch (no column of this step touches them): workgroups
    // that only factor them.  Not the first columns of a level: the private tiles of the levels below are merged into them behind
    // this launch.
    if (ahead_on && st + 1 < fwd_cols.size() && !first_of_level[st + 1])
      for (int32_t c : fwd_cols[st + 1])
        if (!pre[c]) {
          pre[c] = 1;
          wg.push_back(int4{-1, -1, c, c + 1});
        }
    for (int32_t k : ks) {
      const int64_t c0 = col_ptr[k];
      const int m = (int)(col_ptr[k + 1] - c0) - 1;
      const int32_t* rows = col_rows.data() + c0 + 1;
      const int32_t* slots = col_slots.data() + c0 + 1;
      const int gend = group_end[k];
      const auto& pm = priv[group_of[k]];
      for (int J = 0; J < m; ++J) {
        if (rows[J] >= nT) break;
        for (int I = J; I < m; ++I) {
          if (I == 0 && J == 0 && pre[k + 1] && rows[0] == k + 1 && k + 1 < gend) continue;   // listed above
          int32_t d;
          if (split && rows[J] >= gend) {
            const auto it = pm.find((int64_t)rows[I] * (nT + 1) + rows[J]);
            d = it == pm.end() ? -1 : it->second;
          } else d = map[(size_t)rows[I] * nT + rows[J]];
          if (d < 0) return fail(h, MIBA_ERR_INVALID, "tile (%d, %d) is missing from the fill of column %d", rows[I], rows[J], k);
          wg.push_back(int4{slots[I], slots[J], d, 0});
        }
      }
    }
    S.n_upd = (int32_t)(wg.size() - S.upd0);
    fwd.push_back(S);
  }
  {
    std::vector<int32_t> lead_into(nT + 1, 0);   // row k: x_k was formed by the step of row k + 1
    for (auto& ks : bwd_rows) {
      miba_handle_s::SpStep S{(int64_t)wg.size(), 0, 0, 0};
      for (int32_t k : ks) {
        const auto& L = row_list[k];   // (column, slot), columns ascending
        const int have_x = lead_into[k] ? 1 : 0;
        const bool ahead = !L.empty() && L.back().first == k - 1 && k - 1 >= chain_start(k);
        if (L.empty()) wg.push_back(int4{k, -1, 0, have_x | 4});
        for (size_t i = 0; i < L.size(); ++i)
          wg.push_back(int4{k, L[i].second, L[i].first, have_x | ((ahead && i + 1 == L.size()) ? 2 : 0) | (i == 0 ? 4 : 0)});
        if (ahead) lead_into[k - 1] = 1;
      }
      S.n_col = (int32_t)(wg.size() - S.col0);
      bwd.push_back(S);
    }
  }

  int rc;
#define SP_ALLOC(ptr, n) if ((rc = dev_alloc(h, &ptr, (size_t)(n)))) return rc
  SP_ALLOC(h->d_sp_tiles, (size_t)(n_mat + nT + n_priv) * 4096);
  if (split) {
    SP_ALLOC(h->d_sp_merge_ptr, merge_ptr.size());
    SP_ALLOC(h->d_sp_merge_src, merge_src.size() ? merge_src.size() : 1);
    SP_ALLOC(h->d_sp_merge_dst, merge_dst.size() ? merge_dst.size() : 1);
  }
  SP_ALLOC(h->d_sp_Ldiag, (size_t)nT * 4096);   // W = L^-T of every diagonal tile
  SP_ALLOC(h->d_sp_y, (size_t)nT * 64);
  SP_ALLOC(h->d_sp_x, (size_t)nT * 64);
  SP_ALLOC(h->d_sp_map, map.size());
  SP_ALLOC(h->d_sp_pos, nT);
  SP_ALLOC(h->d_sp_grp, nT);
  SP_ALLOC(h->d_sp_wg, wg.size());
#undef SP_ALLOC
  if (!h->d_dflag && (rc = dev_alloc(h, &h->d_dflag, 1))) return rc;
  if ((rc = upload(h, h->d_sp_map, map.data(), map.size()))) return rc;
  if ((rc = upload(h, h->d_sp_pos, pos.data(), (size_t)nT))) return rc;
  if ((rc = upload(h, h->d_sp_grp, grp.data(), (size_t)nT))) return rc;
  if ((rc = upload(h, h->d_sp_wg, wg.data(), wg.size()))) return rc;
  h->sp_fwd.swap(fwd);
  h->sp_bwd.swap(bwd);
  h->sp_merges.swap(merges);
  h->sp_nT = nT;
  h->sp_n_mat = n_mat;
  h->sp_n_arcs = n_arcs;
  h->sp_arc.assign(arc_start.begin(), arc_start.end());
  h->sp_n_priv = n_priv;
  h->sp_fill_tiles = n_off;
  if (split) {
    if ((rc = upload(h, h->d_sp_merge_ptr, merge_ptr.data(), merge_ptr.size()))) return rc;
    if (!merge_src.empty() && (rc = upload(h, h->d_sp_merge_src, merge_src.data(), merge_src.size()))) return rc;
    if (!merge_dst.empty() && (rc = upload(h, h->d_sp_merge_dst, merge_dst.data(), merge_dst.size()))) return rc;
  }
  if (h->cfg.log_level > 0 || getenv("MIBA_SPARSE_DEBUG"))
  {
    int64_t n_pairs = 0;
    for (const miba_handle_s::SpStep& S : h->sp_fwd) n_pairs += S.n_upd;
    fprintf(stderr, "[miba] tile factorisation: %d tile rows, %lld tiles below the diagonal after fill, %d arc(s), %d separator rows in %d set(s), "
            "%lld private tiles, %zu forward steps, %lld tile products\n", nT, (long long)n_off, n_arcs, nS, nG - n_arcs, (long long)n_priv, h->sp_fwd.size(),
            (long long)n_pairs);
  }
  if (!h->dense_plan_tried && (rc = dense_plan(h))) return rc;
  h->sparse_state = 1;
  return MIBA_OK;
}

// exact solution of S y = rhs into d_px through the tiles
int sparse_solve(miba_handle h, int* status) {
  const int nT = h->sp_nT;
  const int64_t n_mat = h->sp_n_mat;
  double* T = h->d_sp_tiles;
  const SDst D{T, 0, h->d_sp_map, h->d_sp_pos, nT};
  int rc = reduced_assemble(h, D, (size_t)(n_mat + nT) * 4096, (size_t)n_mat * 4096);
  if (rc) return rc;
  const int nb64 = nblk((size_t)nT * 64, 256);
  hipLaunchKernelGGL(k_tchol_pad, dim3(nb64), dim3(256), 0, h->st, h->n_cam, nT, T, (const int32_t*)h->d_sp_grp);
  hipLaunchKernelGGL(k_tchol_rhs, dim3(nb64), dim3(256), 0, h->st, h->n_cam, nT, (const double*)h->d_rhs, T, (int)n_mat, (const int32_t*)h->d_sp_grp);
  HIPCHK(h, hipMemsetAsync(h->d_dflag, 0, sizeof(int32_t), h->st));
  // one stream, one chain of launches: a forward step is the column launch and the update launch of all its columns
  if (h->sp_n_priv) HIPCHK(h, hipMemsetAsync(T + (size_t)(n_mat + nT) * 4096, 0, (size_t)h->sp_n_priv * 4096 * sizeof(double), h->st));
  for (size_t i = 0; i < h->sp_fwd.size(); ++i) {
    for (const miba_handle_s::SpMerge& M : h->sp_merges)
      if (M.at == i && M.n > 0)
        hipLaunchKernelGGL(k_tchol_merge, dim3(M.n), dim3(256), 0, h->st, T, (const int32_t*)(h->d_sp_merge_ptr + M.ptr0), (const int32_t*)h->d_sp_merge_src,
                           (const int32_t*)(h->d_sp_merge_dst + M.ptr0));
    const miba_handle_s::SpStep& S = h->sp_fwd[i];
    hipLaunchKernelGGL(k_tchol_col, dim3(S.n_col), dim3(256), 0, h->st, T, (const int4*)(h->d_sp_wg + S.col0), h->d_sp_Ldiag, (int*)h->d_dflag);
    if (S.n_upd) hipLaunchKernelGGL(k_tchol_update, dim3(S.n_upd), dim3(256), 0, h->st, T, (const int4*)(h->d_sp_wg + S.upd0), h->d_sp_Ldiag, (int*)h->d_dflag);
  }
  hipLaunchKernelGGL(k_tchol_gety, dim3(nb64), dim3(256), 0, h->st, nT, (const double*)T, (int)n_mat, h->d_sp_y);
  for (const miba_handle_s::SpStep& S : h->sp_bwd)
    hipLaunchKernelGGL(k_tchol_bwd, dim3(S.n_col), dim3(kNB), 0, h->st, (const double*)T, (const int4*)(h->d_sp_wg + S.col0), (const double*)h->d_sp_Ldiag,
                       h->d_sp_y, h->d_sp_x);
  hipLaunchKernelGGL(k_tchol_unperm, dim3(nblk(9 * (size_t)h->n_cam, 256)), dim3(256), 0, h->st, h->n_cam, (const double*)h->d_sp_x,
                     (const int32_t*)h->d_sp_pos, h->d_px);
  HIPCHK(h, hipGetLastError());
  int32_t flag = 0;
  HIPCHK(h, hipMemcpyAsync(&flag, h->d_dflag, sizeof(int32_t), hipMemcpyDeviceToHost, h->st));
  if (h->d_ctl) HIPCHK(h, hipMemcpyAsync(h->h_ctl, h->d_ctl, sizeof(Ctl), hipMemcpyDeviceToHost, h->st));   // Schur-init failure flag
  rc = sync_stream(h);
  if (rc) return rc;
  *status = flag ? 2 : 0;
  return MIBA_OK;
}

// the linear solve of one LM iteration: exact factorisation where it applies, PCG otherwise
int linear_solve(miba_handle h, const miba_options* o, int* n_it, int* status) {
  int kind = kExactNone;
  int rc = exact_kind(h, o, &kind);
  if (rc) return rc;
  if (kind == kExactDense) {
    *n_it = 1;
    return dense_solve(h, status);
  }
  if (kind == kExactSparse) {
    *n_it = 1;
    return sparse_solve(h, status);
  }
  const double q_tol = (o->linear_solver_type == MIBA_ITERATIVE_SCHUR) ? o->eta : o->eta_exact;
  return pcg_solve(h, q_tol, o->min_linear_solver_iterations, o->max_linear_solver_iterations, o->pcg_check_every, n_it, status);
}

// back-substitution, candidate state, model cost change, step / candidate norms
// reduce = false: the caller all-reduces S_MCC, S_DP2, S_XP2 itself (together with the candidate cost)
// with_cost: the caller wants the candidate's cost in S_COST next (the LM loop).  Where the camera-major kernel runs it takes the cost
// along (k_mf_backsub_cm: the candidate cameras and their tables are formed first, they depend on the PCG's solution alone) and
// h->cand_cost_done tells the caller that eval_cost is not needed; MIBA_K1_FOLD=0 keeps the two kernels apart.
int backsub_and_candidate(miba_handle h, bool reduce = true, bool with_cost = false) {
  const bool cm = h->loss.type == kLossNone && h->k3_cm && h->d_oxy_cm && !h->k4_pm;
  const bool fold = with_cost && cm && h->k1_fold && h->n_tiles > 0;
  h->cand_cost_done = false;
  auto cam_candidate = [&]() -> int {
    const int nb = 120;
    hipLaunchKernelGGL(k_cam_update, dim3(nb), dim3(256), 0, h->st, h->n_cam, h->d_cam, h->d_sc, h->d_px, h->d_cam_active,
                       h->d_cam_cand, h->d_part_small);
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, h->st, h->d_part_small, (size_t)nb, 2, 0, 0, h->d_sum + S_DC2);
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, h->st, h->d_part_small, (size_t)nb, 2, 1, 0, h->d_sum + S_XC2);
    HIPCHK(h, hipGetLastError());
    return MIBA_OK;
  };
  int rc;
  if (fold) {
    if ((rc = cam_candidate())) return rc;
    if ((rc = launch_cam_tables(h, h->d_cam_cand, true))) return rc;
  }
  {
    if (h->n_tiles) {
      Timed t(h, MIBA_K_BACKSUB);
#define MIBA_BACKSUB_ARGS(n_blocks, list)                                                                                 \
  dim3((unsigned)(n_blocks)), dim3(kChunk), 0, h->st, h->d_tiles, list, h->d_grps, h->d_idx, h->d_ocam, h->d_wl, h->d_tile_cams, \
      h->d_oxy, h->d_ct, h->d_cm, h->d_sc, h->d_sp, h->d_ptrec, h->d_px, h->d_yp, h->d_Xc, h->d_part_tile, (const double*)h->d_omega
      if (h->loss.type != kLossNone) {
        hipLaunchKernelGGL((k_mf_backsub<false, true>), MIBA_BACKSUB_ARGS(h->n_tiles, (const int32_t*)nullptr));
        if (h->n_giant) hipLaunchKernelGGL((k_mf_backsub<true, true>), MIBA_BACKSUB_ARGS(h->n_giant, h->d_giant));
      } else {
        if (cm)
          hipLaunchKernelGGL(k_mf_backsub_cm, dim3((unsigned)h->n_tiles), dim3(kChunk), 0, h->st, h->d_tiles, h->d_idx, h->d_wl, h->d_oxy, h->d_ct, h->d_cm,
                             h->d_sc, h->d_sp, h->d_ptrec, h->d_px, with_cost ? (double*)nullptr : h->d_yp /*(the scaled point step: only miba_linear_step hands it out)*/,
                             h->d_Xc, h->d_part_tile,
                             CmArgs{h->d_cmt, h->d_cms, h->d_cmi, h->d_cmslot, h->d_cam64, nullptr}, (const double2*)h->d_oxy_cm,
                             fold ? (const double*)h->d_ct_cand : (const double*)nullptr);
        else hipLaunchKernelGGL((k_mf_backsub<false, false>), MIBA_BACKSUB_ARGS(h->n_tiles, (const int32_t*)nullptr));
        if (h->n_giant) hipLaunchKernelGGL((k_mf_backsub<true, false>), MIBA_BACKSUB_ARGS(h->n_giant, h->d_giant));
      }
#undef MIBA_BACKSUB_ARGS
      // the giant tiles' candidate cost into the free column of their partial sums (behind their back-substitution, which clears it)
      if (fold && h->n_giant)
        hipLaunchKernelGGL(k_mf_cost<false>, dim3((unsigned)h->n_giant), dim3(kChunk), 0, h->st, (int)h->n_giant, h->d_tiles, h->d_giant, h->d_grps,
                           h->d_idx, h->d_ocam, h->d_wl, h->d_tile_cams, h->d_oxy, h->d_ct_cand, h->d_Xc, h->d_part_tile, h->loss, 4, 3);
    }
  }
  HIPCHK(h, hipGetLastError());
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, h->st, h->d_part_tile, (size_t)h->n_tiles, 4, 0, 0, h->d_sum + S_MCC);
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, h->st, h->d_part_tile, (size_t)h->n_tiles, 4, 1, 0, h->d_sum + S_DP2);
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, h->st, h->d_part_tile, (size_t)h->n_tiles, 4, 2, 0, h->d_sum + S_XP2);
  if (fold) {
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, h->st, h->d_part_tile, (size_t)h->n_tiles, 4, 3, 0, h->d_sum + S_COST);
    h->cand_cost_done = true;
  }
  HIPCHK(h, hipGetLastError());
  if (reduce) {
    rc = allreduce(h, h->d_sum + S_MCC, 3, ncclSum);  // S_MCC, S_DP2, S_XP2 are adjacent
    if (rc) return rc;
  }
  if (!fold && (rc = cam_candidate())) return rc;
  return MIBA_OK;
}

int state_norm2(miba_handle h, double* out) {
  const int nb = 120;
  hipLaunchKernelGGL(k_sumsq, dim3(nb), dim3(256), 0, h->st, 9 * (size_t)h->n_cam, h->d_cam, h->d_cam_active, 9, h->d_cfree, h->d_part_small);
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, h->st, h->d_part_small, (size_t)nb, 1, 0, 0, h->d_sum + S_XC2);
  hipLaunchKernelGGL(k_sumsq, dim3(nb), dim3(256), 0, h->st, 3 * (size_t)h->n_pt, h->d_X, (const uint8_t*)nullptr, 3, (const double*)nullptr, h->d_part_small + 128);
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, h->st, h->d_part_small + 128, (size_t)nb, 1, 0, 0, h->d_sum + S_XP2);
  HIPCHK(h, hipGetLastError());
  int rc = allreduce(h, h->d_sum + S_XP2, 1, ncclSum);
  if (rc) return rc;
  HIPCHK(h, hipMemcpyAsync(h->h_scal, h->d_sum, S_COUNT * sizeof(double), hipMemcpyDeviceToHost, h->st));
  rc = sync_stream(h);
  if (rc) return rc;
  *out = h->h_scal[S_XC2] + h->h_scal[S_XP2];
  return MIBA_OK;
}

// device -> pinned host memory by plain stores (few workgroups: the copy shares the device with the solver's kernels)
__global__ __launch_bounds__(256) void k_copy_out(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// options.async_callbacks: copy the current state (cameras, this shard's points) into a pinned snapshot buffer on the second
// stream, behind everything the solver stream has enqueued so far; the solver stream does not wait.  Publishes the buffer for
// iteration `iteration` (snap_buf_of, snap_seq).  changed = false: the state is the one of the iteration before, no copy.
int snapshot_enqueue(miba_handle h, int iteration, bool changed) {
  if (changed || h->snap_last < 0) {
    if (!h->st2) {
      HIPCHK(h, hipStreamCreateWithFlags(&h->st2, hipStreamNonBlocking));
      HIPCHK(h, hipEventCreateWithFlags(&h->ev_state, hipEventDisableTiming));
      for (int b = 0; b < 2; ++b) HIPCHK(h, hipEventCreateWithFlags(&h->ev_snap[b], hipEventDisableTiming));
    }
    const size_t nc = 9 * (size_t)std::max(1, h->n_cam), np = 3 * (size_t)std::max(1, h->n_pt);
    if (h->snap_cam_n < nc || h->snap_X_n < np) {
      for (int b = 0; b < 2; ++b) {
        if (h->snap_cam[b]) (void)hipHostFree(h->snap_cam[b]);
        if (h->snap_X[b]) (void)hipHostFree(h->snap_X[b]);
        h->snap_cam[b] = h->snap_X[b] = nullptr;
      }
      h->snap_cam_n = h->snap_X_n = 0;
      for (int b = 0; b < 2; ++b) {
        HIPCHK(h, hipHostMalloc((void**)&h->snap_cam[b], nc * sizeof(double)));
        HIPCHK(h, hipHostMalloc((void**)&h->snap_X[b], np * sizeof(double)));
      }
      h->snap_cam_n = nc; h->snap_X_n = np;
    }
    const int b = h->snap_last < 0 ? 0 : h->snap_last ^ 1;
    HIPCHK(h, hipEventRecord(h->ev_state, h->st));
    HIPCHK(h, hipStreamWaitEvent(h->st2, h->ev_state, 0));
    // by a kernel that stores into the pinned buffers, not by the copy engine: the solver stream's own small device-to-host
    // copies (the PCG control block the host polls) would queue behind a 107 MB transfer on the engine and stall the PCG
    hipLaunchKernelGGL(k_copy_out, dim3(8), dim3(256), 0, h->st2, h->d_cam, h->snap_cam[b], 9 * (size_t)h->n_cam);
    if (h->n_pt) hipLaunchKernelGGL(k_copy_out, dim3(64), dim3(256), 0, h->st2, h->d_X, h->snap_X[b], 3 * (size_t)h->n_pt);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev_snap[b], h->st2));
    h->snap_recorded[b] = true;
    h->snap_last = b;
  }
  h->snap_buf_of[iteration & 1] = h->snap_last;
  h->snap_seq.store(iteration, std::memory_order_release);
  return MIBA_OK;
}

// the candidate buffers are about to be overwritten: a snapshot copy that still reads one of them (it was the state two
// iterations ago) must have finished.  Stream-ordered, no host wait.
int snapshot_guard(miba_handle h) {
  for (int b = 0; b < 2; ++b)
    if (h->snap_recorded[b]) HIPCHK(h, hipStreamWaitEvent(h->st, h->ev_snap[b], 0));
  return MIBA_OK;
}

int read_scalars(miba_handle h) {
  HIPCHK(h, hipMemcpyAsync(h->h_scal, h->d_sum, S_COUNT * sizeof(double), hipMemcpyDeviceToHost, h->st));
  HIPCHK(h, hipMemcpyAsync(h->h_scal + S_COUNT, h->d_max, S_COUNT * sizeof(double), hipMemcpyDeviceToHost, h->st));
  HIPCHK(h, hipMemcpyAsync(h->h_ctl, h->d_ctl, sizeof(Ctl), hipMemcpyDeviceToHost, h->st));
  const bool peer = h->grp && h->grp->mode == kCommPeer && h->grp->n > 1;
  uint32_t* tmo = (uint32_t*)(h->h_scal + 2 * S_COUNT);   // last slot of the pinned scalar block
  if (peer) HIPCHK(h, hipMemcpyAsync(tmo, &h->grp->psig[h->cfg.rank]->timeout, sizeof(uint32_t), hipMemcpyDeviceToHost, h->st));
  int rc = sync_stream(h);
  if (rc) return rc;
  if (peer && *tmo) return fail(h, MIBA_ERR_RCCL, "peer all-reduce timed out: a rank of the group never arrived");
  return MIBA_OK;
}

// camera tables of the current state and, on first use, the Jacobi scale: nothing is stored per observation
int ensure_jacobian(miba_handle h, const miba_options* o) {
  int rc;
  if (!h->jac_valid || !h->scale_valid) {
    rc = launch_cam_tables(h, h->d_cam);
    if (rc) return rc;
  }
  if (!h->scale_valid) {
    rc = o->jacobi_scaling ? make_scale_mf(h, o) : unit_scale_masked(h);
    if (rc) return rc;
    h->scale_valid = true;
  }
  h->jac_valid = true;
  return MIBA_OK;
}

void print_progress(const miba_iteration& it, bool header) {
  if (header)
    printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius  ls_iter  iter_time  total_time\n");
  printf("%4d % 8.6e  % 3.2e   %3.2e  %3.2e  % 3.2e  %3.2e   %6d   %3.2e   %3.2e\n", it.iteration, it.cost, it.cost_change,
         it.gradient_max_norm, it.step_norm, it.relative_decrease, it.trust_region_radius, it.linear_solver_iterations,
         it.iteration_time_in_seconds, it.cumulative_time_in_seconds);
  fflush(stdout);
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

void miba_default_options(miba_options* o) {
  if (!o) return;
  memset(o, 0, sizeof(*o));
  o->max_num_iterations = 50;
  o->linear_solver_type = MIBA_ITERATIVE_SCHUR;
  o->jacobi_scaling = 1;
  o->max_linear_solver_iterations = 500;
  o->min_linear_solver_iterations = 0;
  o->max_num_consecutive_invalid_steps = 5;
  o->update_state_every_iteration = 0;
  o->minimizer_progress_to_stdout = 0;
  o->pcg_check_every = 4;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->eta = 1e-1;
  o->eta_exact = 1e-12;
  o->max_solver_time_in_seconds = 1e9;
  o->loss_type = MIBA_LOSS_NONE;
  o->async_callbacks = 0;
  o->loss_scale = 1.0;
}

const char* miba_last_error(miba_handle h) { return h ? h->err.c_str() : g_last_error.c_str(); }

int miba_comm_unique_id(void* out128) {
  if (!out128) return fail(nullptr, MIBA_ERR_INVALID, "null output");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  ncclUniqueId id;
  NCCLCHK(nullptr, ncclGetUniqueId(&id));
  memcpy(out128, &id, sizeof(id));
  return MIBA_OK;
}

static int create_impl(const miba_config* cfg, bool make_comm, miba_handle* out) {
  if (!cfg || !out) return fail(nullptr, MIBA_ERR_INVALID, "null argument");
  if (cfg->world_size < 1 || cfg->rank < 0 || cfg->rank >= cfg->world_size)
    return fail(nullptr, MIBA_ERR_INVALID, "bad rank/world_size %d/%d", cfg->rank, cfg->world_size);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, MIBA_ERR_NO_DEVICE, "no HIP device available (libmiba has no CPU fallback)");
  int dev = cfg->device_id;
  if (dev < 0) {
    if (hipGetDevice(&dev) != hipSuccess) return fail(nullptr, MIBA_ERR_NO_DEVICE, "hipGetDevice failed");
  }
  if (dev >= ndev) return fail(nullptr, MIBA_ERR_INVALID, "device %d out of range (%d devices)", dev, ndev);
  miba_handle h = new (std::nothrow) miba_handle_s();
  if (!h) return fail(nullptr, MIBA_ERR_NOMEM, "out of host memory");
  h->cfg = *cfg;
  h->cfg.comm_unique_id = nullptr;
  h->device = dev;
  hipError_t e = hipSetDevice(dev);
  // non-blocking: no implicit synchronisation with the null stream (every copy below is stream-ordered or follows an
  // explicit sync; with several ranks on one device a null-stream copy of one rank must not wait for a peer's kernels)
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_scal, (2 * S_COUNT + 1) * sizeof(double));
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_ctl, sizeof(Ctl));
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_poll, hipEventDisableTiming);
  if (e == hipSuccess) e = static_scratch_for(dev, &h->static_scratch);   // device `dev` is current here
  if (e == hipSuccess) e = hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, dev);
  if (const char* te = getenv("MIBA_TIME_EVERY")) { const int v = atoi(te); if (v >= 1) h->time_every = v; }
  // the PCG tail kernel meets at grid barriers: its grid may not exceed what the device holds at once
  if (e == hipSuccess) {
    int occ = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, pcg_tail<0, 1>, kTailThreads, 0);
    h->tail_occ = std::max(1, occ);
  }
  if (const char* te = getenv("MIBA_TAIL_SPLIT")) h->tail_split = te[0] == '1';
  if (const char* te = getenv("MIBA_XCD_REMAP")) h->xcd_remap = h->k3_remap = te[0] != '0';
  if (const char* te = getenv("MIBA_K4")) h->k4_pm = !strcmp(te, "pm");   // the point-major back-substitution on every tile (tests, A/B)
  if (const char* te = getenv("MIBA_K1_FOLD")) h->k1_fold = te[0] != '0';   // 0: the candidate's cost by its own kernel (tests, A/B)
  if (const char* te = getenv("MIBA_TAIL_TIMEOUT_TICKS")) { const long long v = atoll(te); if (v >= 1) h->tail_timeout = v; }
  if (e != hipSuccess) {
    int rc = fail(nullptr, MIBA_ERR_HIP, "device setup failed: %s", hipGetErrorString(e));
    miba_destroy(h);   // frees what has been created so far (stream, pinned blocks, event)
    return rc;
  }
  const char* force = getenv("MIBA_FORCE_RCCL");
  if (make_comm && (cfg->world_size > 1 || (force && force[0] == '1'))) {
    ncclUniqueId id;
    if (cfg->comm_unique_id) memcpy(&id, cfg->comm_unique_id, sizeof(id));
    else if (cfg->world_size == 1) {
      if (ncclGetUniqueId(&id) != ncclSuccess) { miba_destroy(h); return fail(nullptr, MIBA_ERR_RCCL, "ncclGetUniqueId failed"); }
    } else { miba_destroy(h); return fail(nullptr, MIBA_ERR_INVALID, "world_size > 1 needs comm_unique_id"); }
    ncclResult_t r = ncclCommInitRank(&h->comm, cfg->world_size, id, cfg->rank);
    if (r != ncclSuccess) {
      int rc = fail(nullptr, MIBA_ERR_RCCL, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
      h->comm = nullptr;
      miba_destroy(h);
      return rc;
    }
  }
  *out = h;
  return MIBA_OK;
}

int miba_create(const miba_config* cfg, miba_handle* out) { return create_impl(cfg, true, out); }

namespace {
// rank handle of a group: device state only, the communicator comes from the group
int create_rank(int device, int rank, int world, int log_level, miba_handle* out) {
  miba_config c{};
  c.device_id = device; c.rank = 0; c.world_size = 1; c.log_level = log_level; c.comm_unique_id = nullptr;
  int rc = create_impl(&c, false, out);   // the group makes the communicators
  if (rc) return rc;
  (*out)->cfg.rank = rank;
  (*out)->cfg.world_size = world;
  return MIBA_OK;
}

void destroy_group(RankGroup* G) {
  if (!G) return;
  for (size_t r = 0; r < G->sub.size(); ++r) {
    miba_handle s = G->sub[r];
    if (!s) continue;
    (void)hipSetDevice(s->device);
    if (s->st) (void)hipStreamSynchronize(s->st);
    if (r < G->hbuf.size() && G->hbuf[r]) (void)hipHostFree(G->hbuf[r]);
    if (G->n + r < G->hbuf.size() && G->hbuf[G->n + r]) (void)hipHostFree(G->hbuf[G->n + r]);
    if (r < G->pstage.size() && G->pstage[r]) (void)hipFree(G->pstage[r]);
    if (r < G->psig.size() && G->psig[r]) (void)hipFree(G->psig[r]);
    s->grp = nullptr;
    miba_destroy(s);   // also destroys the rank's RCCL communicator
  }
  delete G;
}
}  // namespace

int miba_create_multi(const int32_t* device_ids, int32_t n_devices, int32_t log_level, miba_handle* out) {
  if (!out || n_devices < 1 || n_devices > kMaxRanks) return fail(nullptr, MIBA_ERR_INVALID, "n_devices must be 1..%d", kMaxRanks);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, MIBA_ERR_NO_DEVICE, "no HIP device available (libmiba has no CPU fallback)");
  std::vector<int> dev(n_devices);
  bool repeated = false;
  for (int r = 0; r < n_devices; ++r) {
    dev[r] = device_ids ? device_ids[r] : r;
    if (dev[r] < 0 || dev[r] >= ndev) return fail(nullptr, MIBA_ERR_INVALID, "device %d out of range (%d devices)", dev[r], ndev);
    for (int q = 0; q < r; ++q) repeated |= dev[q] == dev[r];
  }
  miba_handle g = new (std::nothrow) miba_handle_s();
  RankGroup* G = new (std::nothrow) RankGroup();
  if (!g || !G) { delete g; delete G; return fail(nullptr, MIBA_ERR_NOMEM, "out of host memory"); }
  g->owned = G;
  g->cfg.device_id = dev[0]; g->cfg.rank = 0; g->cfg.world_size = n_devices; g->cfg.log_level = log_level;
  g->device = dev[0];
  G->n = n_devices;
  G->ctl_stop[0].store(0); G->ctl_stop[1].store(0);
  G->sub.assign(n_devices, nullptr);
  // transport: RCCL for distinct devices, the host-staged one when a device repeats (RCCL refuses that);
  // MIBA_ALLREDUCE=peer|host|rccl overrides
  G->mode = repeated ? kCommHost : kCommRccl;
  if (const char* e = getenv("MIBA_ALLREDUCE")) {
    if (!strcmp(e, "peer")) G->mode = kCommPeer;
    else if (!strcmp(e, "host")) G->mode = kCommHost;
    else if (!strcmp(e, "rccl")) G->mode = kCommRccl;
    else { destroy_group(G); delete g; return fail(nullptr, MIBA_ERR_INVALID, "MIBA_ALLREDUCE must be rccl, peer or host"); }
  }
  if (G->mode == kCommRccl && repeated) { destroy_group(G); delete g; return fail(nullptr, MIBA_ERR_INVALID, "RCCL cannot put two ranks on one device"); }
  {
    // The peer kernel of a rank spins until the other ranks' kernels have arrived.  Ranks on ONE device launch on streams that
    // share the process's four hardware queues: with more than three of them a rank's kernel can sit in a queue behind a
    // spinning one and never start, and every all-reduce would run into its time limit.  Such groups use the host transport.
    int max_share = 0;
    for (int r = 0; r < n_devices; ++r) {
      int share = 0;
      for (int q = 0; q < n_devices; ++q) share += dev[q] == dev[r];
      max_share = std::max(max_share, share);
    }
    if (G->mode == kCommPeer && max_share > kPeerMaxRanksPerDevice) {
      if (log_level > 0) fprintf(stderr, "[miba] MIBA_ALLREDUCE=peer with %d ranks on one device (limit %d): using the host transport\n", max_share, kPeerMaxRanksPerDevice);
      G->mode = kCommHost;
    }
  }
  int rc = MIBA_OK;
  for (int r = 0; r < n_devices && rc == MIBA_OK; ++r) {
    rc = create_rank(dev[r], r, n_devices, log_level, &G->sub[r]);
    if (rc == MIBA_OK) {
      G->sub[r]->grp = G;
      // every rank takes the share of the most crowded device: the grid of the PCG tail (and with it the order of its partial
      // sums) must be the same on every rank
      int max_share = 0;
      for (int a = 0; a < n_devices; ++a) {
        int share = 0;
        for (int q = 0; q < n_devices; ++q) share += dev[q] == dev[a];
        max_share = std::max(max_share, share);
      }
      G->sub[r]->dev_share = max_share;
    }
  }
  if (rc == MIBA_OK && n_devices > 1) {
    if (G->mode == kCommRccl) {
      std::vector<ncclComm_t> comms(n_devices);
      const ncclResult_t nr = ncclCommInitAll(comms.data(), n_devices, dev.data());
      if (nr != ncclSuccess) rc = fail(nullptr, MIBA_ERR_RCCL, "ncclCommInitAll failed: %s", ncclGetErrorString(nr));
      else for (int r = 0; r < n_devices; ++r) G->sub[r]->comm = comms[r];
    } else if (G->mode == kCommHost) {
      G->hbuf.assign(2 * (size_t)n_devices, nullptr);
      for (int r = 0; r < 2 * n_devices && rc == MIBA_OK; ++r)
        if (hipHostMalloc((void**)&G->hbuf[r], kPeerChunk * sizeof(double)) != hipSuccess) rc = fail(nullptr, MIBA_ERR_NOMEM, "pinned staging allocation failed");
    } else {
      G->pstage.assign(n_devices, nullptr);
      G->psig.assign(n_devices, nullptr);
      G->xstage.assign(n_devices, nullptr);
      for (int r = 0; r < n_devices && rc == MIBA_OK; ++r) {
        hipError_t e = hipSetDevice(dev[r]);
        for (int q = 0; q < n_devices && e == hipSuccess; ++q)
          if (dev[q] != dev[r]) {
            const hipError_t pe = hipDeviceEnablePeerAccess(dev[q], 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) e = pe;
            (void)hipGetLastError();
          }
        // fine-grained device memory: coherent for the peers' system-scope loads and stores
        if (e == hipSuccess) e = hipExtMallocWithFlags((void**)&G->pstage[r], 2 * kPeerChunk * sizeof(double), hipDeviceMallocFinegrained);
        if (e == hipSuccess) e = hipExtMallocWithFlags((void**)&G->psig[r], sizeof(PeerSignals), hipDeviceMallocFinegrained);
        if (e == hipSuccess) e = hipMemset(G->psig[r], 0, sizeof(PeerSignals));
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) rc = fail(nullptr, MIBA_ERR_HIP, "peer transport setup failed on device %d: %s", dev[r], hipGetErrorString(e));
      }
    }
  }
  if (rc != MIBA_OK) {
    const std::string msg = g_last_error;
    destroy_group(G);
    delete g;
    g_last_error = msg;
    return rc;
  }
  *out = g;
  return MIBA_OK;
}

int miba_destroy(miba_handle h) {
  if (!h) return MIBA_OK;
  if (h->owned) {
    destroy_group(h->owned);
    delete h;
    return MIBA_OK;
  }
  (void)hipSetDevice(h->device);
  if (h->st) (void)hipStreamSynchronize(h->st);
  free_problem(h);
  drain_events(h);
  for (auto& e : h->ev_free) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (auto& e : h->ev_busy) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  if (h->ev_poll) (void)hipEventDestroy(h->ev_poll);
  if (h->comm) (void)ncclCommDestroy(h->comm);
  if (h->h_stage) (void)hipHostFree(h->h_stage);
  for (int b = 0; b < 2; ++b) {
    if (h->snap_cam[b]) (void)hipHostFree(h->snap_cam[b]);
    if (h->snap_X[b]) (void)hipHostFree(h->snap_X[b]);
    if (h->ev_snap[b]) (void)hipEventDestroy(h->ev_snap[b]);
  }
  if (h->ev_state) (void)hipEventDestroy(h->ev_state);
  if (h->st2) { (void)hipStreamSynchronize(h->st2); (void)hipStreamDestroy(h->st2); }
  if (h->h_scal) (void)hipHostFree(h->h_scal);
  if (h->h_ctl) (void)hipHostFree(h->h_ctl);
  if (h->st) (void)hipStreamDestroy(h->st);
  delete h;
  return MIBA_OK;
}

int miba_plan_shards(const miba_problem* p, int32_t world_size, int64_t* n_valid_obs, int64_t* n_tiles, int64_t* n_slots,
                     int64_t* tiles_per_rank, int64_t* obs_per_rank) {
  if (!p) return fail(nullptr, MIBA_ERR_INVALID, "null problem");
  Plan P;
  const char* po = getenv("MIBA_POINT_ORDER");
  int rc = build_plan(P, p->n_cam, p->n_pt, p->n_obs, p->obs_cam, p->obs_pt, p->obs_xy, p->X, world_size, po ? atoi(po) : 1);
  if (rc) return fail(nullptr, rc == -6 ? MIBA_ERR_LIMIT : MIBA_ERR_INVALID, "invalid problem (rc=%d)", rc);
  if (n_valid_obs) *n_valid_obs = P.n_valid;
  if (n_tiles) *n_tiles = (int64_t)P.tiles.size();
  if (n_slots) *n_slots = P.n_slots;
  for (int r = 0; r < world_size; ++r) {
    if (tiles_per_rank) tiles_per_rank[r] = P.shard_tile[r + 1] - P.shard_tile[r];
    if (obs_per_rank) {
      int64_t n = 0;
      for (int64_t t = P.shard_tile[r]; t < P.shard_tile[r + 1]; ++t)
        for (int32_t s = P.tiles[t].slot0; s < P.tiles[t].slot0 + P.tiles[t].nslot; ++s) n += P.idx[s] >= 0;
      obs_per_rank[r] = n;
    }
  }
  return MIBA_OK;
}

int miba_plan_product_stats(const miba_problem* p, double* out10) {
  double* out8 = out10;
  if (!p || !out8) return fail(nullptr, MIBA_ERR_INVALID, "null argument");
  Plan P;
  const char* po = getenv("MIBA_POINT_ORDER");
  int rc = build_plan(P, p->n_cam, p->n_pt, p->n_obs, p->obs_cam, p->obs_pt, p->obs_xy, p->X, 1, po ? atoi(po) : 1);
  if (rc) return fail(nullptr, rc == -6 ? MIBA_ERR_LIMIT : MIBA_ERR_INVALID, "invalid problem (rc=%d)", rc);
  CmLayout L;
  build_cm_layout(L, P.tiles, P.groups, P.idx.data(), P.pt_first.data(), P.pt_count.data());
  CmStats S;
  const bool ok = cm_layout_stats(L, P.tiles, P.groups, P.idx.data(), P.pt_first.data(), P.pt_count.data(), S);
  int64_t n_giant_obs = 0;
  for (const Tile& T : P.tiles)
    if (T.giant) n_giant_obs += T.ncs;
  out8[0] = ok ? 1.0 : 0.0;
  out8[1] = (double)S.n_obs; out8[2] = (double)S.n_entries; out8[3] = (double)S.n_sub; out8[4] = (double)S.n_seg;
  out8[5] = S.pt_mult; out8[6] = S.cam_mult; out8[7] = (double)n_giant_obs; out8[8] = S.rd_extra; out8[9] = S.at_extra;
  return MIBA_OK;
}

int miba_plan_tile_order(int32_t n_rows, const int32_t* eu, const int32_t* ev, int64_t n_edges, int32_t n_arcs, int32_t* pos_out,
                         int32_t* arc_start_out, int32_t* n_arcs_out, int64_t* fill_tiles, double* cost_us, int32_t nested, int32_t* group_start,
                         int32_t* group_level, int32_t* n_groups) {
  if (n_rows < 1 || n_rows > 32768 || n_edges < 0 || (n_edges && (!eu || !ev)) || !pos_out || !arc_start_out || n_arcs < 1 || n_arcs > kMaxArcs)
    return fail(nullptr, MIBA_ERR_INVALID, "bad tile graph");
  const size_t nW = ((size_t)n_rows + 63) / 64;
  std::vector<uint64_t> adj((size_t)n_rows * nW, 0);
  for (int64_t i = 0; i < n_edges; ++i) {
    const int32_t u = eu[i], v = ev[i];
    if (u < 0 || v < 0 || u >= n_rows || v >= n_rows) return fail(nullptr, MIBA_ERR_INVALID, "edge %lld out of range", (long long)i);
    if (u == v) continue;
    adj[(size_t)u * nW + (v >> 6)] |= (uint64_t)1 << (v & 63);
    adj[(size_t)v * nW + (u >> 6)] |= (uint64_t)1 << (u & 63);
  }
  TileOrder O;
  if (!order_tile_graph(adj, n_rows, n_arcs, (int64_t)1 << 40, 0, O, nested != 0))
    return fail(nullptr, MIBA_ERR_LIMIT, "tile graph too large");
  const int na = O.n_arcs;
  for (int32_t v = 0; v < n_rows; ++v) pos_out[v] = O.pos[v];
  for (int a = 0; a <= n_arcs; ++a) arc_start_out[a] = O.arc_start[std::min(a, na)];
  if (n_arcs_out) *n_arcs_out = na;
  if (fill_tiles) *fill_tiles = O.n_off;
  if (cost_us) *cost_us = tile_order_cost(O);
  const int nG = (int)O.group_level.size();
  if (n_groups) *n_groups = nG;
  if (group_start && group_level) {
    for (int g = 0; g < nG && g < 2 * n_arcs; ++g) { group_start[g] = O.group_start[g]; group_level[g] = O.group_level[g]; }
    group_start[std::min(nG, 2 * n_arcs)] = n_rows;
  }
  return MIBA_OK;
}

int miba_plan_point_owner(const miba_problem* p, int32_t world_size, int32_t* owner) {
  if (!p || !owner) return fail(nullptr, MIBA_ERR_INVALID, "null argument");
  Plan P;
  const char* po = getenv("MIBA_POINT_ORDER");
  int rc = build_plan(P, p->n_cam, p->n_pt, p->n_obs, p->obs_cam, p->obs_pt, p->obs_xy, p->X, world_size, po ? atoi(po) : 1);
  if (rc) return fail(nullptr, rc == -6 ? MIBA_ERR_LIMIT : MIBA_ERR_INVALID, "invalid problem (rc=%d)", rc);
  for (int32_t i = 0; i < p->n_pt; ++i) owner[i] = -1;
  for (int r = 0; r < world_size; ++r)
    for (int64_t t = P.shard_tile[r]; t < P.shard_tile[r + 1]; ++t)
      for (int32_t g = P.tiles[t].grp0; g < P.tiles[t].grp0 + P.tiles[t].ngrp; ++g)
        for (int32_t q = P.groups[g].pt0; q < P.groups[g].pt0 + P.groups[g].npt; ++q) owner[P.pt_orig[q]] = r;
  return MIBA_OK;
}

// PlanSorter of build_plan on a GPU: the valid observations ordered by (point in tile order, camera), stable, by a rocPRIM
// radix sort.  Returns non-zero (the planner then sorts on its host threads) when the device cannot hold the scratch.
struct DeviceSortCtx { int device; hipStream_t st; };
static int device_plan_sort(void* ctx, int64_t n_obs, int32_t n_pt, const int32_t* obs_cam, const int32_t* obs_pt, const uint8_t* ok,
                            const int32_t* sorted_of, int32_t n_act, int64_t n_valid, int64_t* obs_sorted, int32_t* sorted_cam) {
  const DeviceSortCtx& C = *(const DeviceSortCtx*)ctx;
  PlanLap lap;
  if (n_obs < 1 || n_obs >= ((int64_t)1 << 32) || hipSetDevice(C.device) != hipSuccess) return 1;
  const size_t n = (size_t)n_obs;
  int32_t *d_cam = nullptr, *d_pt = nullptr, *d_so = nullptr;
  uint8_t* d_ok = nullptr;
  uint64_t *k0 = nullptr, *k1 = nullptr;
  uint32_t *v0 = nullptr, *v1 = nullptr;
  auto cleanup = [&]() {
    (void)hipFree(d_cam); (void)hipFree(d_pt); (void)hipFree(d_so); (void)hipFree(d_ok);
    (void)hipFree(k0); (void)hipFree(k1); (void)hipFree(v0); (void)hipFree(v1);
  };
  hipError_t e = hipMalloc(&d_cam, n * 4);
  if (e == hipSuccess) e = hipMalloc(&d_pt, n * 4);
  if (e == hipSuccess) e = hipMalloc(&d_so, sizeof(int32_t) * (size_t)n_pt);
  if (e == hipSuccess) e = hipMalloc(&d_ok, n);
  if (e == hipSuccess) e = hipMalloc(&k0, n * 8);
  if (e == hipSuccess) e = hipMalloc(&k1, n * 8);
  if (e == hipSuccess) e = hipMalloc(&v0, n * 4);
  if (e == hipSuccess) e = hipMalloc(&v1, n * 4);
  if (e == hipSuccess) e = hipMemcpyAsync(d_cam, obs_cam, n * 4, hipMemcpyHostToDevice, C.st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_pt, obs_pt, n * 4, hipMemcpyHostToDevice, C.st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_so, sorted_of, sizeof(int32_t) * (size_t)n_pt, hipMemcpyHostToDevice, C.st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_ok, ok, n, hipMemcpyHostToDevice, C.st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_sort_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, C.st, n_obs, d_cam, d_pt, d_ok, d_so, k0, v0);
    e = hipGetLastError();
  }
  // key bits: 24 for the camera, enough for the point ranks, and the all-ones key of the filtered observations sorts last
  // only if its bits are inside the sorted range: sort all 64 bits when anything was filtered
  int end_bit = 24;
  while (end_bit < 64 && ((uint64_t)1 << (end_bit - 24)) < (uint64_t)std::max<int32_t>(1, n_act)) ++end_bit;
  if (n_valid != n_obs) end_bit = 64;
  if (e == hipSuccess) e = device_sort_pairs(k0, k1, v0, v1, n, end_bit, C.st);
  raw_vector<uint32_t> out;
  if (e == hipSuccess) {
    // the cameras in sorted order are the low 24 bits of the sorted keys (d_cam has served its purpose and takes them)
    hipLaunchKernelGGL(k_key_cam, dim3((unsigned)((n_valid + 255) / 256)), dim3(256), 0, C.st, n_valid, k1, d_cam);
    e = hipGetLastError();
  }
  if (e == hipSuccess) {
    out.resize((size_t)n_valid);
    e = download_staged(out.data(), v1, (size_t)n_valid * 4, C.st);
    if (e == hipSuccess) e = download_staged(sorted_cam, d_cam, (size_t)n_valid * 4, C.st);
  }
  lap("  device sort (upload, sort, download)");
  cleanup();
  if (e != hipSuccess) { (void)hipGetLastError(); return 1; }
  plan_parallel_for(n_valid, [&](int64_t a, int64_t b) { for (int64_t j = a; j < b; ++j) obs_sorted[j] = (int64_t)out[(size_t)j]; });
  lap("  free + widen");
  return 0;
}

// MIBA_DEVICE_PLAN=0 keeps the whole repack on the host threads (the planner's own sort, host-built measurement arrays)
static bool device_repack_wanted() {
  const char* e = getenv("MIBA_DEVICE_PLAN");
  return !(e && e[0] == '0');
}

static int set_problem_impl(miba_handle h, const miba_problem* p, std::shared_ptr<Plan> shared) {
  if (!h || !p) return fail(h, MIBA_ERR_INVALID, "null argument");
  if (!p->R || !p->T || !p->K || !p->X || (p->n_obs > 0 && (!p->obs_cam || !p->obs_pt || !p->obs_xy)))
    return fail(h, MIBA_ERR_INVALID, "null problem array");
  HIPCHK(h, hipSetDevice(h->device));
  free_problem(h);
  if (p->camera_model != MIBA_MODEL_BAL && p->camera_model != MIBA_MODEL_REFERENCE) return fail(h, MIBA_ERR_INVALID, "unknown camera_model %d", (int)p->camera_model);
  h->cam_model = p->camera_model;
  h->principal.clear();
  if (h->cam_model == MIBA_MODEL_REFERENCE) {
    // The functor's residual, observed - (f P.xy/P.z + o), is the BAL-shaped residual f (-P.xy/P.z) - (o - observed) of the
    // camera (R', -R' t, f, 0, 0): the observations are uploaded as o - observed and the per-observation device code is
    // the BAL code; only the per-camera transforms know the model (k_cam_tables, mf_stage_cam, mf_blocks, k_cam_reduce_mf).
    h->principal.resize(2 * (size_t)p->n_cam);
    for (int c = 0; c < p->n_cam; ++c) { h->principal[2 * (size_t)c] = p->K[6 * (size_t)c + 2]; h->principal[2 * (size_t)c + 1] = p->K[6 * (size_t)c + 3]; }
  }
  auto obs_x = [&](int64_t o) { return h->cam_model ? h->principal[2 * (size_t)p->obs_cam[o]] - p->obs_xy[2 * o] : p->obs_xy[2 * o]; };
  auto obs_y = [&](int64_t o) { return h->cam_model ? h->principal[2 * (size_t)p->obs_cam[o] + 1] - p->obs_xy[2 * o + 1] : p->obs_xy[2 * o + 1]; };
  const char* po = getenv("MIBA_POINT_ORDER");
  // the ranks of a single-process group share one plan (built once by the group), a stand-alone rank builds its own
  if (shared) h->plan_sp = shared; else h->plan_sp = std::make_shared<Plan>();
  Plan& P = h->plan_ref();
  const bool dev_repack = device_repack_wanted();
  DeviceSortCtx sort_ctx{h->device, h->st};
  const PlanSorter sorter{device_plan_sort, &sort_ctx};
  int rc = shared ? 0 : build_plan(P, p->n_cam, p->n_pt, p->n_obs, p->obs_cam, p->obs_pt, p->obs_xy, p->X, h->cfg.world_size,
                                   po ? atoi(po) : 1, dev_repack ? &sorter : nullptr);
  if (rc) return fail(h, rc == -6 ? MIBA_ERR_LIMIT : MIBA_ERR_INVALID, "invalid problem (rc=%d): indices out of range or limits exceeded", rc);
  PlanLap lap;
  h->n_cam = p->n_cam;
  h->n_valid_global = P.n_valid;
  h->pcg_nb = std::min(nblk(p->n_cam, kPcgCams), kPcgMaxBlocks);
  h->red_cpb = 4 * nblk(p->n_cam, 4 * kPcgMaxBlocks);
  h->red_nb = nblk(p->n_cam, h->red_cpb);
  // the tail kernel of a PCG iteration holds grid barriers: its grid must be resident as a whole.  One workgroup (kTailThreads) per
  // compute unit at most, one per sixteen cameras on a small problem; ranks of this process that share the device share its compute units.
  h->tail_nb = std::max(1, std::min(nblk(p->n_cam, 16), std::min(h->n_cu, h->n_cu * h->tail_occ) / std::max(1, h->dev_share)));
  h->t0 = P.shard_tile[h->cfg.rank];
  h->t1 = P.shard_tile[h->cfg.rank + 1];
  h->n_tiles = h->t1 - h->t0;
  // local extents
  std::vector<Tile> tiles(P.tiles.begin() + h->t0, P.tiles.begin() + h->t1);
  int32_t g0 = 0, g1 = 0, tc0 = 0, tc1 = 0;
  if (h->n_tiles) {
    g0 = tiles.front().grp0; g1 = tiles.back().grp0 + tiles.back().ngrp;
    tc0 = tiles.front().cam0; tc1 = tiles.back().cam0 + tiles.back().ncs;
    h->slot_base = tiles.front().slot0;
    h->n_slots = (int64_t)tiles.back().slot0 + tiles.back().nslot - h->slot_base;
  } else {
    h->slot_base = 0; h->n_slots = 0;
  }
  std::vector<Group> grps(P.groups.begin() + g0, P.groups.begin() + g1);
  h->pt_base = grps.empty() ? 0 : grps.front().pt0;
  h->n_pt = grps.empty() ? 0 : grps.back().pt0 + grps.back().npt - h->pt_base;
  const int32_t chunk_base = (int32_t)(h->slot_base / kChunk);
  for (auto& t : tiles) { t.grp0 -= g0; t.cam0 -= tc0; t.slot0 -= (int32_t)h->slot_base; }
  for (auto& g : grps) { g.chunk0 -= chunk_base; g.pt0 -= h->pt_base; }
  // camera-major slab rows for the local (tile, camera-slot) pairs
  const int64_t n_rows = tc1 - tc0;
  h->n_rows = n_rows;
  std::vector<int32_t> cam_row_off(p->n_cam + 1, 0), slab_row(n_rows ? n_rows : 1);
  for (int64_t e = 0; e < n_rows; ++e) cam_row_off[P.tile_cams[tc0 + e] + 1]++;
  for (int c = 0; c < p->n_cam; ++c) cam_row_off[c + 1] += cam_row_off[c];
  {
    std::vector<int32_t> cur(cam_row_off.begin(), cam_row_off.end() - 1);
    for (int64_t e = 0; e < n_rows; ++e) slab_row[e] = cur[P.tile_cams[tc0 + e]]++;
  }
  // camera ranges of the waves of pcg_tail (miba_pcg.hpp): consecutive cameras, at most seven, cut where the slab rows of the
  // range reach the target that gives every wave of the grid the same share; the target grows until one round of the grid takes
  // all ranges, if the seven-camera limit allows that at all
  std::vector<int32_t> tail_rec;
  {
    const int W = h->tail_nb * kTailWaves;
    const int64_t total = cam_row_off[p->n_cam];
    int64_t target = std::max<int64_t>(1, (total + W - 1) / W);
    for (int attempt = 0; attempt < 64; ++attempt) {
      tail_rec.clear();
      for (int c0 = 0; c0 < p->n_cam;) {
        int nc = 1;
        while (nc < 7 && c0 + nc < p->n_cam && cam_row_off[c0 + nc + 1] - cam_row_off[c0] <= target) ++nc;
        tail_rec.push_back(c0); tail_rec.push_back(nc);
        for (int q = 0; q < 8; ++q) tail_rec.push_back(cam_row_off[c0 + std::min(q, nc)]);
        tail_rec.push_back(0); tail_rec.push_back(0);
        c0 += nc;
      }
      if ((int64_t)tail_rec.size() / kTailRec <= W || (int64_t)nblk(p->n_cam, 7) > W) break;
      target += std::max<int64_t>(1, target / 8);
    }
    h->tail_n_vw = (int)(tail_rec.size() / kTailRec);
  }
  // observations in chunk-SoA order, point ranges
  std::vector<double> oxy(dev_repack ? (size_t)1 : (size_t)(h->n_slots ? h->n_slots : 1) * 2, 0.0);
  if (!dev_repack) plan_parallel_for(h->n_slots, [&](int64_t a, int64_t b) {
    for (int64_t s = a; s < b; ++s) {
      const int64_t o = P.slot_orig[h->slot_base + s];
      if (o < 0) continue;
      const size_t chunk = (size_t)(s / kChunk), lane = (size_t)(s % kChunk);
      oxy[chunk * 2 * kChunk + lane] = obs_x(o);
      oxy[chunk * 2 * kChunk + kChunk + lane] = obs_y(o);
    }
  });
  h->n_obs_local = 0;
  for (int32_t q = 0; q < h->n_pt; ++q) h->n_obs_local += P.pt_count[h->pt_base + q];
  std::vector<int2> wlv((size_t)(h->n_slots / kWave) + 1);
  {
    // per wavelet: {first (local) point of the enclosing 256-slot chunk, longest point run in the wavelet}
    std::vector<int32_t> chunk_pt0((size_t)(h->n_slots / kChunk) + 1, 0);
    for (const auto& g : grps)
      for (int32_t ch = g.chunk0; ch < g.chunk0 + g.nchunk; ++ch) chunk_pt0[ch] = g.pt0;
    for (int64_t q = 0; q < h->n_slots / kWave; ++q) {
      wlv[q].x = chunk_pt0[(size_t)(q / 4)];
      wlv[q].y = P.wl_npt[h->slot_base / kWave + q];
    }
  }
  std::vector<int2> pt_rng(h->n_pt ? h->n_pt : 1);
  for (int32_t q = 0; q < h->n_pt; ++q) {
    pt_rng[q].x = P.pt_first[h->pt_base + q] - (int32_t)h->slot_base;
    pt_rng[q].y = P.pt_count[h->pt_base + q];
  }
  lap("local arrays (oxy, ranges)");
  // camera-major sub-tiles of the product kernel (miba_mf2.hpp)
  CmLayout cml;
  {
    const char* e = getenv("MIBA_K3");
    h->k3_cm = !(e && !strcmp(e, "pm"));
    if (h->k3_cm) {
      std::vector<int32_t> pf(h->n_pt ? h->n_pt : 1), pc(h->n_pt ? h->n_pt : 1);
      for (int32_t q = 0; q < h->n_pt; ++q) { pf[q] = pt_rng[q].x; pc[q] = pt_rng[q].y; }
      build_cm_layout(cml, tiles, grps, P.idx.data() + h->slot_base, pf.data(), pc.data());
      // small tiles (small problems: a tile per compute unit matters more there) leave the sub-tiles' lanes mostly empty:
      // the point-major product is the better kernel then
      int64_t segs = 0;
      for (const CmSub& sb : cml.subs) segs += sb.nseg;
      const char* force = getenv("MIBA_K3");
      if (!(force && !strcmp(force, "cm")) && segs * 4 < (int64_t)cml.subs.size() * kCmLanes * 3) { h->k3_cm = 0; cml = CmLayout(); }
    }
  }
  // per (tile, camera slot) at a fixed stride: camera, slab row, accumulator rows (what the product kernel's first loads fetch)
  std::vector<int4> cam64(h->k3_cm ? tiles.size() * (size_t)kMaxCamSlots : 1, make_int4(0, 0, 0, 0));
  if (h->k3_cm)
    for (size_t ti = 0; ti < tiles.size(); ++ti) {
      if (tiles[ti].giant) continue;
      for (int c = 0; c < tiles[ti].ncs; ++c) {
        const size_t e = (size_t)tiles[ti].cam0 + c;
        cam64[ti * kMaxCamSlots + c] = make_int4(P.tile_cams[tc0 + e], slab_row[e], cml.rowinfo[e], 0);
      }
    }
  lap("camera-major sub-tiles (K3)");
  // camera-major copy of the local observations, cut into segments of at most kSegObs observations of one camera
  std::vector<int4> segv;
  std::vector<int32_t> cmo_pt(dev_repack ? (size_t)1 : (size_t)(h->n_obs_local ? h->n_obs_local : 1)), cam_seg_off((size_t)p->n_cam + 1, 0);
  std::vector<double2> cmo_xy(dev_repack ? (size_t)1 : (size_t)(h->n_obs_local ? h->n_obs_local : 1));
  if (!dev_repack) {
    // stable counting sort of the local observations by camera, the points cut into R ranges that are counted
    // and scattered by R threads (ascending point order inside every camera is kept)
    const int R = 16;
    const size_t ncam = (size_t)p->n_cam;
    std::vector<int64_t> cnt((size_t)R * ncam, 0), coff(ncam + 1, 0);
    auto range = [&](int r) { return std::make_pair((int64_t)h->n_pt * r / R, (int64_t)h->n_pt * (r + 1) / R); };
    auto for_ranges = [&](auto fn) {
      std::vector<std::thread> th;
      for (int r = 1; r < R; ++r) th.emplace_back(fn, r);
      fn(0);
      for (auto& x : th) x.join();
    };
    for_ranges([&](int r) {
      int64_t* c = cnt.data() + (size_t)r * ncam;
      const auto qr = range(r);
      for (int64_t q = qr.first; q < qr.second; ++q)
        for (int64_t s = pt_rng[q].x; s < (int64_t)pt_rng[q].x + pt_rng[q].y; ++s) c[P.ocam[h->slot_base + s]]++;
    });
    for (size_t c = 0; c < ncam; ++c) {
      int64_t acc = coff[c];
      for (int r = 0; r < R; ++r) { const int64_t v = cnt[(size_t)r * ncam + c]; cnt[(size_t)r * ncam + c] = acc; acc += v; }
      coff[c + 1] = acc;
    }
    for_ranges([&](int r) {
      int64_t* cur = cnt.data() + (size_t)r * ncam;
      const auto qr = range(r);
      for (int64_t q = qr.first; q < qr.second; ++q)
        for (int64_t s = pt_rng[q].x; s < (int64_t)pt_rng[q].x + pt_rng[q].y; ++s) {
          const int64_t o = P.slot_orig[h->slot_base + s];
          const int64_t d = cur[P.ocam[h->slot_base + s]]++;
          cmo_pt[d] = (int32_t)q;
          cmo_xy[d] = make_double2(obs_x(o), obs_y(o));
        }
    });
    for (int c = 0; c < p->n_cam; ++c) {
      cam_seg_off[c] = (int32_t)segv.size();
      for (int64_t a = coff[c]; a < coff[c + 1]; a += kSegObs)
        segv.push_back(make_int4(c, (int)a, (int)std::min<int64_t>(kSegObs, coff[c + 1] - a), 0));
    }
    cam_seg_off[p->n_cam] = (int32_t)segv.size();
    h->n_seg = (int32_t)segv.size();
  }
  lap("camera-major copy");
#define ALLOC(ptr, n)                        \
  do {                                       \
    rc = dev_alloc(h, &(ptr), (size_t)(n));  \
    if (rc) { free_problem(h); return rc; }  \
  } while (0)
#define UP(dst, src, n)                      \
  do {                                       \
    rc = upload(h, dst, src, (size_t)(n));   \
    if (rc) { free_problem(h); return rc; }  \
  } while (0)
  const size_t nc = (size_t)p->n_cam, np = (size_t)h->n_pt, ns = (size_t)h->n_slots, nt = (size_t)h->n_tiles;
  ALLOC(h->d_tiles, nt); ALLOC(h->d_grps, grps.size());
  ALLOC(h->d_idx, ns); ALLOC(h->d_ocam, ns); ALLOC(h->d_tile_cams, n_rows); ALLOC(h->d_slab_row, n_rows);
  ALLOC(h->d_tail_rec, tail_rec.size());
  ALLOC(h->d_cam_row_off, nc + 1); ALLOC(h->d_pt_rng, np); ALLOC(h->d_cam_active, nc); ALLOC(h->d_wl, wlv.size());
  ALLOC(h->d_oxy, 2 * ns);
  ALLOC(h->d_cam, 9 * nc); ALLOC(h->d_cam_cand, 9 * nc); ALLOC(h->d_ct, (size_t)kCamTabStride * nc); ALLOC(h->d_sc, 9 * nc);
  ALLOC(h->d_ct_cand, (size_t)kCamTabStride * nc); ALLOC(h->d_cm, 10 * nc); ALLOC(h->d_ptrec, (size_t)kPtRec * np);
  std::vector<int32_t> giant;
  for (size_t i = 0; i < tiles.size(); ++i) if (tiles[i].giant) giant.push_back((int32_t)i);
  h->n_giant = (int32_t)giant.size();
  ALLOC(h->d_giant, giant.size());
  ALLOC(h->d_cmt, cml.tiles.size()); ALLOC(h->d_cms, cml.subs.size()); ALLOC(h->d_cmi, cml.idx.size()); ALLOC(h->d_cam64, cam64.size()); ALLOC(h->d_cmslot, cml.slot.size());
  ALLOC(h->d_cmo_pt, h->n_obs_local ? h->n_obs_local : 1); ALLOC(h->d_cmo_xy, h->n_obs_local ? h->n_obs_local : 1);
  ALLOC(h->d_cam_seg_off, nc + 1); ALLOC(h->d_cfree, 9 * nc);
  if (!dev_repack) { ALLOC(h->d_seg, segv.size()); ALLOC(h->d_slabB, (size_t)kK2W * segv.size()); }
  ALLOC(h->d_X, 3 * np); ALLOC(h->d_Xc, 3 * np); ALLOC(h->d_sp, 3 * np);
  h->d_J = nullptr;   // allocated on first use (ensure_J): the matrix-free LM loop never touches it
  h->d_S = nullptr; h->d_Linv = nullptr; h->d_dflag = nullptr; h->dense_N = 0;
  h->d_dseg = nullptr; h->d_dpts = nullptr; h->n_dseg = 0; h->dense_planned = false; h->dense_plan_tried = false; h->d_camacc2 = nullptr;
  h->sparse_state = 0; h->d_sp_tiles = nullptr;
  h->d_omega = nullptr;
  h->d_gather = nullptr;
  ALLOC(h->d_gp, 3 * np); ALLOC(h->d_dgp, 3 * np); ALLOC(h->d_yp, 3 * np);
  // the 64-wide per-(tile, camera) slab is only used by the stored-Jacobian K2 and the one-pass K2-mf
  ALLOC(h->d_slab9, 9 * (size_t)n_rows);
  ALLOC(h->d_camacc, (size_t)kK2W * nc); ALLOC(h->d_Minv, 81 * nc); ALLOC(h->d_Dc2, 9 * nc); ALLOC(h->d_rhs, 9 * nc);
  ALLOC(h->d_px, 9 * nc); ALLOC(h->d_pr, 9 * nc); ALLOC(h->d_pz, 9 * nc); ALLOC(h->d_pp, 9 * nc); ALLOC(h->d_pq, 9 * nc + kPcgMaxBlocks);
  ALLOC(h->d_probe, 9 * nc);
  ALLOC(h->d_part_tile, 4 * nt + 4);
  ALLOC(h->d_part_small, 2 * (size_t)nblk(nc, 64) + 2 * 240 + 512);
  ALLOC(h->d_part_pcg, 5 * (size_t)kPcgMaxBlocks);
  ALLOC(h->d_sum, S_COUNT); ALLOC(h->d_max, S_COUNT);
  ALLOC(h->d_ctl, 2);
  ALLOC(h->d_bar, kBarWords);
  ALLOC(h->d_clk, 4);
  HIPCHK(h, hipMemset(h->d_clk, 0, 4 * sizeof(unsigned long long)));
  if (h->grp && h->grp->n > 1 && h->grp->mode == kCommPeer && (size_t)h->cfg.rank < h->grp->xstage.size()) {
    // exchange buffer of the PCG tail kernel (pcg_tail<MODE, 2>): the q shares of all ranks, two generations.  Fine-grained, so that
    // the peers' system-scope stores land coherently; without it (no memory) the ranks keep the all-reduce kernel between launches.
    void* xs = nullptr;
    const size_t bytes = 2 * (size_t)h->grp->n * 9 * nc * sizeof(double);
    if (hipExtMallocWithFlags(&xs, bytes ? bytes : 8, hipDeviceMallocFinegrained) == hipSuccess) {
      h->allocs.push_back(xs);
      HIPCHK(h, hipMemset(xs, 0, bytes));
      h->grp->xstage[h->cfg.rank] = (double*)xs;
    } else (void)hipGetLastError();
  }
  h->d_tail_dbg = nullptr;
  if (getenv("MIBA_TAIL_DEBUG")) { ALLOC(h->d_tail_dbg, 8 * (size_t)h->tail_nb); HIPCHK(h, hipMemset(h->d_tail_dbg, 0, 64 * (size_t)h->tail_nb)); }
  UP(h->d_tiles, tiles.data(), nt);
  UP(h->d_grps, grps.data(), grps.size());
  UP(h->d_idx, P.idx.data() + h->slot_base, ns);
  UP(h->d_ocam, P.ocam.data() + h->slot_base, ns);
  UP(h->d_tile_cams, P.tile_cams.data() + tc0, n_rows);
  UP(h->d_slab_row, slab_row.data(), n_rows);
  UP(h->d_cam_row_off, cam_row_off.data(), nc + 1);
  UP(h->d_tail_rec, tail_rec.data(), tail_rec.size());
  UP(h->d_pt_rng, pt_rng.data(), np);
  UP(h->d_wl, wlv.data(), wlv.size());
  UP(h->d_cam_active, P.cam_active.data(), nc);
  UP(h->d_giant, giant.data(), giant.size());
  UP(h->d_cmt, cml.tiles.data(), cml.tiles.size());
  UP(h->d_cms, cml.subs.data(), cml.subs.size());
  UP(h->d_cmi, reinterpret_cast<const uint2*>(cml.idx.data()), cml.idx.size());
  UP(h->d_cam64, cam64.data(), cam64.size());
  UP(h->d_cmslot, cml.slot.data(), cml.slot.size());
  if (!dev_repack) {
    UP(h->d_oxy, oxy.data(), 2 * ns);
    UP(h->d_seg, segv.data(), segv.size());
    UP(h->d_cmo_pt, cmo_pt.data(), (size_t)h->n_obs_local);
    UP(h->d_cmo_xy, cmo_xy.data(), (size_t)h->n_obs_local);
    UP(h->d_cam_seg_off, cam_seg_off.data(), nc + 1);
  }
  lap("alloc + upload");
  if (dev_repack) {
    // Measurement arrays on the device: the caller's observations go up once as they are; oxy is gathered
    // through slot_orig, the camera-major copy is a radix sort of (camera, slot) keys.  Same arrays as the host path builds.
    rc = MIBA_OK;
    double2* d_obs = nullptr; int32_t* d_ocam_in = nullptr; int64_t* d_so = nullptr; double2* d_pp = nullptr;
    uint64_t *k0 = nullptr, *k1 = nullptr; uint32_t *v0 = nullptr, *v1 = nullptr; int32_t *d_cnt = nullptr, *d_spt = nullptr;
    auto tmp_free = [&]() {
      (void)hipFree(d_obs); (void)hipFree(d_ocam_in); (void)hipFree(d_so); (void)hipFree(d_pp); (void)hipFree(k0); (void)hipFree(k1);
      (void)hipFree(v0); (void)hipFree(v1); (void)hipFree(d_cnt); (void)hipFree(d_spt);
    };
    const size_t no = (size_t)(p->n_obs ? p->n_obs : 1), ns1 = ns ? ns : 1;
    hipError_t e = hipMalloc(&d_obs, no * sizeof(double2));
    if (e == hipSuccess) e = hipMalloc(&d_so, ns1 * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(&k0, ns1 * 8);
    if (e == hipSuccess) e = hipMalloc(&k1, ns1 * 8);
    if (e == hipSuccess) e = hipMalloc(&v0, ns1 * 4);
    if (e == hipSuccess) e = hipMalloc(&v1, ns1 * 4);
    if (e == hipSuccess) e = hipMalloc(&d_cnt, (nc + 1) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&d_spt, ns1 * sizeof(int32_t));
    if (e == hipSuccess && h->cam_model) {
      e = hipMalloc(&d_ocam_in, no * sizeof(int32_t));
      if (e == hipSuccess) e = hipMalloc(&d_pp, nc * sizeof(double2));
      if (e == hipSuccess) e = hipMemcpyAsync(d_ocam_in, p->obs_cam, (size_t)p->n_obs * sizeof(int32_t), hipMemcpyHostToDevice, h->st);
      if (e == hipSuccess) e = hipMemcpyAsync(d_pp, h->principal.data(), nc * sizeof(double2), hipMemcpyHostToDevice, h->st);
    }
    if (e == hipSuccess && p->n_obs) e = hipMemcpyAsync(d_obs, p->obs_xy, (size_t)p->n_obs * sizeof(double2), hipMemcpyHostToDevice, h->st);
    if (e == hipSuccess && ns) e = hipMemcpyAsync(d_so, P.slot_orig.data() + h->slot_base, ns * sizeof(int64_t), hipMemcpyHostToDevice, h->st);
    if (e == hipSuccess) e = hipMemsetAsync(d_cnt, 0, (nc + 1) * sizeof(int32_t), h->st);
    std::vector<int32_t> cnt(nc + 1, 0);
    if (e == hipSuccess && ns) {
      hipLaunchKernelGGL(k_gather_oxy, dim3(nblk(ns, 256)), dim3(256), 0, h->st, (int64_t)ns, (const int64_t*)d_so, (const double2*)d_obs,
                         (const int32_t*)d_ocam_in, (const double2*)d_pp, h->d_oxy);
      hipLaunchKernelGGL(k_slot_point, dim3(nblk(np, 256)), dim3(256), 0, h->st, (int32_t)np, (const int2*)h->d_pt_rng, d_spt);
      hipLaunchKernelGGL(k_cmo_keys, dim3(nblk(ns, 256)), dim3(256), 0, h->st, (int64_t)ns, (const int32_t*)h->d_idx, (const int32_t*)h->d_ocam, k0, v0,
                         d_cnt);
      e = hipGetLastError();
      if (e == hipSuccess) e = device_sort_pairs(k0, k1, v0, v1, ns, 64, h->st);
      if (e == hipSuccess && h->n_obs_local) {
        hipLaunchKernelGGL(k_cmo_fill, dim3(nblk((size_t)h->n_obs_local, 256)), dim3(256), 0, h->st, h->n_obs_local, (const uint32_t*)v1,
                           (const int32_t*)d_spt, (const double*)h->d_oxy, h->d_cmo_pt, h->d_cmo_xy);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipMemcpyAsync(cnt.data(), d_cnt, (nc + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, h->st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->st);
    tmp_free();
    if (e != hipSuccess) { free_problem(h); return fail(h, MIBA_ERR_HIP, "device repack failed: %s", hipGetErrorString(e)); }
    // segments of at most kSegObs observations of one camera
    int64_t first = 0;
    for (size_t c = 0; c < nc; ++c) {
      cam_seg_off[c] = (int32_t)segv.size();
      for (int64_t a = 0; a < cnt[c]; a += kSegObs)
        segv.push_back(make_int4((int)c, (int)(first + a), (int)std::min<int64_t>(kSegObs, cnt[c] - a), 0));
      first += cnt[c];
    }
    cam_seg_off[nc] = (int32_t)segv.size();
    h->n_seg = (int32_t)segv.size();
    ALLOC(h->d_seg, segv.size()); ALLOC(h->d_slabB, (size_t)kK2W * segv.size());
    UP(h->d_seg, segv.data(), segv.size());
    UP(h->d_cam_seg_off, cam_seg_off.data(), nc + 1);
    lap("device repack (oxy, camera-major copy)");
  }
  h->d_oxy_cm = nullptr;
  if (h->k3_cm && !cml.src.empty()) {
    // measurements in the order of the product's camera-major entries, for the back-substitution over the same sub-tiles
    int32_t* d_src = nullptr;
    const size_t ne = cml.src.size();
    ALLOC(d_src, ne); ALLOC(h->d_oxy_cm, ne);
    UP(d_src, cml.src.data(), ne);
    hipLaunchKernelGGL(k_cm_gather_oxy, dim3(nblk(ne, 256)), dim3(256), 0, h->st, (int64_t)ne, (const int32_t*)d_src, (const double*)h->d_oxy, h->d_oxy_cm);
    HIPCHK(h, hipGetLastError());
    // the source slots are not needed again (4 bytes per entry)
    HIPCHK(h, hipStreamSynchronize(h->st));
    h->allocs.erase(std::remove(h->allocs.begin(), h->allocs.end(), (void*)d_src), h->allocs.end());
    (void)hipFree(d_src);
    lap("measurements in camera-major order");
  }
#undef ALLOC
#undef UP
  hipLaunchKernelGGL(k_fill, dim3(nblk(9 * nc, 256)), dim3(256), 0, h->st, 9 * nc, 1.0, h->d_cfree);   // all blocks free

  HIPCHK(h, hipMemsetAsync(h->d_sum, 0, S_COUNT * sizeof(double), h->st));
  HIPCHK(h, hipMemsetAsync(h->d_max, 0, S_COUNT * sizeof(double), h->st));
  HIPCHK(h, hipMemsetAsync(h->d_ctl, 0, 2 * sizeof(Ctl), h->st));
  HIPCHK(h, hipMemsetAsync(h->d_part_pcg, 0, 5 * (size_t)kPcgMaxBlocks * sizeof(double), h->st));
  HIPCHK(h, hipStreamSynchronize(h->st));
  // the big host-side index arrays are not needed any more, except slot_orig / pt_orig (evaluate, get_state)
  if (!shared) {
    decltype(P.idx)().swap(P.idx);
    decltype(P.ocam)().swap(P.ocam);
  }
  h->has_problem = true;
  memset(&h->kt, 0, sizeof(h->kt));
  lap("finish");
  if (h->cam_model == MIBA_MODEL_REFERENCE) {   // all blocks free, except the intrinsics of the reference's model (src/BundleAdjustment.cpp:77)
    rc = miba_set_constant_blocks(h, nullptr);
    if (rc) { free_problem(h); return rc; }
  }
  return miba_set_state(h, p->R, p->T, p->K, p->X);
}

int miba_set_problem(miba_handle h, const miba_problem* p) {
  if (h && h->owned) {
    if (!p) return fail(h, MIBA_ERR_INVALID, "null argument");
    // one plan for all ranks, built once
    auto P = std::make_shared<Plan>();
    const char* po = getenv("MIBA_POINT_ORDER");
    DeviceSortCtx sort_ctx{h->owned->sub[0]->device, h->owned->sub[0]->st};
    const PlanSorter sorter{device_plan_sort, &sort_ctx};
    const int prc = build_plan(*P, p->n_cam, p->n_pt, p->n_obs, p->obs_cam, p->obs_pt, p->obs_xy, p->X, h->owned->n, po ? atoi(po) : 1,
                               device_repack_wanted() ? &sorter : nullptr);
    if (prc) return fail(h, prc == -6 ? MIBA_ERR_LIMIT : MIBA_ERR_INVALID, "invalid problem (rc=%d): indices out of range or limits exceeded", prc);
    const int rc = group_run(h, [&](miba_handle s, int) { return set_problem_impl(s, p, P); });
    decltype(P->idx)().swap(P->idx);
    decltype(P->ocam)().swap(P->ocam);
    h->has_problem = rc == MIBA_OK;
    return rc;
  }
  return set_problem_impl(h, p, nullptr);
}

int miba_set_state(miba_handle h, const double* R, const double* T, const double* K, const double* X) {
  if (h && h->owned) return group_run(h, [&](miba_handle s, int) { return miba_set_state(s, R, T, K, X); });
  if (!h || !h->has_problem) return fail(h, MIBA_ERR_INVALID, "set_problem has not been called");
  if (!R || !T || !K || !X) return fail(h, MIBA_ERR_INVALID, "null state array");
  HIPCHK(h, hipSetDevice(h->device));
  // the device camera is BAL-shaped: one focal length, no principal point (src/BalDataloader.cpp:143-148).  Intrinsics
  // that say otherwise are refused instead of being silently replaced by (f, f, 0, 0) at miba_get_state.
  for (int c = 0; c < h->n_cam; ++c) {
    const double* k = K + 6 * (size_t)c;
    const bool f_differs = k[1] != k[0] && !(std::isnan(k[0]) && std::isnan(k[1]));
    if (h->cam_model == MIBA_MODEL_REFERENCE) {
      if (f_differs) return fail(h, MIBA_ERR_INVALID, "camera %d: camera model 1 needs fx == fy (got %g, %g)", c, k[0], k[1]);
      if (k[2] != h->principal[2 * (size_t)c] || k[3] != h->principal[2 * (size_t)c + 1])
        return fail(h, MIBA_ERR_INVALID, "camera %d: the principal point is part of the uploaded observations; call miba_set_problem to change it", c);
      continue;
    }
    if (f_differs || (k[2] != 0.0 && !std::isnan(k[2])) || (k[3] != 0.0 && !std::isnan(k[3])))
      return fail(h, MIBA_ERR_INVALID, "camera %d: intrinsics (fx, fy, ox, oy) = (%g, %g, %g, %g) are not BAL-shaped (fx == fy, ox == oy == 0)",
                  c, k[0], k[1], k[2], k[3]);
  }
  std::vector<double> cam(9 * (size_t)h->n_cam);
  for (int c = 0; c < h->n_cam; ++c) {
    double* o = cam.data() + 9 * (size_t)c;
    o[0] = R[3 * c]; o[1] = R[3 * c + 1]; o[2] = R[3 * c + 2];
    o[3] = T[3 * c]; o[4] = T[3 * c + 1]; o[5] = T[3 * c + 2];
    o[6] = K[6 * c]; o[7] = K[6 * c + 4]; o[8] = K[6 * c + 5];   // src/BalDataloader.cpp:143-148
  }
  std::vector<double> pts(3 * (size_t)(h->n_pt ? h->n_pt : 1));
  for (int32_t q = 0; q < h->n_pt; ++q) {
    const size_t o = (size_t)h->plan_ref().pt_orig[h->pt_base + q];
    pts[3 * (size_t)q] = X[3 * o]; pts[3 * (size_t)q + 1] = X[3 * o + 1]; pts[3 * (size_t)q + 2] = X[3 * o + 2];
  }
  int rc = upload(h, h->d_cam, cam.data(), cam.size());
  if (rc) return rc;
  rc = upload(h, h->d_X, pts.data(), 3 * (size_t)h->n_pt);
  if (rc) return rc;
  h->jac_valid = false;
  h->scale_valid = false;
  return MIBA_OK;
}

int miba_set_constant_blocks(miba_handle h, const uint8_t* cam_const) {
  if (h && h->owned) return group_run(h, [&](miba_handle s, int) { return miba_set_constant_blocks(s, cam_const); });
  if (!h || !h->has_problem) return fail(h, MIBA_ERR_INVALID, "set_problem has not been called");
  HIPCHK(h, hipSetDevice(h->device));
  std::vector<double> m(9 * (size_t)h->n_cam, 1.0);
  if (cam_const)
    for (int c = 0; c < h->n_cam; ++c)
      for (int k = 0; k < 9; ++k)
        if (cam_const[3 * (size_t)c + k / 3]) m[9 * (size_t)c + k] = 0.0;
  if (h->cam_model == MIBA_MODEL_REFERENCE)   // the intrinsics block of the reference's model is always constant
    for (int c = 0; c < h->n_cam; ++c) m[9 * (size_t)c + 6] = m[9 * (size_t)c + 7] = m[9 * (size_t)c + 8] = 0.0;
  int rc = upload(h, h->d_cfree, m.data(), m.size());
  if (rc) return rc;
  h->scale_valid = false;
  h->jac_valid = false;
  return MIBA_OK;
}

int miba_get_state(miba_handle h, double* R, double* T, double* K, double* X) {
  if (h && h->owned) {
    // On the calling thread, rank after rank (it may be called from inside an iteration callback, while the rank threads
    // wait for the callback's verdict): cameras from rank 0, every rank's own points.
    for (int r = 0; r < h->owned->n; ++r) {
      miba_handle s = h->owned->sub[r];
      const int rc = miba_get_state(s, r ? nullptr : R, r ? nullptr : T, r ? nullptr : K, X);
      if (rc) { h->err = s->err; return rc; }
    }
    return MIBA_OK;
  }
  if (!h || !h->has_problem) return fail(h, MIBA_ERR_INVALID, "set_problem has not been called");
  // called from inside an iteration callback that runs beside the solver (options.async_callbacks): the state of that
  // iteration is in a pinned snapshot; no device call is made (the solver thread is using the device).  With one process per
  // GPU the snapshot holds this rank's points, and only those are written.
  const int snap = h->snap_read;
  std::vector<double> cam(snap >= 0 ? (size_t)0 : 9 * (size_t)h->n_cam);
  const double* cam_src = snap >= 0 ? h->snap_cam[snap] : cam.data();
  if (snap < 0) {
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(cam.data(), h->d_cam, cam.size() * sizeof(double), hipMemcpyDeviceToHost, h->st));
    HIPCHK(h, hipStreamSynchronize(h->st));
  }
  for (int c = 0; c < h->n_cam; ++c) {
    if (!h->plan_ref().cam_active[c]) continue;   // blocks that are not in the problem stay untouched
    const double* o = cam_src + 9 * (size_t)c;
    if (R) { R[3 * c] = o[0]; R[3 * c + 1] = o[1]; R[3 * c + 2] = o[2]; }
    if (T) { T[3 * c] = o[3]; T[3 * c + 1] = o[4]; T[3 * c + 2] = o[5]; }
    if (K && h->cam_model == MIBA_MODEL_BAL) { K[6 * c] = o[6]; K[6 * c + 1] = o[6]; K[6 * c + 4] = o[7]; K[6 * c + 5] = o[8]; }   // ox, oy are not the solver's
  }
  if (!X) return MIBA_OK;
  if (snap >= 0) {
    const double* pts = h->snap_X[snap];
    const int32_t* po = h->plan_ref().pt_orig.data() + h->pt_base;
    plan_parallel_for((int64_t)h->n_pt, [&](int64_t a, int64_t b) {
      for (int64_t q = a; q < b; ++q) {
        const size_t o = (size_t)po[q];
        X[3 * o] = pts[3 * q]; X[3 * o + 1] = pts[3 * q + 1]; X[3 * o + 2] = pts[3 * q + 2];
      }
    });
    return MIBA_OK;
  }
  const size_t n_act = h->plan_ref().pt_orig.size();
  // pinned staging buffer, kept: per-iteration state pulls (update_state_every_iteration) copy at full PCIe rate
  const size_t need = 3 * (n_act ? n_act : 1);
  if (h->h_stage_n < need) {
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    h->h_stage = nullptr; h->h_stage_n = 0;
    HIPCHK(h, hipHostMalloc((void**)&h->h_stage, need * sizeof(double)));
    h->h_stage_n = need;
  }
  double* pts = h->h_stage;
  if (multi_rank(h) && h->cfg.world_size > 1 && !h->grp) {
    // one process per rank: every rank contributes its own points to a zero-filled buffer; sum = gather
    // (the ranks of a single-process group write their own points straight into the caller's array instead)
    if (!h->d_gather) {
      const int rca = dev_alloc(h, &h->d_gather, 3 * (n_act ? n_act : 1));   // kept until the problem is replaced
      if (rca) return rca;
    }
    double* d_all = h->d_gather;
    hipError_t e = hipMemsetAsync(d_all, 0, 3 * n_act * sizeof(double), h->st);
    if (e == hipSuccess && h->n_pt)
      e = hipMemcpyAsync(d_all + 3 * (size_t)h->pt_base, h->d_X, 3 * (size_t)h->n_pt * sizeof(double), hipMemcpyDeviceToDevice, h->st);
    int rcg = MIBA_OK;
    if (e == hipSuccess) rcg = allreduce(h, d_all, 3 * n_act, ncclSum);
    if (e == hipSuccess && rcg == MIBA_OK) e = hipMemcpyAsync(pts, d_all, 3 * n_act * sizeof(double), hipMemcpyDeviceToHost, h->st);
    if (e == hipSuccess) e = hipStreamSynchronize(h->st);
    if (rcg != MIBA_OK) return rcg;
    if (e != hipSuccess) return fail(h, MIBA_ERR_HIP, "point download failed: %s", hipGetErrorString(e));
    const int32_t* po = h->plan_ref().pt_orig.data();
    plan_parallel_for((int64_t)n_act, [&](int64_t a, int64_t b) {
      for (int64_t s = a; s < b; ++s) {
        const size_t o = (size_t)po[s];
        X[3 * o] = pts[3 * s]; X[3 * o + 1] = pts[3 * s + 1]; X[3 * o + 2] = pts[3 * s + 2];
      }
    });
  } else {
    if (h->n_pt) {
      HIPCHK(h, hipMemcpyAsync(pts, h->d_X, 3 * (size_t)h->n_pt * sizeof(double), hipMemcpyDeviceToHost, h->st));
      HIPCHK(h, hipStreamSynchronize(h->st));
    }
    const int32_t* po = h->plan_ref().pt_orig.data() + h->pt_base;
    plan_parallel_for((int64_t)h->n_pt, [&](int64_t a, int64_t b) {
      for (int64_t q = a; q < b; ++q) {
        const size_t o = (size_t)po[q];
        X[3 * o] = pts[3 * q]; X[3 * o + 1] = pts[3 * q + 1]; X[3 * o + 2] = pts[3 * q + 2];
      }
    });
  }
  return MIBA_OK;
}

int miba_evaluate(miba_handle h, double* cost, double* residuals, double* jac_cam, double* jac_pt, double* grad_cam,
                  double* grad_pt) {
  if (h && h->owned) {
    RankGroup& G = *h->owned;
    if (!G.sub[0]->has_problem) return fail(h, MIBA_ERR_INVALID, "set_problem has not been called");
    const Plan& P = G.sub[0]->plan_ref();
    if (residuals) memset(residuals, 0, sizeof(double) * 2 * (size_t)P.n_obs_in);
    if (jac_cam) memset(jac_cam, 0, sizeof(double) * 18 * (size_t)P.n_obs_in);
    if (jac_pt) memset(jac_pt, 0, sizeof(double) * 6 * (size_t)P.n_obs_in);
    if (grad_pt) memset(grad_pt, 0, sizeof(double) * 3 * (size_t)P.n_pt_in);
    // every rank fills its own observations and points; the replicated outputs (cost, camera gradient) are rank 0's,
    // the other ranks write theirs (the same numbers) into scratch
    std::vector<std::vector<double>> gc(G.n);
    std::vector<double> c(G.n, 0.0);
    if (grad_cam) for (int r = 1; r < G.n; ++r) gc[r].resize(9 * (size_t)P.n_cam);
    return group_run(h, [&](miba_handle s, int r) {
      return miba_evaluate(s, r ? &c[r] : cost, residuals, jac_cam, jac_pt, grad_cam ? (r ? gc[r].data() : grad_cam) : nullptr, grad_pt);
    });
  }
  if (!h || !h->has_problem) return fail(h, MIBA_ERR_INVALID, "set_problem has not been called");
  HIPCHK(h, hipSetDevice(h->device));
  miba_options o;
  miba_default_options(&o);
  int rc;
  if (residuals || jac_cam || jac_pt) {
    rc = eval_jac(h, h->d_cam, h->d_X, false);
    h->jac_valid = false;  // d_J now holds the unscaled Jacobian
  } else {
    // cost only: the matrix-free residual pass; the stored Jacobian (208 B per slot) is not even allocated
    const Loss saved_loss = h->loss;
    h->loss = Loss{kLossNone, 1.0};
    rc = eval_cost(h, h->d_cam, h->d_X);
    h->loss = saved_loss;
  }
  if (rc) return rc;
  HIPCHK(h, hipMemcpyAsync(h->h_scal, h->d_sum, S_COUNT * sizeof(double), hipMemcpyDeviceToHost, h->st));
  rc = sync_stream(h);
  if (rc) return rc;
  if (cost) *cost = h->h_scal[S_COST];
  const Plan& P = h->plan_ref();
  if (residuals || jac_cam || jac_pt) {
    if (!h->grp) {   // (the ranks of a group fill one array that the group has cleared)
      if (residuals) memset(residuals, 0, sizeof(double) * 2 * (size_t)P.n_obs_in);
      if (jac_cam) memset(jac_cam, 0, sizeof(double) * 18 * (size_t)P.n_obs_in);
      if (jac_pt) memset(jac_pt, 0, sizeof(double) * 6 * (size_t)P.n_obs_in);
    }
    std::vector<double> J((size_t)kJComp * (size_t)(h->n_slots ? h->n_slots : 1));
    if (h->n_slots) {
      HIPCHK(h, hipMemcpyAsync(J.data(), h->d_J, J.size() * sizeof(double), hipMemcpyDeviceToHost, h->st));
      HIPCHK(h, hipStreamSynchronize(h->st));
    }
    for (int64_t s = 0; s < h->n_slots; ++s) {
      const int64_t ob = P.slot_orig[h->slot_base + s];
      if (ob < 0) continue;
      const double* c = J.data() + (size_t)(s / kChunk) * kJComp * kChunk + (size_t)(s % kChunk);
      if (jac_cam) for (int k = 0; k < 18; ++k) jac_cam[18 * ob + k] = c[(size_t)k * kChunk];
      if (jac_pt) for (int k = 0; k < 6; ++k) jac_pt[6 * ob + k] = c[(size_t)(18 + k) * kChunk];
      if (residuals) { residuals[2 * ob] = c[(size_t)24 * kChunk]; residuals[2 * ob + 1] = c[(size_t)25 * kChunk]; }
    }
  }
  if (grad_cam || grad_pt) {
    rc = unit_scale(h);
    if (rc) return rc;
    h->scale_valid = false;
    // gradient of the plain (unweighted) problem, consistent with the cost, residuals and Jacobians above: a loss left
    // in the handle by an earlier solve must not weight the sums
    const Loss saved_loss = h->loss;
    h->loss = Loss{kLossNone, 1.0};
    rc = schur_init(h, 1e4, &o);   // F'r and E'r do not depend on the radius
    h->loss = saved_loss;
    if (rc) return rc;
    std::vector<double> acc((size_t)kK2W * h->n_cam), gpv(3 * (size_t)(h->n_pt ? h->n_pt : 1));
    HIPCHK(h, hipMemcpyAsync(acc.data(), h->d_camacc, acc.size() * sizeof(double), hipMemcpyDeviceToHost, h->st));
    if (h->n_pt) HIPCHK(h, hipMemcpyAsync(gpv.data(), h->d_gp, 3 * (size_t)h->n_pt * sizeof(double), hipMemcpyDeviceToHost, h->st));
    rc = sync_stream(h);
    if (rc) return rc;
    if (grad_cam)
      for (int c = 0; c < h->n_cam; ++c)
        for (int k = 0; k < 9; ++k) grad_cam[9 * (size_t)c + k] = acc[(size_t)kK2W * c + 54 + k];
    if (grad_pt) {
      if (!h->grp) memset(grad_pt, 0, sizeof(double) * 3 * (size_t)P.n_pt_in);
      for (int32_t q = 0; q < h->n_pt; ++q) {
        const size_t ob = (size_t)P.pt_orig[h->pt_base + q];
        for (int k = 0; k < 3; ++k) grad_pt[3 * ob + k] = gpv[3 * (size_t)q + k];
      }
    }
  }
  return MIBA_OK;
}

int miba_linear_step(miba_handle h, const miba_options* opt, double radius, double* y_cam, double* y_pt, double* rhs,
                     const double* probe, double* s_probe, int32_t* n_cg, double* model_cost_change) {
  if (h && h->owned) {
    RankGroup& G = *h->owned;
    if (!G.sub[0]->has_problem) return fail(h, MIBA_ERR_INVALID, "bad argument / no problem");
    const Plan& P = G.sub[0]->plan_ref();
    if (y_pt) memset(y_pt, 0, sizeof(double) * 3 * (size_t)P.n_pt_in);
    const size_t nc9 = 9 * (size_t)P.n_cam;
    std::vector<std::vector<double>> tmp(G.n);   // replicated camera-sized outputs of the ranks other than 0
    for (int r = 1; r < G.n; ++r) tmp[r].resize(3 * nc9);
    std::vector<int32_t> it(G.n, 0);
    std::vector<double> mc(G.n, 0.0);
    return group_run(h, [&](miba_handle s, int r) {
      double* t = r ? tmp[r].data() : nullptr;
      return miba_linear_step(s, opt, radius, y_cam ? (r ? t : y_cam) : nullptr, y_pt, rhs ? (r ? t + nc9 : rhs) : nullptr, probe,
                              s_probe ? (r ? t + 2 * nc9 : s_probe) : nullptr, r ? &it[r] : n_cg, r ? &mc[r] : model_cost_change);
    });
  }
  if (!h || !h->has_problem || !opt) return fail(h, MIBA_ERR_INVALID, "bad argument / no problem");
  if (!(radius > 0.0)) return fail(h, MIBA_ERR_INVALID, "radius must be positive");
  HIPCHK(h, hipSetDevice(h->device));
  h->scale_valid = false;
  h->jac_valid = false;
  int rc = set_loss(h, opt);
  if (rc) return rc;
  rc = ensure_jacobian(h, opt);
  if (rc) return rc;
  rc = schur_init(h, radius, opt);
  if (rc) return rc;
  const size_t n = 9 * (size_t)h->n_cam;
  int it = 0, st = 0;
  rc = linear_solve(h, opt, &it, &st);
  if (rc) return rc;
  if (n_cg) *n_cg = it;
  rc = backsub_and_candidate(h);
  if (rc) return rc;
  rc = read_scalars(h);
  if (rc) return rc;
  if (h->h_ctl->fail) return fail(h, MIBA_ERR_INVALID, "Cholesky failure in Schur init");
  if (model_cost_change) *model_cost_change = h->h_scal[S_MCC];
  if (y_cam) HIPCHK(h, hipMemcpy(y_cam, h->d_px, n * sizeof(double), hipMemcpyDeviceToHost));
  if (rhs) HIPCHK(h, hipMemcpy(rhs, h->d_rhs, n * sizeof(double), hipMemcpyDeviceToHost));
  if (y_pt) {
    if (!h->grp) memset(y_pt, 0, sizeof(double) * 3 * (size_t)h->plan_ref().n_pt_in);
    std::vector<double> v(3 * (size_t)(h->n_pt ? h->n_pt : 1));
    if (h->n_pt) HIPCHK(h, hipMemcpy(v.data(), h->d_yp, 3 * (size_t)h->n_pt * sizeof(double), hipMemcpyDeviceToHost));
    for (int32_t q = 0; q < h->n_pt; ++q) {
      const size_t ob = (size_t)h->plan_ref().pt_orig[h->pt_base + q];
      for (int k = 0; k < 3; ++k) y_pt[3 * ob + k] = v[3 * (size_t)q + k];
    }
  }
  if (probe && s_probe) {
    HIPCHK(h, hipMemcpy(h->d_probe, probe, n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemsetAsync(&h->d_ctl->done, 0, sizeof(int32_t), h->st));
    rc = schur_matvec(h, h->d_probe, h->d_pq, h->d_ctl, nullptr);
    if (rc) return rc;
    rc = sync_stream(h);
    if (rc) return rc;
    HIPCHK(h, hipMemcpy(s_probe, h->d_pq, n * sizeof(double), hipMemcpyDeviceToHost));
  }
  return MIBA_OK;
}

int miba_get_reduced_system(miba_handle h, double* S_out) {
  if (h && h->owned) return group_run(h, [&](miba_handle s, int) { return miba_get_reduced_system(s, S_out); });
  if (!h || !h->has_problem || !S_out) return fail(h, MIBA_ERR_INVALID, "bad argument / no problem");
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->jac_valid) return fail(h, MIBA_ERR_INVALID, "call miba_linear_step first (Schur blocks not set up)");
  const size_t n = 9 * (size_t)h->n_cam;
  if ((n + kNB - 1) / kNB * kNB > (size_t)kDenseMaxN) return fail(h, MIBA_ERR_INVALID, "too many cameras for the dense reduced system (%d > %d rows)", (int)n, kDenseMaxN);
  int rc = dense_assemble(h);
  if (rc) return rc;
  rc = sync_stream(h);
  if (rc) return rc;
  const size_t N = (size_t)h->dense_N;
  if (h->grp && h->cfg.rank != 0) return MIBA_OK;   // the all-reduced matrix is the same on every rank: rank 0 hands it out
  HIPCHK(h, hipMemcpy2D(S_out, n * sizeof(double), h->d_S, N * sizeof(double), n * sizeof(double), n, hipMemcpyDeviceToHost));
  for (size_t r = 0; r < n; ++r)
    for (size_t c = r + 1; c < n; ++c) S_out[r * n + c] = S_out[c * n + r];
  return MIBA_OK;
}

int miba_dense_spd_solve(miba_handle h, int32_t n, const double* A, const double* b, double* x, int32_t* status) {
  if (h && h->owned) {   // no collective inside: rank 0's device does it
    const int rc = miba_dense_spd_solve(h->owned->sub[0], n, A, b, x, status);
    if (rc) h->err = h->owned->sub[0]->err;
    return rc;
  }
  if (!h || n < 1 || !A || !b || !x || !status) return fail(h, MIBA_ERR_INVALID, "bad argument");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t N = ((size_t)n + kNB - 1) / kNB * kNB;
  if (N > (size_t)kDenseMaxN) return fail(h, MIBA_ERR_INVALID, "n = %d exceeds the dense limit %d", n, kDenseMaxN);
  double *dA = nullptr, *dL = nullptr, *dv = nullptr;   // dv: b, x
  int32_t* df = nullptr;
  auto cleanup = [&]() { (void)hipFree(dA); (void)hipFree(dL); (void)hipFree(dv); (void)hipFree(df); };
  hipError_t e = hipMalloc(&dA, (N + kNB) * N * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&dL, (N / kNB) * (size_t)kLdiag * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&dv, 2 * N * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&df, sizeof(int32_t));
  if (e != hipSuccess) { cleanup(); return fail(h, MIBA_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e)); }
  // identity on the padding, the caller's matrix in the leading n x n, zero rows below
  std::vector<double> pad((N + kNB) * N, 0.0);
  for (size_t r = 0; r < N; ++r) {
    if (r < (size_t)n) memcpy(&pad[r * N], A + r * (size_t)n, (size_t)n * sizeof(double));
    else pad[r * N + r] = 1.0;
  }
  e = hipMemcpy(dA, pad.data(), pad.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dv, b, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
  if (e != hipSuccess) { cleanup(); return fail(h, MIBA_ERR_HIP, "upload failed: %s", hipGetErrorString(e)); }
  int st = 0;
  int rc = dense_factor_solve(h, dA, (int)N, dL, dv, n, dv + N, n, df, &st);
  if (rc == MIBA_OK) {
    e = hipMemcpy(x, dv + N, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(h, MIBA_ERR_HIP, "download failed: %s", hipGetErrorString(e));
  }
  cleanup();
  *status = st;
  return rc;
}

int miba_dense_spd_solve_static(miba_handle h, int32_t n, const double* d_A, const double* d_b, double* d_x, int32_t* d_status) {
  if (h && h->owned) {   // rank 0's device (the pointers must live there)
    const int rc = miba_dense_spd_solve_static(h->owned->sub[0], n, d_A, d_b, d_x, d_status);
    if (rc) h->err = h->owned->sub[0]->err;
    return rc;
  }
  if (!h || n < 1 || !d_A || !d_b || !d_x || !d_status) return fail(h, MIBA_ERR_INVALID, "bad argument");
  const size_t N = ((size_t)n + kNB - 1) / kNB * kNB;
  if (N > (size_t)kStaticMaxN)
    return fail(h, MIBA_ERR_LIMIT, "n = %d exceeds the static scratch of miba_dense_spd_solve_static (n <= %d); use miba_dense_spd_solve", n, kStaticMaxN);
  if (!h->static_scratch) return fail(h, MIBA_ERR_HIP, "static scratch address was not resolved at miba_create");
  HIPCHK(h, hipSetDevice(h->device));
  // everything below goes through the device address looked up at miba_create, never through the array's name
  double* S = h->static_scratch;
  double* Ldiag = h->static_scratch + kStaticLdiagOff;
  int32_t* flag = (int32_t*)(h->static_scratch + kStaticFlagOff);
  // the array keeps the last call's contents: the whole [(N + 64) x N] image is rewritten (matrix, identity on the
  // padding, zero rows below), the factors are rewritten block by block by the panel kernel, the flag is cleared
  hipLaunchKernelGGL(k_dense_pad, dim3(nblk((N + kNB) * N, 256)), dim3(256), 0, h->st, n, (int)N, d_A, S);
  HIPCHK(h, hipGetLastError());
  int rc = dense_factor_enqueue(h, S, (int)N, Ldiag, d_b, n, d_x, n, flag);
  if (rc) return rc;
  hipLaunchKernelGGL(k_dense_status, dim3(1), dim3(1), 0, h->st, (const int32_t*)flag, d_status);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->st));
  return MIBA_OK;
}

int miba_time_matvec(miba_handle h, const double* probe, int32_t reps, double* avg_ms) {
  if (h && h->owned) {
    std::vector<double> ms(h->owned->n, 0.0);
    return group_run(h, [&](miba_handle s, int r) { return miba_time_matvec(s, probe, reps, r ? &ms[r] : avg_ms); });
  }
  if (!h || !h->has_problem || !probe || !avg_ms || reps < 1) return fail(h, MIBA_ERR_INVALID, "bad argument / no problem");
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->jac_valid) return fail(h, MIBA_ERR_INVALID, "call miba_linear_step first (Jacobian / Schur blocks not set up)");
  const size_t n = 9 * (size_t)h->n_cam;
  HIPCHK(h, hipMemcpy(h->d_probe, probe, n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(h, hipMemsetAsync(&h->d_ctl->done, 0, sizeof(int32_t), h->st));
  int rc = schur_matvec(h, h->d_probe, h->d_pq, h->d_ctl, nullptr);  // warm-up
  if (rc) return rc;
  rc = sync_stream(h);
  if (rc) return rc;
  hipEvent_t a, b;
  HIPCHK(h, hipEventCreate(&a));
  HIPCHK(h, hipEventCreate(&b));
  HIPCHK(h, hipEventRecord(a, h->st));
  for (int i = 0; i < reps; ++i) {
    rc = schur_matvec(h, h->d_probe, h->d_pq, h->d_ctl, nullptr);
    if (rc) return rc;
  }
  HIPCHK(h, hipEventRecord(b, h->st));
  rc = sync_stream(h);
  if (rc) return rc;
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  *avg_ms = (double)ms / reps;
#ifdef MIBA_CM_STAMPS
  {
    std::vector<unsigned long long> all((size_t)kCmStampBlocks * 4 * 16);
    if (hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(g_cm_stamps), all.size() * 8) == hipSuccess) {
      unsigned long long st[16] = {0};
      for (size_t i = 0; i < all.size(); ++i) st[i & 15] += all[i];
      const double nsub = (double)std::max(1ull, st[5]), nw = (double)std::max(1ull, st[9]);
      fprintf(stderr, "[cm stamps] cycles per wave and sub-tile: phase1 %.0f  wait2 %.0f  phase1.5 %.0f  wait3 %.0f  phase2 %.0f | per wave and tile: "
              "prologue %.0f  flush %.0f  total %.0f (sub-tile waves %.0f, tile waves %.0f)\n", st[0] / nsub, st[1] / nsub, st[2] / nsub, st[3] / nsub,
              st[4] / nsub, st[6] / nw, st[7] / nw, st[8] / nw, nsub, nw);
      std::fill(all.begin(), all.end(), 0ull);
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cm_stamps), all.data(), all.size() * 8);
    }
  }
#endif
  return MIBA_OK;
}

// Iteration callbacks beside the solver (options.async_callbacks): one worker thread per solve runs the caller's callback for
// iteration k while the solver works on iteration k + 1.  Before the call it waits for the snapshot of iteration k on every rank of
// the handle and points miba_get_state at it.  One job at a time: the solver submits iteration k after the callback of k - 1
// has returned, and takes that callback's verdict then (a stop therefore takes effect one iteration late).
struct CallbackWorker {
  miba_handle h;
  miba_iteration_callback cb;
  void* user;
  std::mutex m;
  std::condition_variable cv;
  bool has_job = false, busy = false, quit = false;
  miba_iteration job{};
  int verdict = MIBA_SOLVER_CONTINUE;
  std::thread th;   // declared last and started in the constructor's body: run() uses every member above from its first instruction
  CallbackWorker(miba_handle h_, miba_iteration_callback cb_, void* user_) : h(h_), cb(cb_), user(user_) { th = std::thread([this] { run(); }); }
  ~CallbackWorker() {
    { std::lock_guard<std::mutex> l(m); quit = true; }
    cv.notify_all();
    th.join();
  }
  void run() {
    for (;;) {
      miba_iteration it;
      {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return quit || has_job; });
        if (!has_job) return;
        it = job;
        has_job = false;
      }
      std::vector<miba_handle> ranks;
      if (h->grp) ranks = h->grp->sub; else ranks.push_back(h);
      bool ok = true;
      for (miba_handle s : ranks) {
        // the rank's solver thread publishes the snapshot of iteration k at the top of its loop
        while (s->snap_seq.load(std::memory_order_acquire) < it.iteration) {
          { std::lock_guard<std::mutex> l(m); if (quit) { ok = false; break; } }
          if (h->grp && h->grp->abort.load(std::memory_order_acquire)) { ok = false; break; }
          sched_yield();
        }
        if (!ok) break;
        const int b = s->snap_buf_of[it.iteration & 1];
        (void)hipSetDevice(s->device);
        if (b < 0 || hipEventSynchronize(s->ev_snap[b]) != hipSuccess) { ok = false; break; }
        s->snap_read = b;
      }
      int r = MIBA_SOLVER_ABORT;
      if (ok) r = cb(&it, user);
      for (miba_handle s : ranks) s->snap_read = -1;
      {
        std::lock_guard<std::mutex> l(m);
        verdict = r;
        busy = false;
      }
      cv.notify_all();
    }
  }
  // waits for the callback in flight (if any) and returns its verdict
  int wait() {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return !busy && !has_job; });
    const int r = verdict;
    verdict = MIBA_SOLVER_CONTINUE;
    return r;
  }
  void submit(const miba_iteration& it) {
    { std::lock_guard<std::mutex> l(m); job = it; has_job = true; busy = true; }
    cv.notify_all();
  }
};

int miba_solve(miba_handle h, const miba_options* opt, miba_iteration_callback cb, void* user, miba_summary* summary) {
  if (h && h->owned) {
    // every rank runs the same LM loop on its own thread; the callback belongs to rank 0 (the caller's thread), which
    // hands its verdict to the others once per iteration (group_exchange_stop)
    std::vector<miba_summary> sm(h->owned->n);
    h->owned->async_cb.store(cb && opt && opt->async_callbacks ? 1 : 0);
    for (miba_handle s : h->owned->sub) { s->snap_seq.store(-1); s->snap_last = -1; s->snap_buf_of[0] = s->snap_buf_of[1] = -1; }
    const int rcg = group_run(h, [&](miba_handle s, int r) { return miba_solve(s, opt, r ? nullptr : cb, r ? nullptr : user, r ? &sm[r] : summary); });
    h->owned->async_cb.store(0);
    return rcg;
  }
  if (!h || !h->has_problem || !opt) return fail(h, MIBA_ERR_INVALID, "bad argument / no problem");
  HIPCHK(h, hipSetDevice(h->device));
  const miba_options o = *opt;
  const double t_start = now_s();
  double t_callbacks = 0.0, t_linear = 0.0, t_jac = 0.0, t_res = 0.0;
  h->iters.clear();
  miba_summary S;
  memset(&S, 0, sizeof(S));
  S.termination_type = MIBA_NO_CONVERGENCE;
  S.num_residual_blocks = (int32_t)std::min<int64_t>(h->n_valid_global, INT32_MAX);
  S.num_active_cameras = h->plan_ref().n_active_cam;
  S.num_active_points = (int32_t)h->plan_ref().pt_orig.size();
  snprintf(S.message, sizeof(S.message), "Maximum number of iterations reached.");
  int rc = set_loss(h, &o);
  if (rc) return rc;
  {
    int kind = kExactNone;
    rc = exact_kind(h, &o, &kind);   // (first use at this size: the symbolic phase of the tile-sparse factorisation)
    if (rc) return rc;
    S.exact_linear_solver = kind != kExactNone ? 1 : 0;
  }

  // best-state buffers: d_cam_cand / d_Xc double as candidates, so keep the best state on the host side of
  // the swap: x (d_cam, d_X) is always the last ACCEPTED state, which in the monotonic algorithm is also
  // the minimum-cost state.

  // IterationZero
  double x_norm2 = 0.0;
  rc = state_norm2(h, &x_norm2);
  if (rc) return rc;
  double x_norm = sqrt(x_norm2);
  h->scale_valid = false;
  h->jac_valid = false;
  double tj = now_s();
  rc = ensure_jacobian(h, &o);
  if (rc) return rc;
  double radius = o.initial_trust_region_radius;
  double decrease_factor = 2.0;
  int consecutive_invalid = 0;
  rc = schur_init(h, radius, &o);
  if (rc) return rc;
  rc = read_scalars(h);
  if (rc) return rc;
  t_jac += now_s() - tj;
  double x_cost = h->h_scal[S_COST];
  S.initial_cost = x_cost;
  miba_iteration it;
  memset(&it, 0, sizeof(it));
  it.iteration = 0;
  it.cost = x_cost;
  it.gradient_max_norm = fmax(h->h_scal[S_COUNT + S_GMAX_C], h->h_scal[S_COUNT + S_GMAX_P]);
  it.gradient_norm = sqrt(h->h_scal[S_G2_C] + h->h_scal[S_G2_P]);
  it.trust_region_radius = radius;
  it.eta = o.eta;
  it.iteration_time_in_seconds = now_s() - t_start;
  it.cumulative_time_in_seconds = it.iteration_time_in_seconds;
  bool schur_ready = true;   // schur_init done for (current J, current radius)
  bool user_stop = false;
  // callbacks beside the solver: this rank runs the worker (it has the callback), every rank of the handle takes snapshots
  const bool async_cb = cb && o.async_callbacks;
  const bool snap_on = async_cb || (h->grp && h->grp->async_cb.load());
  if (!h->grp) { h->snap_seq.store(-1); h->snap_last = -1; h->snap_buf_of[0] = h->snap_buf_of[1] = -1; }
  std::unique_ptr<CallbackWorker> worker;
  if (async_cb) worker.reset(new CallbackWorker(h, cb, user));

  for (;;) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (it.iteration > 0) {
      if (it.step_is_successful) S.num_successful_steps++; else S.num_unsuccessful_steps++;
    }
    it.trust_region_radius = radius;
    h->iters.push_back(it);
    if (o.minimizer_progress_to_stdout && h->cfg.rank == 0) print_progress(it, it.iteration == 0);
    // verdicts that depend on the caller or on a clock (callback result, time limit): in a single-process group they are
    // rank 0's, handed to the other ranks here, so that every rank leaves the loop in the same iteration
    int stop = 0;
    if (snap_on) {
      rc = snapshot_enqueue(h, it.iteration, it.iteration == 0 || it.step_is_successful);
      if (rc) return rc;
    }
    if (cb) {
      const double tc = now_s();
      int r;
      if (async_cb) {
        r = worker->wait();   // the callback of the iteration before (if it is still running, the solver waits here)
        if (r == MIBA_SOLVER_CONTINUE) worker->submit(it);
      } else r = cb(&it, user);
      t_callbacks += now_s() - tc;
      if (r == MIBA_SOLVER_ABORT) stop = 1;
      else if (r == MIBA_SOLVER_TERMINATE_SUCCESSFULLY) stop = 2;
    }
    if (!stop && now_s() - t_start - t_callbacks >= o.max_solver_time_in_seconds) stop = 3;
    rc = group_exchange_stop(h, it.iteration, &stop);
    if (rc) return rc;
    if (stop == 1) { S.termination_type = MIBA_USER_FAILURE; snprintf(S.message, sizeof(S.message), "User callback returned SOLVER_ABORT."); user_stop = true; }
    else if (stop == 2) { S.termination_type = MIBA_USER_SUCCESS; snprintf(S.message, sizeof(S.message), "User callback returned SOLVER_TERMINATE_SUCCESSFULLY."); user_stop = true; }
    if (user_stop) break;
    if (stop == 3) { snprintf(S.message, sizeof(S.message), "Maximum solver time reached."); break; }
    if (it.iteration >= o.max_num_iterations) break;
    if (it.gradient_max_norm <= o.gradient_tolerance) {
      S.termination_type = MIBA_CONVERGENCE;
      snprintf(S.message, sizeof(S.message), "Gradient tolerance reached.");
      break;
    }
    if (radius < o.min_trust_region_radius) {
      S.termination_type = MIBA_CONVERGENCE;
      snprintf(S.message, sizeof(S.message), "Minimum trust region radius reached.");
      break;
    }
    const double t_it = now_s();
    const miba_iteration prev = it;
    memset(&it, 0, sizeof(it));
    it.iteration = prev.iteration + 1;
    it.eta = o.eta;

    // ComputeTrustRegionStep
    const double tl = now_s();
    if (!schur_ready) {
      rc = schur_init(h, radius, &o);
      if (rc) return rc;
    }
    schur_ready = false;
    int n_cg = 0, cg_status = 0;
    rc = linear_solve(h, &o, &n_cg, &cg_status);
    if (rc) return rc;
    it.linear_solver_iterations = n_cg;
    S.num_linear_solver_iterations += n_cg;
    // The candidate cost is evaluated before the model cost change has been looked at (the step is almost always
    // valid): its partial sum sits next to those of the back-substitution, so one all-reduce of four scalars and one
    // host read serve both.
    if (snap_on && (rc = snapshot_guard(h))) return rc;
    rc = backsub_and_candidate(h, false, true);
    if (rc) return rc;
    t_linear += now_s() - tl;
    const double tr = now_s();
    if (!h->cand_cost_done) {   // (else the back-substitution has taken the candidate's cost along)
      rc = eval_cost(h, h->d_cam_cand, h->d_Xc, false);
      if (rc) return rc;
    }
    static_assert(S_COST + 1 == S_MCC && S_MCC + 1 == S_DP2 && S_DP2 + 1 == S_XP2, "scalar slots must be adjacent");
    rc = allreduce(h, h->d_sum + S_COST, 4, ncclSum);
    if (rc) return rc;
    rc = read_scalars(h);
    if (rc) return rc;
    t_res += now_s() - tr;
    const double mcc = h->h_scal[S_MCC];
    const bool ls_failed = (cg_status == 2) || h->h_ctl->fail || !std::isfinite(mcc);
    it.model_cost_change = ls_failed ? 0.0 : mcc;
    it.step_is_valid = (!ls_failed && mcc > 0.0);
    if (!it.step_is_valid) {
      // HandleInvalidStep
      it.cost = x_cost; it.gradient_max_norm = prev.gradient_max_norm; it.gradient_norm = prev.gradient_norm;
      if (++consecutive_invalid >= o.max_num_consecutive_invalid_steps) {
        S.termination_type = MIBA_FAILURE;
        snprintf(S.message, sizeof(S.message), "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps: %d",
                 o.max_num_consecutive_invalid_steps);
        // Ceres returns from here without appending the unfinished iteration to summary.iterations or counting a step
        break;
      }
      radius *= 0.5;  // StepIsInvalid
      it.iteration_time_in_seconds = now_s() - t_it;
      it.cumulative_time_in_seconds = now_s() - t_start - t_callbacks;
      continue;
    }
    consecutive_invalid = 0;
    double cand_cost = h->h_scal[S_COST];
    if (!std::isfinite(cand_cost)) cand_cost = 1.7976931348623157e308;
    it.step_norm = sqrt(h->h_scal[S_DC2] + h->h_scal[S_DP2]);
    const double cand_norm = sqrt(h->h_scal[S_XC2] + h->h_scal[S_XP2]);
    it.cost_change = x_cost - cand_cost;
    it.relative_decrease = it.cost_change / mcc;
    bool converged = false;
    if (it.step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) {
      S.termination_type = MIBA_CONVERGENCE;
      snprintf(S.message, sizeof(S.message), "Parameter tolerance reached. Relative step_norm: %e <= %e.",
               it.step_norm / (x_norm + o.parameter_tolerance), o.parameter_tolerance);
      converged = true;
    } else if (fabs(it.cost_change) <= o.function_tolerance * x_cost) {
      S.termination_type = MIBA_CONVERGENCE;
      snprintf(S.message, sizeof(S.message), "Function tolerance reached. |cost_change|/cost: %e <= %e",
               fabs(it.cost_change) / x_cost, o.function_tolerance);
      converged = true;
    }
    if (converged) break;   // as Ceres: the iteration that met the tolerance is neither recorded nor counted
    if (it.relative_decrease > o.min_relative_decrease) {
      // HandleSuccessfulStep: x <- candidate, re-evaluate
      std::swap(h->d_cam, h->d_cam_cand);
      std::swap(h->d_X, h->d_Xc);
      x_norm = cand_norm;
      tj = now_s();
      h->jac_valid = false;
      rc = ensure_jacobian(h, &o);
      if (rc) return rc;
      // StepAccepted
      radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * it.relative_decrease - 1.0, 3));
      radius = fmin(o.max_trust_region_radius, radius);
      decrease_factor = 2.0;
      rc = schur_init(h, radius, &o);
      if (rc) return rc;
      schur_ready = true;
      rc = read_scalars(h);
      if (rc) return rc;
      t_jac += now_s() - tj;
      x_cost = h->h_scal[S_COST];
      it.cost = x_cost;
      it.gradient_max_norm = fmax(h->h_scal[S_COUNT + S_GMAX_C], h->h_scal[S_COUNT + S_GMAX_P]);
      it.gradient_norm = sqrt(h->h_scal[S_G2_C] + h->h_scal[S_G2_P]);
      it.step_is_successful = 1;
    } else {
      it.step_is_successful = 0;
      it.cost = cand_cost;
      it.gradient_max_norm = prev.gradient_max_norm; it.gradient_norm = prev.gradient_norm;
      radius = radius / decrease_factor;  // StepRejected
      decrease_factor *= 2.0;
    }
    it.iteration_time_in_seconds = now_s() - t_it;
    it.cumulative_time_in_seconds = now_s() - t_start - t_callbacks;
  }
  if (worker) {
    // the last callback submitted: its verdict can only turn the termination into the caller's
    const double tc = now_s();
    const int r = worker->wait();
    t_callbacks += now_s() - tc;
    if (!user_stop && r == MIBA_SOLVER_ABORT) { S.termination_type = MIBA_USER_FAILURE; snprintf(S.message, sizeof(S.message), "User callback returned SOLVER_ABORT."); }
    else if (!user_stop && r == MIBA_SOLVER_TERMINATE_SUCCESSFULLY) { S.termination_type = MIBA_USER_SUCCESS; snprintf(S.message, sizeof(S.message), "User callback returned SOLVER_TERMINATE_SUCCESSFULLY."); }
    worker.reset();
  }
  rc = sync_stream(h);
  if (rc) return rc;
  S.num_iterations = S.num_successful_steps + S.num_unsuccessful_steps;
  S.final_cost = x_cost;
  S.total_time_in_seconds = now_s() - t_start - t_callbacks;
  S.linear_solver_time_in_seconds = t_linear;
  S.jacobian_time_in_seconds = t_jac;
  S.residual_time_in_seconds = t_res;
  if (summary) *summary = S;
  return MIBA_OK;
}

int miba_get_iterations(miba_handle h, miba_iteration* out, int32_t max, int32_t* n) {
  if (!h) return fail(h, MIBA_ERR_INVALID, "null handle");
  if (h->owned) return miba_get_iterations(h->owned->sub[0], out, max, n);   // identical on every rank
  if (n) *n = (int32_t)h->iters.size();
  if (out)
    for (int32_t i = 0; i < max && i < (int32_t)h->iters.size(); ++i) out[i] = h->iters[i];
  return MIBA_OK;
}

int miba_get_kernel_times(miba_handle h, miba_kernel_times* out) {
  if (!h || !out) return fail(h, MIBA_ERR_INVALID, "null argument");
  if (h->owned) return miba_get_kernel_times(h->owned->sub[0], out);   // rank 0's shard
  if (h->st) { (void)hipStreamSynchronize(h->st); drain_events(h); }
  *out = h->kt;
  out->n_slots = h->n_slots;
  out->n_obs_local = h->n_obs_local;
  out->n_pt_local = h->n_pt;
  out->n_tiles = h->n_tiles;
  out->n_cam = h->n_cam;
  out->product_layout = h->k3_cm;
  out->pcg_launches = h->pcg_launches;
  out->pcg_exchange = h->pcg_exchange;
  out->product_clock_mhz = 0.0;
  if (h->has_problem && h->d_clk) {
    // shader cycles / 10 ns ticks between the start and the end of one workgroup of the last product launch that did work
    unsigned long long c[4] = {0, 0, 0, 0};
    if (hipMemcpy(c, h->d_clk, sizeof(c), hipMemcpyDeviceToHost) == hipSuccess && c[3] > c[1] && c[2] > c[0])
      out->product_clock_mhz = 100.0 * (double)(c[2] - c[0]) / (double)(c[3] - c[1]);
  }
  return MIBA_OK;
}

int miba_reset_kernel_times(miba_handle h) {
  if (!h) return fail(h, MIBA_ERR_INVALID, "null handle");
  if (h->owned) { for (miba_handle s : h->owned->sub) (void)miba_reset_kernel_times(s); return MIBA_OK; }
  if (h->st) { (void)hipStreamSynchronize(h->st); drain_events(h); }
  memset(&h->kt, 0, sizeof(h->kt));
  return MIBA_OK;
}

}  // extern "C"
