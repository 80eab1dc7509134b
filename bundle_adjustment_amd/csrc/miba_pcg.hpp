// miba_pcg.hpp — the tail of one PCG iteration as ONE kernel (ConjugateGradientsSolver::Solve).
//
// One iteration used to be four launches: direction (pcg_chkdir), product over the tiles (k_mf_matvec), camera-side sum
// (k_cam_reduce_mf), vector update (pcg_update); the three small ones cost 33-36 us of kernel time at Final-13682 and all of
// the time on small problems.  pcg_tail does everything after the tiles:
//   A  per camera: sum of the slab rows the tiles have left, camera-side transforms, q = S p, partial p.q
//   -- grid barrier --
//   B  alpha; x += alpha p; r -= alpha q; partial Q; z = M^-1 r; partial r.z
//   -- grid barrier --
//   C  end-of-iteration test (Q tolerance, limits), beta, p = z + beta p, control block of the next iteration
// on a grid that is resident as a whole (at most one workgroup of kTailThreads per CU), with the barrier on one device-scope
// counter.  Every workgroup derives the scalars from the same partial arrays in the same order, so all take the same branch.
// The only data that cross workgroups inside the kernel are those partial sums; they are written and read with device-scope
// atomic stores and loads, so the barrier needs no release / acquire CACHE maintenance: on gfx950 each such fence writes back
// or invalidates the L2 of the XCD, and the fences of a grid serialise (measured, tools/micro/grid_barrier_bench.hip: 34 us
// per barrier for 256 workgroups with them, 3.6 us without; 2.0 us for 128 workgroups).  What the barrier does need is
// ORDER: the partial-sum store and the arrival atomic go to different L2 channels and may complete out of order, so the
// arriving thread waits for its stores to be acknowledged (s_waitcnt vmcnt(0)) before it arrives.
// A barrier that is not reached within two seconds (the grid was not resident after all) ends the solve with a failure status
// instead of hanging.  With several ranks under RCCL or the host transport the camera-side sum stays a kernel of its own (the
// all-reduce sits behind it) and the tail starts at B.  With several ranks of ONE process under the peer transport the exchange is
// part of this kernel (REDUCE == 2): phase A pushes this rank's share of q straight into every rank's
// fine-grained exchange buffer, the last workgroup to finish raises the rank's flag at every peer, everybody waits for all
// flags and sums the N shares of its cameras in rank order (the same bits on every rank); B and C follow as with one rank:
// one launch per PCG iteration after the tiles, no collective call, one cross-device wait.
#pragma once

#include "miba_comm.hpp"
#include "miba_mf.hpp"

namespace miba {

// 512 threads, eight waves: 256 workgroups of them cover Final-13682's 1,955 wave records in one round, and against sixteen
// waves the launch fills the device sooner and the sums over the workgroup are shorter (30.0 -> 27.3 us at Final-13682, 30.7 ->
// 26.7 us per PCG iteration at trafalgar-257).  Sixteen waves keep one round up to 28,672 cameras instead of 14,336.
#ifndef MIBA_TAIL_THREADS
#define MIBA_TAIL_THREADS 512
#endif
constexpr int kTailThreads = MIBA_TAIL_THREADS;
constexpr int kTailWaves = kTailThreads / 64;
constexpr int kTailCams = 7 * kTailWaves;   // cameras per workgroup round: 7 per wave, nine lanes each
constexpr int kTailBatch = 16;              // slab loads in flight per lane (112 rows per wave and batch)

// sum over the workgroup, the same value in every thread; s16: kTailWaves doubles of LDS
__device__ __forceinline__ double tail_block_sum(double v, double* s16) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s16[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < kTailWaves; ++i) t += s16[i];
  return t;
}
// partial sums cross workgroups: device-scope stores and loads (see the head of the file)
__device__ __forceinline__ void tail_put(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double tail_get(const double* p) {
  return __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double tail_sum_partials(const double* part, int n, double* s16) {
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += kTailThreads) v += tail_get(part + i);
  return tail_block_sum(v, s16);
}

// All workgroups of the grid meet.  Two levels of counters that only grow (the host clears them before a solve): sixteen
// workgroups share a group counter, the last of them to arrive adds one to the top counter, everybody waits for the top
// counter to reach the number of groups times the number of barriers so far.  (Arrivals on ONE counter cost 12 ns each, one
// after the other: 3.6 us for 256 workgroups in the microbenchmark, 4.5-6 us inside this kernel; the group counters sit in
// different cache lines and take their arrivals side by side.)  Thread 0 is also the thread that has stored the workgroup's
// partial sums: it waits for those stores to be acknowledged before it arrives -- an explicit s_waitcnt vmcnt(0): outside
// tgsplit mode a workgroup-scope release fence compiles to s_waitcnt lgkmcnt(0) only, which does not cover global stores; the
// fence stays as the compiler barrier.  Returns false when the others did not arrive in time.
constexpr int kBarGroup = 16;
constexpr int kBarStride = 32;   // unsigneds between counters (128 bytes)
__device__ __forceinline__ bool tail_grid_barrier(unsigned* bar, unsigned n_blocks, int* s_ok, long long timeout_ticks) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned g = blockIdx.x / kBarGroup, n_groups = (n_blocks + kBarGroup - 1) / kBarGroup;
    const unsigned size_g = min((unsigned)kBarGroup, n_blocks - g * kBarGroup);
    const unsigned old = __hip_atomic_fetch_add(bar + (1 + g) * kBarStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned round = old / size_g;   // barriers this group has completed before
    if (old % size_g == size_g - 1) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = (round + 1u) * n_groups;
    const long long t0 = wall_clock64();
    int ok = 1;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > timeout_ticks) { ok = 0; break; }   // (2 s of the 100 MHz counter by default)
    }
    *s_ok = ok;
  }
  __syncthreads();
  return *s_ok != 0;
}
constexpr int kBarWords = (1 + 64) * kBarStride;   // top counter + up to 64 groups (1024 workgroups)

// Camera-side transforms of the summed raw slab row (v0..v5 = its rotation and translation parts, own = entry k) as seven
// coefficients per lane, r = c0 v0 + ... + c5 v5 + c6 own, loaded BEFORE the rows are summed so that nothing waits on memory
// afterwards.  The transforms of k_cam_reduce_mf: BAL camera: (M' c)_k | (N a)_i (a itself at small angles) | i_f | f i_k1 |
// f i_k2.  The reference's functor (flag 2): rotation part M~'(t x a - c) = (m x t).a - m.c with m = column k of M~ and
// t = -R~' t~, translation part -a, intrinsics constant.
__device__ __forceinline__ void tail_cam_coef(int k, const double* __restrict__ T, const double* __restrict__ M, double* c /*[7]*/) {
#pragma unroll
  for (int m = 0; m < 7; ++m) c[m] = 0.0;
  const double flag = M[9];
  if (flag == 2.0) {
    if (k < 3) {
      const double t0 = -(T[0] * T[9] + T[3] * T[10] + T[6] * T[11]);
      const double t1 = -(T[1] * T[9] + T[4] * T[10] + T[7] * T[11]);
      const double t2 = -(T[2] * T[9] + T[5] * T[10] + T[8] * T[11]);
      const double m0 = M[k], m1 = M[3 + k], m2 = M[6 + k];
      c[0] = -m0; c[1] = -m1; c[2] = -m2;
      c[3] = m1 * t2 - m2 * t1; c[4] = m2 * t0 - m0 * t2; c[5] = m0 * t1 - m1 * t0;
    } else if (k < 6) c[6] = -1.0;
  } else if (k < 3) { c[0] = M[k]; c[1] = M[3 + k]; c[2] = M[6 + k]; }
  else if (k < 6) {
    const int i = k - 3;
    if (flag == 1.0) c[6] = 1.0;
    else { c[3] = T[3 * i]; c[4] = T[3 * i + 1]; c[5] = T[3 * i + 2]; }
  } else if (k == 6) c[6] = 1.0;
  else c[6] = T[12];
}

// a barrier timed out: the solve ends as a failure in both control blocks (bit 16 of fail tells the host why)
__device__ __forceinline__ void tail_timed_out(Ctl* ctl) {
  for (int b = 0; b < 2; ++b) { ctl[b].done = 1; ctl[b].status = 2; ctl[b].fail |= 16; }
}

// Which cameras a wave takes (built on the host when the problem is set): consecutive cameras, at most seven, cut so that the
// waves have about the same number of slab rows.  Record: {first camera, number of cameras, first slab row of camera 0..7 (the
// entries past the last camera repeat its end), 0, 0}: everything phase A needs to address its rows in ONE load.
constexpr int kTailRec = 12;

struct TailArgs {
  int j;            // PCG iteration (1-based) this launch completes
  int n_cam;
  const int32_t* wave_rec; int n_vw;   // camera ranges of the waves (kTailRec ints each, see below) and their number
  // camera-side sum (phase A)
  const double* slab; const double* ct; const double* cm; const double* sc; const double* d2;
  double diag_w;
  // vectors
  double* p; double* q; const double* b; const double* Minv; double* x; double* r; double* z;
  // scalars
  Ctl* ctl;            // [2], C_k in ctl[k & 1]
  double* part_pq; int n_pq;   // partials of p.q: this grid's (phase A) or, with several ranks, the camera reduction's
  double* part_q;      // partials of Q, one per workgroup
  double* part_rho0; double* part_rho1;   // partials of rho_k in part_rho[k & 1]
  double q_tol; int min_it, max_it;
  int a_on_x;          // MODE 3: the product's input is x (residual refresh), not p
  unsigned* bar;       // barrier counters
  long long bar_timeout;   // ticks of the 100 MHz counter a barrier waits before it gives up (2 s; tests shorten it)
  // several ranks: phases B and C take cameras 7 vw .. 7 vw + 6 per wave (the same on every rank, so that the partial sums the
  // ranks form are the same numbers in the same order) instead of the wave records, whose cuts follow the rank's own slab rows
  int uniform;
  // REDUCE == 2: the exchange of q between the ranks of one process (peer transport)
  double* xstage[kMaxRanks];      // rank j's exchange buffer [2 (parity of epoch)][n_ranks][9 n_cam], fine-grained, peer-mapped
  PeerSignals* xsig[kMaxRanks];   // rank j's signal block
  int n_ranks, rank;
  uint32_t epoch;                 // exchanges so far on this handle (the same on every rank)
  int split;           // 0: one launch.  The form WITHOUT grid barriers (a device shared with other work, where the grid may not be
                       // resident as a whole): phase A as pcg_tail<3>, then 1: this launch ends after phase B, 2: this launch is
                       // phase C (scalars recomputed from the partial sums, vectors from memory)
  long long* dbg;      // MIBA_TAIL_DEBUG: [workgroup][8] time stamps of the phases (100 MHz counter), else null
};
#define TAIL_STAMP(n) do { if (a.dbg && threadIdx.x == 0) a.dbg[blockIdx.x * 8 + n] = wall_clock64(); } while (0)

// two partial-sum arrays in one pass (one trip to memory instead of two)
__device__ __forceinline__ void tail_sum_partials2(const double* pa, int na, const double* pb, int nb, double* s32, double& sa, double& sb) {
  double va = 0.0, vb = 0.0;
  for (int i = threadIdx.x; i < na; i += kTailThreads) va += tail_get(pa + i);
  for (int i = threadIdx.x; i < nb; i += kTailThreads) vb += tail_get(pb + i);
  va = wave_sum(va);
  vb = wave_sum(vb);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { s32[threadIdx.x >> 6] = va; s32[kTailWaves + (threadIdx.x >> 6)] = vb; }
  __syncthreads();
  sa = 0.0; sb = 0.0;
#pragma unroll
  for (int i = 0; i < kTailWaves; ++i) { sa += s32[i]; sb += s32[kTailWaves + i]; }
}

// MODE 0: the whole tail of iteration j.
// MODE 1: residual-reset iteration, first half: A, alpha, x += alpha p (r is rebuilt from S x by MODE 2 after another product).
// MODE 2: second half: A on q = S x (no p.q), r = b - q, Q, z, r.z, then C.
// MODE 3: phase A alone (several ranks: the all-reduce of q and of the p.q partials behind it follows, then the tail without
//         REDUCE): q = S v of this rank's slab rows, v = x when a_on_x is set (no partials then), else p.
// REDUCE 1: phase A is part of this launch (one rank); 0: q and the p.q partials are there already (several ranks, the all-reduce
// has run); 2: phase A and the exchange with the other ranks of the process are part of this launch.
//
// Lane (sub, k) of a wave owns entry k of camera cam0 + sub (cpw cameras per wave, nine lanes each).  When one round covers all
// cameras (single: up to 7 x kTailWaves x 256 = 14,336 of them) the entries of x, p, r, b and the row of M^-1 are loaded before the first
// barrier and stay in registers through all three phases: after a barrier only the partial sums are fetched.
//
// Phase A reads the slab rows of the wave's cameras as ONE contiguous range (the rows of consecutive cameras follow each other):
// lane (g, k) takes rows g, g + 7, g + 14, ... so that a load instruction covers 504 contiguous bytes whatever the row counts
// of the cameras are; a lane's rows run through the cameras in order, and it hands its sum to LDS each time it crosses into
// the next camera; the camera's lanes then add the seven partial sums in a fixed order.
template <int MODE, int REDUCE>
__global__ __launch_bounds__(kTailThreads) void pcg_tail(const TailArgs a) {
  __shared__ double s32[2 * kTailWaves];
  __shared__ int s_ok;
  __shared__ int s_bnd[kTailWaves * 8];
  __shared__ double s_wp[REDUCE ? kTailWaves * 7 * 64 : 1];
  const int j = a.j;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int sub = lane / 9, k = lane - 9 * sub;
  const int nb = gridDim.x;
  // wave records vw0, vw0 + n_waves, ...: consecutive records go to different workgroups, so that every workgroup has work
  // when there are fewer records than waves
  const int vw0 = wv * nb + blockIdx.x, n_waves = nb * kTailWaves;
  const int n_vw_bc = a.uniform ? (a.n_cam + 6) / 7 : a.n_vw;   // records of phases B and C
  // one round: values are carried in registers
  const bool single = (REDUCE == 1 ? a.n_vw : n_vw_bc) <= n_waves && a.split == 0;
  // the first record is fetched together with the control block (one trip to memory instead of two)
  const int rec_first = (vw0 < a.n_vw) ? a.wave_rec[(size_t)kTailRec * vw0 + min(lane, kTailRec - 1)] : 0;
  Ctl c = a.ctl[(j - 1) & 1];   // C_{j-1}
  const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
  if (c.done) {
    if (writer && MODE != 1 && MODE != 3) a.ctl[j & 1] = c;   // the final state in both buffers
    return;
  }
  TAIL_STAMP(0);
  double* part_rho_cur = (j & 1) ? a.part_rho1 : a.part_rho0;
  double* part_rho_next = (j & 1) ? a.part_rho0 : a.part_rho1;
  // carried through the phases when single (entry k of the lane's camera)
  double c_x = 0.0, c_p = 0.0, c_r = 0.0, c_b = 0.0, c_q = 0.0, c_z = 0.0, c_M[9];
#pragma unroll
  for (int m = 0; m < 9; ++m) c_M[m] = 0.0;

  // ---- A: q = S v over the cameras of this wave (v = p, or x in MODE 2) ----
  if (REDUCE) {
    const double* v_in = (MODE == 2 || (MODE == 3 && a.a_on_x)) ? a.x : a.p;
    double xy = 0.0;
    double* wp = s_wp + (size_t)wv * 7 * 64;
    int* bnd = s_bnd + wv * 8;
    for (int vw = vw0; vw < a.n_vw; vw += n_waves) {
      const int rec = (vw == vw0) ? rec_first : a.wave_rec[(size_t)kTailRec * vw + min(lane, kTailRec - 1)];
      const int cam0 = __shfl(rec, 0, 64), ncw = __shfl(rec, 1, 64);
      const bool on = sub < ncw;
      const int cam = on ? cam0 + sub : cam0;
      const size_t i = 9 * (size_t)cam + k;
      const double xv = on ? v_in[i] : 0.0;
      const double scv = on ? a.sc[i] : 0.0, d2v = on ? a.d2[i] : 0.0;
      double cf[7];
      tail_cam_coef(k, a.ct + (size_t)kCamTabStride * cam, a.cm + 10 * (size_t)cam, cf);
      if (lane >= 2 && lane < 10) bnd[lane - 2] = rec;
#pragma unroll
      for (int q7 = 0; q7 < 7; ++q7) wp[q7 * 64 + lane] = 0.0;
      __builtin_amdgcn_wave_barrier();
      const int R0 = __shfl(rec, 2, 64), R1 = __shfl(rec, 2 + ncw, 64);
      int cj = 0, b_next = bnd[1];
      double acc = 0.0;
      const double* sp = a.slab + k;
      const bool row_lane = lane < 63;
      for (int t0 = 0; R0 + 7 * t0 < R1; t0 += kTailBatch) {
        double t[kTailBatch];
#pragma unroll
        for (int m = 0; m < kTailBatch; ++m) {
          const int row = R0 + sub + 7 * (t0 + m);
          t[m] = (row_lane && row < R1) ? sp[9 * (size_t)row] : 0.0;
        }
#pragma unroll
        for (int m = 0; m < kTailBatch; ++m) {
          const int row = R0 + sub + 7 * (t0 + m);
          if (row_lane && row < R1) {
            while (row >= b_next) {   // into the next camera (row < R1 = the last boundary: cj stays below ncw)
              wp[cj * 64 + lane] = acc;
              acc = 0.0;
              ++cj;
              b_next = bnd[cj + 1];
            }
            acc += t[m];
          }
        }
      }
      if (row_lane) wp[cj * 64 + lane] = acc;
      __builtin_amdgcn_wave_barrier();
      double tot = 0.0;
      if (on) {
#pragma unroll
        for (int g = 0; g < 7; ++g) tot += wp[sub * 64 + 9 * g + k];
      }
      __builtin_amdgcn_wave_barrier();
      const int l0 = 9 * sub;   // the first six sums of the camera (rotation and translation parts) to all of its lanes
      const double v0 = __shfl(tot, l0, 64), v1 = __shfl(tot, l0 + 1, 64), v2 = __shfl(tot, l0 + 2, 64);
      const double v3 = __shfl(tot, l0 + 3, 64), v4 = __shfl(tot, l0 + 4, 64), v5 = __shfl(tot, l0 + 5, 64);
      if (on) {
        double rr = cf[0] * v0 + cf[1] * v1 + cf[2] * v2 + cf[3] * v3 + cf[4] * v4 + cf[5] * v5 + cf[6] * tot;
        rr *= scv;
        rr += a.diag_w * d2v * xv;
        if (REDUCE == 2) {
          // this rank's share of q[i] into every rank's exchange buffer (write-through, system scope)
          const size_t at = ((size_t)(a.epoch & 1u) * a.n_ranks + a.rank) * 9 * (size_t)a.n_cam + i;
          for (int jr = 0; jr < a.n_ranks; ++jr) peer_std(a.xstage[jr] + at, rr);
        } else {
          if (single && MODE != 3) { c_q = rr; if (MODE != 2) c_p = xv; else c_x = xv; }
          else a.q[i] = rr;
          xy += xv * rr;
        }
      }
    }
    if (REDUCE == 2) {
      // every push of this workgroup acknowledged, then the workgroup arrives; the last one to arrive raises this rank's flag at
      // every peer; then everybody waits for the flags of all ranks (in the own signal block)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      PeerSignals* me = a.xsig[a.rank];
      if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(&me->xarrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old % (unsigned)nb == (unsigned)nb - 1u) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
          for (int jr = 0; jr < a.n_ranks; ++jr) peer_st(&a.xsig[jr]->xready[a.rank], a.epoch);
        }
        s_ok = 1;
      }
      __syncthreads();
      if (tid < a.n_ranks && !peer_wait(&me->xready[tid], a.epoch, &me->timeout)) s_ok = 0;
      __syncthreads();
      if (!s_ok) {
        if (writer) tail_timed_out(a.ctl);
        return;
      }
      // q = the shares summed in rank order, for the cameras phases B and C give this lane
      const double* mine = a.xstage[a.rank] + (size_t)(a.epoch & 1u) * a.n_ranks * 9 * (size_t)a.n_cam;
      xy = 0.0;
      for (int vw = vw0; vw < n_vw_bc; vw += n_waves) {
        const int cam0 = 7 * vw, ncw = min(7, a.n_cam - cam0);
        if (sub < ncw) {
          const size_t i = 9 * (size_t)(cam0 + sub) + k;
          double v = peer_ldd(mine + i);
          for (int jr = 1; jr < a.n_ranks; ++jr) v += peer_ldd(mine + (size_t)jr * 9 * (size_t)a.n_cam + i);
          const double xv = v_in[i];
          if (single) { c_q = v; if (MODE != 2) c_p = xv; else c_x = xv; }
          else a.q[i] = v;
          xy += xv * v;
        }
      }
    }
    if (MODE == 3) {
      if (!a.a_on_x) {
        const double s = tail_block_sum(xy, s32);
        if (tid == 0) a.part_pq[blockIdx.x] = s;
      }
      return;
    }
    if (MODE != 2) {
      const double s = tail_block_sum(xy, s32);
      if (tid == 0) tail_put(a.part_pq + blockIdx.x, s);
      TAIL_STAMP(1);
      if (!tail_grid_barrier(a.bar, nb, &s_ok, a.bar_timeout)) {
        if (writer) tail_timed_out(a.ctl);
        return;
      }
      TAIL_STAMP(2);
    }
  }

  // ---- B: alpha and the vector updates ----
  // single: the lane's entries of the vectors and its row of M^-1 are requested BEFORE the partial sums, so that they arrive
  // with them (loads return in order): nothing is waited for twice
  const int cam0_1 = a.uniform ? 7 * vw0 : __shfl(rec_first, 0, 64);
  const int ncw_1 = a.uniform ? min(7, a.n_cam - 7 * vw0) : __shfl(rec_first, 1, 64);
  if (single && sub < ncw_1 && vw0 < n_vw_bc) {
    const size_t i = 9 * (size_t)(cam0_1 + sub) + k;
    if (!REDUCE) { c_q = a.q[i]; c_p = a.p[i]; c_x = a.x[i]; }
    else if (MODE != 2) c_x = a.x[i];
    if (MODE != 1) {
      c_b = a.b[i];
      if (MODE == 0) c_r = a.r[i];
#pragma unroll
      for (int m = 0; m < 9; ++m) c_M[m] = a.Minv[81 * (size_t)(cam0_1 + sub) + 9 * m + k];
    }
  }
  const int n_pq = REDUCE ? nb : a.n_pq;
  double rho, pq;
  tail_sum_partials2(part_rho_cur, nb, a.part_pq, n_pq, s32, rho, pq);
  const bool pq_bad = pq <= 0.0 || isinf(pq) || isnan(pq);
  const double alpha = rho / pq;
  const bool alpha_bad = !pq_bad && isinf(alpha);
  TAIL_STAMP(3);
  if (!pq_bad && !alpha_bad && a.split != 2) {
    double a_q = 0.0, a_rho = 0.0;
    for (int vw = vw0; vw < n_vw_bc; vw += n_waves) {
      int cam0 = cam0_1, ncw = ncw_1;
      if (vw != vw0) {
        if (a.uniform) { cam0 = 7 * vw; ncw = min(7, a.n_cam - cam0); }
        else {
          const int rec = a.wave_rec[(size_t)kTailRec * vw + (lane & 1)];
          cam0 = __shfl(rec, 0, 64); ncw = __shfl(rec, 1, 64);
        }
      }
      const bool on = sub < ncw;
      const int cam = on ? cam0 + sub : cam0;
      const size_t i = 9 * (size_t)cam + k;
      double rv = 0.0, xv = 0.0;
      if (on) {
        if (MODE == 2) xv = single ? c_x : a.x[i];
        else { xv = (single ? c_x : a.x[i]) + alpha * (single ? c_p : a.p[i]); a.x[i] = xv; }
        if (MODE != 1) {
          const double bv = single ? c_b : a.b[i];
          const double qv = single ? c_q : a.q[i];
          rv = (MODE == 0) ? (single ? c_r : a.r[i]) - alpha * qv : bv - qv;
          a.r[i] = rv;
          a_q += xv * (bv + rv);
        }
      }
      if (MODE != 1) {
        double zz = 0.0;
#pragma unroll
        for (int m = 0; m < 9; ++m) {
          const double mv = single ? c_M[m] : (on ? a.Minv[81 * (size_t)cam + 9 * m + k] : 0.0);
          zz += mv * __shfl(rv, 9 * sub + m, 64);
        }
        if (on) {
          a.z[i] = zz;
          c_z = zz;
          a_rho += rv * zz;
        }
      }
    }
    if (MODE != 1) {
      // both block sums in one pass
      const double wq = wave_sum(a_q), wr = wave_sum(a_rho);
      __syncthreads();
      if (lane == 0) { s32[wv] = wq; s32[kTailWaves + wv] = wr; }
      __syncthreads();
      if (tid == 0) {
        double sq = 0.0, sr = 0.0;
#pragma unroll
        for (int m = 0; m < kTailWaves; ++m) { sq += s32[m]; sr += s32[kTailWaves + m]; }
        tail_put(a.part_q + blockIdx.x, sq);
        tail_put(part_rho_next + blockIdx.x, sr);
      }
    }
  }
  if (MODE == 1 || a.split == 1) return;
  TAIL_STAMP(4);
  if (a.split == 0 && !tail_grid_barrier(a.bar, nb, &s_ok, a.bar_timeout)) {
    if (writer) tail_timed_out(a.ctl);
    return;
  }
  TAIL_STAMP(5);

  // ---- C: end of iteration j, direction of iteration j + 1 (pcg_chkdir(j + 1)) ----
  // (a single-round MODE 2 has not had p in its registers: requested before the partial sums, like the vectors of phase B)
  if (single && MODE == 2 && REDUCE && sub < ncw_1 && vw0 < n_vw_bc) c_p = a.p[9 * (size_t)(cam0_1 + sub) + k];
  double qs = 0.0, rho_n = 0.0;
  if (!pq_bad && !alpha_bad) tail_sum_partials2(a.part_q, nb, part_rho_next, nb, s32, qs, rho_n);
  c.n_it = j;
  c.pq = pq;
  if (pq_bad) { c.done = 1; c.status = 1; }
  else {
    c.alpha = alpha;
    if (alpha_bad) { c.done = 1; c.status = 2; }
    else {
      const double Q1 = -qs;
      const double zeta = j * (Q1 - c.Q0) / Q1;
      c.Q1 = Q1;
      c.zeta = zeta;
      if (zeta < a.q_tol && j >= a.min_it) { c.done = 1; c.status = 0; }
      else {
        c.Q0 = Q1;
        if (j >= a.max_it) { c.done = 1; c.status = 1; }
      }
    }
  }
  if (c.done) {
    if (writer) a.ctl[j & 1] = c;
    return;
  }
  bool bad = pcg_bad(rho_n);
  double beta = 0.0;
  if (!bad) {
    beta = rho_n / rho;
    bad = pcg_bad(beta);
  }
  c.rho_prev = rho;
  c.rho = rho_n;
  c.beta = beta;
  if (bad) { c.done = 1; c.status = 2; c.n_it = j + 1; }
  if (writer) a.ctl[j & 1] = c;
  if (bad) return;
  for (int vw = vw0; vw < n_vw_bc; vw += n_waves) {
    int cam0 = cam0_1, ncw = ncw_1;
    if (vw != vw0) {
      if (a.uniform) { cam0 = 7 * vw; ncw = min(7, a.n_cam - cam0); }
      else {
        const int rec = a.wave_rec[(size_t)kTailRec * vw + (lane & 1)];
        cam0 = __shfl(rec, 0, 64); ncw = __shfl(rec, 1, 64);
      }
    }
    if (sub < ncw) {
      const size_t i = 9 * (size_t)(cam0 + sub) + k;
      a.p[i] = (single ? c_z : a.z[i]) + beta * (single ? c_p : a.p[i]);
    }
  }
  TAIL_STAMP(6);
}

}  // namespace miba
