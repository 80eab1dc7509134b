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
// on a grid that is resident as a whole (at most one 1024-thread workgroup per CU), with the barrier on one device-scope
// counter.  Every workgroup derives the scalars from the same partial arrays in the same order, so all take the same branch.
// A barrier that is not reached within two seconds (the grid was not resident after all) ends the solve with a failure status
// instead of hanging.  With several ranks the camera-side sum stays a kernel of its own (the all-reduce sits behind it) and
// the tail starts at B.
#pragma once

#include "miba_mf.hpp"

namespace miba {

constexpr int kTailThreads = 1024;
constexpr int kTailWaves = kTailThreads / 64;
constexpr int kTailCams = 7 * kTailWaves;   // cameras per workgroup round: 7 per wave, nine lanes each

// sum over the 1024-thread workgroup, the same value in every thread; s16: 16 doubles of LDS
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
__device__ __forceinline__ double tail_sum_partials(const double* __restrict__ part, int n, double* s16) {
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += kTailThreads) v += part[i];
  return tail_block_sum(v, s16);
}

// All workgroups of the grid meet.  count only grows (the host clears it before a solve): arrival number a belongs to barrier
// a / n_blocks.  Release before arriving, acquire after leaving (device scope: the L2 of another XCD may hold the old lines).
// Returns false when the others did not arrive in time.
__device__ __forceinline__ bool tail_grid_barrier(unsigned* count, unsigned n_blocks, int* s_ok) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned old = __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = (old / n_blocks + 1u) * n_blocks;
    const long long t0 = wall_clock64();
    int ok = 1;
    while (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > 200000000ll) { ok = 0; break; }   // 2 s of the 100 MHz counter
    }
    *s_ok = ok;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return *s_ok != 0;
}

// camera-side transforms of the summed raw slab row v[0..8] of camera c, entry k (the text of k_cam_reduce_mf)
// (own = v[k], the lane's own sum: no register array is indexed with k)
__device__ __forceinline__ double tail_cam_entry(int k, double v0, double v1, double v2, double v3, double v4, double v5, double own,
                                                 const double* __restrict__ T, const double* __restrict__ M) {
  const bool small = M[9] == 1.0;
  double r;
  if (M[9] == 2.0) {
    if (k < 3) {
      const double t0 = -(T[0] * T[9] + T[3] * T[10] + T[6] * T[11]);
      const double t1 = -(T[1] * T[9] + T[4] * T[10] + T[7] * T[11]);
      const double t2 = -(T[2] * T[9] + T[5] * T[10] + T[8] * T[11]);
      const double u0 = t1 * v5 - t2 * v4 - v0, u1 = t2 * v3 - t0 * v5 - v1, u2 = t0 * v4 - t1 * v3 - v2;
      r = M[k] * u0 + M[3 + k] * u1 + M[6 + k] * u2;
    } else if (k < 6) r = -own;
    else r = 0.0;
  } else if (k < 3) r = M[k] * v0 + M[3 + k] * v1 + M[6 + k] * v2;
  else if (k < 6) {
    const int i = k - 3;
    r = small ? own : (T[3 * i] * v3 + T[3 * i + 1] * v4 + T[3 * i + 2] * v5);
  } else if (k == 6) r = own;
  else r = T[12] * own;
  return r;
}

// a barrier timed out: the solve ends as a failure in both control blocks (bit 16 of fail tells the host why)
__device__ __forceinline__ void tail_timed_out(Ctl* ctl) {
  for (int b = 0; b < 2; ++b) { ctl[b].done = 1; ctl[b].status = 2; ctl[b].fail |= 16; }
}

struct TailArgs {
  int j;            // PCG iteration (1-based) this launch completes
  int n_cam;
  // camera-side sum (phase A)
  const int32_t* cam_row_off; const double* slab; const double* ct; const double* cm; const double* sc; const double* d2;
  double diag_w;
  // vectors
  double* p; double* q; const double* b; const double* Minv; double* x; double* r; double* z;
  // scalars
  Ctl* ctl;            // [2], C_k in ctl[k & 1]
  double* part_pq; int n_pq;   // partials of p.q: this grid's (phase A) or, with several ranks, the camera reduction's
  double* part_q;      // partials of Q, one per workgroup
  double* part_rho0; double* part_rho1;   // partials of rho_k in part_rho[k & 1]
  double q_tol; int min_it, max_it;
  unsigned* bar;       // barrier counter
};

// MODE 0: the whole tail of iteration j.
// MODE 1: residual-reset iteration, first half: A, alpha, x += alpha p (r is rebuilt from S x by MODE 2 after another product).
// MODE 2: second half: A on q = S x (no p.q), r = b - q, Q, z, r.z, then C.
// REDUCE: phase A is part of this launch (one rank); otherwise q and the p.q partials are there already.
template <int MODE, bool REDUCE>
__global__ __launch_bounds__(kTailThreads) void pcg_tail(const TailArgs a) {
  __shared__ double s16[kTailWaves];
  __shared__ int s_ok;
  const int j = a.j;
  Ctl c = a.ctl[(j - 1) & 1];   // C_{j-1}
  const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
  if (c.done) {
    if (writer) a.ctl[j & 1] = c;   // the final state in both buffers
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int sub = lane / 9, k = lane - 9 * sub;
  const int nb = gridDim.x;
  const int first = (blockIdx.x * kTailWaves + wv) * 7 + sub, stride = nb * kTailCams;
  const bool lane_on = lane < 63;
  double* part_rho_cur = (j & 1) ? a.part_rho1 : a.part_rho0;
  double* part_rho_next = (j & 1) ? a.part_rho0 : a.part_rho1;

  // ---- A: q = S v over the cameras of this wave (v = p, or x in MODE 2) ----
  if (REDUCE) {
    const double* v_in = (MODE == 2) ? a.x : a.p;
    double xy = 0.0;
    for (int cc = first; cc - sub < a.n_cam; cc += stride) {   // (the whole wave leaves together: the shuffles need every lane)
      const bool on = lane_on && cc < a.n_cam;
      const int cam = on ? cc : 0;
      const int r0 = on ? a.cam_row_off[cam] : 0, r1 = on ? a.cam_row_off[cam + 1] : 0;
      const size_t i = 9 * (size_t)cam + k;
      const double xv = on ? v_in[i] : 0.0;
      const double scv = on ? a.sc[i] : 0.0, d2v = on ? a.d2[i] : 0.0;
      double acc = 0.0;
      const double* sp = a.slab + k;
      int r = r0;
      for (; r + 4 <= r1; r += 4) {
        const double s0 = sp[9 * (size_t)r], s1 = sp[9 * (size_t)(r + 1)], s2 = sp[9 * (size_t)(r + 2)], s3 = sp[9 * (size_t)(r + 3)];
        acc += s0; acc += s1; acc += s2; acc += s3;
      }
      for (; r < r1; ++r) acc += sp[9 * (size_t)r];
      const int l0 = 9 * sub;   // the first six sums of the camera (rotation and translation parts) to all of its lanes
      const double v0 = __shfl(acc, l0, 64), v1 = __shfl(acc, l0 + 1, 64), v2 = __shfl(acc, l0 + 2, 64);
      const double v3 = __shfl(acc, l0 + 3, 64), v4 = __shfl(acc, l0 + 4, 64), v5 = __shfl(acc, l0 + 5, 64);
      if (on) {
        double rr = tail_cam_entry(k, v0, v1, v2, v3, v4, v5, acc, a.ct + (size_t)kCamTabStride * cam, a.cm + 10 * (size_t)cam);
        rr *= scv;
        rr += a.diag_w * d2v * xv;
        a.q[i] = rr;
        xy += xv * rr;
      }
    }
    if (MODE != 2) {
      const double s = tail_block_sum(xy, s16);
      if (tid == 0) a.part_pq[blockIdx.x] = s;
      if (!tail_grid_barrier(a.bar, nb, &s_ok)) {
        if (writer) tail_timed_out(a.ctl);
        return;
      }
    }
  }

  // ---- B: alpha and the vector updates ----
  const int n_pq = REDUCE ? nb : a.n_pq;
  const double rho = tail_sum_partials(part_rho_cur, nb, s16);
  const double pq = tail_sum_partials(a.part_pq, n_pq, s16);
  const bool pq_bad = pq <= 0.0 || isinf(pq) || isnan(pq);
  const double alpha = rho / pq;
  const bool alpha_bad = !pq_bad && isinf(alpha);
  if (!pq_bad && !alpha_bad) {
    double a_q = 0.0, a_rho = 0.0;
    for (int cc = first; cc - sub < a.n_cam; cc += stride) {
      const bool on = lane_on && cc < a.n_cam;
      const size_t i = 9 * (size_t)(on ? cc : 0) + k;
      double rv = 0.0, xv = 0.0, bv = 0.0;
      if (on) {
        if (MODE == 2) xv = a.x[i];
        else { xv = a.x[i] + alpha * a.p[i]; a.x[i] = xv; }
        if (MODE != 1) {
          bv = a.b[i];
          rv = (MODE == 0) ? a.r[i] - alpha * a.q[i] : bv - a.q[i];
          a.r[i] = rv;
          a_q += xv * (bv + rv);
        }
      }
      if (MODE != 1) {
        const double* mrow = a.Minv + 9 * i;
        double zz = 0.0;
#pragma unroll
        for (int m = 0; m < 9; ++m) zz += (on ? mrow[m] : 0.0) * __shfl(rv, 9 * sub + m, 64);
        if (on) {
          a.z[i] = zz;
          a_rho += rv * zz;
        }
      }
    }
    if (MODE != 1) {
      const double sq = tail_block_sum(a_q, s16);
      const double sr = tail_block_sum(a_rho, s16);
      if (tid == 0) { a.part_q[blockIdx.x] = sq; part_rho_next[blockIdx.x] = sr; }
    }
  }
  if (MODE == 1) return;
  if (!tail_grid_barrier(a.bar, nb, &s_ok)) {
    if (writer) tail_timed_out(a.ctl);
    return;
  }

  // ---- C: end of iteration j, direction of iteration j + 1 (pcg_chkdir(j + 1)) ----
  c.n_it = j;
  c.pq = pq;
  if (pq_bad) { c.done = 1; c.status = 1; }
  else {
    c.alpha = alpha;
    if (alpha_bad) { c.done = 1; c.status = 2; }
    else {
      const double qs = tail_sum_partials(a.part_q, nb, s16);
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
  const double rho_n = tail_sum_partials(part_rho_next, nb, s16);
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
  for (int cc = first; cc - sub < a.n_cam; cc += stride) {
    if (lane_on && cc < a.n_cam) {
      const size_t i = 9 * (size_t)cc + k;
      a.p[i] = a.z[i] + beta * a.p[i];
    }
  }
}

}  // namespace miba
