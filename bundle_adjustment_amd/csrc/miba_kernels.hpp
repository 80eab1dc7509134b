// miba_kernels.hpp — gfx950 HIP kernels of the LM iteration (SURVEY.md §8a kernel set).
//
// Replaces what ceres::Solve (/root/reference/src/BundleAdjustment.cpp:108) does per LM iteration:
//   K0  k_cam_tables    per-camera rotation tables            (ceres::AngleAxisToRotationMatrix,
//                                                              src/BAConstraint.cpp:35)
//   K1  k_eval<true>    residual + analytic Jacobian           (BAConstraint::operator()<Jet>,
//                                                              src/BAConstraint.cpp:15-80; BAL model :89-128)
//   K1' k_eval<false>   residual-only cost                     (operator()<double>)
//   K2  k_schur_init    per-point 3x3 Schur elimination, F'F / rhs / gradient camera sums
//   K3  k_schur_matvec  implicit Schur complement product  y = S x
//   K4  k_backsub       back-substitution + model cost change + candidate points
//   K3v pcg_*           PCG vector kernels with device-resident scalars
//
// Layout: observations sorted by point, 256 slots per chunk, a point never straddles a chunk
// (except a "giant" point that owns its whole tile).  Stored Jacobian J[chunk][26][256]
// (F row0 0..8, F row1 9..17, E row0 18..20, E row1 21..23, r 24..25): every wave load is a
// 512-byte contiguous run.  Per-point sums go through an LDS stage and are summed in slot
// order by one thread per point (deterministic); per-camera sums are accumulated in LDS
// per tile (ds_add_f64), written as plain rows to a camera-major slab and summed per
// camera by k_cam_reduce — no global atomics anywhere.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "miba_plan.hpp"

namespace miba {

#define MIBA_DBL_EPS 2.220446049250313e-16

// device-resident PCG / LM control block
struct Ctl {
  double rho, rho_prev, pq, alpha, beta, Q0, Q1, zeta, norm_b2;
  int32_t done;      // 1 => every later PCG kernel returns immediately
  int32_t status;    // 0 success, 1 no convergence, 2 failure
  int32_t n_it;      // iterations completed
  int32_t fail;      // Cholesky failure flags from K2 / cam_blocks
};

// scalar slots produced by reductions
enum { S_COST = 0, S_MCC, S_DP2, S_XP2, S_DC2, S_XC2, S_GMAX_C, S_G2_C, S_GMAX_P, S_G2_P, S_COUNT = 16 };

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}
// sum over a 256-thread block (exactly four wavefronts: a smaller block would read unwritten LDS slots);
// result valid in every thread; s4 = 4 doubles of LDS
__device__ __forceinline__ double block_sum(double v, double* s4) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (s4[0] + s4[1]) + (s4[2] + s4[3]);
}
__device__ __forceinline__ double block_max(double v, double* s4) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmax(fmax(s4[0], s4[1]), fmax(s4[2], s4[3]));
}

// ---------------------------------------------------------------------------------------------
// K0: per-camera table  [R 0..8 | t 9..11 | f k1 k2 12..14 | - | dR_0 16..24 | dR_1 25..33 | dR_2 34..42]
// R row-major; dR_j = dR/dw_j (row-major).  Branches as ceres/rotation.h.
// ---------------------------------------------------------------------------------------------
// model 1 (the reference's live functor, src/BAConstraint.cpp:15-80: P = R^-1 (X - t), no distortion): the table holds the
// camera the BAL-shaped per-observation code needs, R~ = R', t~ = -R' t, f, k1 = k2 = 0, dR~_j = (dR_j)', M~ = J_r(-w) = M',
// and the flag 2 in slot 15 / cm[9]: only the per-camera transforms (staging, reduction, F rows) differ between the models.
__global__ void k_cam_tables(int n_cam, const double* __restrict__ cam /*[n][9]*/, double* __restrict__ ct,
                             double* __restrict__ cm /*[n][10] or null*/, int model) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cam) return;
  const double* p = cam + 9 * (size_t)c;
  double* o = ct + (size_t)kCamTabStride * c;
  const double w0 = p[0], w1 = p[1], w2 = p[2];
  const double th2 = w0 * w0 + w1 * w1 + w2 * w2;
  double R[9], dR[27];
  if (th2 > MIBA_DBL_EPS) {
    const double th = sqrt(th2);
    const double s = sin(th), c_ = cos(th);
    const double omc = (c_ > 0.0) ? s * s / (1.0 + c_) : 1.0 - c_;
    const double k[3] = {w0 / th, w1 / th, w2 / th};
    R[0] = c_ + k[0] * k[0] * omc;      R[1] = -k[2] * s + k[0] * k[1] * omc; R[2] = k[1] * s + k[0] * k[2] * omc;
    R[3] = k[2] * s + k[0] * k[1] * omc; R[4] = c_ + k[1] * k[1] * omc;      R[5] = -k[0] * s + k[1] * k[2] * omc;
    R[6] = -k[1] * s + k[0] * k[2] * omc; R[7] = k[0] * s + k[1] * k[2] * omc; R[8] = c_ + k[2] * k[2] * omc;
    // dR_j = -s k_j I + c k_j [k]x + s [dk]x + s k_j k k' + omc (dk k' + k dk'),  dk = (e_j - k k_j)/th
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double kj = k[j];
      double dk[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) dk[i] = ((i == j ? 1.0 : 0.0) - k[i] * kj) / th;
      double* D = dR + 9 * j;
      const double a = c_ * kj;  // coefficient of [k]x
      // symmetric part
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          D[r * 3 + q] = (r == q ? -s * kj : 0.0) + s * kj * k[r] * k[q] + omc * (dk[r] * k[q] + k[r] * dk[q]);
      // skew parts: a [k]x + s [dk]x
      const double v0 = a * k[0] + s * dk[0], v1 = a * k[1] + s * dk[1], v2 = a * k[2] + s * dk[2];
      D[1] += -v2; D[2] += v1;
      D[3] += v2;  D[5] += -v0;
      D[6] += -v1; D[7] += v0;
    }
  } else {
    R[0] = 1.0; R[1] = -w2; R[2] = w1;
    R[3] = w2;  R[4] = 1.0; R[5] = -w0;
    R[6] = -w1; R[7] = w0;  R[8] = 1.0;
#pragma unroll
    for (int i = 0; i < 27; ++i) dR[i] = 0.0;
    // dR_j = [e_j]x
    dR[0 * 9 + 5] = -1.0; dR[0 * 9 + 7] = 1.0;
    dR[1 * 9 + 2] = 1.0;  dR[1 * 9 + 6] = -1.0;
    dR[2 * 9 + 1] = -1.0; dR[2 * 9 + 3] = 1.0;
  }
  if (cm) {
    // d(R X)/dw = -R [X]x M with M = J_r(w) = I - a [w]x + b [w]x^2, a = (1-cos)/th^2, b = (th-sin)/th^3
    // (closed form of the same derivative as dR_j; b by series below th = 0.1 to avoid cancellation).
    // Small-angle branch: d(RX)/dw = -[X]x, i.e. M = I and N = I in place of R (flag = 1).
    double* m = cm + 10 * (size_t)c;
    if (th2 > MIBA_DBL_EPS) {
      const double th = sqrt(th2);
      const double s = sin(th), c_ = cos(th);
      const double omc = (c_ > 0.0) ? s * s / (1.0 + c_) : 1.0 - c_;
      const double a = omc / th2;
      double b;
      if (th < 0.1) b = 1.0 / 6.0 - th2 * (1.0 / 120.0 - th2 * (1.0 / 5040.0 - th2 * (1.0 / 362880.0 - th2 / 39916800.0)));
      else b = (th - s) / (th2 * th);
      // [w]x^2 = w w' - th2 I
      m[0] = 1.0 + b * (w0 * w0 - th2); m[1] = a * w2 + b * w0 * w1;        m[2] = -a * w1 + b * w0 * w2;
      m[3] = -a * w2 + b * w1 * w0;      m[4] = 1.0 + b * (w1 * w1 - th2);  m[5] = a * w0 + b * w1 * w2;
      m[6] = a * w1 + b * w2 * w0;       m[7] = -a * w0 + b * w2 * w1;      m[8] = 1.0 + b * (w2 * w2 - th2);
      m[9] = 0.0;
    } else {
      m[0] = 1.0; m[1] = 0.0; m[2] = 0.0; m[3] = 0.0; m[4] = 1.0; m[5] = 0.0; m[6] = 0.0; m[7] = 0.0; m[8] = 1.0;
      m[9] = 1.0;
    }
    if (model == 1) {
      // M~ = M'; never "small": the small-angle rotation is inverted like any other (R~ = (I + [w]x)' to 1e-16)
      double t;
      t = m[1]; m[1] = m[3]; m[3] = t;
      t = m[2]; m[2] = m[6]; m[6] = t;
      t = m[5]; m[5] = m[7]; m[7] = t;
      m[9] = 2.0;
    }
  }
  if (model == 1) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) o[3 * r + q] = R[3 * q + r];
#pragma unroll
    for (int i = 0; i < 3; ++i) o[9 + i] = -(R[i] * p[3] + R[3 + i] * p[4] + R[6 + i] * p[5]);   // -R' t
    o[12] = p[6]; o[13] = 0.0; o[14] = 0.0;
    o[15] = 2.0;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) o[16 + 9 * j + 3 * r + q] = dR[9 * j + 3 * q + r];
    return;
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) o[i] = R[i];
  o[9] = p[3]; o[10] = p[4]; o[11] = p[5];
  o[12] = p[6]; o[13] = p[7]; o[14] = p[8];
  o[15] = 0.0;
#pragma unroll
  for (int i = 0; i < 27; ++i) o[16 + i] = dR[i];
}

// ---------------------------------------------------------------------------------------------
// K1 / K1': residual (+ Jacobian) per observation slot.  One block per tile.
// ---------------------------------------------------------------------------------------------
template <bool WITH_JAC>
__global__ __launch_bounds__(256) void k_eval(const Tile* __restrict__ tiles, const Group* __restrict__ grps,
                                              const int32_t* __restrict__ idx, const int32_t* __restrict__ ocam,
                                              const double* __restrict__ oxy /*[chunk][2][256]*/,
                                              const double* __restrict__ ct, const double* __restrict__ X /*[npt][3]*/,
                                              const double* __restrict__ sc /*[ncam][9] or null*/,
                                              const double* __restrict__ sp /*[npt][3] or null*/,
                                              double* __restrict__ J, double* __restrict__ partial_cost) {
  __shared__ double s4[4];
  const Tile t = tiles[blockIdx.x];
  const int tid = threadIdx.x;
  double acc = 0.0;
  for (int g = 0; g < t.ngrp; ++g) {
    const Group G = grps[t.grp0 + g];
    for (int ch = 0; ch < G.nchunk; ++ch) {
      const size_t chunk = (size_t)(G.chunk0 + ch);
      const size_t slot = chunk * kChunk + tid;
      const int32_t w = idx[slot];
      double* Jc = J + chunk * (size_t)(kJComp * kChunk) + tid;
      if (w < 0) {
        if (WITH_JAC) {
#pragma unroll
          for (int k = 0; k < kJComp; ++k) Jc[(size_t)k * kChunk] = 0.0;
        }
        continue;
      }
      const int cam = ocam[slot];
      const int p = G.pt0 + (w & 0xFF);
      const double* T = ct + (size_t)kCamTabStride * cam;
      const double x0 = X[3 * (size_t)p], x1 = X[3 * (size_t)p + 1], x2 = X[3 * (size_t)p + 2];
      double R[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) R[i] = T[i];
      const double P0 = R[0] * x0 + R[1] * x1 + R[2] * x2 + T[9];
      const double P1 = R[3] * x0 + R[4] * x1 + R[5] * x2 + T[10];
      const double P2 = R[6] * x0 + R[7] * x1 + R[8] * x2 + T[11];
      const double f = T[12], k1 = T[13], k2 = T[14];
      const double iz = 1.0 / P2;
      const double px = -P0 * iz, py = -P1 * iz;
      const double n = px * px + py * py;
      const double d = 1.0 + n * (k1 + k2 * n);
      const double ox = oxy[chunk * (2 * kChunk) + tid], oy = oxy[chunk * (2 * kChunk) + kChunk + tid];
      const double r0 = f * d * px - ox, r1 = f * d * py - oy;
      acc += r0 * r0 + r1 * r1;
      if (!WITH_JAC) continue;
      const double gg = k1 + 2.0 * k2 * n;
      const double m00 = f * (d + 2.0 * gg * px * px), m01 = f * (2.0 * gg * px * py), m11 = f * (d + 2.0 * gg * py * py);
      // A = M * dp/dP,  dp/dP = [-iz 0 P0 iz^2; 0 -iz P1 iz^2]
      const double q0 = P0 * iz * iz, q1 = P1 * iz * iz;
      double A[6];
      A[0] = -m00 * iz; A[1] = -m01 * iz; A[2] = m00 * q0 + m01 * q1;
      A[3] = -m01 * iz; A[4] = -m11 * iz; A[5] = m01 * q0 + m11 * q1;
      double Fv[18], Ev[6];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        Ev[j] = A[0] * R[j] + A[1] * R[3 + j] + A[2] * R[6 + j];
        Ev[3 + j] = A[3] * R[j] + A[4] * R[3 + j] + A[5] * R[6 + j];
      }
      const bool model2 = T[15] == 2.0;   // the reference's functor: P = R~ (X - t)
      // rotation columns: A * (dR_j X), or A * (dR~_j (X - t)) with t = -R~' t~
      double y0 = x0, y1 = x1, y2 = x2;
      if (model2) {
        y0 += R[0] * T[9] + R[3] * T[10] + R[6] * T[11];
        y1 += R[1] * T[9] + R[4] * T[10] + R[7] * T[11];
        y2 += R[2] * T[9] + R[5] * T[10] + R[8] * T[11];
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double* D = T + 16 + 9 * j;
        const double v0 = D[0] * y0 + D[1] * y1 + D[2] * y2;
        const double v1 = D[3] * y0 + D[4] * y1 + D[5] * y2;
        const double v2 = D[6] * y0 + D[7] * y1 + D[8] * y2;
        Fv[j] = A[0] * v0 + A[1] * v1 + A[2] * v2;
        Fv[9 + j] = A[3] * v0 + A[4] * v1 + A[5] * v2;
      }
      if (model2) {
        // dP/dt = -R~: the translation columns are -E; the intrinsics block is constant: zero columns
#pragma unroll
        for (int j = 0; j < 3; ++j) { Fv[3 + j] = -Ev[j]; Fv[12 + j] = -Ev[3 + j]; }
#pragma unroll
        for (int j = 6; j < 9; ++j) { Fv[j] = 0.0; Fv[9 + j] = 0.0; }
      } else {
#pragma unroll
        for (int j = 0; j < 3; ++j) { Fv[3 + j] = A[j]; Fv[12 + j] = A[3 + j]; }
        Fv[6] = d * px;         Fv[15] = d * py;
        Fv[7] = f * n * px;     Fv[16] = f * n * py;
        Fv[8] = f * n * n * px; Fv[17] = f * n * n * py;
      }
      if (sc) {
        const double* s = sc + 9 * (size_t)cam;
#pragma unroll
        for (int k = 0; k < 9; ++k) { const double v = s[k]; Fv[k] *= v; Fv[9 + k] *= v; }
      }
      if (sp) {
        const double* s = sp + 3 * (size_t)p;
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double v = s[k]; Ev[k] *= v; Ev[3 + k] *= v; }
      }
#pragma unroll
      for (int k = 0; k < 18; ++k) Jc[(size_t)k * kChunk] = Fv[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) Jc[(size_t)(18 + k) * kChunk] = Ev[k];
      Jc[(size_t)24 * kChunk] = r0;
      Jc[(size_t)25 * kChunk] = r1;
    }
  }
  const double tot = block_sum(acc, s4);
  if (tid == 0) partial_cost[blockIdx.x] = 0.5 * tot;
}

// 3x3 SPD inverse through Cholesky (Eigen llt().solve(I) in Ceres). a = {a00,a10,a11,a20,a21,a22}.
// Returns false if not positive definite.
// 1/x and 1/sqrt(x) to within an ulp or two: v_rcp_f64 / v_rsq_f64 and two Newton steps.  The correctly rounded
// division and square root cost about thirty instructions each on CDNA, and every wavelet pass of K2 / K4 executes
// them whether or not a lane needs the result.
__device__ __forceinline__ double mf_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
}
__device__ __forceinline__ double mf_rsqrt(double d) {
  double r = __builtin_amdgcn_rsq(d);
  r = r * fma(-0.5 * d * r, r, 1.5);
  r = r * fma(-0.5 * d * r, r, 1.5);
  return r;
}

// inverse of a 3x3 SPD matrix through its Cholesky factor; the pivots' reciprocal square roots are all it needs
__device__ __forceinline__ bool spd3_inverse(const double* a, double* ci /*c00,c10,c11,c20,c21,c22*/) {
  const double d0 = a[0];
  if (!(d0 > 0.0)) return false;
  const double i00 = mf_rsqrt(d0);
  const double l10 = a[1] * i00, l20 = a[3] * i00;
  const double d1 = a[2] - l10 * l10;
  if (!(d1 > 0.0)) return false;
  const double i11 = mf_rsqrt(d1);
  const double l21 = (a[4] - l20 * l10) * i11;
  const double d2 = a[5] - l20 * l20 - l21 * l21;
  if (!(d2 > 0.0)) return false;
  const double i22 = mf_rsqrt(d2);
  const double i10 = -l10 * i00 * i11;
  const double i21 = -l21 * i11 * i22;
  const double i20 = -(l20 * i00 + l21 * i10) * i22;
  ci[0] = i00 * i00 + i10 * i10 + i20 * i20;
  ci[1] = i10 * i11 + i20 * i21;
  ci[2] = i11 * i11 + i21 * i21;
  ci[3] = i20 * i22;
  ci[4] = i21 * i22;
  ci[5] = i22 * i22;
  return true;
}

constexpr int kK2W = 64;   // width of a camera row of the Schur init: upper triangle of F'F (45), F'(r - E z) (9), F'r (9), one spare

// ---------------------------------------------------------------------------------------------
// Camera-major slab reductions (deterministic: rows are summed in row order per camera).
// One wave per camera.  cam_row_off[c]..cam_row_off[c+1] = rows of camera c on this shard.
// ---------------------------------------------------------------------------------------------
// W = 64: out[c][0..63] = sum rows (lane = component)
// sc != null: the rows hold sums of the UNSCALED Jacobian (K2-mf); entry (i,j) of F'F is scaled by
// sc_i sc_j, F'v and F'r by sc_i.
__global__ __launch_bounds__(256) void k_cam_reduce64(int n_cam, const int32_t* __restrict__ cam_row_off,
                                                      const double* __restrict__ slab, const double* __restrict__ sc,
                                                      double* __restrict__ out) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= n_cam) return;
  const int lane = threadIdx.x & 63;
  double a0 = 0.0;
  for (int r = cam_row_off[c]; r < cam_row_off[c + 1]; ++r) a0 += slab[(size_t)kK2W * r + lane];
  if (sc && lane < 63) {
    const double* s = sc + 9 * (size_t)c;
    if (lane >= 45) a0 *= s[(lane - 45) % 9];
    else {
      int i = 0, off = 0;
      while (lane >= off + (9 - i)) { off += 9 - i; ++i; }
      a0 *= s[i] * s[i + (lane - off)];
    }
  }
  out[(size_t)kK2W * c + lane] = a0;
}

// Scalars of the Schur init in the unused 64th entry of the camera sums (multi-rank runs: one all-reduce for the
// stage).  unpack = 0: local cost, |g_p|^2 and max |g_p| into cameras 0, 1 and 2 + rank; unpack = 1 (after the
// all-reduce): totals back into the scalar slots, the max over the per-rank entries.
__global__ void k_pack_scalars(double* __restrict__ acc, double* __restrict__ sum, double* __restrict__ mx, int rank, int unpack,
                               int world) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (!unpack) {
    acc[(size_t)kK2W * 0 + 63] = sum[S_COST];
    acc[(size_t)kK2W * 1 + 63] = sum[S_G2_P];
    acc[(size_t)kK2W * (2 + rank) + 63] = mx[S_GMAX_P];
  } else {
    sum[S_COST] = acc[(size_t)kK2W * 0 + 63];
    sum[S_G2_P] = acc[(size_t)kK2W * 1 + 63];
    double m = 0.0;
    for (int r = 0; r < world; ++r) m = fmax(m, acc[(size_t)kK2W * (2 + r) + 63]);
    mx[S_GMAX_P] = m;
  }
}

// ---------------------------------------------------------------------------------------------
// Camera blocks: B = F'F + D^2 -> M^-1 (9x9 Cholesky inverse), rhs, unscaled gradient norms.
// acc[c][64] = {F'F upper 45, rhs 9, F'r 9}.  One thread per camera.
// ---------------------------------------------------------------------------------------------
// acc2 (SCHUR_JACOBI, else null): upper triangle of sum G'WG per camera, subtracted from B before the inversion,
// so that M is the block diagonal of S itself (the LM diagonal is still taken from F'F).
__global__ __launch_bounds__(64) void k_cam_blocks(int n_cam, const double* __restrict__ acc, const double* __restrict__ acc2, const uint8_t* __restrict__ active,
                             const double* __restrict__ sc, double inv_radius, double dmin, double dmax,
                             double* __restrict__ Minv /*[n][81], column-major blocks*/, double* __restrict__ Dc2 /*[n][9]*/,
                             double* __restrict__ rhs /*[n][9]*/, double* __restrict__ partial /*[nblk][2]*/,
                             Ctl* __restrict__ ctl) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  double gmax = 0.0, g2 = 0.0;
  if (c < n_cam) {
    // (every loop below has compile-time bounds and is unrolled: L stays in registers.  With the early exit the factorisation
    // loop used to have, L was indexed dynamically and lived in scratch memory: 1.2 KB per lane, which the runtime backs for
    // every wave slot of the device, 0.64 GB held for as long as the process lives.)
    const double* a = acc + (size_t)kK2W * c;
    double L[81];
    {
      int o = 0;
#pragma unroll
      for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = i; j < 9; ++j) { L[j * 9 + i] = a[o]; L[i * 9 + j] = a[o]; ++o; }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const double d2 = fmin(fmax(L[i * 9 + i], dmin), dmax) * inv_radius;
      Dc2[9 * (size_t)c + i] = d2;
      L[i * 9 + i] += d2;
      rhs[9 * (size_t)c + i] = a[45 + i];
      if (active[c] && sc[9 * (size_t)c + i] != 0.0) {       // scale 0: constant parameter block, not in the problem
        const double gu = a[54 + i] / sc[9 * (size_t)c + i];  // gradient of the unscaled problem
        gmax = fmax(gmax, fabs(gu));
        g2 += gu * gu;
      }
    }
    if (acc2) {
      const double* a2 = acc2 + (size_t)kK2W * c;
      int o = 0;
#pragma unroll
      for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = i; j < 9; ++j) { L[j * 9 + i] -= a2[o]; if (j != i) L[i * 9 + j] -= a2[o]; ++o; }
    }
    // in-place lower Cholesky (a non-positive pivot is noted and replaced by 1: the result is discarded then)
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      double d = L[j * 9 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) d -= L[j * 9 + k] * L[j * 9 + k];
      if (!(d > 0.0)) { ok = false; d = 1.0; }
      d = sqrt(d);
      L[j * 9 + j] = d;
#pragma unroll
      for (int i = j + 1; i < 9; ++i) {
        double s = L[i * 9 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= L[i * 9 + k] * L[j * 9 + k];
        L[i * 9 + j] = s / d;
      }
    }
    double* Mi = Minv + 81 * (size_t)c;
    if (!ok) ctl->fail = 1;
    // solve L L' col = e_j for each j
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      double col[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        double s = (i == j) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[i * 9 + k] * col[k];
        col[i] = s / L[i * 9 + i];
      }
#pragma unroll
      for (int i = 8; i >= 0; --i) {
        double s = col[i];
#pragma unroll
        for (int k = i + 1; k < 9; ++k) s -= L[k * 9 + i] * col[k];
        col[i] = s / L[i * 9 + i];
      }
#pragma unroll
      for (int i = 0; i < 9; ++i) Mi[j * 9 + i] = ok ? col[i] : 0.0;   // column-major: entry (i, j) at 9 j + i (the PCG kernels read it lane = row)
    }
  }
  // launched with 64-thread blocks (one wavefront): wave reductions, NOT the 256-thread block_sum / block_max
  const double m = wave_max(gmax);
  const double s = wave_sum(g2);
  if (threadIdx.x == 0) { partial[2 * (size_t)blockIdx.x] = m; partial[2 * (size_t)blockIdx.x + 1] = s; }
}

// out[slot] = sum (or max) over n entries of partial[i*stride + off]; single 1024-thread block.
__global__ __launch_bounds__(1024) void k_reduce(const double* __restrict__ partial, size_t n, int stride, int off,
                                                 int is_max, double* __restrict__ out) {
  __shared__ double s[16];
  double v = 0.0;
  for (size_t i = threadIdx.x; i < n; i += 1024) {
    const double x = partial[i * stride + off];
    v = is_max ? fmax(v, x) : v + x;
  }
  v = is_max ? wave_max(v) : wave_sum(v);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = s[0];
    for (int i = 1; i < 16; ++i) t = is_max ? fmax(t, s[i]) : t + s[i];
    *out = t;
  }
}

// scale = 1/(1+sqrt(n2)); inactive entries -> 1
__global__ void k_make_scale(size_t n, const double* __restrict__ n2, double* __restrict__ scale) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) scale[i] = 1.0 / (1.0 + sqrt(n2[i]));
}
// a[i] *= m[i]  (constant-block mask onto the Jacobi scale)
__global__ void k_mul(size_t n, const double* __restrict__ m, double* __restrict__ a) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] *= m[i];
}
__global__ void k_fill(size_t n, double v, double* __restrict__ a) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = v;
}

// candidate cameras: cand = cam - sc .* yc ; partial {|cam-cand|^2, |cand|^2} over active cameras
__global__ __launch_bounds__(256) void k_cam_update(int n_cam, const double* __restrict__ cam, const double* __restrict__ sc,
                                                    const double* __restrict__ yc, const uint8_t* __restrict__ active,
                                                    double* __restrict__ cand, double* __restrict__ partial) {
  __shared__ double s4[4];
  double d2 = 0.0, x2 = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 9 * (size_t)n_cam; i += (size_t)gridDim.x * blockDim.x) {
    const double c = cam[i] - yc[i] * sc[i];
    cand[i] = c;
    if (active[i / 9] && sc[i] != 0.0) {   // constant blocks (scale 0) are not part of the reduced program's norms
      const double e = cam[i] - c;
      d2 += e * e;
      x2 += c * c;
    }
  }
  const double a = block_sum(d2, s4);
  const double b = block_sum(x2, s4);
  if (threadIdx.x == 0) { partial[2 * (size_t)blockIdx.x] = a; partial[2 * (size_t)blockIdx.x + 1] = b; }
}

// sum of squares of active entries (|x|^2 at start): stride = entries per block (9 cams / 3 pts)
__global__ __launch_bounds__(256) void k_sumsq(size_t n, const double* __restrict__ a, const uint8_t* __restrict__ active, int per,
                                               const double* __restrict__ keep /*0 = constant entry, or null*/,
                                               double* __restrict__ partial) {
  __shared__ double s4[4];
  double x2 = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if ((!active || active[i / per]) && (!keep || keep[i] != 0.0)) x2 += a[i] * a[i];
  const double s = block_sum(x2, s4);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------
// PCG vector kernels (ConjugateGradientsSolver).  Fixed grid of kPcgBlocks
// blocks, grid-stride; dot products leave one partial per block which the consumer kernels
// sum in block order (deterministic, and identical on every rank).
// ---------------------------------------------------------------------------------------------
constexpr int kPcgBlocks = 120;     // blocks of pcg_dot
constexpr int kPcgCams = 28;        // cameras per block round of the fused kernels (252 of 256 threads)
constexpr int kPcgMaxBlocks = 4096; // cap on the number of partials a consumer sums

// sum of n per-block partials, in a fixed order; every thread gets the result
__device__ __forceinline__ double sum_partials(const double* __restrict__ part, int n, double* s4) {
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += kChunk) v += part[i];
  return block_sum(v, s4);
}

__device__ __forceinline__ bool pcg_bad(double v) { return v == 0.0 || isinf(v) || isnan(v); }

// The control block is double-buffered: C_k (state after k completed iterations) lives in ctl[k & 1], so a
// kernel that produces C_k never writes the buffer its sibling blocks are still reading (C_{k-1}).
//
// One PCG iteration j (1-based) is the product over the tiles and pcg_tail(j) (miba_pcg.hpp), which is
//   update(j):   alpha = rho_j / pq; x += alpha p; r -= alpha q; Q partials; z = M^-1 r; rho_{j+1} partials
//   chkdir(j+1): end-of-iteration test of j (C_{j-1} -> C_j), then rho_{j+1}, beta, p = z + beta p
// pcg_chkdir below is that second half as a kernel of its own; a solve launches it once, for j = 1 (p = z).
// Every block recomputes the scalars from the same partial arrays in the same order, so all blocks (and all
// ranks) take identical decisions.  part_rho is double-buffered by parity of the iteration that consumes it.

// x = 0, r = b, z = M^-1 b; partials |b|^2 and rho_1; C_0
__global__ __launch_bounds__(256) void pcg_start(int n_cam, const double* __restrict__ b, const double* __restrict__ Minv,
                                                 double* __restrict__ x, double* __restrict__ r, double* __restrict__ z,
                                                 Ctl* __restrict__ ctl /*[2]*/, double* __restrict__ part_nb,
                                                 double* __restrict__ part_rho1) {
  __shared__ double s_r[kPcgCams * 9];
  __shared__ double s4[4];
  const int tid = threadIdx.x;
  const int cl = tid / 9, k = tid % 9;
  double a_nb = 0.0, a_rho = 0.0;
  for (int c0 = blockIdx.x * kPcgCams; c0 < n_cam; c0 += gridDim.x * kPcgCams) {
    const int c = c0 + cl;
    const bool act = tid < kPcgCams * 9 && c < n_cam;
    const size_t i = 9 * (size_t)c + k;
    double v = 0.0;
    if (act) { v = b[i]; x[i] = 0.0; r[i] = v; s_r[tid] = v; a_nb += v * v; }
    __syncthreads();
    if (act) {
      const double* m = Minv + 81 * (size_t)c + k;   // row k of the camera's 9x9 block (stored column-major)
      double zz = 0.0;
#pragma unroll
      for (int j = 0; j < 9; ++j) zz += m[9 * j] * s_r[9 * cl + j];
      z[i] = zz;
      a_rho += v * zz;
    }
    __syncthreads();
  }
  const double s_nb = block_sum(a_nb, s4);
  const double s_rho = block_sum(a_rho, s4);
  if (tid == 0) {
    part_nb[blockIdx.x] = s_nb;
    part_rho1[blockIdx.x] = s_rho;
    if (blockIdx.x == 0) {
      Ctl c;
      c.rho = 1.0; c.rho_prev = 1.0; c.pq = 0.0; c.alpha = 0.0; c.beta = 0.0; c.Q0 = 0.0; c.Q1 = 0.0; c.zeta = 0.0;
      c.norm_b2 = 0.0; c.done = 0; c.status = 1; c.n_it = 0; c.fail = ctl[0].fail;
      ctl[0] = c;
      ctl[1] = c;
    }
  }
}
// finish start: norm_b == 0 => done (success, x = 0)
__global__ void pcg_start2(const double* __restrict__ part_nb, int nb, Ctl* __restrict__ ctl) {
  __shared__ double s4[4];
  const double nb2 = sum_partials(part_nb, nb, s4);
  if (threadIdx.x == 0) {
    ctl[0].norm_b2 = nb2; ctl[1].norm_b2 = nb2;
    if (nb2 == 0.0) { ctl[0].done = 1; ctl[0].status = 0; ctl[1].done = 1; ctl[1].status = 0; }
  }
}

__global__ __launch_bounds__(256) void pcg_chkdir(int j, size_t n, const double* __restrict__ z, double* __restrict__ p,
                                                  Ctl* __restrict__ ctl, const double* __restrict__ part_rho_prev,
                                                  const double* __restrict__ part_rho_cur, int nb,
                                                  const double* __restrict__ part_pq, int n_pq,
                                                  const double* __restrict__ part_q, double q_tol, int min_it, int max_it) {
  __shared__ double s4[4];
  const Ctl* in = (j >= 2) ? &ctl[j & 1] : &ctl[0];   // C_{j-2} (C_0 for j == 1)
  Ctl* out = &ctl[(j - 1) & 1];                       // C_{j-1}
  Ctl c = *in;
  const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
  if (c.done) {
    if (writer && j >= 2) *out = c;   // keep the final state in both buffers
    return;
  }
  double last_rho = c.rho;
  if (j >= 2) {
    const int it = j - 1;
    const double rho_it = sum_partials(part_rho_prev, nb, s4);
    const double pq = sum_partials(part_pq, n_pq, s4);
    c.n_it = it;
    c.pq = pq;
    if (pq <= 0.0 || isinf(pq) || isnan(pq)) { c.done = 1; c.status = 1; }
    else {
      const double alpha = rho_it / pq;
      c.alpha = alpha;
      if (isinf(alpha)) { c.done = 1; c.status = 2; }
      else {
        const double qs = sum_partials(part_q, nb, s4);
        const double Q1 = -qs;
        const double zeta = it * (Q1 - c.Q0) / Q1;
        c.Q1 = Q1;
        c.zeta = zeta;
        if (zeta < q_tol && it >= min_it) { c.done = 1; c.status = 0; }
        else {
          c.Q0 = Q1;
          if (it >= max_it) { c.done = 1; c.status = 1; }
        }
      }
    }
    if (c.done) {
      if (writer) *out = c;
      return;
    }
    last_rho = rho_it;
  }
  const double rho = sum_partials(part_rho_cur, nb, s4);
  bool bad = pcg_bad(rho);
  double beta = 0.0;
  if (!bad && j > 1) {
    beta = rho / last_rho;
    bad = pcg_bad(beta);
  }
  c.rho_prev = last_rho;
  c.rho = rho;
  c.beta = beta;
  if (bad) { c.done = 1; c.status = 2; c.n_it = j; }
  if (writer && (j >= 2 || bad)) *out = c;
  if (bad) return;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = (j == 1) ? z[i] : z[i] + beta * p[i];
}

}  // namespace miba
