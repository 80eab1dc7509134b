// miba_mf.hpp — matrix-free implicit-Schur product (K3-mf).
//
// Same operator as k_schur_matvec<0> (ImplicitSchurComplement::RightMultiply),
// but the Jacobian blocks are rebuilt per observation from the camera, the point and nothing
// else: MI355X has ~10 fp64 flop per HBM byte, and a stored 2x9 + 2x3 block costs 192 bytes per
// observation per PCG iteration.  Per observation only the 4-byte index word is read from HBM.
//
// Factorisation used (A = d r / d P_cam, E = A R, E' = A N with N = R, or I in the small-angle branch):
//   F x = E' (m x X + tau) + p (d x_f + f n x_k1 + f n^2 x_k2),   m = M (sc.x)_w,  tau = N' (sc.x)_t
//   F' w: with e' = E'' w, s = p.w :  rot = M' sum (X x e'),  tr = N sum e',  (f,k1,k2) = (sum d s, f sum n s, f sum n^2 s)
// so per (tile, camera) only 9 raw sums are accumulated in LDS and the camera-side 3x3 transforms
// are applied once per camera after the cross-tile reduction (k_cam_reduce_mf).
// Column scaling (Jacobi) is applied to the vectors, not to J: F_s x = F (sc.x), F_s' w = sc.(F' w),
// E_s y = E (sp.y).
#pragma once

#include <cfloat>

#include "miba_kernels.hpp"

namespace miba {

#ifndef MIBA_ACC_COPIES
#define MIBA_ACC_COPIES 4
#endif
constexpr int kAccCopies = MIBA_ACC_COPIES;  // copies of the per-tile camera accumulators (fewer same-address LDS atomics)
constexpr int kCamLds = 26;  // camera record stride in LDS: 13 x 16 bytes, read with ds_read_b128 (256 B/clk; the
                             // ds_read2_b64 the compiler would otherwise form runs at half that on gfx950)

// one camera record from LDS into registers: 13 ds_read_b128
__device__ __forceinline__ void mf_read_cam(double* cd /*[kCamLds]*/, const double* s_cam, int cs) {
  const double2* p = reinterpret_cast<const double2*>(s_cam) + (kCamLds / 2) * cs;
#pragma unroll
  for (int k = 0; k < kCamLds / 2; ++k) {
    const double2 v = p[k];
    cd[2 * k] = v.x; cd[2 * k + 1] = v.y;
  }
}

// [R 0..8 | t 9..11 | f k1 k2 12..14 | m 15..17 | tau 18..20 | xi 21..23 | small 24]
__device__ __forceinline__ void mf_stage_cam(double* dst, int cam, const double* __restrict__ ct,
                                             const double* __restrict__ cm, const double* __restrict__ sc,
                                             const double* __restrict__ x) {
  const double* T = ct + (size_t)kCamTabStride * cam;
  const double* M = cm + 10 * (size_t)cam;
  const double* s = sc + 9 * (size_t)cam;
  const double* xv = x + 9 * (size_t)cam;
  double xs[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) xs[k] = s[k] * xv[k];
#pragma unroll
  for (int k = 0; k < 15; ++k) dst[k] = T[k];
  if (M[9] == 2.0) {
    // camera model of the reference's functor, P = R~ (X - t): F x = E (m' x X + tau') with m' = -M~ x_w,
    // tau' = -(m' x t) - x_t, t = -R~' t~ (derivation: DESIGN.md, camera model #2); everything per observation is unchanged
    double mm[3], t[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) mm[i] = -(M[3 * i] * xs[0] + M[3 * i + 1] * xs[1] + M[3 * i + 2] * xs[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = -(T[i] * T[9] + T[3 + i] * T[10] + T[6 + i] * T[11]);
    dst[15] = mm[0]; dst[16] = mm[1]; dst[17] = mm[2];
    dst[18] = -(mm[1] * t[2] - mm[2] * t[1]) - xs[3];
    dst[19] = -(mm[2] * t[0] - mm[0] * t[2]) - xs[4];
    dst[20] = -(mm[0] * t[1] - mm[1] * t[0]) - xs[5];
    dst[21] = 0.0; dst[22] = 0.0; dst[23] = 0.0;   // constant intrinsics
    dst[24] = 0.0;
    return;
  }
  const bool small = M[9] == 1.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) dst[15 + i] = M[3 * i] * xs[0] + M[3 * i + 1] * xs[1] + M[3 * i + 2] * xs[2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    dst[18 + i] = small ? xs[3 + i] : (T[i] * xs[3] + T[3 + i] * xs[4] + T[6 + i] * xs[5]);  // N' x_t
  dst[21] = xs[6]; dst[22] = xs[7]; dst[23] = xs[8];
  dst[24] = small ? 1.0 : 0.0;
}

// Robust loss on every residual block (ceres::LossFunction; the reference passes nullptr, src/BundleAdjustment.cpp:69).
// HuberLoss(a) and CauchyLoss(a) as Ceres evaluates them on s = |r|^2: rho(s) and rho'(s).  Both have rho'' <= 0, for
// which Ceres' Corrector multiplies the block's residual and Jacobian rows by sqrt(rho') and nothing else: in J'J, J'r
// and the model cost change every observation's terms carry the weight om = rho', the cost is 1/2 sum rho.
// The point-side pass of K2 leaves om per slot for K3 and K4; kernels are templated on LOSS, so the default
// (no loss, what the reference runs) executes the same code as before.
enum { kLossNone = 0, kLossHuber = 1, kLossCauchy = 2 };
struct Loss { int type; double a; };
__device__ __forceinline__ void mf_loss(const Loss L, double s, double& rho, double& om) {
  const double b = L.a * L.a;
  if (L.type == kLossHuber && s > b) {
    const double r = sqrt(s);
    rho = 2.0 * L.a * r - b;
    om = fmax(DBL_MIN, L.a / r);
  } else if (L.type == kLossCauchy) {
    const double sum = 1.0 + s / b;
    rho = b * log(sum);
    om = fmax(DBL_MIN, 1.0 / sum);
  } else {
    rho = s;
    om = 1.0;
  }
}

struct MfGeo {
  double E[6];   // d r / d X       = A R   (unscaled)
  double Ep[6];  // A N: equals E, or A in the small-angle branch
  double px, py, n, d, f;
};

// cd: camera slot data (LDS or registers), X: point.  SMALL = false: no camera of the tile is in the
// small-angle branch, Ep is E.
template <bool SMALL = true>
__device__ __forceinline__ void mf_geom(const double* cd, double x0, double x1, double x2, MfGeo& g) {
  const double P0 = cd[0] * x0 + cd[1] * x1 + cd[2] * x2 + cd[9];
  const double P1 = cd[3] * x0 + cd[4] * x1 + cd[5] * x2 + cd[10];
  const double P2 = cd[6] * x0 + cd[7] * x1 + cd[8] * x2 + cd[11];
  const double f = cd[12], k1 = cd[13], k2 = cd[14];
  const double iz = mf_rcp(P2);
  const double px = -P0 * iz, py = -P1 * iz;
  const double n = px * px + py * py;
  const double d = 1.0 + n * (k1 + k2 * n);
  const double gg = k1 + 2.0 * k2 * n;
  const double m00 = f * (d + 2.0 * gg * px * px), m01 = f * (2.0 * gg * px * py), m11 = f * (d + 2.0 * gg * py * py);
  const double q0 = P0 * iz * iz, q1 = P1 * iz * iz;
  double A[6];
  A[0] = -m00 * iz; A[1] = -m01 * iz; A[2] = m00 * q0 + m01 * q1;
  A[3] = -m01 * iz; A[4] = -m11 * iz; A[5] = m01 * q0 + m11 * q1;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    g.E[j] = A[0] * cd[j] + A[1] * cd[3 + j] + A[2] * cd[6 + j];
    g.E[3 + j] = A[3] * cd[j] + A[4] * cd[3 + j] + A[5] * cd[6 + j];
  }
  if (SMALL) {
    const bool small = cd[24] == 1.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) g.Ep[j] = small ? A[j] : g.E[j];
  } else {
#pragma unroll
    for (int j = 0; j < 6; ++j) g.Ep[j] = g.E[j];
  }
  g.px = px; g.py = py; g.n = n; g.d = d; g.f = f;
}

// u = F_s x for this observation
__device__ __forceinline__ void mf_forward(const MfGeo& g, const double* cd, double x0, double x1, double x2, double& u0,
                                           double& u1) {
  // v = m x X + tau
  const double v0 = cd[16] * x2 - cd[17] * x1 + cd[18];
  const double v1 = cd[17] * x0 - cd[15] * x2 + cd[19];
  const double v2 = cd[15] * x1 - cd[16] * x0 + cd[20];
  const double si = g.d * cd[21] + g.f * g.n * (cd[22] + g.n * cd[23]);
  u0 = g.Ep[0] * v0 + g.Ep[1] * v1 + g.Ep[2] * v2 + g.px * si;
  u1 = g.Ep[3] * v0 + g.Ep[4] * v1 + g.Ep[5] * v2 + g.py * si;
}

// raw camera-side sums of F' w for this observation: out[0..2] = X x e', out[3..5] = e', out[6..8] = (d s, n s, n^2 s)
__device__ __forceinline__ void mf_transpose(const MfGeo& g, double x0, double x1, double x2, double w0, double w1, double* out) {
  const double e0 = g.Ep[0] * w0 + g.Ep[3] * w1;
  const double e1 = g.Ep[1] * w0 + g.Ep[4] * w1;
  const double e2 = g.Ep[2] * w0 + g.Ep[5] * w1;
  out[0] = x1 * e2 - x2 * e1;
  out[1] = x2 * e0 - x0 * e2;
  out[2] = x0 * e1 - x1 * e0;
  out[3] = e0; out[4] = e1; out[5] = e2;
  const double s = g.px * w0 + g.py * w1;
  out[6] = g.d * s;
  out[7] = g.n * s;
  out[8] = g.n * g.n * s;
}

// One wavefront = one 64-slot wavelet: a point's observations never leave their wavelet, so the
// per-point sum of E'u is a segmented scan over lanes (shuffles) and every lane applies its point's
// 3x3 block itself: no workgroup barrier and no LDS exchange inside the tile loop, waves of a
// workgroup run independently and hide each other's memory latency.  The workgroup only shares the
// per-tile camera data and the per-tile camera accumulators in LDS (one barrier after staging, one
// before the flush).
// value of v in the lane whose byte address (lane * 4) is given: two ds_bpermute_b32, no LDS storage
__device__ __forceinline__ double lane_read(double v, int lane_byte_addr) {
  const int lo = __builtin_amdgcn_ds_bpermute(lane_byte_addr, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(lane_byte_addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

// value of v in lane - 1 (0 in lane 0): wavefront shift through DPP, on the VALU instead of the LDS crossbar
__device__ __forceinline__ double lane_shr1(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// segmented inclusive scan of three values over the lanes of each point (a point's lanes are contiguous, pos =
// position in the run, maxrun = longest run in the wavelet, wave-uniform).  Steps 1 and 2 shift through DPP.
__device__ __forceinline__ void mf_seg_scan3(double& v0, double& v1, double& v2, int lane, int pos, int maxrun) {
  if (maxrun > 1) {
    const double in = (pos >= 1) ? 1.0 : 0.0;
    v0 = fma(lane_shr1(v0), in, v0); v1 = fma(lane_shr1(v1), in, v1); v2 = fma(lane_shr1(v2), in, v2);
  }
  if (maxrun > 2) {
    const double in = (pos >= 2) ? 1.0 : 0.0;
    v0 = fma(lane_shr1(lane_shr1(v0)), in, v0); v1 = fma(lane_shr1(lane_shr1(v1)), in, v1);
    v2 = fma(lane_shr1(lane_shr1(v2)), in, v2);
  }
  for (int d = 4; d < maxrun; d <<= 1) {
    const int src = (lane - d) << 2;   // byte address of the source lane (wraps harmlessly: masked by pos >= d)
    const double t0 = lane_read(v0, src), t1 = lane_read(v1, src), t2 = lane_read(v2, src);
    const double in = (pos >= d) ? 1.0 : 0.0;
    v0 = fma(t0, in, v0); v1 = fma(t1, in, v1); v2 = fma(t2, in, v2);
  }
}

// gather of X and W of a point record (lanes of one point read the same bytes)
__device__ __forceinline__ void mf_load_rec9(double* rec, const double* __restrict__ ptrec, int32_t w, int chunk_pt0) {
  const int pl = (w >= 0) ? (w & 0xFF) : 0;   // pad lanes read the chunk's first point: no branch, finite data
  const double* rp = ptrec + (size_t)kPtRec * (size_t)(chunk_pt0 + pl);
#pragma unroll
  for (int k = 0; k < 3; ++k) rec[k] = rp[k];       // X
#pragma unroll
  for (int k = 0; k < 6; ++k) rec[3 + k] = rp[6 + k];   // W
}

// Wavelet loop of one wave over its tile (wavelets wv, wv+4, ...), software-pipelined: the index word of
// wavelet i+2 and the point records of wavelet i+1 are in flight while wavelet i is computed.
template <bool SMALL, bool LOSS>
__device__ __forceinline__ void mf_matvec_tile(const Tile& t, const int32_t* __restrict__ idx, const int2* __restrict__ wl,
                                               const double* __restrict__ ptrec, const double* __restrict__ omega,
                                               const double* s_cam, double* s_acc) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nwl = t.nslot / kWave;
  const int acc_stride = t.ncs * 9;
  const int32_t* idxp = idx + (size_t)t.slot0 + lane;
  const int2* wlp = wl + (t.slot0 >> 6);   // per wavelet: {first point of the enclosing chunk, longest point run}
  int q = wv;
  if (q >= nwl) return;
  // the prefetches are unconditional (indices clamped to the wave's last wavelet): a load that is skipped on
  // some path would force every wait to drain the whole queue
  const int q_last = wv + 4 * ((nwl - 1 - wv) >> 2);
  int32_t w_cur = idxp[(size_t)q * kWave];
  int2 m_cur = wlp[q];
  const int q1 = min(q + 4, q_last);
  int32_t w_nxt = idxp[(size_t)q1 * kWave];
  int2 m_nxt = wlp[q1];
  // loss weights rho' ride in the same pipeline as the index words (LOSS only)
  const double* omp = LOSS ? omega + (size_t)t.slot0 + lane : nullptr;
  double om_cur = LOSS ? omp[(size_t)q * kWave] : 1.0, om_nxt = LOSS ? omp[(size_t)q1 * kWave] : 1.0;
  // one wavelet: index word w, point record rec (of the lane's point), longest point run maxrun
  auto wavelet = [&](const double* rec, const int32_t w, const int maxrun, const double om_slot /*loss weight rho'*/) {
    const bool valid = w >= 0;
    const double om = (LOSS && valid) ? om_slot : 1.0;   // pad slots hold no weight
    const int cs = valid ? ((w >> 8) & 0xFF) : 0, pl = w & 0xFF;
    const int pos = valid ? ((w >> 16) & 0x7F) : 0, cnt = valid ? (w >> 23) : 1;
    MfGeo geo;
    double u0, u1;
    double cd[kCamLds];
    mf_read_cam(cd, s_cam, cs);
    mf_geom<SMALL>(cd, rec[0], rec[1], rec[2], geo);
    mf_forward(geo, cd, rec[0], rec[1], rec[2], u0, u1);
    double v0 = valid ? geo.E[0] * u0 + geo.E[3] * u1 : 0.0;
    double v1 = valid ? geo.E[1] * u0 + geo.E[4] * u1 : 0.0;
    double v2 = valid ? geo.E[2] * u0 + geo.E[5] * u1 : 0.0;
    if (LOSS) { v0 *= om; v1 *= om; v2 *= om; }
    mf_seg_scan3(v0, v1, v2, lane, pos, maxrun);
    // total of the run sits in its last lane
    const int last = (lane - pos + cnt - 1) << 2;
    const double a0 = lane_read(v0, last), a1 = lane_read(v1, last), a2 = lane_read(v2, last);
    if (valid) {
      // rec[3..8] = diag(sp) C^-1 diag(sp);  y = -that * sum E'u
      const double y0 = -(rec[3] * a0 + rec[4] * a1 + rec[6] * a2);
      const double y1 = -(rec[4] * a0 + rec[5] * a1 + rec[7] * a2);
      const double y2 = -(rec[6] * a0 + rec[7] * a1 + rec[8] * a2);
      double w0 = u0 + geo.E[0] * y0 + geo.E[1] * y1 + geo.E[2] * y2;
      double w1 = u1 + geo.E[3] * y0 + geo.E[4] * y1 + geo.E[5] * y2;
      if (LOSS) { w0 *= om; w1 *= om; }
      double o[9];
      mf_transpose(geo, rec[0], rec[1], rec[2], w0, w1, o);
      // consecutive points go to different accumulator copies: lanes of one ds_add rarely share an address
      double* dst = s_acc + (pl & (kAccCopies - 1)) * acc_stride + 9 * cs;
#pragma unroll
      for (int k = 0; k < 9; ++k) atomicAdd(dst + k, o[k]);
    }
  };
  // two wavelets per trip, the point records alternating between two register sets: no register copies at the end
  // of a trip (3 % on the product, measured A/B inside one run: tools/ab_k3.sh)
  double rec_a[9], rec_b[9];
  mf_load_rec9(rec_a, ptrec, w_cur, m_cur.x);
  for (;;) {
    {
      mf_load_rec9(rec_b, ptrec, w_nxt, m_nxt.x);
      const int q2 = min(q + 8, q_last);
      const int32_t w_n2 = idxp[(size_t)q2 * kWave];
      const int2 m_n2 = wlp[q2];
      const double om_n2 = LOSS ? omp[(size_t)q2 * kWave] : 1.0;
      wavelet(rec_a, w_cur, m_cur.y, om_cur);
      w_cur = w_nxt; m_cur = m_nxt; w_nxt = w_n2; m_nxt = m_n2;
      if (LOSS) { om_cur = om_nxt; om_nxt = om_n2; }
    }
    q += 4;
    if (q >= nwl) break;
    {
      mf_load_rec9(rec_a, ptrec, w_nxt, m_nxt.x);
      const int q2 = min(q + 8, q_last);
      const int32_t w_n2 = idxp[(size_t)q2 * kWave];
      const int2 m_n2 = wlp[q2];
      const double om_n2 = LOSS ? omp[(size_t)q2 * kWave] : 1.0;
      wavelet(rec_b, w_cur, m_cur.y, om_cur);
      w_cur = w_nxt; m_cur = m_nxt; w_nxt = w_n2; m_nxt = m_n2;
      if (LOSS) { om_cur = om_nxt; om_nxt = om_n2; }
    }
    q += 4;
    if (q >= nwl) break;
  }
}

template <bool GIANT, bool LOSS>
__device__ __forceinline__ void mf_matvec_block(int tile_index, double* s_cam, double* s_acc, double* s4,
                                                   const Tile* __restrict__ tiles,
                                                   const Group* __restrict__ grps, const int32_t* __restrict__ idx,
                                                   const int32_t* __restrict__ ocam, const int2* __restrict__ wl,
                                                   const int32_t* __restrict__ tile_cams, const int32_t* __restrict__ slab_row,
                                                   const double* __restrict__ ct, const double* __restrict__ cm,
                                                   const double* __restrict__ sc, const double* __restrict__ X,
                                                   const double* __restrict__ ptrec /*[npt][kPtRec]*/,
                                                   const double* __restrict__ omega /*[slot] loss weights, LOSS only*/,
                                                   const double* __restrict__ x, double* __restrict__ slab /*[rows][9]*/) {
  const Tile t = tiles[tile_index];
  if (!GIANT && t.giant) return;
  const int tid = threadIdx.x;
  if (!GIANT) {
    bool small = false;
    if (tid < t.ncs) {
      mf_stage_cam(s_cam + kCamLds * tid, tile_cams[t.cam0 + tid], ct, cm, sc, x);
      small = s_cam[kCamLds * tid + 24] != 0.0;
    }
    for (int i = tid; i < kAccCopies * t.ncs * 9; i += kChunk) s_acc[i] = 0.0;
    const int any_small = __syncthreads_or(small);
    if (any_small) mf_matvec_tile<true, LOSS>(t, idx, wl, ptrec, omega, s_cam, s_acc);
    else mf_matvec_tile<false, LOSS>(t, idx, wl, ptrec, omega, s_cam, s_acc);
    __syncthreads();  // every wave's LDS atomics have landed
    const int acc_stride = t.ncs * 9;
    for (int i = tid; i < acc_stride; i += kChunk) {
      double v = s_acc[i];
#pragma unroll
      for (int c = 1; c < kAccCopies; ++c) v += s_acc[c * acc_stride + i];
      slab[9 * (size_t)slab_row[t.cam0 + i / 9] + i % 9] = v;
    }
  } else {
    for (int g = 0; g < t.ngrp; ++g) {
      const Group G = grps[t.grp0 + g];
      // giant point: its observations span G.nchunk chunks, one camera slot (= slab row) each
      const int p = G.pt0;
      const double x0 = X[3 * (size_t)p], x1 = X[3 * (size_t)p + 1], x2 = X[3 * (size_t)p + 2];
      double a0 = 0.0, a1 = 0.0, a2 = 0.0;
      for (int ch = 0; ch < G.nchunk; ++ch) {
        const size_t slot = (size_t)(G.chunk0 + ch) * kChunk + tid;
        double v0 = 0.0, v1 = 0.0, v2 = 0.0;
        if (idx[slot] >= 0) {
          double cd[kCamLds];
          mf_stage_cam(cd, ocam[slot], ct, cm, sc, x);
          MfGeo geo;
          double u0, u1;
          mf_geom(cd, x0, x1, x2, geo);
          mf_forward(geo, cd, x0, x1, x2, u0, u1);
          v0 = geo.E[0] * u0 + geo.E[3] * u1; v1 = geo.E[1] * u0 + geo.E[4] * u1; v2 = geo.E[2] * u0 + geo.E[5] * u1;
          if (LOSS) { const double om = omega[slot]; v0 *= om; v1 *= om; v2 *= om; }
        }
        a0 += block_sum(v0, s4); a1 += block_sum(v1, s4); a2 += block_sum(v2, s4);
      }
      const double* W = ptrec + (size_t)kPtRec * (size_t)p + 6;   // diag(sp) C^-1 diag(sp)
      const double y0 = -(W[0] * a0 + W[1] * a1 + W[3] * a2);
      const double y1 = -(W[1] * a0 + W[2] * a1 + W[4] * a2);
      const double y2 = -(W[3] * a0 + W[4] * a1 + W[5] * a2);
      for (int ch = 0; ch < G.nchunk; ++ch) {
        const size_t slot = (size_t)(G.chunk0 + ch) * kChunk + tid;
        if (idx[slot] < 0) continue;
        double cd[kCamLds];
        mf_stage_cam(cd, ocam[slot], ct, cm, sc, x);
        MfGeo geo;
        double u0, u1;
        mf_geom(cd, x0, x1, x2, geo);
        mf_forward(geo, cd, x0, x1, x2, u0, u1);
        double w0 = u0 + geo.E[0] * y0 + geo.E[1] * y1 + geo.E[2] * y2;
        double w1 = u1 + geo.E[3] * y0 + geo.E[4] * y1 + geo.E[5] * y2;
        if (LOSS) { const double om = omega[slot]; w0 *= om; w1 *= om; }
        double o[9];
        mf_transpose(geo, x0, x1, x2, w0, w1, o);
        double* dst = slab + 9 * (size_t)slab_row[t.cam0 + (int)(slot - (size_t)t.slot0)];
#pragma unroll
        for (int k = 0; k < 9; ++k) dst[k] = o[k];
      }
    }
  }
}

// K3-mf launch: the first n_giant blocks take the giant tiles of tile_list (they are the slowest blocks, so they
// start first), block n_giant + i takes tile i (a giant tile returns at once there): the few giant points run
// beside the ordinary tiles instead of in a launch of their own.
template <bool LOSS>
__global__ __launch_bounds__(256) void k_mf_matvec(int n_giant, const Tile* __restrict__ tiles, const int32_t* __restrict__ tile_list,
                                                   const Group* __restrict__ grps, const int32_t* __restrict__ idx,
                                                   const int32_t* __restrict__ ocam, const int2* __restrict__ wl,
                                                   const int32_t* __restrict__ tile_cams, const int32_t* __restrict__ slab_row,
                                                   const double* __restrict__ ct, const double* __restrict__ cm,
                                                   const double* __restrict__ sc, const double* __restrict__ X,
                                                   const double* __restrict__ ptrec /*[npt][kPtRec]*/,
                                                   const double* __restrict__ omega,
                                                   const double* __restrict__ x, double* __restrict__ slab /*[rows][9]*/,
                                                   const Ctl* __restrict__ ctl) {
  __shared__ __align__(16) double s_cam[kMaxCamSlots * kCamLds];
  __shared__ double s_acc[kAccCopies * kMaxCamSlots * 9];
  __shared__ double s4[4];
  if (ctl->done) return;
  if ((int)blockIdx.x >= n_giant)
    mf_matvec_block<false, LOSS>(blockIdx.x - n_giant, s_cam, s_acc, s4, tiles, grps, idx, ocam, wl, tile_cams, slab_row, ct, cm, sc, X,
                                 ptrec, omega, x, slab);
  else
    mf_matvec_block<true, LOSS>(tile_list[blockIdx.x], s_cam, s_acc, s4, tiles, grps, idx, ocam, wl, tile_cams, slab_row, ct, cm, sc, X,
                                ptrec, omega, x, slab);
}

// =============================================================================================
// Matrix-free Schur init (K2-mf) and back-substitution (K4-mf): the LM loop never stores J.
// =============================================================================================

// camera record of K2-mf: [R 0..8 | t 9..11 | f k1 k2 12..14 | M 15..23 | small 24]
__device__ __forceinline__ void mf_stage_cam_init(double* dst, int cam, const double* __restrict__ ct,
                                                  const double* __restrict__ cm) {
  const double* T = ct + (size_t)kCamTabStride * cam;
  const double* M = cm + 10 * (size_t)cam;
#pragma unroll
  for (int k = 0; k < 15; ++k) dst[k] = T[k];
#pragma unroll
  for (int k = 0; k < 9; ++k) dst[15 + k] = M[k];
  dst[24] = M[9];
}

// Unscaled Jacobian blocks and residual of one observation (same derivative as k_eval<true>, through
// d(R X)/dw = -N [X]x M):  F row i = [ (X x ep_i)' M | A_i | d p_i, f n p_i, f n^2 p_i ],  E = A R.
__device__ __forceinline__ void mf_blocks(const double* cd, double x0, double x1, double x2, double ox, double oy, double* F,
                                          double* E, double& r0, double& r1) {
  const double P0 = cd[0] * x0 + cd[1] * x1 + cd[2] * x2 + cd[9];
  const double P1 = cd[3] * x0 + cd[4] * x1 + cd[5] * x2 + cd[10];
  const double P2 = cd[6] * x0 + cd[7] * x1 + cd[8] * x2 + cd[11];
  const double f = cd[12], k1 = cd[13], k2 = cd[14];
  const double iz = mf_rcp(P2);
  const double px = -P0 * iz, py = -P1 * iz;
  const double n = px * px + py * py;
  const double d = 1.0 + n * (k1 + k2 * n);
  r0 = f * d * px - ox;
  r1 = f * d * py - oy;
  const double gg = k1 + 2.0 * k2 * n;
  const double m00 = f * (d + 2.0 * gg * px * px), m01 = f * (2.0 * gg * px * py), m11 = f * (d + 2.0 * gg * py * py);
  const double q0 = P0 * iz * iz, q1 = P1 * iz * iz;
  double A[6];
  A[0] = -m00 * iz; A[1] = -m01 * iz; A[2] = m00 * q0 + m01 * q1;
  A[3] = -m01 * iz; A[4] = -m11 * iz; A[5] = m01 * q0 + m11 * q1;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    E[j] = A[0] * cd[j] + A[1] * cd[3 + j] + A[2] * cd[6 + j];
    E[3 + j] = A[3] * cd[j] + A[4] * cd[3 + j] + A[5] * cd[6 + j];
  }
  if (cd[24] == 2.0) {
    // the reference's functor (P = R~ (X - t)): rotation columns ((t - X) x e_i)' M~, translation columns -e_i, constant
    // intrinsics (zero columns); t = -R~' t~
    const double t0 = -(cd[0] * cd[9] + cd[3] * cd[10] + cd[6] * cd[11]) - x0;
    const double t1 = -(cd[1] * cd[9] + cd[4] * cd[10] + cd[7] * cd[11]) - x1;
    const double t2 = -(cd[2] * cd[9] + cd[5] * cd[10] + cd[8] * cd[11]) - x2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const double e0 = E[3 * i], e1 = E[3 * i + 1], e2 = E[3 * i + 2];
      const double g0 = t1 * e2 - t2 * e1, g1 = t2 * e0 - t0 * e2, g2 = t0 * e1 - t1 * e0;
#pragma unroll
      for (int j = 0; j < 3; ++j) F[9 * i + j] = g0 * cd[15 + j] + g1 * cd[18 + j] + g2 * cd[21 + j];
      F[9 * i + 3] = -e0; F[9 * i + 4] = -e1; F[9 * i + 5] = -e2;
      F[9 * i + 6] = 0.0; F[9 * i + 7] = 0.0; F[9 * i + 8] = 0.0;
    }
    return;
  }
  const bool small = cd[24] == 1.0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const double e0 = small ? A[3 * i] : E[3 * i], e1 = small ? A[3 * i + 1] : E[3 * i + 1], e2 = small ? A[3 * i + 2] : E[3 * i + 2];
    const double g0 = x1 * e2 - x2 * e1, g1 = x2 * e0 - x0 * e2, g2 = x0 * e1 - x1 * e0;
#pragma unroll
    for (int j = 0; j < 3; ++j) F[9 * i + j] = g0 * cd[15 + j] + g1 * cd[18 + j] + g2 * cd[21 + j];
#pragma unroll
    for (int j = 0; j < 3; ++j) F[9 * i + 3 + j] = A[3 * i + j];
  }
  F[6] = d * px;          F[15] = d * py;
  F[7] = f * n * px;      F[16] = f * n * py;
  F[8] = f * n * n * px;  F[17] = f * n * n * py;
}

// per-point part of the Schur init from the raw sums a = {E'E (00,10,11,20,21,22), E'r}: scaled C^-1 (ci),
// scaled gradient (g), scaled diag(E'E) (dg), W = sp C^-1 sp and z = sp.(C^-1 g).  false: C not SPD.
__device__ __forceinline__ bool mf_point_solve(const double* a, double s0, double s1, double s2, double inv_radius, double dmin,
                                               double dmax, double* ci, double* g, double* dg, double* W, double* z) {
  double c[6] = {s0 * a[0] * s0, s1 * a[1] * s0, s1 * a[2] * s1, s2 * a[3] * s0, s2 * a[4] * s1, s2 * a[5] * s2};
  dg[0] = c[0]; dg[1] = c[2]; dg[2] = c[5];
  c[0] += fmin(fmax(c[0], dmin), dmax) * inv_radius;
  c[2] += fmin(fmax(c[2], dmin), dmax) * inv_radius;
  c[5] += fmin(fmax(c[5], dmin), dmax) * inv_radius;
  const bool ok = spd3_inverse(c, ci);
  if (!ok) {
#pragma unroll
    for (int k = 0; k < 6; ++k) ci[k] = 0.0;
  }
  g[0] = s0 * a[6]; g[1] = s1 * a[7]; g[2] = s2 * a[8];
  W[0] = s0 * ci[0] * s0; W[1] = s1 * ci[1] * s0; W[2] = s1 * ci[2] * s1;
  W[3] = s2 * ci[3] * s0; W[4] = s2 * ci[4] * s1; W[5] = s2 * ci[5] * s2;
  z[0] = s0 * (ci[0] * g[0] + ci[1] * g[1] + ci[3] * g[2]);
  z[1] = s1 * (ci[1] * g[0] + ci[2] * g[1] + ci[4] * g[2]);
  z[2] = s2 * (ci[3] * g[0] + ci[4] * g[1] + ci[5] * g[2]);
  return ok;
}

// gradient norms of the unscaled problem for one point: g / sp is E'r of the unscaled blocks, a[6..8]
__device__ __forceinline__ void mf_grad_norms(const double* a, double& gmax, double& g2) {
  gmax = fmax(gmax, fmax(fabs(a[6]), fmax(fabs(a[7]), fabs(a[8]))));
  g2 += a[6] * a[6] + a[7] * a[7] + a[8] * a[8];
}

// residual and point block E = A R only (the point-side pass needs no F)
__device__ __forceinline__ void mf_resid_E(const double* cd, double x0, double x1, double x2, double ox, double oy, double* E,
                                           double& r0, double& r1) {
  const double P0 = cd[0] * x0 + cd[1] * x1 + cd[2] * x2 + cd[9];
  const double P1 = cd[3] * x0 + cd[4] * x1 + cd[5] * x2 + cd[10];
  const double P2 = cd[6] * x0 + cd[7] * x1 + cd[8] * x2 + cd[11];
  const double f = cd[12], k1 = cd[13], k2 = cd[14];
  const double iz = mf_rcp(P2);
  const double px = -P0 * iz, py = -P1 * iz;
  const double n = px * px + py * py;
  const double d = 1.0 + n * (k1 + k2 * n);
  r0 = f * d * px - ox;
  r1 = f * d * py - oy;
  const double gg = k1 + 2.0 * k2 * n;
  const double m00 = f * (d + 2.0 * gg * px * px), m01 = f * (2.0 * gg * px * py), m11 = f * (d + 2.0 * gg * py * py);
  const double q0 = P0 * iz * iz, q1 = P1 * iz * iz;
  double A[6];
  A[0] = -m00 * iz; A[1] = -m01 * iz; A[2] = m00 * q0 + m01 * q1;
  A[3] = -m01 * iz; A[4] = -m11 * iz; A[5] = m01 * q0 + m11 * q1;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    E[j] = A[0] * cd[j] + A[1] * cd[3 + j] + A[2] * cd[6 + j];
    E[3 + j] = A[3] * cd[j] + A[4] * cd[3 + j] + A[5] * cd[6 + j];
  }
}

// K2-mf, camera side: one wave per segment of at most kSegObs observations of ONE camera (camera-major copy of the
// observation list).  Every lane keeps the 63 sums (upper triangle of F'F, F'v, F'r) of its observations in
// registers; no atomics, no LDS until one cross-lane reduction per segment.  v = r - E z with the point's z from
// the point-side pass (point record).  Row sg of slab gets the segment's sums; k_cam_reduce64 adds a camera's segments.
constexpr int kSegObs = 1024;
template <bool LOSS>
__global__ __launch_bounds__(256) void k_mf_camrows(int n_seg, const int4* __restrict__ seg /*{cam, first, count, -}*/,
                                                    const int32_t* __restrict__ cmo_pt, const double2* __restrict__ cmo_xy,
                                                    const double* __restrict__ ct, const double* __restrict__ cm,
                                                    const double* __restrict__ ptrec, const Loss loss,
                                                    double* __restrict__ slab /*[n_seg][64]*/, const int xcd_remap) {
  __shared__ double s_red[4][64 * 9];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // Workgroups go to the eight XCDs in turn (blockIdx mod 8), each with an L2 of its own; the segments are in camera order and
  // neighbouring cameras gather the same point records: XCD x takes the x-th eighth of the segments, so that those gathers meet in
  // one L2 instead of being fetched by eight (the grid is padded to a multiple of eight workgroups; MIBA_XCD_REMAP, host side)
  const int blk = xcd_remap ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int sg = blk * 4 + wv;
  if (sg >= n_seg) return;
  const int4 S = seg[sg];
  const int cam = __builtin_amdgcn_readfirstlane(S.x);
  const int first = __builtin_amdgcn_readfirstlane(S.y), n = __builtin_amdgcn_readfirstlane(S.z);
  double cd[kCamLds];
  mf_stage_cam_init(cd, cam, ct, cm);
  double acc[63];
#pragma unroll
  for (int k = 0; k < 63; ++k) acc[k] = 0.0;
  // software pipeline with clamped (unconditional) prefetches: point index / measurement two ahead, point record one ahead
  const int32_t* ptp = cmo_pt + first;
  const double2* xyp = cmo_xy + first;
  int i = lane;
  int32_t pt_c = ptp[min(i, n - 1)];
  double2 xy_c = xyp[min(i, n - 1)];
  int32_t pt_n = ptp[min(i + 64, n - 1)];
  double2 xy_n = xyp[min(i + 64, n - 1)];
  double q_c[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) q_c[k] = ptrec[(size_t)kPtRec * (size_t)pt_c + k];   // X, z
  const int n_it = (n + 63) >> 6;
  for (int it = 0; it < n_it; ++it, i += 64) {
    double q_n[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) q_n[k] = ptrec[(size_t)kPtRec * (size_t)pt_n + k];
    const int32_t pt_n2 = ptp[min(i + 128, n - 1)];
    const double2 xy_n2 = xyp[min(i + 128, n - 1)];
    if (i < n) {
      double F[18], E[6], r0, r1;
      mf_blocks(cd, q_c[0], q_c[1], q_c[2], xy_c.x, xy_c.y, F, E, r0, r1);
      const double v0 = r0 - (E[0] * q_c[3] + E[1] * q_c[4] + E[2] * q_c[5]);
      const double v1 = r1 - (E[3] * q_c[3] + E[4] * q_c[4] + E[5] * q_c[5]);
      double Fw[LOSS ? 18 : 1];   // om F: the one weighted factor of F'F, F'v and F'r
      if (LOSS) {
        double rho, om;
        mf_loss(loss, r0 * r0 + r1 * r1, rho, om);
#pragma unroll
        for (int k = 0; k < 18; ++k) Fw[k] = om * F[k];
      }
      const double* Fl = LOSS ? Fw : F;
      int o = 0;
#pragma unroll
      for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int b = a; b < 9; ++b) {
          acc[o] = fma(Fl[a], F[b], fma(Fl[9 + a], F[9 + b], acc[o]));
          ++o;
        }
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        acc[45 + k] = fma(Fl[k], v0, fma(Fl[9 + k], v1, acc[45 + k]));
        acc[54 + k] = fma(Fl[k], r0, fma(Fl[9 + k], r1, acc[54 + k]));
      }
    }
    pt_n = pt_n2; xy_c = xy_n; xy_n = xy_n2;
#pragma unroll
    for (int k = 0; k < 6; ++k) q_c[k] = q_n[k];
  }
  // cross-lane sums, nine components per round through the wave's LDS patch (LDS operations of one wave execute in
  // order, so no barrier is needed between a round's writes and its reads)
  double* sr = s_red[wv];
  const int sub = lane / 9, k9 = lane % 9;
  double* row = slab + (size_t)kK2W * sg;
#pragma unroll
  for (int r = 0; r < 7; ++r) {
#pragma unroll
    for (int k = 0; k < 9; ++k) sr[lane * 9 + k] = acc[9 * r + k];
    __builtin_amdgcn_wave_barrier();
    double a = 0.0;
    if (lane < 63)
      for (int l = sub; l < 64; l += 7) a += sr[l * 9 + k9];
    double tot = a;
#pragma unroll
    for (int s2 = 1; s2 < 7; ++s2) tot += __shfl(a, k9 + 9 * s2, 64);
    if (lane < 9) row[9 * r + lane] = tot;
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) row[63] = 0.0;
}

// K2-mf, point side: residual, cost and per-point Schur elimination straight from the state (no F, no camera rows:
// k_mf_camrows takes X and z from the point records this kernel writes).  The point scaling sp enters through
// mf_point_solve.  partial_cost[tile] = 0.5 sum r^2.
template <bool GIANT, bool LOSS>
__global__ __launch_bounds__(256) void k_mf_init(const Tile* __restrict__ tiles, const int32_t* __restrict__ tile_list,
                                                 const Group* __restrict__ grps, const int32_t* __restrict__ idx,
                                                 const int32_t* __restrict__ ocam, const int2* __restrict__ wl,
                                                 const int32_t* __restrict__ tile_cams, const int32_t* __restrict__ slab_row,
                                                 const double* __restrict__ oxy, const double* __restrict__ ct,
                                                 const double* __restrict__ cm, const double* __restrict__ X,
                                                 const double* __restrict__ sp, double inv_radius, double dmin, double dmax,
                                                 double* __restrict__ gp, double* __restrict__ dgp /*diag E'E, or null*/,
                                                 double* __restrict__ ptrec,
                                                 double* __restrict__ partial /*[tile][4]: cost, max |g_p|, |g_p|^2*/,
                                                 Ctl* __restrict__ ctl, const Loss loss, double* __restrict__ omega /*[slot], LOSS*/) {
  __shared__ __align__(16) double s_cam[GIANT ? 2 : kMaxCamSlots * kCamLds];
  __shared__ double s4[4];
  const int tile_id = GIANT ? tile_list[blockIdx.x] : blockIdx.x;
  const Tile t = tiles[tile_id];
  if (!GIANT && t.giant) return;
  const int tid = threadIdx.x;
  double cost = 0.0, gmax = 0.0, g2 = 0.0;   // gradient norms of the unscaled problem: g / sp
  if (!GIANT) {
    if (tid < t.ncs) mf_stage_cam_init(s_cam + kCamLds * tid, tile_cams[t.cam0 + tid], ct, cm);
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6;
    const int nwl = t.nslot / kWave;
    for (int q = wv; q < nwl; q += 4) {
      const size_t slot = (size_t)t.slot0 + (size_t)q * kWave + lane;
      const int32_t w = idx[slot];
      const int2 wq = wl[(t.slot0 >> 6) + q];   // {first point of the enclosing chunk, longest point run}
      const int maxrun = wq.y, gpt0 = wq.x;
      const bool valid = w >= 0;
      const int cs = (w >> 8) & 0xFF, pl = w & 0xFF;
      const int pos = valid ? ((w >> 16) & 0x7F) : 0, cnt = valid ? (w >> 23) : 1;
      const size_t p = (size_t)(gpt0 + pl);
      double E[6], r0 = 0.0, r1 = 0.0, v[9];
      double x0 = 0, x1 = 0, x2 = 0, s0 = 1, s1 = 1, s2 = 1;
#pragma unroll
      for (int k = 0; k < 9; ++k) v[k] = 0.0;
      if (valid) {
        x0 = X[3 * p]; x1 = X[3 * p + 1]; x2 = X[3 * p + 2];
        s0 = sp[3 * p]; s1 = sp[3 * p + 1]; s2 = sp[3 * p + 2];
        const size_t chunk = slot / kChunk, l = slot % kChunk;
        const double ox = oxy[chunk * (2 * kChunk) + l], oy = oxy[chunk * (2 * kChunk) + kChunk + l];
        double cd[kCamLds];
        mf_read_cam(cd, s_cam, cs);
        mf_resid_E(cd, x0, x1, x2, ox, oy, E, r0, r1);
        v[0] = E[0] * E[0] + E[3] * E[3];
        v[1] = E[1] * E[0] + E[4] * E[3];
        v[2] = E[1] * E[1] + E[4] * E[4];
        v[3] = E[2] * E[0] + E[5] * E[3];
        v[4] = E[2] * E[1] + E[5] * E[4];
        v[5] = E[2] * E[2] + E[5] * E[5];
        v[6] = E[0] * r0 + E[3] * r1;
        v[7] = E[1] * r0 + E[4] * r1;
        v[8] = E[2] * r0 + E[5] * r1;
        if (LOSS) {
          double rho, om;
          mf_loss(loss, r0 * r0 + r1 * r1, rho, om);
          cost += rho;
          omega[slot] = om;
#pragma unroll
          for (int k = 0; k < 9; ++k) v[k] *= om;
        } else {
          cost += r0 * r0 + r1 * r1;
        }
      }
      mf_seg_scan3(v[0], v[1], v[2], lane, pos, maxrun);
      mf_seg_scan3(v[3], v[4], v[5], lane, pos, maxrun);
      mf_seg_scan3(v[6], v[7], v[8], lane, pos, maxrun);
      const int last = (lane - pos + cnt - 1) << 2;
      double a[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) a[k] = lane_read(v[k], last);
      double ci[6], g[3], dg[3], W[6], z[3];
      const bool ok = mf_point_solve(a, s0, s1, s2, inv_radius, dmin, dmax, ci, g, dg, W, z);
      if (valid) {
        if (pos == 0) {

          if (!ok) ctl->fail = 1;
#pragma unroll
          for (int k = 0; k < 3; ++k) gp[3 * p + k] = g[k];
          if (dgp) {
#pragma unroll
            for (int k = 0; k < 3; ++k) dgp[3 * p + k] = dg[k];
          }
          mf_grad_norms(a, gmax, g2);
          double* rp = ptrec + (size_t)kPtRec * p;
          rp[0] = x0; rp[1] = x1; rp[2] = x2;
#pragma unroll
          for (int k = 0; k < 6; ++k) rp[6 + k] = W[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) rp[3 + k] = z[k];
        }
      }
    }
  } else {
    for (int gi = 0; gi < t.ngrp; ++gi) {
      const Group G = grps[t.grp0 + gi];
      const size_t p = (size_t)G.pt0;
      const double x0 = X[3 * p], x1 = X[3 * p + 1], x2 = X[3 * p + 2];
      const double s0 = sp[3 * p], s1 = sp[3 * p + 1], s2 = sp[3 * p + 2];
      double a[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) a[k] = 0.0;
      for (int ch = 0; ch < G.nchunk; ++ch) {
        const size_t chunk = (size_t)(G.chunk0 + ch);
        const size_t slot = chunk * kChunk + tid;
        double v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = 0.0;
        if (idx[slot] >= 0) {
          double cd[kCamLds], E[6], r0, r1;
          mf_stage_cam_init(cd, ocam[slot], ct, cm);
          mf_resid_E(cd, x0, x1, x2, oxy[chunk * (2 * kChunk) + tid], oxy[chunk * (2 * kChunk) + kChunk + tid], E, r0, r1);
          v[0] = E[0] * E[0] + E[3] * E[3]; v[1] = E[1] * E[0] + E[4] * E[3]; v[2] = E[1] * E[1] + E[4] * E[4];
          v[3] = E[2] * E[0] + E[5] * E[3]; v[4] = E[2] * E[1] + E[5] * E[4]; v[5] = E[2] * E[2] + E[5] * E[5];
          v[6] = E[0] * r0 + E[3] * r1; v[7] = E[1] * r0 + E[4] * r1; v[8] = E[2] * r0 + E[5] * r1;
          if (LOSS) {
            double rho, om;
            mf_loss(loss, r0 * r0 + r1 * r1, rho, om);
            cost += rho;
            omega[slot] = om;
#pragma unroll
            for (int k = 0; k < 9; ++k) v[k] *= om;
          } else {
            cost += r0 * r0 + r1 * r1;
          }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) a[k] += block_sum(v[k], s4);
      }
      double ci[6], g[3], dg[3], W[6], z[3];
      const bool ok = mf_point_solve(a, s0, s1, s2, inv_radius, dmin, dmax, ci, g, dg, W, z);
      if (tid == 0) {
        if (!ok) ctl->fail = 1;
#pragma unroll
        for (int k = 0; k < 3; ++k) gp[3 * p + k] = g[k];
        if (dgp) {
#pragma unroll
          for (int k = 0; k < 3; ++k) dgp[3 * p + k] = dg[k];
        }
        mf_grad_norms(a, gmax, g2);
        double* rp = ptrec + (size_t)kPtRec * p;
        rp[0] = x0; rp[1] = x1; rp[2] = x2;
#pragma unroll
        for (int k = 0; k < 6; ++k) rp[6 + k] = W[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) rp[3 + k] = z[k];
      }
    }
  }
  const double tot = block_sum(cost, s4);
  const double bm = block_max(gmax, s4);
  const double b2 = block_sum(g2, s4);
  if (tid == 0) {
    partial[4 * (size_t)tile_id + 0] = 0.5 * tot;
    partial[4 * (size_t)tile_id + 1] = bm;
    partial[4 * (size_t)tile_id + 2] = b2;
    partial[4 * (size_t)tile_id + 3] = 0.0;
  }
}

// The point-major wavelet walk of K4 over one tile whose cameras are staged in s_cam (with y_c applied): per observation u = F y_c
// and E'u, per point z = zg - W sum E'u, the candidate point and yp, per observation m = u + E z; adds to mcc, dp2, xp2.
template <bool LOSS>
__device__ __forceinline__ void mf_backsub_tile(const Tile& t, const int32_t* __restrict__ idx, const int2* __restrict__ wl,
                                                const double* __restrict__ oxy, const double* __restrict__ sp,
                                                const double* __restrict__ ptrec, const double* s_cam, double* __restrict__ yp,
                                                double* __restrict__ Xc, const double* __restrict__ omega, double& mcc, double& dp2, double& xp2) {
  const int tid = threadIdx.x;
  {
    const int lane = tid & 63, wv = tid >> 6;
    const int nwl = t.nslot / kWave;
    for (int q = wv; q < nwl; q += 4) {
      const size_t slot = (size_t)t.slot0 + (size_t)q * kWave + lane;
      const int32_t w = idx[slot];
      const int2 wq = wl[(t.slot0 >> 6) + q];   // {first point of the enclosing chunk, longest point run}
      const int maxrun = wq.y, gpt0 = wq.x;
      const bool valid = w >= 0;
      const int cs = (w >> 8) & 0xFF, pl = w & 0xFF;
      const int pos = valid ? ((w >> 16) & 0x7F) : 0, cnt = valid ? (w >> 23) : 1;
      const size_t p = (size_t)(gpt0 + pl);
      double rec[kPtRec];
#pragma unroll
      for (int k = 0; k < kPtRec; ++k) rec[k] = 0.0;
      MfGeo geo;
      double u0 = 0, u1 = 0, v0 = 0, v1 = 0, v2 = 0, r0 = 0, r1 = 0, om = 1.0;
      if (valid) {
        const double* rp = ptrec + (size_t)kPtRec * p;
#pragma unroll
        for (int k = 0; k < kPtRec; ++k) rec[k] = rp[k];
        double cd[kCamLds];
        mf_read_cam(cd, s_cam, cs);
        mf_geom(cd, rec[0], rec[1], rec[2], geo);
        mf_forward(geo, cd, rec[0], rec[1], rec[2], u0, u1);
        v0 = geo.E[0] * u0 + geo.E[3] * u1;
        v1 = geo.E[1] * u0 + geo.E[4] * u1;
        v2 = geo.E[2] * u0 + geo.E[5] * u1;
        if (LOSS) { om = omega[slot]; v0 *= om; v1 *= om; v2 *= om; }
        const size_t chunk = slot / kChunk, l = slot % kChunk;
        r0 = geo.f * geo.d * geo.px - oxy[chunk * (2 * kChunk) + l];
        r1 = geo.f * geo.d * geo.py - oxy[chunk * (2 * kChunk) + kChunk + l];
      }
      mf_seg_scan3(v0, v1, v2, lane, pos, maxrun);
      const int last = (lane - pos + cnt - 1) << 2;
      const double a0 = lane_read(v0, last), a1 = lane_read(v1, last), a2 = lane_read(v2, last);
      if (valid) {
        const double z0 = rec[3] - (rec[6] * a0 + rec[7] * a1 + rec[9] * a2);
        const double z1 = rec[4] - (rec[7] * a0 + rec[8] * a1 + rec[10] * a2);
        const double z2 = rec[5] - (rec[9] * a0 + rec[10] * a1 + rec[11] * a2);
        const double m0 = u0 + geo.E[0] * z0 + geo.E[1] * z1 + geo.E[2] * z2;
        const double m1 = u1 + geo.E[3] * z0 + geo.E[4] * z1 + geo.E[5] * z2;
        mcc += om * (m0 * (r0 - 0.5 * m0) + m1 * (r1 - 0.5 * m1));
        if (pos == 0) {
          if (yp) { yp[3 * p] = z0 * mf_rcp(sp[3 * p]); yp[3 * p + 1] = z1 * mf_rcp(sp[3 * p + 1]); yp[3 * p + 2] = z2 * mf_rcp(sp[3 * p + 2]); }
          const double c0 = rec[0] - z0, c1 = rec[1] - z1, c2 = rec[2] - z2;
          Xc[3 * p] = c0; Xc[3 * p + 1] = c1; Xc[3 * p + 2] = c2;
          const double e0 = rec[0] - c0, e1 = rec[1] - c1, e2 = rec[2] - c2;   // step norm exactly as Ceres: |x - x_cand|
          dp2 += e0 * e0 + e1 * e1 + e2 * e2;
          xp2 += c0 * c0 + c1 * c1 + c2 * c2;
        }
      }
    }
  }
}

// K4-mf: back-substitution  z_p = sp.(C^-1 (g_p - sum E_s' F_s y_c)) = zg - W sum E'u, candidate points
// Xc = X - z, yp = z / sp, and partial[4*tile + {0,1,2}] = {sum m.(r - m/2), |X - Xc|^2, |Xc|^2}, m = u + E z.
template <bool GIANT, bool LOSS>
__global__ __launch_bounds__(256) void k_mf_backsub(const Tile* __restrict__ tiles, const int32_t* __restrict__ tile_list,
                                                    const Group* __restrict__ grps, const int32_t* __restrict__ idx,
                                                    const int32_t* __restrict__ ocam, const int2* __restrict__ wl,
                                                    const int32_t* __restrict__ tile_cams, const double* __restrict__ oxy,
                                                    const double* __restrict__ ct, const double* __restrict__ cm,
                                                    const double* __restrict__ sc, const double* __restrict__ sp,
                                                    const double* __restrict__ ptrec, const double* __restrict__ yc,
                                                    double* __restrict__ yp, double* __restrict__ Xc,
                                                    double* __restrict__ partial, const double* __restrict__ omega /*LOSS*/) {
  __shared__ __align__(16) double s_cam[GIANT ? 2 : kMaxCamSlots * kCamLds];
  __shared__ double s4[4];
  const int tile_id = GIANT ? tile_list[blockIdx.x] : blockIdx.x;
  const Tile t = tiles[tile_id];
  if (!GIANT && t.giant) return;
  const int tid = threadIdx.x;
  double mcc = 0.0, dp2 = 0.0, xp2 = 0.0;
  if (!GIANT) {
    if (tid < t.ncs) mf_stage_cam(s_cam + kCamLds * tid, tile_cams[t.cam0 + tid], ct, cm, sc, yc);
    __syncthreads();
    mf_backsub_tile<LOSS>(t, idx, wl, oxy, sp, ptrec, s_cam, yp, Xc, omega, mcc, dp2, xp2);
  } else {
    for (int gi = 0; gi < t.ngrp; ++gi) {
      const Group G = grps[t.grp0 + gi];
      const size_t p = (size_t)G.pt0;
      double rec[kPtRec];
#pragma unroll
      for (int k = 0; k < kPtRec; ++k) rec[k] = ptrec[(size_t)kPtRec * p + k];
      double a0 = 0.0, a1 = 0.0, a2 = 0.0;
      for (int ch = 0; ch < G.nchunk; ++ch) {
        const size_t slot = (size_t)(G.chunk0 + ch) * kChunk + tid;
        double v0 = 0.0, v1 = 0.0, v2 = 0.0;
        if (idx[slot] >= 0) {
          double cd[kCamLds];
          mf_stage_cam(cd, ocam[slot], ct, cm, sc, yc);
          MfGeo geo;
          double u0, u1;
          mf_geom(cd, rec[0], rec[1], rec[2], geo);
          mf_forward(geo, cd, rec[0], rec[1], rec[2], u0, u1);
          v0 = geo.E[0] * u0 + geo.E[3] * u1; v1 = geo.E[1] * u0 + geo.E[4] * u1; v2 = geo.E[2] * u0 + geo.E[5] * u1;
          if (LOSS) { const double om = omega[slot]; v0 *= om; v1 *= om; v2 *= om; }
        }
        a0 += block_sum(v0, s4); a1 += block_sum(v1, s4); a2 += block_sum(v2, s4);
      }
      const double z0 = rec[3] - (rec[6] * a0 + rec[7] * a1 + rec[9] * a2);
      const double z1 = rec[4] - (rec[7] * a0 + rec[8] * a1 + rec[10] * a2);
      const double z2 = rec[5] - (rec[9] * a0 + rec[10] * a1 + rec[11] * a2);
      if (tid == 0) {
        yp[3 * p] = z0 * mf_rcp(sp[3 * p]); yp[3 * p + 1] = z1 * mf_rcp(sp[3 * p + 1]); yp[3 * p + 2] = z2 * mf_rcp(sp[3 * p + 2]);
        const double c0 = rec[0] - z0, c1 = rec[1] - z1, c2 = rec[2] - z2;
        Xc[3 * p] = c0; Xc[3 * p + 1] = c1; Xc[3 * p + 2] = c2;
        const double e0 = rec[0] - c0, e1 = rec[1] - c1, e2 = rec[2] - c2;
        dp2 += e0 * e0 + e1 * e1 + e2 * e2;
        xp2 += c0 * c0 + c1 * c1 + c2 * c2;
      }
      for (int ch = 0; ch < G.nchunk; ++ch) {
        const size_t chunk = (size_t)(G.chunk0 + ch);
        const size_t slot = chunk * kChunk + tid;
        if (idx[slot] < 0) continue;
        double cd[kCamLds];
        mf_stage_cam(cd, ocam[slot], ct, cm, sc, yc);
        MfGeo geo;
        double u0, u1;
        mf_geom(cd, rec[0], rec[1], rec[2], geo);
        mf_forward(geo, cd, rec[0], rec[1], rec[2], u0, u1);
        const double m0 = u0 + geo.E[0] * z0 + geo.E[1] * z1 + geo.E[2] * z2;
        const double m1 = u1 + geo.E[3] * z0 + geo.E[4] * z1 + geo.E[5] * z2;
        const double r0 = geo.f * geo.d * geo.px - oxy[chunk * (2 * kChunk) + tid];
        const double r1 = geo.f * geo.d * geo.py - oxy[chunk * (2 * kChunk) + kChunk + tid];
        mcc += (LOSS ? omega[slot] : 1.0) * (m0 * (r0 - 0.5 * m0) + m1 * (r1 - 0.5 * m1));
      }
    }
  }
  const double q0 = block_sum(mcc, s4);
  const double q1 = block_sum(dp2, s4);
  const double q2 = block_sum(xp2, s4);
  if (tid == 0) {
    partial[4 * (size_t)tile_id + 0] = q0;
    partial[4 * (size_t)tile_id + 1] = q1;
    partial[4 * (size_t)tile_id + 2] = q2;
    partial[4 * (size_t)tile_id + 3] = 0.0;
  }
}

// K1'-mf: cost 0.5 sum |r|^2 of a (candidate) state.  Same tile walk as the other matrix-free kernels: the tile's
// cameras (R, t, f, k1, k2) are staged in LDS, every lane gathers its point and reads its measurement.
// The first n_giant blocks take the giant tiles of tile_list, block n_giant + i takes tile i.
constexpr int kCamCost = 16;   // 15 doubles used: 8 x ds_read_b128

// |r|^2 of one observation of the camera record cd (R 0..8, t 9..11, f 12, k1 13, k2 14) at the point x
__device__ __forceinline__ double mf_cost_obs(const double* cd, const double x0, const double x1, const double x2, const double ox, const double oy) {
  const double P0 = cd[0] * x0 + cd[1] * x1 + cd[2] * x2 + cd[9];
  const double P1 = cd[3] * x0 + cd[4] * x1 + cd[5] * x2 + cd[10];
  const double P2 = cd[6] * x0 + cd[7] * x1 + cd[8] * x2 + cd[11];
  const double iz = mf_rcp(P2);
  const double px = -P0 * iz, py = -P1 * iz;
  const double n = px * px + py * py;
  const double d = 1.0 + n * (cd[13] + cd[14] * n);
  const double r0 = cd[12] * d * px - ox, r1 = cd[12] * d * py - oy;
  return r0 * r0 + r1 * r1;
}

__device__ __forceinline__ void mf_read_cam_cost(double* cd, const double* s_cam16, const int cs) {
  const double2* c2 = reinterpret_cast<const double2*>(s_cam16) + (kCamCost / 2) * cs;
#pragma unroll
  for (int k = 0; k < kCamCost / 2; ++k) { const double2 v = c2[k]; cd[2 * k] = v.x; cd[2 * k + 1] = v.y; }
}

// sum of |r|^2 (rho(|r|^2) with a loss) over this thread's observations of the non-giant tile t, cameras staged in s_cam16.
// FRESH: the points were written by this workgroup earlier in the same launch (k_mf_backsub_cm's point-major tiles): read past the
// compute unit's vector cache.
template <bool LOSS, bool FRESH>
__device__ __forceinline__ double mf_cost_tile(const Tile& t, const int32_t* __restrict__ idx, const int2* __restrict__ wl, const double* __restrict__ oxy,
                                               const double* s_cam16, const double* X, const Loss& loss) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int nwl = t.nslot / kWave;
  double cost = 0.0;
  for (int q = wv; q < nwl; q += 4) {
    const size_t slot = (size_t)t.slot0 + (size_t)q * kWave + lane;
    const int32_t w = idx[slot];
    if (w < 0) continue;
    const size_t p = (size_t)(wl[(t.slot0 >> 6) + q].x + (w & 0xFF));
    double x0, x1, x2;
    if (FRESH) {
      x0 = __hip_atomic_load(X + 3 * p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      x1 = __hip_atomic_load(X + 3 * p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      x2 = __hip_atomic_load(X + 3 * p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      x0 = X[3 * p]; x1 = X[3 * p + 1]; x2 = X[3 * p + 2];
    }
    const size_t chunk = slot / kChunk, l = slot % kChunk;
    const double ox = oxy[chunk * (2 * kChunk) + l], oy = oxy[chunk * (2 * kChunk) + kChunk + l];
    double cd[kCamCost];
    mf_read_cam_cost(cd, s_cam16, (w >> 8) & 0xFF);
    double rho = mf_cost_obs(cd, x0, x1, x2, ox, oy);
    if (LOSS) { double om; mf_loss(loss, rho, rho, om); }
    cost += rho;
  }
  return cost;
}

// partial_cost[tile * pstride + poff] = the tile's cost (pstride 1, poff 0: a vector of its own; 4, 3: the free column of the
// back-substitution's partial sums, when that kernel has taken the other tiles' candidate cost along)
template <bool LOSS>
__global__ __launch_bounds__(256) void k_mf_cost(int n_giant, const Tile* __restrict__ tiles, const int32_t* __restrict__ tile_list,
                                                 const Group* __restrict__ grps, const int32_t* __restrict__ idx,
                                                 const int32_t* __restrict__ ocam, const int2* __restrict__ wl,
                                                 const int32_t* __restrict__ tile_cams, const double* __restrict__ oxy,
                                                 const double* __restrict__ ct, const double* __restrict__ X,
                                                 double* __restrict__ partial_cost, const Loss loss, const int pstride, const int poff) {
  __shared__ __align__(16) double s_cam[kMaxCamSlots * kCamCost];
  __shared__ double s4[4];
  const bool giant = (int)blockIdx.x < n_giant;
  const int tile_id = giant ? tile_list[blockIdx.x] : (int)blockIdx.x - n_giant;
  const Tile t = tiles[tile_id];
  if (!giant && t.giant) return;
  const int tid = threadIdx.x;
  double cost = 0.0;
  if (!giant) {
    if (tid < t.ncs) {
      const double* T = ct + (size_t)kCamTabStride * tile_cams[t.cam0 + tid];
#pragma unroll
      for (int k = 0; k < 15; ++k) s_cam[kCamCost * tid + k] = T[k];
    }
    __syncthreads();
    cost = mf_cost_tile<LOSS, false>(t, idx, wl, oxy, s_cam, X, loss);
  } else {
    for (int gi = 0; gi < t.ngrp; ++gi) {
      const Group G = grps[t.grp0 + gi];
      const size_t p = (size_t)G.pt0;
      const double x0 = X[3 * p], x1 = X[3 * p + 1], x2 = X[3 * p + 2];
      for (int ch = 0; ch < G.nchunk; ++ch) {
        const size_t chunk = (size_t)(G.chunk0 + ch);
        const size_t slot = chunk * kChunk + tid;
        if (idx[slot] < 0) continue;
        const double* cd = ct + (size_t)kCamTabStride * ocam[slot];
        double rho = mf_cost_obs(cd, x0, x1, x2, oxy[chunk * (2 * kChunk) + tid], oxy[chunk * (2 * kChunk) + kChunk + tid]);
        if (LOSS) { double om; mf_loss(loss, rho, rho, om); }
        cost += rho;
      }
    }
  }
  const double tot = block_sum(cost, s4);
  if (tid == 0) partial_cost[(size_t)tile_id * pstride + poff] = 0.5 * tot;
}

// Jacobi scale from the raw (unit-scale) camera sums: sc = 1 / (1 + sqrt(diag F'F))
__global__ void k_scale_from_acc(int n_cam, const double* __restrict__ acc, double* __restrict__ sc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * n_cam) return;
  const int c = i / 9, k = i % 9;
  sc[i] = 1.0 / (1.0 + sqrt(acc[(size_t)kK2W * c + (k * 9 - (k * (k - 1)) / 2)]));
}

// Per camera: sum the raw slab rows in row order, then apply the camera-side transforms:
//   out = diag_w * Dc2 .* x + sc .* [ M' c ; N a ; i_f ; f i_k1 ; f i_k2 ]
__global__ __launch_bounds__(256) void k_cam_reduce_mf(int n_cam, int cams_per_block, const int32_t* __restrict__ cam_row_off,
                                                       const double* __restrict__ slab, const double* __restrict__ ct,
                                                       const double* __restrict__ cm, const double* __restrict__ sc,
                                                       const double* __restrict__ d2, const double* __restrict__ x,
                                                       double diag_w, double* __restrict__ out,
                                                       double* __restrict__ part_xy /*per block x.out, or null*/,
                                                       const Ctl* __restrict__ ctl) {
  __shared__ double s4[4];
  if (ctl->done) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c_end = min(n_cam, (int)(blockIdx.x + 1) * cams_per_block);
  const int sub = lane / 9, k = lane % 9;
  double xy = 0.0;
  for (int c = blockIdx.x * cams_per_block + wv; c < c_end; c += 4) {
    const int r0 = cam_row_off[c], r1 = cam_row_off[c + 1];
    double acc = 0.0;
    if (lane < 63)
      for (int r = r0 + sub; r < r1; r += 7) acc += slab[9 * (size_t)r + k];
    double tot = acc;
#pragma unroll
    for (int s = 1; s < 7; ++s) tot += __shfl(acc, k + 9 * s, 64);
    // broadcast the 9 totals (held by lanes 0..8) to every lane
    double v[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) v[j] = __shfl(tot, j, 64);
    if (lane < 9) {
      const double* T = ct + (size_t)kCamTabStride * c;
      const double* M = cm + 10 * (size_t)c;
      const bool small = M[9] == 1.0;
      double r;
      if (M[9] == 2.0) {
        // the reference's functor: rotation part M~' (t x a - c), translation part -a, constant intrinsics; t = -R~' t~
        if (lane < 3) {
          const double t0 = -(T[0] * T[9] + T[3] * T[10] + T[6] * T[11]);
          const double t1 = -(T[1] * T[9] + T[4] * T[10] + T[7] * T[11]);
          const double t2 = -(T[2] * T[9] + T[5] * T[10] + T[8] * T[11]);
          const double u0 = t1 * v[5] - t2 * v[4] - v[0], u1 = t2 * v[3] - t0 * v[5] - v[1], u2 = t0 * v[4] - t1 * v[3] - v[2];
          r = M[lane] * u0 + M[3 + lane] * u1 + M[6 + lane] * u2;
        } else if (lane < 6) r = -v[lane];
        else r = 0.0;
      } else if (lane < 3) r = M[lane] * v[0] + M[3 + lane] * v[1] + M[6 + lane] * v[2];                // (M' c)_lane
      else if (lane < 6) {
        const int i = lane - 3;
        r = small ? v[3 + i] : (T[3 * i] * v[3] + T[3 * i + 1] * v[4] + T[3 * i + 2] * v[5]);           // (N a)_i
      } else if (lane == 6) r = v[6];
      else r = T[12] * v[lane];
      r *= sc[9 * (size_t)c + lane];
      const double xv = x[9 * (size_t)c + lane];
      if (d2) r += diag_w * d2[9 * (size_t)c + lane] * xv;
      out[9 * (size_t)c + lane] = r;
      xy += xv * r;
    }
  }
  if (part_xy) {
    const double s = block_sum(xy, s4);
    if (threadIdx.x == 0) part_xy[blockIdx.x] = s;
  }
}

}  // namespace miba
