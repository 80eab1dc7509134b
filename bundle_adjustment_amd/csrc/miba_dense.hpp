// miba_dense.hpp — exact solve of the reduced camera system for problems with few cameras
// (what Ceres does for DENSE_SCHUR / SPARSE_SCHUR / *_NORMAL_CHOLESKY, the solver
// type the reference app asks for at app/main.cpp:19, reaches through src/BundleAdjustment.cpp:20,109).
//
//   S = B - sum_p G_p' W_p G_p,   G = E' F_s (3 x 9 per observation),  W_p = sp C_p^-1 sp (point record),
//   B = F_s'F_s + D_c^2 per camera (the block diagonal K2 has already reduced),
// the same operator the matrix-free product applies, assembled densely (row-major, N = 9 n_cam rounded
// up to 64, lower triangle), factored by a blocked right-looking Cholesky (diagonal blocks in the registers of
// one wave, trailing update on the fp64 matrix cores, v_mfma_f64_16x16x4_f64) that carries the forward
// substitution along, then a blocked backward substitution.
#pragma once

#include "miba_mf.hpp"

namespace miba {

constexpr int kNB = 64;             // Cholesky block
constexpr int kDenseMaxN = 18432;   // 2048 cameras: S is 2.7 GB

typedef double d4_t __attribute__((ext_vector_type(4)));

// Where the assembly kernels put entry (a, b) of camera block (ci, cj) of S: a dense row-major matrix (N x N, map == nullptr), or
// the 64 x 64 tiles of the tile-sparse factorisation (miba_sparse.hpp).  There a tile row holds kTileCams = 7 consecutive cameras
// (63 rows and one padding row), the tile rows are permuted by the fill-reducing ordering (pos[group] = tile row), and
// map[tile row * nT + tile column] is the tile's slot in S (lower block triangle only: a block whose tile row comes before its
// tile column is written transposed; entries above the diagonal of a diagonal tile are dropped, the factorisation never reads them).
constexpr int kTileCams = 7;
struct SDst {
  double* S;
  int N;
  const int32_t* map;
  const int32_t* pos;
  int nT;
};
// The camera block (ci, cj) is located once (the tile lookups are two dependent trips to memory), its entries by arithmetic.
struct SBlock {
  double* p;    // kind 0: entry (a, b) at p[a * ld + b]; kind 1: transposed, p[b * ld + a]; kind 2: p is a diagonal tile
  int ld, kind;
  int lr0, lc0; // kind 2: the block's first row and column inside the tile
};
__device__ __forceinline__ SBlock sdst_block(const SDst D, int ci, int cj) {
  SBlock B;
  if (!D.map) {
    B.p = D.S + (size_t)(9 * ci) * D.N + 9 * cj; B.ld = D.N; B.kind = 0; B.lr0 = 0; B.lc0 = 0;
    return B;
  }
  const int gi = ci / kTileCams, gj = cj / kTileCams;
  const int tr = D.pos[gi], tc = D.pos[gj];
  B.lr0 = 9 * (ci - kTileCams * gi); B.lc0 = 9 * (cj - kTileCams * gj);
  B.ld = 64;
  if (tr > tc) { B.kind = 0; B.p = D.S + (size_t)D.map[(size_t)tr * D.nT + tc] * 4096 + (size_t)B.lr0 * 64 + B.lc0; }
  else if (tr < tc) { B.kind = 1; B.p = D.S + (size_t)D.map[(size_t)tc * D.nT + tr] * 4096 + (size_t)B.lc0 * 64 + B.lr0; }
  else { B.kind = 2; B.p = D.S + (size_t)D.map[(size_t)tr * D.nT + tr] * 4096; }
  return B;
}
// same_cam: the block is a camera's own diagonal block, whose entries above the diagonal are dropped
__device__ __forceinline__ double* sdst_entry(const SBlock& B, bool same_cam, int a, int b) {
  if (B.kind == 0) return B.p + (size_t)a * B.ld + b;
  if (B.kind == 1) return B.p + (size_t)b * B.ld + a;
  const int lr = B.lr0 + a, lc = B.lc0 + b;
  if (lr >= lc) return B.p + lr * 64 + lc;
  return same_cam ? nullptr : B.p + lc * 64 + lr;
}

// G = E' F_s (3x9, row a = point coordinate) of one observation and T = W G
__device__ __forceinline__ void dense_obs_G(const double* cd, const double* s /*[9] camera scale*/, double x0, double x1, double x2,
                                            double* G) {
  double F[18], E[6], r0, r1;
  mf_blocks(cd, x0, x1, x2, 0.0, 0.0, F, E, r0, r1);
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const double f0 = F[k] * s[k], f1 = F[9 + k] * s[k];
#pragma unroll
    for (int a = 0; a < 3; ++a) G[9 * a + k] = E[a] * f0 + E[3 + a] * f1;
  }
}

__device__ __forceinline__ void dense_WG(const double* W /*00 10 11 20 21 22*/, const double* G, double* T) {
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    T[k] = W[0] * G[k] + W[1] * G[9 + k] + W[3] * G[18 + k];
    T[9 + k] = W[1] * G[k] + W[2] * G[9 + k] + W[4] * G[18 + k];
    T[18 + k] = W[3] * G[k] + W[4] * G[9 + k] + W[5] * G[18 + k];
  }
}

// S[cam_i block row][cam_j block column] -= T_i' G_j
__device__ __forceinline__ void dense_sub_block(const SDst D, int cam_i, int cam_j, const double* T, const double* Gj) {
  const SBlock B = sdst_block(D, cam_i, cam_j);
#pragma unroll
  for (int a = 0; a < 9; ++a)
#pragma unroll
    for (int b = 0; b < 9; ++b) {
      const double v = -(T[a] * Gj[b] + T[9 + a] * Gj[9 + b] + T[18 + a] * Gj[18 + b]);
      double* dst = sdst_entry(B, cam_i == cam_j, a, b);
      if (dst) atomicAdd(dst, v);
    }
}

// K5: point side of the explicit Schur complement.  Wavelet layout of K3: a wave owns whole points; lane i
// visits the other observations of its point by rotating through the run (lane shuffles) and handles the
// pairs whose camera index is not larger than its own, so only the lower block triangle is written.
// LOSS: G of every observation carries its loss weight rho' (the per-slot weights K2 has left, miba_mf.hpp); this
// slot-walking kernel is the assembly used with a loss, because the planned instance lists name no slots.
template <bool LOSS>
__global__ __launch_bounds__(256) void k_dense_pairs(const Tile* __restrict__ tiles, const int32_t* __restrict__ idx,
                                                     const int2* __restrict__ wl, const int32_t* __restrict__ tile_cams,
                                                     const double* __restrict__ ct, const double* __restrict__ cm,
                                                     const double* __restrict__ sc, const double* __restrict__ ptrec,
                                                     const double* __restrict__ omega, const SDst D) {
  __shared__ __align__(16) double s_cam[kMaxCamSlots * kCamLds];
  __shared__ double s_sc[kMaxCamSlots * 9];
  __shared__ int s_id[kMaxCamSlots];
  const Tile t = tiles[blockIdx.x];
  if (t.giant) return;
  const int tid = threadIdx.x;
  if (tid < t.ncs) {
    const int cam = tile_cams[t.cam0 + tid];
    mf_stage_cam_init(s_cam + kCamLds * tid, cam, ct, cm);
    s_id[tid] = cam;
#pragma unroll
    for (int k = 0; k < 9; ++k) s_sc[9 * tid + k] = sc[9 * (size_t)cam + k];
  }
  __syncthreads();
  const int lane = tid & 63, wv = tid >> 6;
  const int nwl = t.nslot / kWave;
  for (int q = wv; q < nwl; q += 4) {
    const size_t slot = (size_t)t.slot0 + (size_t)q * kWave + lane;
    const int32_t w = idx[slot];
    const int2 wq = wl[(t.slot0 >> 6) + q];
    const int maxrun = wq.y;
    const bool valid = w >= 0;
    const int cs = valid ? ((w >> 8) & 0xFF) : 0, pl = valid ? (w & 0xFF) : 0;
    const int pos = valid ? ((w >> 16) & 0x7F) : 0, cnt = valid ? (w >> 23) : 1;
    const double* rp = ptrec + (size_t)kPtRec * (size_t)(wq.x + pl);
    double G[27], T[27];
    double cd[kCamLds];
    mf_read_cam(cd, s_cam, cs);
    dense_obs_G(cd, s_sc + 9 * cs, rp[0], rp[1], rp[2], G);
    if (!valid) {
#pragma unroll
      for (int m = 0; m < 27; ++m) G[m] = 0.0;
    } else if (LOSS) {
      const double om = omega[slot];
#pragma unroll
      for (int m = 0; m < 27; ++m) G[m] *= om;
    }
    dense_WG(rp + 6, G, T);
    const int cam = valid ? s_id[cs] : -1;
    for (int d = 0; d < maxrun; ++d) {
      int pj = pos + d;
      if (pj >= cnt) pj -= cnt;
      const int src = (lane - pos + pj) << 2;
      double Gj[27];
#pragma unroll
      for (int m = 0; m < 27; ++m) Gj[m] = lane_read(G[m], src);
      const int camj = __builtin_amdgcn_ds_bpermute(src, cam);
      if (valid && d < cnt && cam >= camj) dense_sub_block(D, cam, camj, T, Gj);
    }
  }
}

// giant points (more than 64 observations): the point's chunks against each other through LDS
template <bool LOSS>
__global__ __launch_bounds__(256) void k_dense_pairs_giant(const Tile* __restrict__ tiles, const int32_t* __restrict__ tile_list,
                                                           const Group* __restrict__ grps, const int32_t* __restrict__ idx,
                                                           const int32_t* __restrict__ ocam, const double* __restrict__ ct,
                                                           const double* __restrict__ cm, const double* __restrict__ sc,
                                                           const double* __restrict__ ptrec, const double* __restrict__ omega,
                                                           const SDst D) {
  __shared__ double s_G[kChunk * 27];
  __shared__ int s_camj[kChunk];
  const Tile t = tiles[tile_list[blockIdx.x]];
  const int tid = threadIdx.x;
  for (int g = 0; g < t.ngrp; ++g) {
    const Group Gr = grps[t.grp0 + g];
    const double* rp = ptrec + (size_t)kPtRec * (size_t)Gr.pt0;
    const double x0 = rp[0], x1 = rp[1], x2 = rp[2];
    for (int ci = 0; ci < Gr.nchunk; ++ci) {
      const size_t slot_i = (size_t)(Gr.chunk0 + ci) * kChunk + tid;
      const bool valid_i = idx[slot_i] >= 0;
      const int cam_i = valid_i ? ocam[slot_i] : -1;
      double T[27];
      {
        double Gi[27];
#pragma unroll
        for (int m = 0; m < 27; ++m) Gi[m] = 0.0;
        if (valid_i) {
          double cd[kCamLds];
          mf_stage_cam_init(cd, cam_i, ct, cm);
          dense_obs_G(cd, sc + 9 * (size_t)cam_i, x0, x1, x2, Gi);
          if (LOSS) {
            const double om = omega[slot_i];
#pragma unroll
            for (int m = 0; m < 27; ++m) Gi[m] *= om;
          }
        }
        dense_WG(rp + 6, Gi, T);
      }
      for (int cj = 0; cj < Gr.nchunk; ++cj) {
        __syncthreads();   // the previous round's readers are done with s_G
        {
          const size_t slot_j = (size_t)(Gr.chunk0 + cj) * kChunk + tid;
          const bool valid_j = idx[slot_j] >= 0;
          const int cam_j = valid_j ? ocam[slot_j] : -1;
          double Gj[27];
#pragma unroll
          for (int m = 0; m < 27; ++m) Gj[m] = 0.0;
          if (valid_j) {
            double cd[kCamLds];
            mf_stage_cam_init(cd, cam_j, ct, cm);
            dense_obs_G(cd, sc + 9 * (size_t)cam_j, x0, x1, x2, Gj);
            if (LOSS) {
              const double om = omega[slot_j];
#pragma unroll
              for (int m = 0; m < 27; ++m) Gj[m] *= om;
            }
          }
#pragma unroll
          for (int m = 0; m < 27; ++m) s_G[27 * tid + m] = Gj[m];
          s_camj[tid] = cam_j;
        }
        __syncthreads();
        if (valid_i) {
          for (int j = 0; j < kChunk; ++j) {
            const int camj = s_camj[j];
            if (camj < 0 || cam_i < camj) continue;
            dense_sub_block(D, cam_i, camj, T, s_G + 27 * j);
          }
        }
      }
    }
  }
}

// K5, planned form (the default when the host could list the pair instances): the host sorts every
// (camera i >= camera j, shared point) instance by block, one wave sums a segment of at most kDenseSeg
// instances of ONE block with the two camera records in scalar registers and 81 accumulators per lane,
// then a cross-lane reduction through LDS.  No atomics unless a block was cut into several segments
// (seg.w = 1), so S is reproducible bit for bit for all but the very largest blocks.
constexpr int kDenseSeg = 2048;
// rows [A0, A1) of the 9 x 9 block of one segment, by one wave.  81 accumulators per lane, and the 81 values they are updated
// from, leave room for one wave per SIMD only, and a single wave issues an fp64 instruction every 8 cycles; two waves that
// take half of the block's rows each (45 + 36 accumulators; the geometry of both observations computed by both, 1.3 times the
// instructions) run at two waves per SIMD, twice the issue rate.
template <int A0, int A1>
__device__ __forceinline__ void dense_block_rows(const int lane, double* sr, const int ci, const int cj, const int first, const int n,
                                                 const int use_atomic, const int32_t* __restrict__ pts, const double* __restrict__ ct,
                                                 const double* __restrict__ cm, const double* __restrict__ sc,
                                                 const double* __restrict__ ptrec, const SBlock B) {
  constexpr int NA = A1 - A0;
  double cdi[kCamLds], cdj[kCamLds], sci[9], scj[9];
  mf_stage_cam_init(cdi, ci, ct, cm);
  mf_stage_cam_init(cdj, cj, ct, cm);
#pragma unroll
  for (int k = 0; k < 9; ++k) { sci[k] = sc[9 * (size_t)ci + k]; scj[k] = sc[9 * (size_t)cj + k]; }
  double acc[NA * 9];
#pragma unroll
  for (int k = 0; k < NA * 9; ++k) acc[k] = 0.0;
  // the point id of the next pass is in flight while a pass computes (its record is not: the registers for it would cost the
  // second wave per SIMD, which hides that trip better)
  const int32_t* ptp = pts + first;
  int32_t pt_c = ptp[min(lane, n - 1)];
  for (int i = lane; i < n; i += 64) {
    const int32_t pt_n = ptp[min(i + 64, n - 1)];
    double rec_c[9];
    {
      const double* rp = ptrec + (size_t)kPtRec * (size_t)pt_c;
#pragma unroll
      for (int k = 0; k < 3; ++k) rec_c[k] = rp[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) rec_c[3 + k] = rp[6 + k];
    }
    // G_i' W G_j = F_i' (E_i W E_j') F_j: the 2 x 2 matrix M = E_i W E_j' first, then N = M F_j (2 x 9) and F_i' N, 2 multiply-adds
    // per entry of the block where T_i' G_j (T = W G_i, G = E'F, 3 x 9) takes 3, and neither G is formed (a fifth fewer instructions).
    double Fi[18], Ei[6], Fj[18], Ej[6], r0, r1;
    mf_blocks(cdi, rec_c[0], rec_c[1], rec_c[2], 0.0, 0.0, Fi, Ei, r0, r1);   // (only the columns A0 .. A1 - 1 of F_i are used below)
    if (ci != cj) mf_blocks(cdj, rec_c[0], rec_c[1], rec_c[2], 0.0, 0.0, Fj, Ej, r0, r1);
    else {
#pragma unroll
      for (int m = 0; m < 18; ++m) Fj[m] = Fi[m];
#pragma unroll
      for (int m = 0; m < 6; ++m) Ej[m] = Ei[m];
    }
    const double* W = rec_c + 3;   // 00 10 11 20 21 22
    double V[6], M[4], N[18];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      V[3 * r + 0] = Ei[3 * r] * W[0] + Ei[3 * r + 1] * W[1] + Ei[3 * r + 2] * W[3];
      V[3 * r + 1] = Ei[3 * r] * W[1] + Ei[3 * r + 1] * W[2] + Ei[3 * r + 2] * W[4];
      V[3 * r + 2] = Ei[3 * r] * W[3] + Ei[3 * r + 1] * W[4] + Ei[3 * r + 2] * W[5];
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int q = 0; q < 2; ++q) M[2 * r + q] = V[3 * r] * Ej[3 * q] + V[3 * r + 1] * Ej[3 * q + 1] + V[3 * r + 2] * Ej[3 * q + 2];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const double f0 = Fj[k] * scj[k], f1 = Fj[9 + k] * scj[k];
      N[k] = M[0] * f0 + M[1] * f1;
      N[9 + k] = M[2] * f0 + M[3] * f1;
    }
#pragma unroll
    for (int a = A0; a < A1; ++a) {
      const double g0 = Fi[a] * sci[a], g1 = Fi[9 + a] * sci[a];
#pragma unroll
      for (int b = 0; b < 9; ++b) acc[9 * (a - A0) + b] = fma(g0, N[b], fma(g1, N[9 + b], acc[9 * (a - A0) + b]));
    }
    pt_c = pt_n;
  }
  const int sub = lane / 9, k9 = lane % 9;
#pragma unroll
  for (int r = A0; r < A1; ++r) {   // block row r
#pragma unroll
    for (int k = 0; k < 9; ++k) sr[lane * 9 + k] = acc[9 * (r - A0) + k];
    __builtin_amdgcn_wave_barrier();
    double a = 0.0;
    if (lane < 63)
      for (int l = sub; l < 64; l += 7) a += sr[l * 9 + k9];
    double tot = a;
#pragma unroll
    for (int s2 = 1; s2 < 7; ++s2) tot += __shfl(a, k9 + 9 * s2, 64);
    if (lane < 9) {
      double* dst = sdst_entry(B, ci == cj, r, lane);
      if (dst) {
        if (use_atomic) atomicAdd(dst, -tot);
        else *dst = -tot;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_dense_blocks(int n_seg, const int4* __restrict__ seg /*{ci, first, count, cj | atomic << 30}*/,
                                                      const int32_t* __restrict__ pts, const double* __restrict__ ct,
                                                      const double* __restrict__ cm, const double* __restrict__ sc,
                                                      const double* __restrict__ ptrec, const SDst D, const int xcd_remap) {
  __shared__ double s_red[2][64 * 9];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // (XCD x takes the x-th eighth of the block-ordered segments: neighbouring blocks gather the same point records, see k_mf_camrows)
  const int sg = xcd_remap ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  // one segment per workgroup, two waves, half of the block's rows each: a workgroup leaves when its segment is done (with two
  // segments per workgroup the waves of the shorter one held their slots until the longer one had finished: 1.5 of 2 waves resident)
  if (sg >= n_seg) return;
  const int4 Sg = seg[sg];
  const int ci = __builtin_amdgcn_readfirstlane(Sg.x);
  const int first = __builtin_amdgcn_readfirstlane(Sg.y), n = __builtin_amdgcn_readfirstlane(Sg.z);
  const int w4 = __builtin_amdgcn_readfirstlane(Sg.w);
  const int use_atomic = w4 >> 30, cj = w4 & 0x3FFFFFFF;
  const SBlock B = sdst_block(D, ci, cj);   // (wave-uniform, requested here: the lookups are back long before the stores)
  if (wv == 0) dense_block_rows<0, 5>(lane, s_red[wv], ci, cj, first, n, use_atomic, pts, ct, cm, sc, ptrec, B);
  else dense_block_rows<5, 9>(lane, s_red[wv], ci, cj, first, n, use_atomic, pts, ct, cm, sc, ptrec, B);
}

// SCHUR_JACOBI preconditioner (ceres::SCHUR_JACOBI): the block diagonal of S itself needs, per camera,
// sum over its observations of G' W G (G = E'F unscaled here, the column scale is applied by k_cam_reduce64
// like for F'F).  Camera-major pass shaped like k_mf_camrows: one wave per segment of one camera, camera record
// in scalar registers, the 45 upper-triangle sums per lane, one LDS cross-lane reduction.  Row sg of slab.
// LOSS: G carries the observation's loss weight rho' (E and F each carry its square root), recomputed from the residual.
template <bool LOSS>
__global__ __launch_bounds__(256) void k_mf_schurdiag(int n_seg, const int4* __restrict__ seg /*{cam, first, count, -}*/,
                                                      const int32_t* __restrict__ cmo_pt, const double2* __restrict__ cmo_xy,
                                                      const double* __restrict__ ct,
                                                      const double* __restrict__ cm, const double* __restrict__ ptrec, const Loss loss,
                                                      double* __restrict__ slab /*[n_seg][64]*/) {
  __shared__ double s_red[4][64 * 9];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int sg = blockIdx.x * 4 + wv;
  if (sg >= n_seg) return;
  const int4 Sg = seg[sg];
  const int cam = __builtin_amdgcn_readfirstlane(Sg.x);
  const int first = __builtin_amdgcn_readfirstlane(Sg.y), n = __builtin_amdgcn_readfirstlane(Sg.z);
  double cd[kCamLds], one[9];
  mf_stage_cam_init(cd, cam, ct, cm);
#pragma unroll
  for (int k = 0; k < 9; ++k) one[k] = 1.0;
  double acc[45];
#pragma unroll
  for (int k = 0; k < 45; ++k) acc[k] = 0.0;
  const int32_t* ptp = cmo_pt + first;
  int32_t pt_c = ptp[min(lane, n - 1)];
  for (int i = lane; i < n; i += 64) {
    const int32_t pt_n = ptp[min(i + 64, n - 1)];
    const double* rp = ptrec + (size_t)kPtRec * (size_t)pt_c;
    double G[27], T[27];
    dense_obs_G(cd, one, rp[0], rp[1], rp[2], G);
    if (LOSS) {
      const double2 xy = cmo_xy[first + i];
      double E[6], r0, r1, rho, om;
      mf_resid_E(cd, rp[0], rp[1], rp[2], xy.x, xy.y, E, r0, r1);
      mf_loss(loss, r0 * r0 + r1 * r1, rho, om);
#pragma unroll
      for (int m = 0; m < 27; ++m) G[m] *= om;
    }
    dense_WG(rp + 6, G, T);
    int o = 0;
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
      for (int b = a; b < 9; ++b) {
        acc[o] = fma(G[a], T[b], fma(G[9 + a], T[9 + b], fma(G[18 + a], T[18 + b], acc[o])));
        ++o;
      }
    pt_c = pt_n;
  }
  double* sr = s_red[wv];
  const int sub = lane / 9, k9 = lane % 9;
  double* row = slab + (size_t)kK2W * sg;
#pragma unroll
  for (int r = 0; r < 5; ++r) {
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
  if (lane < 19) row[45 + lane] = 0.0;
}

// S += block diagonal B = F_s'F_s + D_c^2 (acc: upper triangle of F_s'F_s per camera), unit diagonal on the padding
__global__ void k_dense_diag(int n_cam, const double* __restrict__ acc /*[n][64]*/, const double* __restrict__ Dc2, const SDst D) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 81 * (size_t)n_cam) {
    const int c = (int)(i / 81), a = (int)(i % 81) / 9, b = (int)(i % 9);
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    double v = acc[(size_t)kK2W * c + (9 * lo - (lo * (lo - 1)) / 2 + (hi - lo))];
    if (a == b) v += Dc2[9 * (size_t)c + a];
    const SBlock B = sdst_block(D, c, c);
    double* dst = sdst_entry(B, true, a, b);
    if (dst) *dst += v;
  }
  const size_t r = 9 * (size_t)n_cam + i;
  if (!D.map && r < (size_t)D.N) D.S[r * D.N + r] = 1.0;   // (the tiles set their own padding: k_tchol_pad)
}

// ---- blocked Cholesky, lower, right-looking --------------------------------------------------
// The matrix is stored with 64 extra rows: row N holds the right-hand side, so the panel solve and the
// trailing update carry the forward substitution L y = b along (row N of the factor is y').  Per 64-column
// step two launches: k_chol_panel (every workgroup factors the 64 x 64 diagonal block itself, in the
// registers of one wave, then solves its rows of the panel against it) and k_chol_update (fp64 MFMA).
// The diagonal blocks of S stay as they were; their factors go to Ldiag (per block 64 x 64 + 64 reciprocal
// diagonal entries) for the backward substitution.
constexpr int kLdiag = kNB * kNB + kNB;
constexpr int kLs = kNB + 2;   // LDS row stride of the diagonal factor

__device__ __forceinline__ double lane_bcast(double v, int src_lane) {   // src_lane: wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane), __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}

// wave 0: Cholesky factor of the diagonal block at k0, lane = row, the row in registers.  Column j of the
// factor goes to LDS as soon as it exists (Lt[j][r] = L[r][j]) and is read back as wave-uniform operands
// for the rank-1 update (LDS operations of one wave execute in order: no barrier).  Lr = 1 / diagonal.
// Returns true if a pivot was not positive.
__device__ __forceinline__ bool chol_diag_wave(const double* __restrict__ diag /*top-left entry of the block*/, size_t stride, int lane,
                                               double* Lt, double* Lr) {
  double a[kNB];
  const double* row = diag + (size_t)lane * stride;
#pragma unroll
  for (int c = 0; c < kNB; ++c) a[c] = row[c];
#pragma unroll
  for (int c = 0; c < kNB; ++c) a[c] = (c <= lane) ? a[c] : 0.0;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < kNB; ++j) {
    double d = lane_bcast(a[j], j);
    if (!(d > 0.0)) { bad = true; d = 1.0; }
    const double rd = mf_rsqrt(d);
    const double l = (lane >= j) ? a[j] * rd : 0.0;
    Lt[j * kLs + lane] = l;
    if (lane == 0) Lr[j] = rd;
    // all operands of the step are requested before the first FMA (the reads pipeline in the LDS; taken one by one, each
    // costs its round trip)
    double t[kNB];
#pragma unroll
    for (int c = j + 1; c < kNB; ++c) t[c] = Lt[j * kLs + c];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = j + 1; c < kNB; ++c) a[c] = fma(-l, t[c], a[c]);
    // pin the row after every step: left alone, the compiler turns the unrolled nest into one deferred dot product
    // per column, keeps every earlier column live for it and spills
#pragma unroll
    for (int c = j + 1; c < kNB; ++c) asm volatile("" : "+v"(a[c]));
  }
  return bad;
}

// One wave per workgroup: 64 rows of the panel each.  (Four waves sharing one factor in LDS were slower: the
// wave-uniform reads cost full LDS return bandwidth per wave, and one wave per CU has the LDS pipe to itself.)
__global__ __launch_bounds__(64) void k_chol_panel(double* __restrict__ S, int N, int NR, int k0, double* __restrict__ Ldiag /*[kLdiag]*/,
                                                   int* __restrict__ flag) {
  __shared__ __align__(16) double Lt[kNB * kLs];   // Lt[k][c] = L[c][k]
  __shared__ double Lr[kNB];
  const int lane = threadIdx.x;
  {
    const bool bad = chol_diag_wave(S + (size_t)k0 * N + k0, (size_t)N, lane, Lt, Lr);
    if (bad && blockIdx.x == 0 && lane == 0) *flag = 1;
  }
  if (blockIdx.x == 0) {
    for (int i = lane; i < kNB * kNB; i += kNB) Ldiag[i] = Lt[(i & 63) * kLs + (i >> 6)];
    Ldiag[kNB * kNB + lane] = Lr[lane];
  }
  // my row of the panel: x L11' = a, right-looking forward substitution (each solved entry updates all the later
  // ones: independent FMAs), the factor read as wave-uniform LDS operands
  const int r = k0 + kNB + (int)blockIdx.x * kNB + lane;
  if (r >= NR) return;
  double* row = S + (size_t)r * N + k0;
  double x[kNB];
#pragma unroll
  for (int c = 0; c < kNB; ++c) x[c] = row[c];
#pragma unroll
  for (int k = 0; k < kNB; ++k) {
    const double xk = x[k] * Lr[k];
    x[k] = xk;
    double t[kNB];   // (operands requested together, as in chol_diag_wave)
#pragma unroll
    for (int c = k + 1; c < kNB; ++c) t[c] = Lt[k * kLs + c];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = k + 1; c < kNB; ++c) x[c] = fma(-xk, t[c], x[c]);
#pragma unroll
    for (int c = k + 1; c < kNB; ++c) asm volatile("" : "+v"(x[c]));
  }
#pragma unroll
  for (int c = 0; c < kNB; ++c) row[c] = x[c];
}

// trailing update A22 -= L21 L21' on the lower block triangle: workgroup = 64 x 64 tile (I >= J), wave = 16 rows,
// on the fp64 matrix cores (A fragment: lane l holds A[l & 15][4 kk + (l >> 4)]; B[k][n] = L21[col tile row n][k];
// D: row (l >> 4) + 4 reg, col l & 15).  Row tiles run to NR (the right-hand-side row).
__global__ __launch_bounds__(256) void k_chol_update(double* __restrict__ S, int N, int k0) {
  __shared__ double bt[kNB][kNB + 1];
  const int I = blockIdx.y, J = blockIdx.x;
  if (J > I) return;
  const int tid = threadIdx.x;
  const int base = k0 + kNB;
  const int cj = base + kNB * J;
  const int l = tid & 63, wv = tid >> 6;
  const int ar = l & 15, ak = l >> 4;
  const int ri = base + kNB * I + 16 * wv;
  // both operands requested before the first wait (as a loop "load, store to LDS" the staging took sixteen trips to memory)
  double bv[16], af[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) bv[q] = S[(size_t)(cj + 4 * q + wv) * N + k0 + l];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) af[kk] = S[(size_t)(ri + ar) * N + k0 + 4 * kk + ak];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < 16; ++q) bt[4 * q + wv][l] = bv[q];
  __syncthreads();
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    d4_t c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) c = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], bt[16 * nt + ar][4 * kk + ak], c, 0, 0, 0);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) S[(size_t)(ri + ak + 4 * reg) * N + cj + 16 * nt + ar] -= c[reg];
  }
}

// backward step k of L' x = y (y = row N of the factored matrix, updated in place): x_k = L_kk^-T y_k by one wave
// (lane = unknown, broadcasts of the solved entries), then y_j -= (L_kj)' x_k for the columns to the left
__global__ __launch_bounds__(256) void k_chol_bwd(double* __restrict__ S, int N, int k0, const double* __restrict__ Ldiag,
                                                  double* __restrict__ x, int n_out) {
  __shared__ double Ls[kNB * kLs];
  __shared__ double xk[kNB];
  const int tid = threadIdx.x;
  double* y = S + (size_t)N * N;
  {
    double lv[16];   // (requested together: as a loop "load, store to LDS" this took sixteen trips to memory)
#pragma unroll
    for (int q = 0; q < 16; ++q) lv[q] = Ldiag[tid + 256 * q];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 16; ++q) Ls[(4 * q + (tid >> 6)) * kLs + (tid & 63)] = lv[q];
  }
  __syncthreads();
  if (tid < kNB) {
    double yv = y[k0 + tid];
    const double rdiag = Ldiag[kNB * kNB + tid];
    double xv = 0.0;
    for (int k = kNB - 1; k >= 0; --k) {
      const double xs = lane_bcast(yv * rdiag, k);          // x_k = y_k / L_kk, from lane k
      if (tid == k) xv = xs;
      if (tid < k) yv = fma(-Ls[k * kLs + tid], xs, yv);     // y_i -= L[k][i] x_k
    }
    xk[tid] = xv;
    if (blockIdx.x == 0 && k0 + tid < n_out) x[k0 + tid] = xv;
  }
  __syncthreads();
  const int c = (int)blockIdx.x * 256 + tid;
  if (c < k0) {
    double sacc = 0.0;
#pragma unroll 8
    for (int r = 0; r < kNB; ++r) sacc = fma(S[(size_t)(k0 + r) * N + c], xk[r], sacc);
    y[c] -= sacc;
  }
}

// row N of the matrix = right-hand side (zero on the padding)
__global__ void k_dense_rhs(int n, int N, const double* __restrict__ v, double* __restrict__ S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) S[(size_t)N * N + i] = (i < n) ? v[i] : 0.0;
}

// padded image of a caller's n x n matrix: [(N + 64) x N], identity on the padding diagonal, zero rows below
__global__ void k_dense_pad(int n, int N, const double* __restrict__ A, double* __restrict__ S) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)(N + kNB) * N) return;
  const int r = (int)(i / N), c = (int)(i % N);
  double v = 0.0;
  if (r < n && c < n) v = A[(size_t)r * n + c];
  else if (r < N && r == c) v = 1.0;
  S[i] = v;
}

// status word of the factorisation as the ABI reports it: 0 ok, 2 a pivot was not positive
__global__ void k_dense_status(const int32_t* __restrict__ flag, int32_t* __restrict__ status) { *status = *flag ? 2 : 0; }

}  // namespace miba
