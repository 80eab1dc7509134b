// miba_mf2.hpp — the implicit-Schur product (K3) over camera-major sub-tiles.
//
// Same operator and the same per-observation algebra as k_mf_matvec (miba_mf.hpp: ImplicitSchurComplement::RightMultiply,
// SURVEY App. B.3), different ownership.  The point-major wavelets of miba_mf.hpp make the point side cheap (lane scans) and
// pay for the camera side through LDS: 13 ds_read_b128 of the camera record and 9 ds_add_f64 per observation, which is what
// bounds that kernel (LDS array 81 % busy, 272 LDS cycles per 64 observations, about 150 of them the atomic adds).  Here a
// lane owns a SEGMENT: kCmSeg observations of one camera (layout: build_cm_layout, miba_plan.hpp).  Per sub-tile
//   1  the lane reads its camera record once (12 ds_read_b128 per kCmSeg observations) and, per observation, the point from
//      LDS, rebuilds the Jacobian blocks, forms u = F x and adds E'u to its point's three sums in LDS (3 ds_add_f64);
//      E, u and p stay in registers
//   -- workgroup barrier --
//   1.5  one thread per point: y = -W a (W from the point record), in place
//   -- workgroup barrier --
//   2  per observation w = u + E y and the nine raw camera sums F'w, accumulated in REGISTERS over the segment; one set of
//      nine ds_add_f64 per segment adds them to the tile's camera accumulators.
// LDS instructions per observation: about 4 + 3 + 3 + 9/4 instead of 36, no lane scans, no run broadcast, and the point
// records are read as whole consecutive points (no 96-byte gathers).  Tiles with a camera in Ceres' small-angle branch, giant
// points and solves with a robust loss take the point-major code of miba_mf.hpp (same launch, per-tile decision).
#pragma once

#include "miba_mf.hpp"

namespace miba {

#ifndef MIBA_CM_WAVES
#define MIBA_CM_WAVES 3   // waves per SIMD the register allocation aims at (168 VGPRs)
#endif
constexpr int kCmPtLds = 6;   // arrays of kCmMaxPts doubles per sub-tile in LDS: X0, X1, X2, a0, a1, a2 (a: sum of E'u, then y)

struct CmArgs {
  const CmTile* tiles;
  const CmSub* subs;
  const uint2* idx;   // [sub-tile][256] four 16-bit entries
  const uint8_t* slot;   // [sub-tile][256] LDS slot of the sub-tile's k-th point
  const int4* cam64;   // [tile][64] per camera slot: {camera, slab row, first accumulator row | rows << 8, 0}: addressed by the
                       // block index alone, so that a workgroup's first loads already fetch them
  unsigned long long* clk;   // [4] measurement: shader-clock and 100 MHz counter at the start and the end of ONE workgroup (the
                             // middle one of the grid), from which the host derives the clock the kernel ran at; may be null
};

// MIBA_CM_STAMPS (debug builds): cycles per phase, summed over the waves, in a device array that miba_time_matvec prints
#ifdef MIBA_CM_STAMPS
constexpr int kCmStampBlocks = 16384;
__device__ unsigned long long g_cm_stamps[kCmStampBlocks * 4 * 16];   // [block][wave][16]: plain stores, summed on the host
#define CM_T(var) const long long var = clock64()
#define CM_ACC_DECL long long cm_acc[6] = {0, 0, 0, 0, 0, 0}
#define CM_ACC(i, a, b) cm_acc[i] += (b) - (a)
#define CM_ACC_FLUSH do { if ((threadIdx.x & 63) == 0 && blockIdx.x < kCmStampBlocks) { for (int i_ = 0; i_ < 6; ++i_) g_cm_stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + i_] += (unsigned long long)cm_acc[i_]; } } while (0)
#define CM_ADD(i, a, b) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < kCmStampBlocks) g_cm_stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (i)] += (unsigned long long)((b) - (a)); } while (0)
#else
#define CM_T(var) do {} while (0)
#define CM_ACC_DECL do {} while (0)
#define CM_ACC(i, a, b) do {} while (0)
#define CM_ACC_FLUSH do {} while (0)
#define CM_ADD(i, a, b) do {} while (0)
#endif

// 8-byte LDS read that stays a ds_read_b64 (256 B/clk): left to itself the compiler pairs the reads of two point arrays into a
// ds_read2st64_b64, which the LDS serves at half that rate
__device__ __forceinline__ double lds_read64(const double* p) {
  typedef const volatile __attribute__((address_space(3))) double lds_cvd;
  return *(lds_cvd*)p;
}

// one sub-tile's worth of LDS point slots: 256 x 6 doubles; two of them (double-buffered by sub-tile parity)
constexpr int kCmPtBuf = kCmMaxPts * kCmPtLds;

__device__ __forceinline__ void mf_matvec_cm_tile(const Tile& t, const CmTile ctile, const CmArgs& A, const double* __restrict__ ptrec,
                                                  const double* s_cam, double* s_acc, double* s_pt) {
  const int tid = threadIdx.x;
  // stage of a sub-tile: X of its k-th point into slot sl of the three X arrays, the point's sums cleared
  auto stage_write = [&](double* buf, const int sl, const double x0, const double x1, const double x2, const bool on) {
    if (on) {
      buf[sl] = x0; buf[kCmMaxPts + sl] = x1; buf[2 * kCmMaxPts + sl] = x2;
      buf[3 * kCmMaxPts + sl] = 0.0; buf[4 * kCmMaxPts + sl] = 0.0; buf[5 * kCmMaxPts + sl] = 0.0;
    }
  };
  const CmSub* subs = A.subs + ctile.sub0;
  // prologue: points of sub-tile 0 (their range comes with the tile record: nothing here waits for the sub-tile record)
  int my_slot = A.slot[(size_t)ctile.sub0 * kCmLanes + tid];
  {
    const bool on = tid < ctile.npt0;
    const double* rp = ptrec + (size_t)kPtRec * (size_t)(ctile.pt0 + (on ? tid : 0));
    stage_write(s_pt, my_slot, rp[0], rp[1], rp[2], on);
  }
  uint2 w = A.idx[(size_t)ctile.sub0 * kCmLanes + tid];
  CmSub sb = subs[0];
  CM_ACC_DECL;
  __syncthreads();
  for (int s = 0; s < ctile.nsub; ++s) {
    double* buf = s_pt + (s & 1) * kCmPtBuf;
    double* buf_next = s_pt + ((s + 1) & 1) * kCmPtBuf;
    // requests for the next sub-tile (points, index words) and this sub-tile's W rows: in flight during phase 1
    CM_T(c0);
    const bool more = s + 1 < ctile.nsub;
    // (the next record is only needed at the end of this trip: its point range follows from this one)
    const CmSub sn_rec = subs[more ? s + 1 : s];
    const int n_pt0 = sb.pt0 + sb.npt, n_npt = sb.npt_next;
    const bool on_n = more && tid < n_npt;
    const double* rn = ptrec + (size_t)kPtRec * (size_t)(on_n ? n_pt0 + tid : sb.pt0);
    const double nx0 = rn[0], nx1 = rn[1], nx2 = rn[2];
    const size_t nxt = (size_t)(ctile.sub0 + (more ? s + 1 : s)) * kCmLanes + tid;
    const uint2 w_next = A.idx[nxt];
    const int slot_next = A.slot[nxt];
    const bool on_p = tid < sb.npt;
    const double* rw = ptrec + (size_t)kPtRec * (size_t)(sb.pt0 + (on_p ? tid : 0)) + 6;
    double W[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) W[k] = rw[k];

    // ---- phase 1 ----
    const unsigned e[kCmSeg] = {w.x & 0xFFFFu, w.x >> 16, w.y & 0xFFFFu, w.y >> 16};
    const bool seg_on = (e[0] & 0x8000u) != 0;
    const int cs = (e[0] >> 8) & 63;
    double sE[kCmSeg][6], su0[kCmSeg], su1[kCmSeg], spx[kCmSeg], spy[kCmSeg];
    double k1 = 0.0, k2 = 0.0;
    if (seg_on) {
      double cd[kCamLds];
      mf_read_cam(cd, s_cam, cs);
      k1 = cd[13]; k2 = cd[14];
#pragma unroll
      for (int j = 0; j < kCmSeg; ++j) {
        double* pp = buf + (e[j] & 0xFF);
        const double x0 = lds_read64(pp), x1 = lds_read64(pp + kCmMaxPts), x2 = lds_read64(pp + 2 * kCmMaxPts);
        MfGeo geo;
        mf_geom<false>(cd, x0, x1, x2, geo);
        double u0, u1;
        mf_forward(geo, cd, x0, x1, x2, u0, u1);
        if (e[j] & 0x8000u) {
          atomicAdd(pp + 3 * kCmMaxPts, geo.E[0] * u0 + geo.E[3] * u1);
          atomicAdd(pp + 4 * kCmMaxPts, geo.E[1] * u0 + geo.E[4] * u1);
          atomicAdd(pp + 5 * kCmMaxPts, geo.E[2] * u0 + geo.E[5] * u1);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) sE[j][k] = geo.E[k];
        su0[j] = u0; su1[j] = u1; spx[j] = geo.px; spy[j] = geo.py;
      }
    }
    CM_T(c1);
    __syncthreads();
    CM_T(c2);
    // ---- phase 1.5: y = -W a per point; the next sub-tile's points go into the other buffer (every thread has left phase 2
    // of the sub-tile before, which read it) ----
    if (on_p) {
      double* pp = buf + my_slot;
      const double a0 = pp[3 * kCmMaxPts], a1 = pp[4 * kCmMaxPts], a2 = pp[5 * kCmMaxPts];
      pp[3 * kCmMaxPts] = -(W[0] * a0 + W[1] * a1 + W[3] * a2);
      pp[4 * kCmMaxPts] = -(W[1] * a0 + W[2] * a1 + W[4] * a2);
      pp[5 * kCmMaxPts] = -(W[3] * a0 + W[4] * a1 + W[5] * a2);
    }
    stage_write(buf_next, slot_next, nx0, nx1, nx2, on_n);
    CM_T(c3);
    __syncthreads();
    CM_T(c4);
    // ---- phase 2 ----
    if (seg_on) {
      double o[9];
      // the six reads of step j + 1 are issued before step j is computed
      double r[2][6];
      {
        const double* pp = buf + (e[0] & 0xFF);
#pragma unroll
        for (int k = 0; k < 6; ++k) r[0][k] = lds_read64(pp + k * kCmMaxPts);
      }
#pragma unroll
      for (int j = 0; j < kCmSeg; ++j) {
        if (j + 1 < kCmSeg) {
          const double* pp = buf + (e[j + 1] & 0xFF);
#pragma unroll
          for (int k = 0; k < 6; ++k) r[(j + 1) & 1][k] = lds_read64(pp + k * kCmMaxPts);
        }
        const double x0 = r[j & 1][0], x1 = r[j & 1][1], x2 = r[j & 1][2];
        const double y0 = r[j & 1][3], y1 = r[j & 1][4], y2 = r[j & 1][5];
        const bool valid = (e[j] & 0x8000u) != 0;
        double w0 = su0[j] + sE[j][0] * y0 + sE[j][1] * y1 + sE[j][2] * y2;
        double w1 = su1[j] + sE[j][3] * y0 + sE[j][4] * y1 + sE[j][5] * y2;
        w0 = valid ? w0 : 0.0;
        w1 = valid ? w1 : 0.0;
        const double e0 = sE[j][0] * w0 + sE[j][3] * w1;
        const double e1 = sE[j][1] * w0 + sE[j][4] * w1;
        const double e2 = sE[j][2] * w0 + sE[j][5] * w1;
        const double n = spx[j] * spx[j] + spy[j] * spy[j];
        const double d = 1.0 + n * (k1 + k2 * n);
        const double sg = spx[j] * w0 + spy[j] * w1;
        const double ns = n * sg;
        if (j == 0) {
          o[0] = x1 * e2 - x2 * e1; o[1] = x2 * e0 - x0 * e2; o[2] = x0 * e1 - x1 * e0;
          o[3] = e0; o[4] = e1; o[5] = e2;
          o[6] = d * sg; o[7] = ns; o[8] = n * ns;
        } else {
          o[0] += x1 * e2 - x2 * e1; o[1] += x2 * e0 - x0 * e2; o[2] += x0 * e1 - x1 * e0;
          o[3] += e0; o[4] += e1; o[5] += e2;
          o[6] += d * sg; o[7] += ns; o[8] += n * ns;
        }
      }
      double* dst = s_acc + 9 * ((e[1] >> 8) & 127);   // the lane's accumulator row (one of its camera's)
#pragma unroll
      for (int k = 0; k < 9; ++k) atomicAdd(dst + k, o[k]);
    }
    w = w_next;
    my_slot = slot_next;
    sb = sn_rec;
    CM_T(c5);
    CM_ACC(0, c0, c1); CM_ACC(1, c1, c2); CM_ACC(2, c2, c3); CM_ACC(3, c3, c4); CM_ACC(4, c4, c5); CM_ACC(5, 0, 1);
  }
  CM_ACC_FLUSH;
}

// K3 launch: blocks [0, n_giant) take the giant tiles of tile_list, block n_giant + i takes tile i (as k_mf_matvec).
__global__ __launch_bounds__(256, MIBA_CM_WAVES) void k_mf_matvec_cm(int n_giant, const Tile* __restrict__ tiles, const int32_t* __restrict__ tile_list,
                                                         const Group* __restrict__ grps, const int32_t* __restrict__ idx,
                                                         const int32_t* __restrict__ ocam, const int2* __restrict__ wl,
                                                         const int32_t* __restrict__ tile_cams, const int32_t* __restrict__ slab_row,
                                                         const double* __restrict__ ct, const double* __restrict__ cm,
                                                         const double* __restrict__ sc, const double* __restrict__ X,
                                                         const double* __restrict__ ptrec, const double* __restrict__ x,
                                                         double* __restrict__ slab, const Ctl* __restrict__ ctl, const CmArgs A, const int n_tiles_remap) {
  // [camera records | accumulator copies (4 point-major, 2 camera-major) | rest of the point slots of the camera-major path]
  constexpr int kAccDoubles = kAccCopies * kMaxCamSlots * 9;
  constexpr int kCmAcc = kCmMaxRows * 9;
  constexpr int kExtra = 2 * kCmPtBuf - (kAccDoubles - kCmAcc);
  static_assert(kExtra > 0, "point slots overlap the unused accumulator copies");
  __shared__ __align__(16) double s_cam[kMaxCamSlots * kCamLds];
  __shared__ __align__(16) double s_acc[kAccDoubles + kExtra];
  __shared__ double s4[4];
  // first trip to memory: everything that the block index alone addresses
  // n_tiles_remap > 0: the tile blocks (padded to a multiple of eight) take the tiles so that XCD x works on the x-th eighth of them
  // (neighbouring tiles share cameras: their tables and slab rows then meet in one L2)
  int ti = max(0, (int)blockIdx.x - n_giant);
  if (n_tiles_remap > 0 && (int)blockIdx.x >= n_giant) {
    const int b = (int)blockIdx.x - n_giant, per = ((int)gridDim.x - n_giant) >> 3;
    ti = (b & 7) * per + (b >> 3);
    if (ti >= n_tiles_remap) return;
  }
  const int tid = threadIdx.x;
  const int done = ctl->done;
  const Tile t = tiles[ti];
  const CmTile ctile = A.tiles[ti];
  const int4 c64 = A.cam64[(size_t)ti * kMaxCamSlots + (tid & (kMaxCamSlots - 1))];
  int4 f64[3];   // the flush's cameras: entry tid + 256 m of the tile's (camera, component) pairs
#pragma unroll
  for (int m = 0; m < 3; ++m) f64[m] = A.cam64[(size_t)ti * kMaxCamSlots + min((tid + m * kChunk) / 9, kMaxCamSlots - 1)];
  if (done) return;
  CM_T(k0);
  // in-kernel clock (MI355X_MICROARCH.md: s_memtime ticks are shader cycles, s_memrealtime ticks are 10 ns): one workgroup stamps
  const bool clk_on = A.clk != nullptr && blockIdx.x == (gridDim.x >> 1);
  unsigned long long clk_c0 = 0, clk_r0 = 0;
  if (clk_on) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
  if ((int)blockIdx.x < n_giant) {
    mf_matvec_block<true, false>(tile_list[blockIdx.x], s_cam, s_acc, s4, tiles, grps, idx, ocam, wl, tile_cams, slab_row, ct, cm, sc, X, ptrec,
                                 nullptr, x, slab);
    return;
  }
  if (t.giant) return;
  bool small = false;
  if (tid < t.ncs) {
    mf_stage_cam(s_cam + kCamLds * tid, c64.x, ct, cm, sc, x);
    small = s_cam[kCamLds * tid + 24] != 0.0;
  }
  const int acc_stride = t.ncs * 9;
  // a camera in Ceres' small-angle branch (or no camera-major layout for the tile): the point-major code
  const int any_small = __syncthreads_or(small || ctile.nsub == 0);
  if (any_small) {
    for (int i = tid; i < kAccCopies * acc_stride; i += kChunk) s_acc[i] = 0.0;
    __syncthreads();
    mf_matvec_tile<true, false>(t, idx, wl, ptrec, nullptr, s_cam, s_acc);
    __syncthreads();  // every wave's LDS atomics have landed
    for (int i = tid; i < acc_stride; i += kChunk) {
      double v = s_acc[i];
#pragma unroll
      for (int c = 1; c < kAccCopies; ++c) v += s_acc[c * acc_stride + i];
      slab[9 * (size_t)slab_row[t.cam0 + i / 9] + i % 9] = v;
    }
    return;
  }
  for (int i = tid; i < ctile.nrows * 9; i += kChunk) s_acc[i] = 0.0;
  CM_T(k1);
  CM_ADD(6, k0, k1);
  mf_matvec_cm_tile(t, ctile, A, ptrec, s_cam, s_acc, s_acc + kCmAcc);
  CM_T(k2);
  __syncthreads();  // every wave's LDS atomics have landed
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const int i = tid + m * kChunk;
    if (i < acc_stride) {
      const int cs = i / 9, k = i - 9 * cs, r0 = f64[m].z & 0xFF, nr = f64[m].z >> 8;
      double v = s_acc[9 * r0 + k];
      for (int c = 1; c < nr; ++c) v += s_acc[9 * (r0 + c) + k];
      slab[9 * (size_t)f64[m].y + k] = v;
    }
  }
  CM_T(k3);
  CM_ADD(7, k2, k3); CM_ADD(8, k0, k3); CM_ADD(9, 0, 1);
  if (clk_on) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { A.clk[0] = clk_c0; A.clk[1] = clk_r0; A.clk[2] = c1; A.clk[3] = r1; }
  }
}

// ---- K4 (back-substitution + model cost change + candidate points) over the same camera-major sub-tiles ----------------------------
// The back-substitution is the product's first half with y_c in the cameras, then per point z = zg - W sum E'u instead of y = -W a,
// then per observation m = u + E z and m.(r - m/2): no camera-side sums at all.  On the point-major wavelets it pays 13 ds_read_b128
// per observation for the camera record and three lane scans (0.41 ms at Final-13682); here the record is read once per four
// observations and the point sums are three LDS adds.  The measurements come in camera-major order (oxy_cm, gathered once per problem
// through the layout's source slots).  Tiles with a camera in the small-angle branch or without a layout take mf_backsub_tile
// (miba_mf.hpp) inside the same launch; giant tiles keep their own launch (k_mf_backsub<true, .>), solves with a loss the old kernel.
// THE CANDIDATE COST RIDES ALONG (ct_cand != null: the camera tables of the candidate cameras, which depend on the PCG's solution
// alone and are written before this launch): phase 1.5 leaves the candidate point in the X slots of the sub-tile's LDS buffer,
// phase 2 projects it through the candidate camera and adds |r|^2 — what the cost kernel K1' (k_mf_cost) would read the index
// words, the measurements and the points a second time for (0.17 ms at Final-13682).  The tile's cost goes to partial[4 ti + 3].

// measurements of the camera-major entries: out[entry] = (ox, oy) of the observation in local slot src[entry]; padding entries get 0
__global__ void k_cm_gather_oxy(int64_t n_entries, const int32_t* __restrict__ src, const double* __restrict__ oxy, double2* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_entries) return;
  const int32_t sl = src[e];
  double2 v = make_double2(0.0, 0.0);
  if (sl >= 0) {
    const int64_t chunk = sl / kChunk, l = sl % kChunk;
    v.x = oxy[chunk * (2 * kChunk) + l];
    v.y = oxy[chunk * (2 * kChunk) + kChunk + l];
  }
  out[e] = v;
}

#ifndef MIBA_K4CM_WAVES
// Waves per SIMD the register allocation aims at.  Two: phase 1 holds the camera record (52 registers), the stash of four observations
// (80), the points' zg and W (18) and the next sub-tile's requests; at three waves (168 registers) the compiler spills freshly loaded
// values inside the loop, and a spill waits for every request issued before it (measured at Final-13682: 0.63 ms with three waves
// and spills, 0.48 ms with three waves and zg / W requested after phase 1, 0.32 ms with two waves and no spill; point-major 0.42 ms).
#define MIBA_K4CM_WAVES 2
#endif
__global__ __launch_bounds__(256, MIBA_K4CM_WAVES) void k_mf_backsub_cm(const Tile* __restrict__ tiles, const int32_t* __restrict__ idx,
                                                          const int2* __restrict__ wl, const double* __restrict__ oxy,
                                                          const double* __restrict__ ct, const double* __restrict__ cm,
                                                          const double* __restrict__ sc, const double* __restrict__ sp,
                                                          const double* __restrict__ ptrec, const double* __restrict__ yc,
                                                          double* __restrict__ yp, double* __restrict__ Xc, double* __restrict__ partial,
                                                          const CmArgs A, const double2* __restrict__ oxy_cm, const double* __restrict__ ct_cand) {
  __shared__ __align__(16) double s_cam[kMaxCamSlots * kCamLds];
  __shared__ __align__(16) double s_cam2[kMaxCamSlots * kCamCost];   // the candidate cameras: R, t, f, k1, k2
  __shared__ __align__(16) double s_pt[2 * kCmPtBuf];
  __shared__ double s4[4];
  const int ti = blockIdx.x, tid = threadIdx.x;
  const Tile t = tiles[ti];
  const CmTile ctile = A.tiles[ti];
  const int4 c64 = A.cam64[(size_t)ti * kMaxCamSlots + (tid & (kMaxCamSlots - 1))];
  if (t.giant) return;
  bool small = false;
  if (tid < t.ncs) {
    mf_stage_cam(s_cam + kCamLds * tid, c64.x, ct, cm, sc, yc);
    small = s_cam[kCamLds * tid + 24] != 0.0;
    if (ct_cand) {
      const double* T = ct_cand + (size_t)kCamTabStride * c64.x;
#pragma unroll
      for (int k = 0; k < 15; ++k) s_cam2[kCamCost * tid + k] = T[k];
    }
  }
  double mcc = 0.0, dp2 = 0.0, xp2 = 0.0, cost = 0.0;
  const int any_small = __syncthreads_or(small || ctile.nsub == 0);
  if (any_small) {
    mf_backsub_tile<false>(t, idx, wl, oxy, sp, ptrec, s_cam, yp, Xc, nullptr, mcc, dp2, xp2);
    if (ct_cand) {
      __syncthreads();   // the tile's candidate points are in memory
      cost = mf_cost_tile<false, true>(t, idx, wl, oxy, s_cam2, Xc, Loss{kLossNone, 0.0});
    }
  } else {
    auto stage_write = [&](double* buf, const int sl, const double x0, const double x1, const double x2, const bool on) {
      if (on) {
        buf[sl] = x0; buf[kCmMaxPts + sl] = x1; buf[2 * kCmMaxPts + sl] = x2;
        buf[3 * kCmMaxPts + sl] = 0.0; buf[4 * kCmMaxPts + sl] = 0.0; buf[5 * kCmMaxPts + sl] = 0.0;
      }
    };
    const CmSub* subs = A.subs + ctile.sub0;
    int my_slot = A.slot[(size_t)ctile.sub0 * kCmLanes + tid];
    {
      const bool on = tid < ctile.npt0;
      const double* rp = ptrec + (size_t)kPtRec * (size_t)(ctile.pt0 + (on ? tid : 0));
      stage_write(s_pt, my_slot, rp[0], rp[1], rp[2], on);
    }
    uint2 w = A.idx[(size_t)ctile.sub0 * kCmLanes + tid];
    CmSub sb = subs[0];
    __syncthreads();
    for (int s = 0; s < ctile.nsub; ++s) {
      double* buf = s_pt + (s & 1) * kCmPtBuf;
      double* buf_next = s_pt + ((s + 1) & 1) * kCmPtBuf;
      // requests for the next sub-tile (points, index words), this sub-tile's measurements and its points' zg, W, sp
      const bool more = s + 1 < ctile.nsub;
      const CmSub sn_rec = subs[more ? s + 1 : s];
      const int n_pt0 = sb.pt0 + sb.npt, n_npt = sb.npt_next;
      const bool on_n = more && tid < n_npt;
      const double* rn = ptrec + (size_t)kPtRec * (size_t)(on_n ? n_pt0 + tid : sb.pt0);
      const double nx0 = rn[0], nx1 = rn[1], nx2 = rn[2];
      const size_t cur = (size_t)(ctile.sub0 + s) * kCmLanes + tid;
      const size_t nxt = (size_t)(ctile.sub0 + (more ? s + 1 : s)) * kCmLanes + tid;
      const uint2 w_next = A.idx[nxt];
      const int slot_next = A.slot[nxt];
      const bool on_p = tid < sb.npt;
      const size_t pg = (size_t)(sb.pt0 + (on_p ? tid : 0));
      const double* rw = ptrec + (size_t)kPtRec * pg + 3;
      double Z[9];   // zg 0..2, W 3..8
#pragma unroll
      for (int k = 0; k < 9; ++k) Z[k] = rw[k];

      // ---- phase 1: u = F y_c, E'u into the point's sums, the residual ----
      const unsigned e[kCmSeg] = {w.x & 0xFFFFu, w.x >> 16, w.y & 0xFFFFu, w.y >> 16};
      const bool seg_on = (e[0] & 0x8000u) != 0;
      const int cs = (e[0] >> 8) & 63;
      double sE[kCmSeg][6], su0[kCmSeg], su1[kCmSeg], sr0[kCmSeg], sr1[kCmSeg];
      if (seg_on) {
        double cd[kCamLds];
        mf_read_cam(cd, s_cam, cs);
#pragma unroll
        for (int j = 0; j < kCmSeg; ++j) {
          double* pp = buf + (e[j] & 0xFF);
          const double x0 = lds_read64(pp), x1 = lds_read64(pp + kCmMaxPts), x2 = lds_read64(pp + 2 * kCmMaxPts);
          MfGeo geo;
          mf_geom<false>(cd, x0, x1, x2, geo);
          double u0, u1;
          mf_forward(geo, cd, x0, x1, x2, u0, u1);
          if (e[j] & 0x8000u) {
            atomicAdd(pp + 3 * kCmMaxPts, geo.E[0] * u0 + geo.E[3] * u1);
            atomicAdd(pp + 4 * kCmMaxPts, geo.E[1] * u0 + geo.E[4] * u1);
            atomicAdd(pp + 5 * kCmMaxPts, geo.E[2] * u0 + geo.E[5] * u1);
          }
#pragma unroll
          for (int k = 0; k < 6; ++k) sE[j][k] = geo.E[k];
          su0[j] = u0; su1[j] = u1;
          sr0[j] = geo.f * geo.d * geo.px;   // the projection; the measurement is subtracted in phase 2
          sr1[j] = geo.f * geo.d * geo.py;
        }
      }
      __syncthreads();
      // ---- phase 1.5: per point z = zg - W a (left in the sums' place), candidate point, yp; the next sub-tile's points ----
      // this lane's measurements: requested here, used in phase 2 (not in registers during phase 1)
      double2 oo[kCmSeg];
#pragma unroll
      for (int j = 0; j < kCmSeg; ++j) oo[j] = oxy_cm[cur * kCmSeg + j];
      // (the point scale is only needed for yp, which nothing in the launch waits for: requested here, used after phase 2, so
      // that it is not in registers during phase 1)
      double s0 = 1.0, s1 = 1.0, s2 = 1.0;
      if (yp) { s0 = sp[3 * pg]; s1 = sp[3 * pg + 1]; s2 = sp[3 * pg + 2]; }   // (yp == null: the LM loop, which has no use for the scaled point step)
      double z0 = 0.0, z1 = 0.0, z2 = 0.0;
      if (on_p) {
        double* pp = buf + my_slot;
        const double a0 = pp[3 * kCmMaxPts], a1 = pp[4 * kCmMaxPts], a2 = pp[5 * kCmMaxPts];
        z0 = Z[0] - (Z[3] * a0 + Z[4] * a1 + Z[6] * a2);
        z1 = Z[1] - (Z[4] * a0 + Z[5] * a1 + Z[7] * a2);
        z2 = Z[2] - (Z[6] * a0 + Z[7] * a1 + Z[8] * a2);
        pp[3 * kCmMaxPts] = z0; pp[4 * kCmMaxPts] = z1; pp[5 * kCmMaxPts] = z2;
        const double x0 = pp[0], x1 = pp[kCmMaxPts], x2 = pp[2 * kCmMaxPts];
        const double c0 = x0 - z0, c1 = x1 - z1, c2 = x2 - z2;
        Xc[3 * pg] = c0; Xc[3 * pg + 1] = c1; Xc[3 * pg + 2] = c2;
        const double d0 = x0 - c0, d1 = x1 - c1, d2 = x2 - c2;   // step norm exactly as Ceres: |x - x_cand|
        dp2 += d0 * d0 + d1 * d1 + d2 * d2;
        xp2 += c0 * c0 + c1 * c1 + c2 * c2;
        pp[0] = c0; pp[kCmMaxPts] = c1; pp[2 * kCmMaxPts] = c2;   // (X has done its work in this sub-tile: phase 2 projects the candidate)
      }
      stage_write(buf_next, slot_next, nx0, nx1, nx2, on_n);
      __syncthreads();
      // ---- phase 2: m = u + E z, model cost change; the candidate's residual ----
      if (seg_on) {
#pragma unroll
        for (int j = 0; j < kCmSeg; ++j) {
          const double* pp = buf + (e[j] & 0xFF);
          const double z0 = lds_read64(pp + 3 * kCmMaxPts), z1 = lds_read64(pp + 4 * kCmMaxPts), z2 = lds_read64(pp + 5 * kCmMaxPts);
          const double m0 = su0[j] + sE[j][0] * z0 + sE[j][1] * z1 + sE[j][2] * z2;
          const double m1 = su1[j] + sE[j][3] * z0 + sE[j][4] * z1 + sE[j][5] * z2;
          if (e[j] & 0x8000u) mcc += m0 * ((sr0[j] - oo[j].x) - 0.5 * m0) + m1 * ((sr1[j] - oo[j].y) - 0.5 * m1);
        }
        if (ct_cand) {
          double cd2[kCamCost];
          mf_read_cam_cost(cd2, s_cam2, cs);
#pragma unroll
          for (int j = 0; j < kCmSeg; ++j) {
            const double* pp = buf + (e[j] & 0xFF);
            const double c0 = lds_read64(pp), c1 = lds_read64(pp + kCmMaxPts), c2 = lds_read64(pp + 2 * kCmMaxPts);
            const double rho = mf_cost_obs(cd2, c0, c1, c2, oo[j].x, oo[j].y);
            if (e[j] & 0x8000u) cost += rho;
          }
        }
      }
      if (yp && on_p) { yp[3 * pg] = z0 * mf_rcp(s0); yp[3 * pg + 1] = z1 * mf_rcp(s1); yp[3 * pg + 2] = z2 * mf_rcp(s2); }
      w = w_next;
      my_slot = slot_next;
      sb = sn_rec;
    }
  }
  const double q0 = block_sum(mcc, s4);
  const double q1 = block_sum(dp2, s4);
  const double q2 = block_sum(xp2, s4);
  const double q3 = block_sum(cost, s4);
  if (tid == 0) {
    partial[4 * (size_t)ti + 0] = q0;
    partial[4 * (size_t)ti + 1] = q1;
    partial[4 * (size_t)ti + 2] = q2;
    partial[4 * (size_t)ti + 3] = 0.5 * q3;
  }
}

}  // namespace miba
