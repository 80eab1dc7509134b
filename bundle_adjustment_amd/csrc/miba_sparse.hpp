// miba_sparse.hpp — exact solve of the reduced camera system beyond the dense limit ("block-sparse explicit
// Schur + Cholesky"; what ceres::SPARSE_SCHUR / SPARSE_NORMAL_CHOLESKY, the solver type the reference app asks for at
// app/main.cpp:19, do for a problem of Final-13682's size).
//
// S is stored as 64 x 64 tiles of its lower block triangle: a tile exists where two cameras of its row and column ranges see
// a common point, or where the factorisation fills in (symbolic elimination of the tile graph on the host, once per problem).
// The right-hand side rides along as one more tile row (row nT), exactly as row N does in the dense factorisation, so the
// forward substitution costs nothing.  Numeric factorisation: right-looking over tile columns with the dense kernels' own
// building blocks — the diagonal tile factored in the registers of one wave, panel tiles solved against it, trailing tiles
// updated on the fp64 matrix cores — addressed through per-column tile lists instead of row and column offsets.
#pragma once

#include "miba_dense.hpp"

namespace miba {

// 16 x 16 blocks of matrices held in LDS as operands of v_mfma_f64_16x16x4_f64 (lane l: r = l & 15, g = l >> 4):
// A fragment kk = A[r][4 kk + g], B fragment kk = B[4 kk + g][r], D register reg = D[g + 4 reg][r].

// One wave: Cholesky factor L of a diagonal tile, then W = L^-T, both left in LDS.  Only W is ever used (Winv[c * 64 + i] =
// W[i][c] = (L^-1)[c][i], i.e. L^-1 row-major): the tiles of the column are multiplied by it on the matrix cores (k_tchol_col),
// and the backward substitution uses it too; the factor of the diagonal tile itself is never needed again.
// Both halves are blocked by 16 so that the chain of dependent steps is 4 x 16 + 16 instead of 64 + 64 steps of up to 63
// FMAs each (one wave issues an fp64 FMA every 8 cycles):
//   factor   per block column: the 16 columns right-looking with lane = row (chol_diag_wave's step, 15 FMAs at most), then
//            the trailing blocks A_pq -= L_p L_q' on the matrix cores, operands straight from the factor in LDS;
//   inverse  M = L^-1: the four diagonal blocks at once (lane = block, column: sixteen substitution steps), then the blocks at
//            distance 1, 2, 3 from the diagonal, M_ab = -M_aa sum_c L_ac M_cb, two matrix-core stages each.
// LOAD = false: the tile is in As already (k_tchol_update, which has just formed it).
template <bool LOAD = true>
__device__ __forceinline__ bool tchol_diag_wave(const double* __restrict__ A, const int lane, double* As /*[64][kLs]: the tile, then M = L^-1, row-major*/,
                                                double* Lt /*[64][kLs]: Lt[k][r] = L[r][k]*/, double* Lr /*[64]: 1 / L[k][k]*/, double (*Sb)[256]) {
  const int r = lane & 15, g = lane >> 4;
  if (LOAD) {
#pragma unroll
    for (int r0 = 0; r0 < kNB; r0 += 16) {
      double v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = A[(size_t)(r0 + i) * kNB + lane];
#pragma unroll
      for (int i = 0; i < 16; ++i) As[(r0 + i) * kLs + lane] = v[i];
    }
  }
  bool bad = false;
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) {
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = As[lane * kLs + 16 * jb + c];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int J = 16 * jb + j;
      double d = lane_bcast(a[j], J);
      if (!(d > 0.0)) { bad = true; d = 1.0; }
      const double rd = mf_rsqrt(d);
      const double l = (lane >= J) ? a[j] * rd : 0.0;
      Lt[J * kLs + lane] = l;
      if (lane == 0) Lr[J] = rd;
      double t[16];   // (requested together: taken one by one, each read costs its LDS round trip)
#pragma unroll
      for (int c = j + 1; c < 16; ++c) t[c] = Lt[J * kLs + 16 * jb + c];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = j + 1; c < 16; ++c) a[c] = fma(-l, t[c], a[c]);
#pragma unroll
      for (int c = j + 1; c < 16; ++c) asm volatile("" : "+v"(a[c]));
    }
    // trailing blocks (p, q), jb < q <= p: A_pq -= L_p,jb L_q,jb'   (the accumulator starts at -A_pq and is stored negated)
#pragma unroll
    for (int p = jb + 1; p < 4; ++p)
#pragma unroll
      for (int q = jb + 1; q <= p; ++q) {
        d4_t c;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) c[reg] = -As[(16 * p + g + 4 * reg) * kLs + 16 * q + r];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          c = __builtin_amdgcn_mfma_f64_16x16x4f64(Lt[(16 * jb + 4 * kk + g) * kLs + 16 * p + r], Lt[(16 * jb + 4 * kk + g) * kLs + 16 * q + r], c, 0, 0, 0);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) As[(16 * p + g + 4 * reg) * kLs + 16 * q + r] = -c[reg];
      }
  }
  // M_aa = L_aa^-1, a = g: lane (g, r) solves x L_aa' = e_r, which is column r of M_aa
  {
    double x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) x[c] = (c == r) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const double xk = x[k] * Lr[16 * g + k];
      x[k] = xk;
      double t[16];
#pragma unroll
      for (int c = k + 1; c < 16; ++c) t[c] = Lt[(16 * g + k) * kLs + 16 * g + c];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = k + 1; c < 16; ++c) x[c] = fma(-xk, t[c], x[c]);
#pragma unroll
      for (int c = k + 1; c < 16; ++c) asm volatile("" : "+v"(x[c]));
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) As[(16 * g + c) * kLs + 16 * g + r] = x[c];
  }
  // blocks at distance dd below the diagonal: S = sum_{c = b}^{a - 1} L_ac M_cb, then M_ab = -M_aa S
#pragma unroll
  for (int dd = 1; dd < 4; ++dd) {
#pragma unroll
    for (int a = dd; a < 4; ++a) {
      const int b = a - dd;
      d4_t c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int cc = b; cc < a; ++cc)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          c = __builtin_amdgcn_mfma_f64_16x16x4f64(Lt[(16 * cc + 4 * kk + g) * kLs + 16 * a + r], As[(16 * cc + 4 * kk + g) * kLs + 16 * b + r], c, 0, 0, 0);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) Sb[a - dd][(g + 4 * reg) * 16 + r] = c[reg];
    }
#pragma unroll
    for (int a = dd; a < 4; ++a) {
      const int b = a - dd;
      d4_t c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(As[(16 * a + r) * kLs + 16 * a + 4 * kk + g], Sb[a - dd][(4 * kk + g) * 16 + r], c, 0, 0, 0);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) As[(16 * a + g + 4 * reg) * kLs + 16 * b + r] = -c[reg];
    }
  }
  return bad;   // (M is lower block triangular: the blocks of As above the diagonal still hold what is left of the tile)
}

// Column step, first launch.  A launch takes one column of EVERY group of a level of the dissected order at once (arcs, then nested
// separator sets level by level; the columns of a level's groups do not touch each other: sparse_symbolic in miba.hip): workgroup e = wg[blockIdx.x] = {slot of the column's diagonal tile,
// slot of its own tile (one of the tiles below the diagonal, the right-hand-side tile among them), column k, index b of the tile in
// the column} turns its tile into X = A W = A L^-T.  Every workgroup factors and inverts the diagonal tile itself,
// on its first wave, while the loads of its own tile are in flight: a launch that does it once and hands W over through memory
// costs the column another launch ramp and two trips to memory (9 us where the product itself is 1.5).  The column's workgroup 0 leaves
// W for the backward substitution.  A NEGATIVE diagonal slot says that W is in memory already: the update launch of the step before,
// whose workgroup for (k, k) saw the tile's last update, has factored it there, beside the other updates of that launch instead of
// at the head of this one (15 of a column launch's 21 us).  Product on the fp64 matrix cores: wave = 16 rows, A fragment lane l = A[l & 15][4 kk + (l >> 4)],
// B[k][n] = W[k][16 nt + n] = M[16 nt + n][k] (zero for k >= 16 (nt + 1): M is lower block triangular), D: row (l >> 4) + 4 reg,
// column l & 15.  In place: a wave reads all of its sixteen rows before it writes any of them, and nobody else touches them.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_tchol_col(double* __restrict__ tiles, const int4* __restrict__ wg /*of this step*/,
                                                   double* __restrict__ Winv_all /*[column][64 * 64]*/, int* __restrict__ flag) {
  __shared__ __align__(16) double As[kNB * kLs];
  __shared__ __align__(16) double Lt[kNB * kLs];
  __shared__ double Lr[kNB];
  __shared__ double Sb[3][256];
  const int tid = threadIdx.x;
  const int l = tid & 63, wv = tid >> 6;
  const int ar = l & 15, ak = l >> 4;
  const int4 e = wg[blockIdx.x];
  double* A = tiles + (size_t)e.y * 4096;
  double af[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) af[kk] = A[(size_t)(16 * wv + ar) * 64 + 4 * kk + ak];
  if (e.x < 0) {
    const double* Winv = Winv_all + (size_t)e.z * 4096;
    double wv16[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) wv16[q] = Winv[tid + 256 * q];
#pragma unroll
    for (int q = 0; q < 16; ++q) As[(4 * q + wv) * kLs + l] = wv16[q];   // element tid + 256 q = row 4 q + wv, column l
  } else if (wv == 0) {
    const bool bad = tchol_diag_wave(tiles + (size_t)e.x * 4096, l, As, Lt, Lr, Sb);
    if (e.w == 0) {
      if (bad && l == 0) *flag = 1;
      double* Winv = Winv_all + (size_t)e.z * 4096;
#pragma unroll 16
      for (int c = 0; c < kNB; ++c) Winv[c * kNB + l] = (ak <= (c >> 4)) ? As[c * kLs + l] : 0.0;
    }
  }
  __syncthreads();
  d4_t c[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    c[nt] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4 * (nt + 1); ++kk) c[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], As[(16 * nt + ar) * kLs + 4 * kk + ak], c[nt], 0, 0, 0);
  }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) A[(size_t)(16 * wv + ak + 4 * reg) * 64 + 16 * nt + ar] = c[nt][reg];
}

// Column step, second launch, trailing update: tile (r_I, r_J) -= L(r_I, k) L(r_J, k)' for the tiles I >= J below the diagonal of every
// column of the step.  Workgroup e = wg[blockIdx.x] = {slot of L(r_I, k), slot of L(r_J, k), destination slot}: the pairs and their
// destinations are listed once, in the symbolic phase (the tile map for a destination inside the matrix; for a column of an arc whose
// update lands in separator x separator tiles, or in the right-hand side of a separator column, the arc's PRIVATE tile: two arcs of
// one step would otherwise add to the same tile at the same time; k_tchol_merge adds the private tiles to the matrix once all arcs
// are done).
// e.w = c + 1 > 0: the destination is the diagonal tile of column c, which the NEXT step takes, and this is its last update: the
// workgroup goes on to factor the tile (W_c = L_cc^-T into Winv_all, what k_tchol_col would otherwise do at the head of the next
// launch) while the launch's other workgroups are still at their updates.  A negative operand slot: no update, the tile was complete
// before this launch (no column of this step touches it): the workgroup only factors it.
__global__ __launch_bounds__(256) void k_tchol_update(double* __restrict__ tiles, const int4* __restrict__ wg /*of this step*/,
                                                      double* __restrict__ Winv_all, int* __restrict__ flag) {
  __shared__ __align__(16) double Lt[kNB * kLs];   // first B', staged with a row stride of 65; the factor's L' afterwards
  __shared__ __align__(16) double As[kNB * kLs];
  __shared__ double Lr[kNB];
  __shared__ double Sb[3][256];
  double (*bt)[kNB + 1] = reinterpret_cast<double (*)[kNB + 1]>(Lt);
  const int4 e = wg[blockIdx.x];
  double* Dt = tiles + (size_t)e.z * 4096;
  const int tid = threadIdx.x;
  const int l = tid & 63, wv = tid >> 6;
  const int ar = l & 15, ak = l >> 4;
  if (e.x >= 0) {
    const double* A = tiles + (size_t)e.x * 4096;
    const double* B = tiles + (size_t)e.y * 4096;
    // both operands and the destination are requested together, before the first wait: one trip to memory.  (As a loop
    // "load, store to LDS" the staging of B took sixteen trips, one per element and thread.)  The destination is negated as
    // the accumulator's start.
    double bv[16], af[16];
    d4_t c[4];
#pragma unroll
    for (int q = 0; q < 16; ++q) bv[q] = B[tid + 256 * q];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) af[kk] = A[(size_t)(16 * wv + ar) * 64 + 4 * kk + ak];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) c[nt][reg] = -Dt[(size_t)(16 * wv + ak + 4 * reg) * 64 + 16 * nt + ar];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 16; ++q) bt[4 * q + wv][l] = bv[q];   // element tid + 256 q = row 4 q + wv, column l
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) c[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], bt[16 * nt + ar][4 * kk + ak], c[nt], 0, 0, 0);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) Dt[(size_t)(16 * wv + ak + 4 * reg) * 64 + 16 * nt + ar] = -c[nt][reg];
    }
    if (e.w <= 0) return;   // (the same for the whole workgroup)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) As[(16 * wv + ak + 4 * reg) * kLs + 16 * nt + ar] = -c[nt][reg];
  } else {
    double tv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) tv[q] = Dt[tid + 256 * q];
#pragma unroll
    for (int q = 0; q < 16; ++q) As[(4 * q + wv) * kLs + l] = tv[q];   // element tid + 256 q = row 4 q + wv, column l
  }
  __syncthreads();   // the tile is in As, and nobody reads B' any more
  if (wv == 0) {
    const bool bad = tchol_diag_wave<false>(nullptr, l, As, Lt, Lr, Sb);
    if (bad && l == 0) *flag = 1;
    double* Winv = Winv_all + (size_t)(e.w - 1) * 4096;
#pragma unroll 16
    for (int cc = 0; cc < kNB; ++cc) Winv[cc * kNB + l] = (ak <= (cc >> 4)) ? As[cc * kLs + l] : 0.0;
  }
}

// the arcs' private tiles into the matrix: workgroup d adds the private tiles src[ptr[d]] .. src[ptr[d + 1] - 1] (in that order:
// arcs ascending) to tile dst[d]
__global__ __launch_bounds__(256) void k_tchol_merge(double* __restrict__ tiles, const int32_t* __restrict__ ptr, const int32_t* __restrict__ src,
                                                     const int32_t* __restrict__ dst) {
  const int d = blockIdx.x, tid = threadIdx.x;
  const int p0 = ptr[d], p1 = ptr[d + 1];
  double* D = tiles + (size_t)dst[d] * 4096;
  double v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] = D[tid + 256 * q];
  for (int p = p0; p < p1; ++p) {
    const double* S = tiles + (size_t)src[p] * 4096;
    double w[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) w[q] = S[tid + 256 * q];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] += w[q];
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) D[tid + 256 * q] = v[q];
}

// right-hand side into row 0 of the tiles of the extra tile row (slots rhs0 .. rhs0 + nT - 1, cleared beforehand); tile row p holds
// the cameras of group grp[p], nine entries each, entry 63 and the cameras beyond n_cam are padding
__global__ void k_tchol_rhs(int n_cam, int nT, const double* __restrict__ v, double* __restrict__ tiles, int rhs0, const int32_t* __restrict__ grp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nT * 64) return;
  const int p = i >> 6, l = i & 63;
  const int cam = kTileCams * grp[p] + l / 9;
  tiles[(size_t)(rhs0 + p) * 4096 + l] = (l < 63 && cam < n_cam) ? v[9 * (size_t)cam + l % 9] : 0.0;
}
// y (the forward-substituted right-hand side the factorisation has left in the extra tile row) as one vector
__global__ void k_tchol_gety(int nT, const double* __restrict__ tiles, int rhs0, double* __restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nT * 64) return;
  y[i] = tiles[(size_t)(rhs0 + (i >> 6)) * 4096 + (i & 63)];
}
// unit diagonal on the padding rows of the diagonal tiles (slot p = diagonal tile of tile row p)
__global__ void k_tchol_pad(int n_cam, int nT, double* __restrict__ tiles, const int32_t* __restrict__ grp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nT * 64) return;
  const int p = i >> 6, l = i & 63;
  if (l == 63 || kTileCams * grp[p] + l / 9 >= n_cam) tiles[(size_t)p * 4096 + (size_t)l * 64 + l] = 1.0;
}
// the solution back in camera order
__global__ void k_tchol_unperm(int n_cam, const double* __restrict__ xp, const int32_t* __restrict__ pos, double* __restrict__ x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * n_cam) return;
  const int cam = i / 9, g = cam / kTileCams;
  x[i] = xp[(size_t)pos[g] * 64 + 9 * (cam - kTileCams * g) + (i - 9 * cam)];
}

// Backward step of L' x = y, one tile row of every arc per launch (a separator row alone).  Workgroup e = wg[blockIdx.x] = {row k, slot
// of its tile (k, J) or -1 when the row has no tile (then the workgroup only forms x_k), column J, flags}.  x_k = W_k y_k (W_k = L_kk^-T
// from the column's first launch) comes from the launch before this one when flag 1 is set, else every workgroup (one wave) forms it;
// then the workgroup takes its tile: y_J -= L(k, J)' x_k.  The workgroup with flag 2 holds the tile with J = k - 1: after its update
// y_(k-1) is final (the rows above k were applied by earlier launches), so it goes on to x_(k-1) = W_(k-1) y_(k-1) with W requested
// at the start, and the next launch finds it in memory instead of waiting for W and y itself.  Flag 4: the row's first workgroup
// (it stores x_k).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_tchol_bwd(const double* __restrict__ tiles, const int4* __restrict__ wg /*of this step*/,
                                                  const double* __restrict__ Winv_all, double* __restrict__ y, double* __restrict__ x) {
  __shared__ double ys[kNB];
  __shared__ double xk[kNB];
  const int tid = threadIdx.x;
  const int4 e = wg[blockIdx.x];
  const int k = e.x;
  const bool work = e.y >= 0;
  const bool have_x = (e.w & 1) != 0, lead = (e.w & 2) != 0, first = (e.w & 4) != 0;
  // everything the step reads is requested before the first wait: the tile (one column per lane), the entry of y it updates,
  // W of the next step in the workgroup that runs ahead.  (Left to the compiler the tile's loads came one by one, each
  // waited for before the next: the 64 FMAs form one chain, and it saw no reason to keep registers for loads in flight.)
  const double* T = tiles + (size_t)(work ? e.y : 0) * 4096;
  const int col = work ? e.z : 0;
  double tv[kNB], wn[kNB];
  double y_old = 0.0;
  if (work) {
#pragma unroll
    for (int r = 0; r < kNB; ++r) tv[r] = T[(size_t)r * 64 + tid];
    y_old = y[col * kNB + tid];
  }
  if (lead) {
    const double* Wn = Winv_all + (size_t)(k - 1) * 4096;
#pragma unroll
    for (int c = 0; c < kNB; ++c) wn[c] = Wn[c * kNB + tid];
  }
  __builtin_amdgcn_sched_barrier(0);
  if (have_x) {
    xk[tid] = x[k * kNB + tid];
  } else {
    const double* Wk = Winv_all + (size_t)k * 4096;
    ys[tid] = y[k * kNB + tid];
    double wk[kNB];
#pragma unroll
    for (int c = 0; c < kNB; ++c) wk[c] = Wk[c * kNB + tid];
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    double xv = 0.0;
#pragma unroll
    for (int c = 0; c < kNB; ++c) xv = fma(wk[c], ys[c], xv);
    xk[tid] = xv;
    if (first) x[k * kNB + tid] = xv;
  }
  __syncthreads();
  if (!work) return;
  double sacc = 0.0;
#pragma unroll
  for (int r = 0; r < kNB; ++r) sacc = fma(tv[r], xk[r], sacc);
  const double yn = y_old - sacc;
  y[col * kNB + tid] = yn;
  if (lead) {
    ys[tid] = yn;
    __syncthreads();
    double xv = 0.0;
#pragma unroll
    for (int c = 0; c < kNB; ++c) xv = fma(wn[c], ys[c], xv);
    x[(k - 1) * kNB + tid] = xv;
  }
}

}  // namespace miba
