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

// x L' = a for one row a held in registers (x overwrites it): right-looking forward substitution against the diagonal factor in
// LDS (Lt[k][c] = L[c][k], Lr = 1 / diagonal), every solved entry updates all the later ones with independent FMAs
__device__ __forceinline__ void tchol_row_solve(double* x, const double* Lt, const double* Lr) {
#pragma unroll
  for (int k = 0; k < kNB; ++k) {
    const double xk = x[k] * Lr[k];
    x[k] = xk;
    double t[kNB];
#pragma unroll
    for (int c = k + 1; c < kNB; ++c) t[c] = Lt[k * kLs + c];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = k + 1; c < kNB; ++c) x[c] = fma(-xk, t[c], x[c]);
#pragma unroll
    for (int c = k + 1; c < kNB; ++c) asm volatile("" : "+v"(x[c]));
  }
}

// Column step k, first launch (one wave): Cholesky factor L of the diagonal tile (chol_diag_wave, miba_dense.hpp), then
// W = L^-T by solving x L' = e_lane against it.  Only W leaves the kernel (Winv[c * 64 + i] = W[i][c]): the panel tiles are
// multiplied by it on the matrix cores (k_tchol_panel), and the backward substitution uses it too; the factor of the diagonal
// tile itself is never needed again.  (One sweep for both, sharing the LDS operands, would save a third of the time; the
// doubly unrolled 64-step nest does not get through the compiler in a quarter of an hour.)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_tchol_diag(const double* __restrict__ tiles, const int32_t* __restrict__ slots /*of this column*/,
                                                   double* __restrict__ Winv /*[64 * 64]*/, int* __restrict__ flag) {
  __shared__ __align__(16) double Lt[kNB * kLs];
  __shared__ double Lr[kNB];
  const int lane = threadIdx.x;
  const bool bad = chol_diag_wave(tiles + (size_t)slots[0] * 4096, 64, lane, Lt, Lr);
  if (bad && lane == 0) *flag = 1;
  double x[kNB];
#pragma unroll
  for (int c = 0; c < kNB; ++c) x[c] = (c == lane) ? 1.0 : 0.0;
  tchol_row_solve(x, Lt, Lr);
#pragma unroll
  for (int c = 0; c < kNB; ++c) Winv[c * kNB + lane] = x[c];
}

// Column step k, second launch: panel tile b of the column (slots[b]: the tiles below the diagonal, the right-hand-side tile
// last) becomes X = A W = A L^-T on the fp64 matrix cores: wave = 16 rows, A fragment lane l = A[l & 15][4 kk + (l >> 4)],
// B[k][n] = W[k][16 nt + n] = Winv[(16 nt + n) * 64 + k], D: row (l >> 4) + 4 reg, column l & 15.  In place: a wave reads all
// of its sixteen rows before it writes any of them, and nobody else touches them.
__global__ __launch_bounds__(256) void k_tchol_panel(double* __restrict__ tiles, const int32_t* __restrict__ slots, const double* __restrict__ Winv) {
  __shared__ double wt[kNB][kNB + 1];
  const int tid = threadIdx.x;
  double* A = tiles + (size_t)slots[blockIdx.x] * 4096;
  for (int i = tid; i < kNB * kNB; i += 256) wt[i >> 6][i & 63] = Winv[i];
  const int l = tid & 63, wv = tid >> 6;
  const int ar = l & 15, ak = l >> 4;
  double af[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) af[kk] = A[(size_t)(16 * wv + ar) * 64 + 4 * kk + ak];
  __syncthreads();
  d4_t c[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    c[nt] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) c[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], wt[16 * nt + ar][4 * kk + ak], c[nt], 0, 0, 0);
  }
  // (the wave's rows are all in af and c by now: the stores below cannot reach a value another lane of the wave still needs,
  // MFMA operands are read when the instruction issues)
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) A[(size_t)(16 * wv + ak + 4 * reg) * 64 + 16 * nt + ar] = c[nt][reg];
}

// trailing update of column step k: tile (r_I, r_J) -= L(r_I, k) L(r_J, k)' for the tiles I >= J below the diagonal of the
// column (rows[] / slots[] list them, diagonal excluded); the destination comes from the tile map (symbolic fill guarantees it)
__global__ __launch_bounds__(256) void k_tchol_update(double* __restrict__ tiles, const int32_t* __restrict__ rows,
                                                      const int32_t* __restrict__ slots, const int32_t* __restrict__ map, int nT) {
  __shared__ double bt[kNB][kNB + 1];
  const int I = blockIdx.y, J = blockIdx.x;
  if (J > I) return;
  const int rI = rows[I], rJ = rows[J];
  if (rJ >= nT) return;                       // the right-hand-side row is never a column
  const int32_t dslot = map[(size_t)rI * nT + rJ];
  if (dslot < 0) return;                      // cannot happen after symbolic fill; never write through a bad slot
  const double* A = tiles + (size_t)slots[I] * 4096;
  const double* B = tiles + (size_t)slots[J] * 4096;
  double* Dt = tiles + (size_t)dslot * 4096;
  const int tid = threadIdx.x;
  for (int i = tid; i < kNB * kNB; i += 256) bt[i >> 6][i & 63] = B[i];
  __syncthreads();
  const int l = tid & 63, wv = tid >> 6;
  const int ar = l & 15, ak = l >> 4;
  double af[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) af[kk] = A[(size_t)(16 * wv + ar) * 64 + 4 * kk + ak];
  // the destination is requested together with the operands (one trip to memory, not two), negated as the accumulator's start
  d4_t c[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) c[nt][reg] = -Dt[(size_t)(16 * wv + ak + 4 * reg) * 64 + 16 * nt + ar];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) c[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], bt[16 * nt + ar][4 * kk + ak], c[nt], 0, 0, 0);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) Dt[(size_t)(16 * wv + ak + 4 * reg) * 64 + 16 * nt + ar] = -c[nt][reg];
  }
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

// backward step k of L' x = y: x_k = W_k y_k (every workgroup, one wave; W_k = L_kk^-T from the panel step), then workgroup b takes
// tile (k, J_b) of tile row k: y_J -= L(k, J)' x_k
__global__ __launch_bounds__(64) void k_tchol_bwd(const double* __restrict__ tiles, int k, int n_row_tiles, const int32_t* __restrict__ cols,
                                                  const int32_t* __restrict__ slots, const double* __restrict__ Winv, double* __restrict__ y,
                                                  double* __restrict__ x) {
  __shared__ double ys[kNB];
  __shared__ double xk[kNB];
  const int tid = threadIdx.x;
  ys[tid] = y[k * kNB + tid];
  __syncthreads();
  double xv = 0.0;
#pragma unroll 16
  for (int c = 0; c < kNB; ++c) xv = fma(Winv[c * kNB + tid], ys[c], xv);
  xk[tid] = xv;
  if (blockIdx.x == 0) x[k * kNB + tid] = xv;
  __syncthreads();
  if ((int)blockIdx.x >= n_row_tiles) return;
  const double* T = tiles + (size_t)slots[blockIdx.x] * 4096;
  double sacc = 0.0;
#pragma unroll 16
  for (int r = 0; r < kNB; ++r) sacc = fma(T[(size_t)r * 64 + tid], xk[r], sacc);
  y[cols[blockIdx.x] * kNB + tid] -= sacc;
}

}  // namespace miba
