// miba_comm.hpp — the library's own all-reduce for the ranks of a single-process multi-device group (SURVEY 8f rank 4,
// "custom xGMI all-reduce"; the reference has no communication code at all).
//
// MI355X nodes connect the 8 GPUs point to point over xGMI, and the vectors reduced per PCG iteration are small
// (9 doubles per camera: 1 MB at Final-13682): a ring all-reduce pays 14 latency hops, a direct exchange two.  Every rank
// runs k_peer_allreduce on its own stream at the same point of the (identical) call sequence:
//   1. copy the own input into the own fine-grained staging buffer, tell every peer (ready flag, system scope);
//   2. wait for every peer's ready flag; reduce the own slice [n r / N, n (r + 1) / N) over the ranks IN RANK ORDER, and push
//      the result into the out-half of every rank's staging buffer (reduce-scatter and all-gather in one step: each slice
//      is summed by exactly one rank, so every rank ends with the same bits);
//   3. tell every peer (done flag), wait for every peer's done flag, copy the out-half into the caller's buffer.
// All cross-device traffic touches only the fine-grained (hipDeviceMallocFinegrained) staging buffers and signal blocks,
// with system-scope atomics and fences; flags carry the epoch (the count of all-reduces so far), so nothing is reset
// between calls.  Every spin is bounded: a peer that never arrives sets `timeout` and the host reports MIBA_ERR_RCCL.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace miba {

constexpr int kMaxRanks = 8;
constexpr int kPeerMaxRanksPerDevice = 3;          // ranks of one group that may share a device under the peer transport (see miba_create_multi)
constexpr int kPeerBlocks = 32;                  // workgroups per rank kernel (all resident; they wait on each other)
constexpr size_t kPeerChunk = (size_t)1 << 20;   // doubles per staging half (8 MiB): larger buffers go in pieces

struct PeerSignals {            // one per rank, in that rank's fine-grained device memory
  uint32_t ready[kMaxRanks];    // ready[j] = epoch: rank j's input of this epoch is staged (written by rank j)
  uint32_t done[kMaxRanks];     // done[j]  = epoch: rank j has pushed its reduced slice to everybody (written by rank j)
  uint32_t blocks_staged;       // local counters of the own kernel's blocks (monotonic: epoch * kPeerBlocks)
  uint32_t blocks_pushed;
  uint32_t timeout;             // a bounded spin ran out
  uint32_t pad[13];
  // exchange inside the PCG tail kernel (pcg_tail<MODE, 2>, miba_pcg.hpp): xready[j] = epoch: rank j's share of q of that
  // exchange has arrived in this rank's exchange buffer (written by rank j); xarrive: workgroups of the own tail kernel that
  // have finished pushing (only grows)
  uint32_t xready[kMaxRanks];
  uint32_t xarrive;
  uint32_t pad2[23];
};
static_assert(sizeof(PeerSignals) == 256, "signal block layout");

struct PeerArgs {
  double* stage[kMaxRanks];     // [in kPeerChunk | out kPeerChunk] of every rank
  PeerSignals* sig[kMaxRanks];
  int n_ranks, rank;
  uint32_t epoch;
  int op_max;
};

__device__ __forceinline__ uint32_t peer_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void peer_st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double peer_ldd(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void peer_std(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// one thread spins until *p >= want (epochs only grow) or about two seconds have passed; false = timed out
__device__ __forceinline__ bool peer_wait(const uint32_t* p, uint32_t want, uint32_t* timeout) {
  const long long t0 = wall_clock64();   // 100 MHz
  while ((int32_t)(peer_ld(p) - want) < 0) {
    __builtin_amdgcn_s_sleep(8);
    if (peer_ld(timeout) != 0) return false;
    if (wall_clock64() - t0 > 200000000LL) { peer_st(timeout, 1u); return false; }
  }
  return true;
}

__global__ __launch_bounds__(256) void k_peer_allreduce(const PeerArgs A, double* __restrict__ buf, size_t n) {
  __shared__ int s_ok;
  const int N = A.n_ranks, r = A.rank;
  PeerSignals* me = A.sig[r];
  double* my_in = A.stage[r];
  const size_t gsz = (size_t)gridDim.x * blockDim.x, gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  // 1. stage the own input, then announce it
  for (size_t i = gid; i < n; i += gsz) peer_std(my_in + i, buf[i]);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t old = __hip_atomic_fetch_add(&me->blocks_staged, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_SYSTEM);
    if (old + 1 == A.epoch * (uint32_t)gridDim.x)   // the last block of this kernel: everything is staged
      for (int j = 0; j < N; ++j) peer_st(&A.sig[j]->ready[r], A.epoch);
  }
  // 2. every peer staged?
  if (threadIdx.x == 0) s_ok = 1;
  __syncthreads();
  if ((int)threadIdx.x < N && !peer_wait(&me->ready[threadIdx.x], A.epoch, &me->timeout)) s_ok = 0;
  __syncthreads();
  if (!s_ok) return;
  __threadfence_system();
  const size_t lo = n * (size_t)r / N, hi = n * (size_t)(r + 1) / N;
  for (size_t i = lo + gid; i < hi; i += gsz) {
    double v = peer_ldd(A.stage[0] + i);
    for (int j = 1; j < N; ++j) {
      const double w = peer_ldd(A.stage[j] + i);
      v = A.op_max ? fmax(v, w) : v + w;
    }
    for (int j = 0; j < N; ++j) peer_std(A.stage[j] + kPeerChunk + i, v);
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t old = __hip_atomic_fetch_add(&me->blocks_pushed, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_SYSTEM);
    if (old + 1 == A.epoch * (uint32_t)gridDim.x)
      for (int j = 0; j < N; ++j) peer_st(&A.sig[j]->done[r], A.epoch);
  }
  // 3. every slice arrived?
  if ((int)threadIdx.x < N && !peer_wait(&me->done[threadIdx.x], A.epoch, &me->timeout)) s_ok = 0;
  __syncthreads();
  if (!s_ok) return;
  __threadfence_system();
  const double* my_out = my_in + kPeerChunk;
  for (size_t i = gid; i < n; i += gsz) buf[i] = peer_ldd(my_out + i);
}

}  // namespace miba
