// miba_sort.hpp — device side of the repack of miba_set_problem (replaces the observation loop of
// BundleAdjustment::createProblem, /root/reference/src/BundleAdjustment.cpp:44-90, together with miba_plan.hpp):
//   * the stable sort of the valid observations by (point in tile order, camera): rocPRIM radix sort of 64-bit keys
//     (point rank << 24 | camera) with the file index as value, the planner's PlanSorter;
//   * the measurement arrays in slot order (oxy) and the camera-major copy of the observations (cmo_pt, cmo_xy, per-camera
//     counts), gathered and sorted on the device from the caller's arrays uploaded once.
#pragma once

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>

namespace miba {

constexpr uint64_t kSortInvalid = ~0ull;

// key = rank of the observation's point in tile order << 24 | camera (n_cam < 2^24); filtered observations sort to the end
__global__ void k_sort_keys(int64_t n_obs, const int32_t* __restrict__ obs_cam, const int32_t* __restrict__ obs_pt,
                            const uint8_t* __restrict__ ok, const int32_t* __restrict__ sorted_of, uint64_t* __restrict__ key,
                            uint32_t* __restrict__ val) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_obs) return;
  key[i] = ok[i] ? (((uint64_t)(uint32_t)sorted_of[obs_pt[i]]) << 24) | (uint64_t)(uint32_t)obs_cam[i] : kSortInvalid;
  val[i] = (uint32_t)i;
}

// camera of every sorted key (its low 24 bits)
__global__ void k_key_cam(int64_t n, const uint64_t* __restrict__ key, int32_t* __restrict__ cam) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cam[i] = (int32_t)(key[i] & 0xFFFFFFu);
}

// measurements in slot order: oxy[chunk][2][256] from the caller's obs_xy through slot_orig; model 1 uploads o - observed
__global__ void k_gather_oxy(int64_t n_slots, const int64_t* __restrict__ slot_orig, const double2* __restrict__ obs_xy,
                             const int32_t* __restrict__ obs_cam, const double2* __restrict__ principal /*or null*/,
                             double* __restrict__ oxy) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  const int64_t o = slot_orig[s];
  double2 v = make_double2(0.0, 0.0);
  if (o >= 0) {
    v = obs_xy[o];
    if (principal) { const double2 c = principal[obs_cam[o]]; v.x = c.x - v.x; v.y = c.y - v.y; }
  }
  const int64_t chunk = s / 256, lane = s % 256;
  oxy[chunk * 512 + lane] = v.x;
  oxy[chunk * 512 + 256 + lane] = v.y;
}

// camera-major copy: key = camera << 32 | slot (ascending slot = ascending point inside a camera); pads sort to the end
__global__ void k_cmo_keys(int64_t n_slots, const int32_t* __restrict__ idx, const int32_t* __restrict__ ocam,
                           uint64_t* __restrict__ key, uint32_t* __restrict__ val, int32_t* __restrict__ cam_count) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  const bool valid = idx[s] >= 0;
  key[s] = valid ? (((uint64_t)(uint32_t)ocam[s]) << 32) | (uint64_t)(uint32_t)s : kSortInvalid;
  val[s] = (uint32_t)s;
  if (valid) atomicAdd(cam_count + ocam[s], 1);
}

// point of every slot (local layout-order id), from the point ranges
__global__ void k_slot_point(int32_t n_pt, const int2* __restrict__ pt_rng, int32_t* __restrict__ slot_pt) {
  const int32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_pt) return;
  const int2 r = pt_rng[q];
  for (int32_t s = r.x; s < r.x + r.y; ++s) slot_pt[s] = q;
}

__global__ void k_cmo_fill(int64_t n_obs_local, const uint32_t* __restrict__ sorted_slot, const int32_t* __restrict__ slot_pt,
                           const double* __restrict__ oxy, int32_t* __restrict__ cmo_pt, double2* __restrict__ cmo_xy) {
  const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= n_obs_local) return;
  const int64_t s = sorted_slot[d];
  cmo_pt[d] = slot_pt[s];
  const int64_t chunk = s / 256, lane = s % 256;
  cmo_xy[d] = make_double2(oxy[chunk * 512 + lane], oxy[chunk * 512 + 256 + lane]);
}

// stable sort of n (key, value) pairs on the stream; scratch allocated and freed here
inline hipError_t device_sort_pairs(uint64_t* key_in, uint64_t* key_out, uint32_t* val_in, uint32_t* val_out, size_t n, int end_bit,
                                    hipStream_t st) {
  size_t tmp_bytes = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, key_in, key_out, val_in, val_out, n, 0, end_bit, st);
  if (e != hipSuccess) return e;
  void* tmp = nullptr;
  e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1);
  if (e != hipSuccess) return e;
  e = rocprim::radix_sort_pairs(tmp, tmp_bytes, key_in, key_out, val_in, val_out, n, 0, end_bit, st);
  const hipError_t e2 = hipStreamSynchronize(st);
  (void)hipFree(tmp);
  return e != hipSuccess ? e : e2;
}

// ---- instance list of the atomic-free assembly of the Schur complement (k_dense_blocks, miba_dense.hpp) on the device ----
// Every pair of observations (a, b) of one point with camera(a) >= camera(b) is one instance of camera block
// (camera(a), camera(b)).  The instances are written out point by point as key = camera(a) << 24 | camera(b), value = point,
// and sorted by key (stable: ascending points inside a block, the order the host plan produces).

// instances per point (a point's slots hold ascending cameras); *dup is set if a point sees one camera twice (the plan then
// falls back to the host's, which counts such pairs from both sides)
__global__ void k_inst_count(int32_t n_pt, const int2* __restrict__ pt_rng, const int32_t* __restrict__ ocam, uint64_t* __restrict__ cnt,
                             int32_t* __restrict__ dup) {
  const int32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_pt) return;
  const int2 r = pt_rng[q];
  bool d = false;
  for (int32_t a = 1; a < r.y; ++a) d |= ocam[r.x + a] <= ocam[r.x + a - 1];
  if (d) *dup = 1;
  cnt[q] = (uint64_t)r.y * (uint64_t)(r.y + 1) / 2;
}
// points of up to 64 observations: one thread writes all pairs of its point
__global__ void k_inst_fill(int32_t n_pt, const int2* __restrict__ pt_rng, const int32_t* __restrict__ ocam, const uint64_t* __restrict__ off,
                            uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
  const int32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_pt) return;
  const int2 r = pt_rng[q];
  if (r.y > 64) return;
  uint64_t o = off[q];
  for (int32_t a = 0; a < r.y; ++a) {
    const uint64_t ca = (uint64_t)(uint32_t)ocam[r.x + a] << 24;
    for (int32_t b = 0; b <= a; ++b, ++o) { key[o] = ca | (uint64_t)(uint32_t)ocam[r.x + b]; val[o] = (uint32_t)q; }
  }
}
// longer tracks: one workgroup per point, a thread per observation a
__global__ void k_inst_fill_long(const int32_t* __restrict__ long_pts, const int2* __restrict__ pt_rng, const int32_t* __restrict__ ocam,
                                 const uint64_t* __restrict__ off, uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
  const int32_t q = long_pts[blockIdx.x];
  const int2 r = pt_rng[q];
  for (int32_t a = threadIdx.x; a < r.y; a += blockDim.x) {
    uint64_t o = off[q] + (uint64_t)a * (uint64_t)(a + 1) / 2;
    const uint64_t ca = (uint64_t)(uint32_t)ocam[r.x + a] << 24;
    for (int32_t b = 0; b <= a; ++b, ++o) { key[o] = ca | (uint64_t)(uint32_t)ocam[r.x + b]; val[o] = (uint32_t)q; }
  }
}
// first instance of every block of the sorted list
__global__ void k_inst_heads(int64_t n, const uint64_t* __restrict__ key, uint32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (i == 0 || key[i] != key[i - 1]) ? 1u : 0u;
}
__global__ void k_inst_blocks(int64_t n, const uint64_t* __restrict__ key, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ rank,
                              uint32_t* __restrict__ bstart, uint64_t* __restrict__ bkey) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flag[i]) { bstart[rank[i]] = (uint32_t)i; bkey[rank[i]] = key[i]; }
}
template <typename T>
inline hipError_t device_exclusive_scan(const T* in, T* out, size_t n, hipStream_t st) {
  size_t tmp_bytes = 0;
  hipError_t e = rocprim::exclusive_scan(nullptr, tmp_bytes, in, out, (T)0, n, rocprim::plus<T>(), st);
  if (e != hipSuccess) return e;
  void* tmp = nullptr;
  e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1);
  if (e != hipSuccess) return e;
  e = rocprim::exclusive_scan(tmp, tmp_bytes, in, out, (T)0, n, rocprim::plus<T>(), st);
  const hipError_t e2 = hipStreamSynchronize(st);
  (void)hipFree(tmp);
  return e != hipSuccess ? e : e2;
}

}  // namespace miba
