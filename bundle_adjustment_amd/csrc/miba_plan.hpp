// miba_plan.hpp — host-side repack of the problem into the device layout.
//
// Replaces the host loop of BundleAdjustment::createProblem
// (/root/reference/src/BundleAdjustment.cpp:44-86): the same observation filter (:49-60), but
// instead of one heap-allocated cost function per observation (BAConstraint::create,
// src/BAConstraint.cpp:82-86) the observations are grouped by point and cut into tiles (whole
// points, at most kMaxCamSlots distinct cameras); inside a tile the points are packed into 64-slot
// wavelets that never split a point, four wavelets to a 256-slot chunk.  Pure C++ (no HIP): used by
// miba_set_problem and miba_plan_shards.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>
#include <chrono>
#include <cstdio>
#include <array>
#include <thread>
#include <utility>
#include <vector>

namespace miba {

// fn(begin, end) over [0, n) on up to 16 host threads (MIBA_PLAN_THREADS overrides); ranges are disjoint
template <typename F>
inline void plan_parallel_for(int64_t n, F fn, int64_t min_parallel = (1 << 16)) {
  unsigned t = std::thread::hardware_concurrency();
  if (const char* e = getenv("MIBA_PLAN_THREADS")) t = (unsigned)atoi(e);
  if (t < 1) t = 1;
  if (t > 16) t = 16;
  if ((int64_t)t > n) t = (unsigned)std::max<int64_t>(1, n);
  if (n <= min_parallel || t == 1) { fn(0, n); return; }
  std::vector<std::thread> th;
  for (unsigned k = 1; k < t; ++k) th.emplace_back(fn, n * k / t, n * (k + 1) / t);
  fn(0, n / t);
  for (auto& x : th) x.join();
}

// std::vector whose resize() leaves new elements uninitialised: the planner's big arrays (one entry per observation) are
// filled by its worker threads, and a value-initialising resize would first fault in and zero every page on one thread
template <typename T>
struct default_init_alloc : std::allocator<T> {
  template <typename U> struct rebind { using other = default_init_alloc<U>; };
  using std::allocator<T>::allocator;
  template <typename U> void construct(U* p) noexcept { ::new ((void*)p) U; }
  template <typename U, typename... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
template <typename T> using raw_vector = std::vector<T, default_init_alloc<T>>;

// the same with the worker's index: fn(t, begin, end), t < returned count; range t precedes range t + 1
inline unsigned plan_threads(int64_t n, int64_t min_parallel = (1 << 16)) {
  unsigned t = std::thread::hardware_concurrency();
  if (const char* e = getenv("MIBA_PLAN_THREADS")) t = (unsigned)atoi(e);
  if (t < 1) t = 1;
  if (t > 16) t = 16;
  if (n <= min_parallel) t = 1;
  return t;
}
template <typename F>
inline void plan_parallel_ranges(int64_t n, unsigned t, F fn) {
  if (t <= 1) { fn(0u, (int64_t)0, n); return; }
  std::vector<std::thread> th;
  for (unsigned k = 1; k < t; ++k) th.emplace_back(fn, k, n * k / t, n * (k + 1) / t);
  fn(0u, (int64_t)0, n / t);
  for (auto& x : th) x.join();
}

// MIBA_DEBUG_PLAN=1: phase times of the host tiler on stderr
struct PlanLap {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  bool on = getenv("MIBA_DEBUG_PLAN") != nullptr;
  void operator()(const char* what) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[miba plan] %-28s %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
    t0 = t1;
  }
};


constexpr int kChunk = 256;        // observation slots per chunk = threads per block
constexpr int kWave = 64;          // a point never straddles a 64-slot wavelet (one wavefront)
constexpr int kMaxCamSlots = 64;   // distinct cameras per (non-giant) tile: LDS accumulator rows
constexpr int kSpanSeg = 1 << 15;     // points per independently tiled segment of the point sequence
constexpr int kMaxChunksPerTile = 8;   // default below 4 M observations (31 above); MIBA_TILE_CHUNKS overrides (1..64)
constexpr int kJComp = 26;         // doubles stored per observation: F 18, E 6, r 2
constexpr int kPtRec = 12;        // per-point record: X 0..2 | z = sp.(C^-1 g) 3..5 | W = diag(sp) C^-1 diag(sp) 6..11
constexpr int kCamTabStride = 48;  // per-camera table row (R 9, t 3, f k1 k2, pad, dR 27, pad)

struct Tile {   // 32 bytes
  int32_t grp0, ngrp;   // groups [grp0, grp0+ngrp)
  int32_t cam0, ncs;    // camera slots [cam0, cam0+ncs) in tile_cams / slab_row
  int32_t giant;        // 1: single point with more than kMaxCamSlots observations
  int32_t slot0;        // first observation slot of the tile
  int32_t nslot;        // slots (multiple of kChunk)
  int32_t pad;
};

struct Group {  // 16 bytes: a run of chunks that holds whole points
  int32_t chunk0, nchunk;  // nchunk > 1 only for a giant point
  int32_t pt0, npt;        // local (sorted) point ids [pt0, pt0+npt)
};

struct Plan {
  int32_t n_cam = 0, n_pt_in = 0;
  int64_t n_obs_in = 0, n_valid = 0;
  // sorted active points (all shards)
  std::vector<int32_t> pt_orig;          // sorted point -> caller's point id
  std::vector<uint8_t> cam_active;       // [n_cam]
  int32_t n_active_cam = 0;
  // global tiling (all shards)
  std::vector<Tile> tiles;
  std::vector<Group> groups;
  raw_vector<int32_t> idx;               // [n_slots] pl | cs<<8, -1 = pad
  raw_vector<int32_t> ocam;              // [n_slots] global camera id (0 for pads)
  raw_vector<int64_t> slot_orig;         // [n_slots] caller's observation index, -1 = pad
  std::vector<int32_t> pt_first;         // [n_pt] first slot of sorted point
  std::vector<int32_t> pt_count;         // [n_pt]
  std::vector<int32_t> tile_cams;        // per (tile, cam slot): global camera id
  std::vector<int32_t> wl_npt;           // per 64-slot wavelet: longest point run
  int64_t n_slots = 0;
  // sharding: rank r owns tiles [shard_tile[r], shard_tile[r+1])
  std::vector<int64_t> shard_tile;
};

// point_order: 0 = caller's order, 1 = sort by smallest camera index (camera locality)
// optional sorter of build_plan: fills obs_sorted[n_valid] with the indices of the valid observations ordered by
// (sorted_of[obs_pt], obs_cam), stable, and sorted_cam[n_valid] with their cameras in that order; returns 0, or non-zero to
// make build_plan use its host sort
struct PlanSorter {
  int (*fn)(void* ctx, int64_t n_obs, int32_t n_pt, const int32_t* obs_cam, const int32_t* obs_pt, const uint8_t* ok,
            const int32_t* sorted_of /*[n_pt]*/, int32_t n_act, int64_t n_valid, int64_t* obs_sorted, int32_t* sorted_cam);
  void* ctx;
};

inline int build_plan(Plan& P, int32_t n_cam, int32_t n_pt, int64_t n_obs, const int32_t* obs_cam,
                      const int32_t* obs_pt, const double* obs_xy, const double* X, int world_size,
                      int point_order, const PlanSorter* sorter = nullptr) {
  PlanLap lap;
  P = Plan();
  P.n_cam = n_cam; P.n_pt_in = n_pt; P.n_obs_in = n_obs;
  if (n_cam <= 0 || n_pt <= 0 || n_obs < 0 || world_size < 1) return -1;
  if (n_cam >= (1 << 24)) return -6;
  int max_chunks = 0;   // chosen after the filter, when the number of valid observations is known
  if (const char* e = getenv("MIBA_TILE_CHUNKS")) { const int v = atoi(e); if (v >= 1 && v <= 64) max_chunks = v; }
  // --- filter (src/BundleAdjustment.cpp:49-60) and count
  std::vector<int32_t> cnt(n_pt, 0);
  std::vector<int32_t> mincam(n_pt, INT32_MAX);
  std::vector<uint8_t> ok(n_obs ? n_obs : 1, 0);
  P.cam_active.assign(n_cam, 0);
  {
    // threaded over observation ranges; counts and minima per point through relaxed atomics (order-independent results)
    std::vector<int64_t> n_ok(17, 0);
    std::vector<int> bad(17, 0);
    unsigned slot = 0;
    std::mutex mu;
    plan_parallel_for(n_obs, [&](int64_t a, int64_t b) {
      unsigned me;
      { std::lock_guard<std::mutex> lk(mu); me = slot++; }
      int64_t nv = 0;
      for (int64_t i = a; i < b; ++i) {
        const int32_t p = obs_pt[i], c = obs_cam[i];
        if (p == -1) continue;
        if (p < 0 || p >= n_pt || c < 0 || c >= n_cam) { bad[me] = 1; return; }
        if (std::isnan(obs_xy[2 * i])) continue;
        if (X && std::isnan(X[3 * (size_t)p])) continue;
        ok[i] = 1;
        __atomic_fetch_add(&cnt[p], 1, __ATOMIC_RELAXED);
        int32_t cur = __atomic_load_n(&mincam[p], __ATOMIC_RELAXED);
        while (c < cur && !__atomic_compare_exchange_n(&mincam[p], &cur, c, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
        __atomic_store_n(&P.cam_active[c], (uint8_t)1, __ATOMIC_RELAXED);
        ++nv;
      }
      n_ok[me] = nv;
    });
    for (int v : bad) if (v) return -1;
    for (int64_t v : n_ok) P.n_valid += v;
  }
  lap("filter");
  // Tile size: 31 chunks once that still leaves a few thousand tiles.  (Rounds 1-2: 16 chunks, fewer tile prologues and less
  // padding than 8: 3.5 % on the LM iteration at Final-13682.  Round 3: the product kernel walks a tile as camera-major sub-tiles of
  // 256 four-observation segments; 31 chunks are eight of them with 98 % of the lanes in use, 16 chunks are 4.1 of them, i.e.
  // five with 81 %: 70.1 / 73.7 / 74.4 LM iterations per second with 15 / 23 / 31 chunks, tools/ab_bench_r3.sh.)  Below that the largest power of
  // two that leaves about 600 tiles: a small problem is latency-bound, a workgroup walks its chunks one after the other, and
  // 256 compute units want a tile each (trafalgar-257, 226 k observations: 37.8 us per PCG iteration with 8 chunks, 32.5-34 us
  // with 1-4; dubrovnik-356, 1.26 M: 49.6 us with 8, 55.9 with 2).
  if (max_chunks == 0) {
    if (P.n_valid >= ((int64_t)4 << 20)) max_chunks = 31;   // eight full camera-major sub-tiles of the product kernel (build_cm_layout)
    else {
      const int64_t want = P.n_valid / (kChunk * 600);
      max_chunks = 1;
      while (max_chunks < kMaxChunksPerTile && 2 * max_chunks <= want) max_chunks *= 2;
    }
  }
  P.n_active_cam = 0;
  for (int c = 0; c < n_cam; ++c) P.n_active_cam += P.cam_active[c];
  // --- order active points (threaded over point ranges: 20 M points at the largest configuration)
  unsigned nthr = plan_threads(n_pt);
  while (nthr > 1 && (size_t)nthr * (size_t)n_cam > ((size_t)1 << 26)) nthr /= 2;   // (the per-worker camera histograms)
  {
    // active points in the caller's order, or stably counting-sorted by smallest camera index: every worker counts its range
    // (per camera for the sort), the ranges' offsets follow from the counts, every worker writes its range
    const bool by_cam = point_order == 1;
    const size_t nbin = by_cam ? (size_t)n_cam : 1;
    std::vector<int64_t> hist((size_t)nthr * nbin, 0);
    plan_parallel_ranges(n_pt, nthr, [&](unsigned t, int64_t a, int64_t b) {
      int64_t* h = hist.data() + (size_t)t * nbin;
      for (int64_t p = a; p < b; ++p)
        if (cnt[p] > 0) h[by_cam ? mincam[p] : 0]++;
    });
    int64_t run = 0;
    for (size_t c = 0; c < nbin; ++c)
      for (unsigned t = 0; t < nthr; ++t) { const int64_t v = hist[(size_t)t * nbin + c]; hist[(size_t)t * nbin + c] = run; run += v; }
    P.pt_orig.resize((size_t)run);
    plan_parallel_ranges(n_pt, nthr, [&](unsigned t, int64_t a, int64_t b) {
      int64_t* h = hist.data() + (size_t)t * nbin;
      for (int64_t p = a; p < b; ++p)
        if (cnt[p] > 0) P.pt_orig[(size_t)h[by_cam ? mincam[p] : 0]++] = (int32_t)p;
    });
  }
  lap("order points");
  const int32_t n_act = (int32_t)P.pt_orig.size();
  raw_vector<int32_t> sorted_of((size_t)n_pt);
  plan_parallel_for(n_pt, [&](int64_t a, int64_t b) { for (int64_t p = a; p < b; ++p) sorted_of[p] = -1; });
  plan_parallel_for(n_act, [&](int64_t a, int64_t b) { for (int64_t q = a; q < b; ++q) sorted_of[P.pt_orig[q]] = (int32_t)q; });
  // --- bucket observations by sorted point (stable: file order), then sort each by camera
  raw_vector<int64_t> off((size_t)n_act + 1);
  {
    // prefix sum of the observation counts in sorted order, in two threaded passes
    const unsigned nt2 = plan_threads(n_act);
    std::vector<int64_t> part(nt2 + 1, 0);
    plan_parallel_ranges(n_act, nt2, [&](unsigned t, int64_t a, int64_t b) {
      int64_t sum = 0;
      for (int64_t q = a; q < b; ++q) sum += cnt[P.pt_orig[q]];
      part[t + 1] = sum;
    });
    for (unsigned t = 0; t < nt2; ++t) part[t + 1] += part[t];
    off[0] = 0;
    plan_parallel_ranges(n_act, nt2, [&](unsigned t, int64_t a, int64_t b) {
      int64_t run = part[t];
      for (int64_t q = a; q < b; ++q) { run += cnt[P.pt_orig[q]]; off[q + 1] = run; }
    });
  }
  // observations of every sorted point, ascending camera, file order among equal cameras: a stable sort of the valid
  // observations by (sorted point, camera).  On the GPU when the caller brings a sorter (miba_set_problem: rocPRIM radix
  // sort), else bucket + per-point sort on the host threads; both give the same permutation.
  raw_vector<int64_t> obs_sorted(P.n_valid ? P.n_valid : 1);
  raw_vector<int32_t> sorted_cam(P.n_valid ? P.n_valid : 1);
  bool sorted_on_device = false;
  if (sorter && sorter->fn && P.n_valid > 0)
    sorted_on_device = sorter->fn(sorter->ctx, n_obs, n_pt, obs_cam, obs_pt, ok.data(), sorted_of.data(), n_act, P.n_valid, obs_sorted.data(),
                                  sorted_cam.data()) == 0;
  if (!sorted_on_device) {
    std::vector<int64_t> cur(off.begin(), off.end() - 1);
    for (int64_t i = 0; i < n_obs; ++i)
      if (ok[i]) obs_sorted[cur[sorted_of[obs_pt[i]]]++] = i;
    plan_parallel_for(n_act, [&](int64_t s0, int64_t s1) {
      for (int64_t s = s0; s < s1; ++s)
        std::stable_sort(obs_sorted.begin() + off[s], obs_sorted.begin() + off[s + 1],
                         [&](int64_t a, int64_t b) { return obs_cam[a] < obs_cam[b]; });
    });
    plan_parallel_for(P.n_valid, [&](int64_t a, int64_t b) {
      for (int64_t j = a; j < b; ++j) sorted_cam[j] = obs_cam[obs_sorted[j]];
    });
  }
  lap("bucket + per-point sort");
  // --- tiling, phase 1: cut the point sequence into tiles.  A tile holds whole points, at most kMaxCamSlots
  // distinct cameras and about max_chunks chunks of observations; a giant point (more observations than a
  // wavefront) is a tile of its own.
  struct Span { int32_t s0, s1, giant; };
  std::vector<Span> spans;
  {
    // The point sequence is cut into segments of kSpanSeg points that are tiled independently (a tile never crosses a
    // segment boundary: under 2 % more tiles at Final-13682), so the greedy walk, which carries the open tile's camera set
    // from point to point, runs on worker threads; the result does not depend on their number.
    const int64_t cap = (int64_t)max_chunks * kChunk - (max_chunks > 1 ? kWave : 0);
    const int32_t n_seg = (n_act + kSpanSeg - 1) / kSpanSeg;
    std::vector<std::vector<Span>> seg_spans((size_t)n_seg);
    auto tile_segment = [&](int32_t sg, std::vector<int32_t>& stamp, int32_t& span_id) {
      std::vector<Span>& out = seg_spans[(size_t)sg];
      const int32_t s_begin = sg * kSpanSeg, s_stop = std::min(n_act, s_begin + kSpanSeg);
      int32_t open_s0 = -1, ncs = 0;
      int64_t used = 0;
      auto close = [&](int32_t s_end) {
        if (open_s0 >= 0) out.push_back(Span{open_s0, s_end, 0});
        open_s0 = -1;
      };
      for (int32_t s = s_begin; s < s_stop; ++s) {
        const int64_t a = off[s], b = off[s + 1];
        const int32_t L = (int32_t)(b - a);
        if (L > kWave) {
          close(s);
          out.push_back(Span{s, s + 1, 1});
          continue;
        }
        if (open_s0 >= 0) {
          int add = 0;
          int32_t prev = -1;
          for (int64_t j = a; j < b; ++j) {
            const int32_t c = sorted_cam[j];
            if (c != prev && stamp[c] != span_id) ++add;
            prev = c;
          }
          if (ncs + add > kMaxCamSlots || used + L > cap) close(s);
        }
        if (open_s0 < 0) { open_s0 = s; ++span_id; ncs = 0; used = 0; }
        for (int64_t j = a; j < b; ++j) {
          const int32_t c = sorted_cam[j];
          if (stamp[c] != span_id) { stamp[c] = span_id; ++ncs; }
        }
        used += L;
      }
      close(s_stop);
    };
    plan_parallel_for(n_seg, [&](int64_t g0, int64_t g1) {
      std::vector<int32_t> stamp(n_cam, -1);
      int32_t span_id = 0;
      for (int64_t sg = g0; sg < g1; ++sg) tile_segment((int32_t)sg, stamp, span_id);
    }, 1);
    size_t total = 0;
    for (auto& v : seg_spans) total += v.size();
    spans.reserve(total);
    for (auto& v : seg_spans) spans.insert(spans.end(), v.begin(), v.end());
  }
  lap("phase 1 (tile spans)");
  // --- phase 2: lay every tile out.  Inside a tile the point order is free, so its points are packed
  // first-fit-decreasing into 64-slot wavelets (a point never straddles a wavelet): little padding, and points
  // of similar length share a wavelet, which keeps the segmented scans of the kernels short.  Points are
  // renumbered in layout order.
  // Tiles are independent of each other, so contiguous ranges of tiles (balanced by observation count) are laid
  // out by worker threads into buffers of their own with range-relative offsets, then copied to their final
  // places (every tile is a whole number of chunks, so the ranges concatenate).
  struct Part {
    std::vector<Tile> tiles;
    std::vector<Group> groups;
    std::vector<int32_t> idx, ocam, tile_cams, new_orig, pt_first, pt_count;
    std::vector<int64_t> slot_orig;
    int64_t slots = 0;
  };
  auto layout_range = [&](size_t t_begin, size_t t_end, Part& Q) {
    std::vector<int32_t> stamp(n_cam, -1), slot_of(n_cam, 0);
    std::vector<int32_t> order, bin_free;
    std::vector<size_t> bin_order;
    std::vector<std::vector<int32_t>> bins;
    int64_t& slots = Q.slots;
    if (t_begin < t_end) {
      // capacity up front (observations of the range + padding): the arrays grow by push_back, without reallocation copies
      const int64_t n_obs_rng = off[spans[t_end - 1].s1] - off[spans[t_begin].s0];
      const size_t est = (size_t)(n_obs_rng + n_obs_rng / 16 + (int64_t)(t_end - t_begin) * kChunk + 4096);
      const size_t n_pts_rng = (size_t)(spans[t_end - 1].s1 - spans[t_begin].s0);
      Q.idx.reserve(est); Q.ocam.reserve(est); Q.slot_orig.reserve(est);
      Q.new_orig.reserve(n_pts_rng); Q.pt_first.reserve(n_pts_rng); Q.pt_count.reserve(n_pts_rng);
      Q.tiles.reserve(t_end - t_begin); Q.tile_cams.reserve((t_end - t_begin) * (size_t)kMaxCamSlots);
      Q.groups.reserve(est / kChunk + (t_end - t_begin));
    }
    auto pad_to = [&](int64_t mult) {
      while (slots % mult) { Q.idx.push_back(-1); Q.ocam.push_back(0); Q.slot_orig.push_back(-1); ++slots; }
    };
    for (size_t ti = t_begin; ti < t_end; ++ti) {
      const Span sp = spans[ti];
      Tile T{};
      T.grp0 = (int32_t)Q.groups.size();
      T.cam0 = (int32_t)Q.tile_cams.size();
      T.giant = sp.giant;
      T.slot0 = (int32_t)slots;
      if (sp.giant) {
        // one camera slot (= slab row) per observation, no LDS accumulation
        const int32_t s = sp.s0;
        const int64_t a = off[s], b = off[s + 1];
        const int32_t L = (int32_t)(b - a);
        const int32_t np = (int32_t)Q.new_orig.size();
        Group G{};
        G.chunk0 = (int32_t)(slots / kChunk);
        G.nchunk = (L + kChunk - 1) / kChunk;
        G.pt0 = np; G.npt = 1;
        Q.pt_first.push_back((int32_t)slots);
        Q.pt_count.push_back(L);
        Q.new_orig.push_back(P.pt_orig[s]);
        for (int64_t j = a; j < b; ++j) {
          const int64_t i = obs_sorted[j];
          Q.idx.push_back(0);
          Q.ocam.push_back(sorted_cam[j]);
          Q.slot_orig.push_back(i);
          Q.tile_cams.push_back(sorted_cam[j]);
          ++slots;
        }
        pad_to(kChunk);
        T.ncs = L;
        Q.groups.push_back(G);
        T.ngrp = 1;
      } else {
        const int32_t np_t = sp.s1 - sp.s0;
        order.resize(np_t);
        for (int32_t k = 0; k < np_t; ++k) order[k] = sp.s0 + k;
        std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return off[x + 1] - off[x] > off[y + 1] - off[y]; });
        bins.clear(); bin_free.clear();
        size_t first_open = 0;   // bins before it are full
        for (int32_t s : order) {
          const int32_t L = (int32_t)(off[s + 1] - off[s]);
          size_t bsel = bins.size();
          for (size_t bi = first_open; bi < bins.size(); ++bi)
            if (bin_free[bi] >= L) { bsel = bi; break; }
          if (bsel == bins.size()) { bins.emplace_back(); bin_free.push_back(kWave); }
          bins[bsel].push_back(s);
          bin_free[bsel] -= L;
          while (first_open < bins.size() && bin_free[first_open] == 0) ++first_open;
        }
        const int32_t tid = (int32_t)ti;
        // Wavelets of long points (the first bins) alternate with wavelets of short ones (the last bins): every stretch of the
        // tile then holds about the same number of points per observation, which is what the camera-major sub-tiles of the
        // product kernel want (at most kCmMaxPts points each; a stretch of two-observation points alone would hit that limit
        // at half the lanes).  The kernels that walk wavelets do not care about their order.
        bin_order.resize(bins.size());
        for (size_t k = 0, lo = 0, hi = bins.size(); k < bins.size(); ++k) bin_order[k] = (k & 1) ? --hi : lo++;
        for (size_t b0 = 0; b0 < bins.size(); b0 += 4) {
          Group G{};
          G.chunk0 = (int32_t)(slots / kChunk);
          G.nchunk = 1;
          G.pt0 = (int32_t)Q.new_orig.size();
          for (size_t bk = b0; bk < std::min(b0 + 4, bins.size()); ++bk) {
            const size_t bi = bin_order[bk];
            for (int32_t s : bins[bi]) {
              const int64_t a = off[s], b = off[s + 1];
              const int32_t L = (int32_t)(b - a);
              const int32_t np = (int32_t)Q.new_orig.size();
              const int32_t pl = np - G.pt0;
              Q.pt_first.push_back((int32_t)slots);
              Q.pt_count.push_back(L);
              Q.new_orig.push_back(P.pt_orig[s]);
              for (int64_t j = a; j < b; ++j) {
                const int64_t i = obs_sorted[j];
                const int32_t c = sorted_cam[j];   // (sequential; obs_cam[i] would be a random gather)
                if (stamp[c] != tid) {
                  stamp[c] = tid;
                  slot_of[c] = T.ncs++;
                  Q.tile_cams.push_back(c);
                }
                // point-in-chunk | camera-slot-in-tile << 8 | position in the point's run << 16 | run length << 23
                Q.idx.push_back(pl | (slot_of[c] << 8) | ((int32_t)(j - a) << 16) | (L << 23));
                Q.ocam.push_back(c);
                Q.slot_orig.push_back(i);
                ++slots;
              }
              G.npt++;
            }
            pad_to(kWave);
          }
          pad_to(kChunk);
          Q.groups.push_back(G);
          T.ngrp++;
        }
      }
      T.nslot = (int32_t)(slots - T.slot0);
      Q.tiles.push_back(T);
    }
  };
  // ranges of tiles with about equal observation counts
  unsigned n_part = std::thread::hardware_concurrency();
  if (const char* e = getenv("MIBA_PLAN_THREADS")) n_part = (unsigned)atoi(e);
  if (n_part < 1) n_part = 1;
  if (n_part > 16) n_part = 16;
  if (off[n_act] < (1 << 18) || spans.size() < 4 * (size_t)n_part) n_part = 1;
  std::vector<size_t> cut(n_part + 1, spans.size());
  cut[0] = 0;
  {
    size_t ti = 0;
    for (unsigned k = 1; k < n_part; ++k) {
      const int64_t target = off[n_act] * (int64_t)k / n_part;
      while (ti < spans.size() && off[spans[ti].s0] < target) ++ti;
      cut[k] = ti;
    }
  }
  std::vector<Part> parts(n_part);
  {
    std::vector<std::thread> th;
    for (unsigned k = 1; k < n_part; ++k) th.emplace_back([&, k] { layout_range(cut[k], cut[k + 1], parts[k]); });
    layout_range(cut[0], cut[1], parts[0]);
    for (auto& x : th) x.join();
  }
  // offsets of every range in the concatenated arrays
  std::vector<int64_t> o_slot(n_part + 1, 0), o_grp(n_part + 1, 0), o_cam(n_part + 1, 0), o_pt(n_part + 1, 0), o_tile(n_part + 1, 0);
  for (unsigned k = 0; k < n_part; ++k) {
    o_slot[k + 1] = o_slot[k] + parts[k].slots;
    o_grp[k + 1] = o_grp[k] + (int64_t)parts[k].groups.size();
    o_cam[k + 1] = o_cam[k] + (int64_t)parts[k].tile_cams.size();
    o_pt[k + 1] = o_pt[k] + (int64_t)parts[k].new_orig.size();
    o_tile[k + 1] = o_tile[k] + (int64_t)parts[k].tiles.size();
  }
  const int64_t slots = o_slot[n_part];
  if (slots >= (int64_t)INT32_MAX - kChunk || o_cam[n_part] >= INT32_MAX) return -6;   // slot and row indices are int32 on the device
  std::vector<int32_t> new_orig((size_t)o_pt[n_part]);   // layout-order point -> caller's point id
  P.pt_first.assign((size_t)o_pt[n_part], 0);
  P.pt_count.assign((size_t)o_pt[n_part], 0);
  P.idx.resize((size_t)slots); P.ocam.resize((size_t)slots); P.slot_orig.resize((size_t)slots);
  P.tile_cams.resize((size_t)o_cam[n_part]);
  P.groups.resize((size_t)o_grp[n_part]);
  P.tiles.resize((size_t)o_tile[n_part]);
  {
    auto place = [&](unsigned k) {
      Part& Q = parts[k];
      std::copy(Q.idx.begin(), Q.idx.end(), P.idx.begin() + o_slot[k]);
      std::copy(Q.ocam.begin(), Q.ocam.end(), P.ocam.begin() + o_slot[k]);
      std::copy(Q.slot_orig.begin(), Q.slot_orig.end(), P.slot_orig.begin() + o_slot[k]);
      std::copy(Q.tile_cams.begin(), Q.tile_cams.end(), P.tile_cams.begin() + o_cam[k]);
      std::copy(Q.new_orig.begin(), Q.new_orig.end(), new_orig.begin() + o_pt[k]);
      std::copy(Q.pt_count.begin(), Q.pt_count.end(), P.pt_count.begin() + o_pt[k]);
      for (size_t i = 0; i < Q.pt_first.size(); ++i) P.pt_first[(size_t)o_pt[k] + i] = Q.pt_first[i] + (int32_t)o_slot[k];
      for (size_t i = 0; i < Q.groups.size(); ++i) {
        Group G = Q.groups[i];
        G.chunk0 += (int32_t)(o_slot[k] / kChunk);
        G.pt0 += (int32_t)o_pt[k];
        P.groups[(size_t)o_grp[k] + i] = G;
      }
      for (size_t i = 0; i < Q.tiles.size(); ++i) {
        Tile T = Q.tiles[i];
        T.grp0 += (int32_t)o_grp[k];
        T.cam0 += (int32_t)o_cam[k];
        T.slot0 += (int32_t)o_slot[k];
        P.tiles[(size_t)o_tile[k] + i] = T;
      }
      Q = Part();
    };
    std::vector<std::thread> th;
    for (unsigned k = 1; k < n_part; ++k) th.emplace_back(place, k);
    place(0);
    for (auto& x : th) x.join();
  }
  lap("phase 2 (layout)");
  if (slots >= (int64_t)INT32_MAX - kChunk) return -6;   // slot indices are int32 on the device
  P.pt_orig.swap(new_orig);
  P.n_slots = slots;
  P.wl_npt.assign((size_t)(slots / kWave) + 1, 0);
  for (int32_t s = 0; s < n_act; ++s) {
    const size_t wq = (size_t)(P.pt_first[s] / kWave);
    P.wl_npt[wq] = std::max(P.wl_npt[wq], P.pt_count[s]);   // longest run in the wavelet
  }
  // --- shard tiles by cumulative slot count
  P.shard_tile.assign(world_size + 1, 0);
  const int64_t nt = (int64_t)P.tiles.size();
  {
    int64_t t = 0, acc = 0;
    for (int r = 0; r < world_size; ++r) {
      P.shard_tile[r] = t;
      const int64_t target = (slots * (int64_t)(r + 1)) / world_size;
      while (t < nt && acc + P.tiles[t].nslot / 2 < target) { acc += P.tiles[t].nslot; ++t; }
    }
    P.shard_tile[world_size] = nt;
  }
  lap("wavelets + shards");
  return 0;
}


// ---- elimination order of the tile graph of the exact solve (sparse_symbolic in miba.hip; host only) ----------------------------
// adj: nT rows of nW = ceil(nT / 64) words, bit u of row v set when tile rows u and v share an entry (no diagonal); destroyed.
// Out: the order (grp[k] = tile row eliminated at step k, pos = its inverse), per step the rows below the diagonal after fill
// (col_nodes, as tile rows), their total n_off, the number of arcs the order was dissected into, the first step of every arc
// (arc_start[n_arcs] = first separator step, nT when there is none), and the GROUPS of the order: group g holds the steps
// [group_start[g], group_start[g + 1]) and has level group_level[g]; the arcs are the groups of level 0, the separator sets follow
// by level.  Groups of one level share no tile and every column of a group holds rows of its own group and of higher levels only, so
// the groups of a level are factored side by side.  false: more than max_tiles tiles.
constexpr int kMaxArcs = 64;   // arcs of the dissected order
struct TileOrder {
  int n_arcs = 1;
  std::vector<int32_t> grp, pos, arc_start, group_start, group_level;
  std::vector<std::vector<int32_t>> col_nodes;
  int64_t n_off = 0;
};
inline bool order_tile_graph(std::vector<uint64_t>& adj, const int nT, const int n_arcs_wanted, const int64_t max_tiles, const int log_level,
                             TileOrder& out, const bool nested = true) {
  const size_t nW = ((size_t)nT + 63) / 64;
  int n_arcs = n_arcs_wanted;
  // Dissection of the tile graph into arcs that can be factored side by side (the elimination order of a band is a chain: at
  // Final-13682 1,955 tile columns, each waiting for the one before it).  The tile rows are cut by index into n_arcs equal ranges; an
  // edge between two ranges puts its higher end into a separator (rows ascending: a valid cover of the crossing edges; on a band these
  // are the first rows behind every cut, on a ring also the last rows, which close it).  Interiors of different arcs then share no
  // edge, and eliminating an interior row creates fill only inside its arc and towards the separators.
  // NESTED: the cuts form a binary tree over the arcs (the root cut halves the range, its children halve the halves, ...).  An edge
  // between arcs a < b crosses the cut that is their lowest common ancestor; its higher end joins THAT cut's separator set unless the
  // lower end already sits in a set of that height or above (which covers the edge).  A set's level is the height of its cut (1 for a
  // cut between two arcs).  Two sets of one height lie in disjoint subtrees, and every edge that leaves a subtree has its higher end in
  // a set above the subtree's root: after the levels below are eliminated, sets of one level share no edge, so the sets of a level are
  // factored side by side like the arcs, and the separator's chain is (levels) x (rows of a set) steps instead of all its rows.
  // Not nested: every separator row in ONE set of level 1 (the first form of round 3).
  // Only camera numberings with locality (what BAL files and the stand-in have) dissect this way; if the separators take more than a
  // third of the rows the order stays plain minimum degree, one chain.
  if (nT < 8 * n_arcs) n_arcs = 1;
  if (n_arcs > kMaxArcs) n_arcs = kMaxArcs;
  std::vector<int32_t> lvl(nT, 0), gid(nT, 0);   // level and group of a tile row: arcs are groups 0 .. n_arcs - 1 (level 0), cut c is group n_arcs + c
  std::vector<int32_t> cut_height;               // per cut
  std::vector<uint64_t> adj_keep;
  if (n_arcs > 1) {
    if (nested) adj_keep = adj;   // (for the fall-back below)
    // cuts of the binary tree, and the cut that separates arcs a < b
    std::vector<int32_t> lca((size_t)n_arcs * n_arcs, -1);
    struct Node { int lo, hi; };
    std::vector<Node> stack{{0, n_arcs}};
    std::vector<std::array<int, 3>> cuts;   // lo, mid, hi
    while (!stack.empty()) {
      const Node nd = stack.back(); stack.pop_back();
      if (nd.hi - nd.lo < 2) continue;
      const int mid = nd.lo + (nd.hi - nd.lo + 1) / 2;
      cuts.push_back({nd.lo, mid, nd.hi});
      stack.push_back({nd.lo, mid}); stack.push_back({mid, nd.hi});
    }
    cut_height.assign(cuts.size(), 1);
    // height = 1 + the larger height of the two halves' own cuts (cuts of [lo, hi) with fewer than two arcs do not exist: height 0)
    auto height_of = [&](auto&& self, int lo, int hi) -> int {
      if (hi - lo < 2) return 0;
      const int mid = lo + (hi - lo + 1) / 2;
      return 1 + std::max(self(self, lo, mid), self(self, mid, hi));
    };
    for (size_t c = 0; c < cuts.size(); ++c) {
      cut_height[c] = nested ? height_of(height_of, cuts[c][0], cuts[c][2]) : 1;
      for (int a2 = cuts[c][0]; a2 < cuts[c][1]; ++a2)
        for (int b2 = cuts[c][1]; b2 < cuts[c][2]; ++b2) lca[(size_t)a2 * n_arcs + b2] = (int32_t)c;
    }
    auto arc_of = [&](int v) { return (int)std::min<int64_t>(n_arcs - 1, (int64_t)v * n_arcs / nT); };
    int n_sep = 0;
    for (int v = 0; v < nT; ++v) {
      const int av = arc_of(v);
      const uint64_t* row = &adj[(size_t)v * nW];
      int best = 0, best_cut = -1;
      bool past = false;
      for (size_t w = 0; w < nW && !past; ++w)
        for (uint64_t m = row[w]; m; m &= m - 1) {
          const int u = (int)(64 * w + __builtin_ctzll(m));
          if (u >= v) { past = true; break; }
          const int au = arc_of(u);
          if (au == av) continue;
          const int c = lca[(size_t)std::min(au, av) * n_arcs + std::max(au, av)];
          const int hgt = cut_height[c];
          if (lvl[u] >= hgt) continue;   // the lower end covers this edge
          if (hgt > best) { best = hgt; best_cut = c; }
        }
      lvl[v] = best;
      gid[v] = best ? n_arcs + (nested ? best_cut : 0) : av;
      n_sep += best > 0;
    }
    if (3 * n_sep > nT) {
      if (log_level > 0) fprintf(stderr, "[miba] tile graph: %d of %d tile rows in the separators of %d arcs: not dissected\n", n_sep, nT, n_arcs);
      n_arcs = 1;
      std::fill(lvl.begin(), lvl.end(), 0);
      std::fill(gid.begin(), gid.end(), 0);
      std::vector<uint64_t>().swap(adj_keep);
    }
  }
  int max_level = 0;
  for (int v = 0; v < nT; ++v) max_level = std::max(max_level, (int)lvl[v]);
  std::vector<int32_t> left(max_level + 1, 0);
  for (int v = 0; v < nT; ++v) ++left[lvl[v]];

  // minimum degree on the tile graph (ties: the lowest tile row, which keeps a banded problem banded), level by level
  auto popcnt_row = [&](int v) { int c = 0; for (size_t w = 0; w < nW; ++w) c += __builtin_popcountll(adj[(size_t)v * nW + w]); return c; };
  std::vector<int32_t> deg(nT), pos(nT, -1), grp(nT);
  for (int v = 0; v < nT; ++v) deg[v] = popcnt_row(v);
  std::vector<std::vector<int32_t>> col_nodes(nT);
  int64_t n_off = 0;
  int cur = 0;
  bool ok_groups = true;
  for (int k = 0; k < nT; ++k) {
    while (left[cur] == 0) ++cur;
    int v = -1;
    for (int u = 0; u < nT; ++u)
      if (pos[u] < 0 && lvl[u] == cur && (v < 0 || deg[u] < deg[v])) v = u;
    pos[v] = k;
    grp[k] = v;
    --left[cur];
    std::vector<int32_t>& nb = col_nodes[k];
    const uint64_t* av = &adj[(size_t)v * nW];
    for (size_t w = 0; w < nW; ++w)
      for (uint64_t m = av[w]; m; m &= m - 1) nb.push_back((int32_t)(64 * w + __builtin_ctzll(m)));
    n_off += (int64_t)nb.size();
    if (2 * (int64_t)nT + n_off > max_tiles) return false;   // does not fit
    for (int32_t u : nb) {
      if (!(gid[u] == gid[v] || lvl[u] > lvl[v])) ok_groups = false;   // (cannot happen by construction: checked, not assumed)
      uint64_t* au = &adj[(size_t)u * nW];
      for (size_t w = 0; w < nW; ++w) au[w] |= av[w];
      au[v >> 6] &= ~((uint64_t)1 << (v & 63));
      au[u >> 6] &= ~((uint64_t)1 << (u & 63));
      deg[u] = popcnt_row(u);
    }
  }
  std::vector<uint64_t>().swap(adj);
  if (!ok_groups) {
    if (log_level > 0) fprintf(stderr, "[miba] tile graph: the groups of %d arcs are not independent: %s\n", n_arcs, nested ? "one separator set" : "not dissected");
    if (nested && !adj_keep.empty()) return order_tile_graph(adj_keep, nT, n_arcs_wanted, max_tiles, log_level, out, false);
    return false;
  }
  // the steps of one group side by side in the order (a stable regrouping: the groups of a level do not interact, so the fill is the
  // same): the arcs, then the separator sets level by level
  std::vector<int32_t> arc_start(n_arcs + 1, nT), group_start, group_level;
  if (n_arcs > 1) {
    const int n_gid = n_arcs + (int)std::max<size_t>(1, cut_height.size());
    std::vector<int32_t> g_level(n_gid, -1);
    for (int v = 0; v < nT; ++v) g_level[gid[v]] = lvl[v];
    for (int a2 = 0; a2 < n_arcs; ++a2) g_level[a2] = 0;   // (an arc without interior rows keeps its entry in arc_start)
    std::vector<int32_t> g_order;
    for (int l2 = 0; l2 <= max_level; ++l2)
      for (int g = 0; g < n_gid; ++g)
        if (g_level[g] == l2) g_order.push_back(g);
    std::vector<std::vector<int32_t>> steps_of(n_gid);
    for (int k = 0; k < nT; ++k) steps_of[gid[grp[k]]].push_back(k);
    std::vector<int32_t> grp2(nT);
    std::vector<std::vector<int32_t>> nodes2(nT);
    int at = 0;
    for (int g : g_order) {
      if (g < n_arcs) arc_start[g] = at;
      else if (arc_start[n_arcs] == nT && at < nT) arc_start[n_arcs] = at;
      if (!steps_of[g].empty() || g < n_arcs) { group_start.push_back(at); group_level.push_back(g_level[g]); }
      for (int32_t k : steps_of[g]) { grp2[at] = grp[k]; nodes2[at].swap(col_nodes[k]); ++at; }
    }
    group_start.push_back(nT);
    if (arc_start[n_arcs] == nT && max_level == 0) arc_start[n_arcs] = nT;
    grp.swap(grp2);
    col_nodes.swap(nodes2);
    for (int k = 0; k < nT; ++k) pos[grp[k]] = k;
  } else {
    arc_start[0] = 0;
    group_start = {0, nT};
    group_level = {0};
  }
  out.n_arcs = n_arcs;
  out.grp.swap(grp);
  out.pos.swap(pos);
  out.col_nodes.swap(col_nodes);
  out.arc_start.swap(arc_start);
  out.group_start.swap(group_start);
  out.group_level.swap(group_level);
  out.n_off = n_off;
  return true;
}

// The forward steps of an order: the groups of a level in lock-step (step st of the level takes column group_start[g] + st of every
// group g of the level that has one), level after level.  first_of_level[i]: step i is the first of a level above 0 (the private
// tiles of the levels below are merged in front of it).
inline void tile_order_steps(const TileOrder& O, std::vector<std::vector<int32_t>>& steps, std::vector<uint8_t>& first_of_level) {
  steps.clear(); first_of_level.clear();
  const int nG = (int)O.group_level.size();
  int max_level = 0;
  for (int g = 0; g < nG; ++g) max_level = std::max(max_level, (int)O.group_level[g]);
  for (int l2 = 0; l2 <= max_level; ++l2) {
    int longest = 0;
    for (int g = 0; g < nG; ++g)
      if (O.group_level[g] == l2) longest = std::max(longest, (int)(O.group_start[g + 1] - O.group_start[g]));
    for (int st = 0; st < longest; ++st) {
      std::vector<int32_t> ks;
      for (int g = 0; g < nG; ++g)
        if (O.group_level[g] == l2 && O.group_start[g] + st < O.group_start[g + 1]) ks.push_back(O.group_start[g] + st);
      steps.push_back(ks);
      first_of_level.push_back(st == 0 && l2 > 0);
    }
  }
}

// What the numeric phase pays for an order, in microseconds (sparse_symbolic chooses the number of arcs by it).  A forward step is two
// dependent launches for one column of every group of a level: the column launch (about 5 us: the diagonal tiles come factored) and
// the update launch, whose longest path is the workgroup that also factors a diagonal tile of the next step (about 27 us however little
// work the launch holds; measured at Final-13682, whose columns hold a dozen tiles), and past that the update's tile products at what
// the memory system sustains on them (three tiles in, one out per product: 0.02 us each); a backward step is one launch, about 6 us.
inline double tile_order_cost(const TileOrder& O) {
  auto pairs = [&](int k) { const double m = (double)O.col_nodes[k].size() + 1.0; return 0.5 * m * (m + 1.0); };
  const double t_step = 32.0, t_pair = 0.02, t_back = 6.0;
  std::vector<std::vector<int32_t>> steps;
  std::vector<uint8_t> first;
  tile_order_steps(O, steps, first);
  double us = 0.0;
  for (const auto& ks : steps) {
    double w = 0.0;
    for (int32_t k : ks) w += pairs(k);
    us += std::max(t_step, t_pair * w) + t_back;
  }
  return us;
}


// ---- camera-major sub-tile layout of the product kernel (K3, miba_mf2.hpp) ------------------------------------------
// The product's per-observation work wants two things at once: the camera record (24 doubles) and a place to add nine
// camera-side sums, and the point (3 doubles) and a place to add three point-side sums.  The point-major wavelets of the
// other kernels pay for the camera side through LDS (13 reads and 9 atomic adds per observation).  Here a tile's points are
// cut into SUB-TILES of at most kCmLanes segments; a segment is kCmSeg observations of ONE camera, owned by one lane: the lane
// reads the camera record once per segment and keeps the nine sums in registers; the point side goes through LDS (3 atomic
// adds per observation), with a workgroup barrier between the two halves of the product.
//   idx [sub-tile][lane] : four 16-bit entries, step j in bits 16 j .. 16 j + 15: LDS slot of the point (bits 0-7; slot[sub-tile][k]
//                          is the slot of the sub-tile's k-th point) | camera slot
//                          in tile (8-13) | valid (15).  Padding entries repeat the segment's first (point, camera) pair with
//                          valid = 0, so that their geometry is that of a real observation; a lane without a segment holds 0.
//   segment s of a sub-tile (cameras ascending) sits in lane (s & 3) * 64 + (s >> 2): the segments of one camera are spread over
//   the four waves.  Bits 8-14 of entry 1 hold the lane's camera ACCUMULATOR ROW instead of the camera slot: an atomic add
//   costs 8 LDS cycles per lane on the same address, so a camera with m segments in one wave of some sub-tile of the tile gets
//   min(8, m rounded up to a power of two) rows (fewer when the tile's kCmMaxRows rows would not suffice), used in turn by its
//   lanes; the flush at the end of the tile adds a camera's rows up (rowinfo).
constexpr int kCmSeg = 4;
constexpr int kCmLanes = 256;
constexpr int kCmMaxPts = 256;
constexpr int kCmMaxRows = 128;   // camera accumulator rows of a tile in LDS (9 doubles each)
struct CmTile { int32_t sub0, nsub, nrows, pt0, npt0, pad0, pad1, pad2; };   // per local tile (nsub = 0: giant); nrows: camera accumulator
// rows in LDS; pt0, npt0: the point range of the first sub-tile (the workgroup requests those points with the tile's first loads)
struct CmSub { int32_t pt0, npt, nseg, npt_next; };    // local point range, segments in use, points of the tile's next sub-tile (0: last)
struct CmLayout {
  std::vector<CmTile> tiles;
  std::vector<CmSub> subs;
  raw_vector<uint64_t> idx;                       // [subs.size()][kCmLanes]
  raw_vector<uint8_t> slot;                       // [subs.size()][kCmLanes] LDS slot of the sub-tile's k-th point
  raw_vector<int32_t> src;                        // [subs.size()][kCmLanes][kCmSeg] local observation slot of every entry, -1: padding
                                                  // (what the back-substitution's measurements are gathered through)
  std::vector<int32_t> rowinfo;                   // per (tile, camera slot), parallel to tile_cams: first accumulator row | rows << 8
  int64_t n_entries = 0, n_valid = 0;             // lane-steps laid out, and those that hold an observation
};

// tiles / grps: the shard's (local numbering); idx: the shard's slot words (point-in-chunk | camera slot << 8 | ...);
// pt_first / pt_count: per local point, first local slot and number of observations
inline void build_cm_layout(CmLayout& L, const std::vector<Tile>& tiles, const std::vector<Group>& grps, const int32_t* idx,
                            const int32_t* pt_first_local, const int32_t* pt_count_local) {
  L = CmLayout();
  const size_t nt = tiles.size();
  L.tiles.assign(nt, CmTile{0, 0, 0, 0, 0, 0, 0, 0});
  {
    size_t n_rows = 0;
    for (const Tile& T : tiles) n_rows = std::max(n_rows, (size_t)T.cam0 + (size_t)T.ncs);
    L.rowinfo.assign(n_rows ? n_rows : 1, 0);
  }
  if (!nt) return;
  struct Part { std::vector<CmSub> subs; std::vector<uint64_t> idx; std::vector<uint8_t> slot; std::vector<int32_t> src, nsub, nrows; int64_t n_valid = 0; };
  const unsigned nthr = std::max(1u, std::min(plan_threads((int64_t)nt, 64), (unsigned)nt));
  std::vector<Part> parts(nthr);
  plan_parallel_ranges((int64_t)nt, nthr, [&](unsigned th, int64_t ta, int64_t tb) {
    Part& Q = parts[th];
    Q.nsub.assign((size_t)(tb - ta), 0);
    Q.nrows.assign((size_t)(tb - ta), 0);
    int cnt[kMaxCamSlots], seg_off[kMaxCamSlots + 1];
    int next_t[kMaxCamSlots][16];
    int rd_cnt[32][32], at_cnt[64][16];
    const bool no_color = getenv("MIBA_CM_COLOR") && getenv("MIBA_CM_COLOR")[0] == '0';   // (diagnostic: slots in point order)
    std::vector<int32_t> place, porder, poff;
    for (int64_t ti = ta; ti < tb; ++ti) {
      const Tile& T = tiles[(size_t)ti];
      if (T.giant || T.ngrp == 0) continue;
      const int32_t p_begin = grps[(size_t)T.grp0].pt0;
      const Group& Gl = grps[(size_t)(T.grp0 + T.ngrp - 1)];
      const int32_t p_end = Gl.pt0 + Gl.npt;
      int32_t p = p_begin;
      while (p < p_end) {
        // greedy cut: whole points while the segments fit the lanes and the points fit their 8-bit ids
        for (int c = 0; c < kMaxCamSlots; ++c) cnt[c] = 0;
        int segs = 0;
        int32_t q = p;
        for (; q < p_end && q - p < kCmMaxPts; ++q) {
          const int32_t s0 = pt_first_local[q], n = pt_count_local[q];
          int add = 0;
          for (int32_t s = s0; s < s0 + n; ++s) {
            const int cs = (idx[s] >> 8) & 0xFF;
            if (cnt[cs] % kCmSeg == 0) ++add;
            ++cnt[cs];
          }
          if (segs + add > kCmLanes) {
            for (int32_t s = s0; s < s0 + n; ++s) --cnt[(idx[s] >> 8) & 0xFF];
            break;
          }
          segs += add;
        }
        // (a point has at most kWave observations here, so q > p)
        seg_off[0] = 0;
        for (int c = 0; c < kMaxCamSlots; ++c) seg_off[c + 1] = seg_off[c] + (cnt[c] + kCmSeg - 1) / kCmSeg;
        const size_t base = Q.idx.size();
        Q.idx.resize(base + kCmLanes, 0);
        uint64_t* out = Q.idx.data() + base;
        Q.src.resize((base + kCmLanes) * kCmSeg, -1);
        int32_t* src_out = Q.src.data() + base * kCmSeg;
        auto lane_of = [](int seg) { return (seg & 3) * 64 + (seg >> 2); };
        // Position of an observation in its camera's row.  The three point-side atomic adds of a wave instruction cost
        // 8 LDS cycles times the largest number of lanes on ONE address (tools/micro/lds_atomic_bench.hip), so the
        // observations of a point should sit in different (wave, step) classes: position pos of camera c is executed by
        // wave (seg_off[c] + pos / 4) & 3 at step pos & 3, class (4 seg_off[c] + pos) & 15.  Every observation takes the next
        // free position of a class its point has not used yet (the classes of a row are filled evenly, so nothing is padded
        // beyond the row's last segment); with more than sixteen observations the least used class.
        for (int c = 0; c < kMaxCamSlots; ++c)
          for (int k = 0; k < 16; ++k) next_t[c][k] = 0;
        // (lane | step << 8 | camera slot << 16) of every observation, points in order; the longest points choose first (they
        // need the most distinct classes)
        place.resize((size_t)(seg_off[kMaxCamSlots] * kCmSeg));
        porder.resize((size_t)(q - p));
        poff.resize((size_t)(q - p) + 1);
        poff[0] = 0;
        for (int32_t r = p; r < q; ++r) { porder[(size_t)(r - p)] = r; poff[(size_t)(r - p) + 1] = poff[(size_t)(r - p)] + pt_count_local[r]; }
        std::stable_sort(porder.begin(), porder.end(), [&](int32_t x, int32_t y) { return pt_count_local[x] > pt_count_local[y]; });
        for (const int32_t r : porder) {
          const int32_t s0 = pt_first_local[r], n = pt_count_local[r];
          int used[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          for (int32_t s = s0; s < s0 + n; ++s) {
            const int cs = (idx[s] >> 8) & 0xFF;
            const int shift = (4 * seg_off[cs]) & 15;
            int best = -1, best_used = 1 << 30;
            for (int kk = 0; kk < 16; ++kk) {
              const int k = (kk + (int)(r - p) + (int)(s - s0)) & 15;
              const int pos = ((k - shift) & 15) + 16 * next_t[cs][k];
              if (pos >= cnt[cs]) continue;
              if (used[k] < best_used) { best = k; best_used = used[k]; if (!best_used) break; }
            }
            const int pos = ((best - shift) & 15) + 16 * next_t[cs][best]++;
            ++used[best];
            place[(size_t)poff[(size_t)(r - p)] + (size_t)(s - s0)] = lane_of(seg_off[cs] + pos / kCmSeg) | ((pos % kCmSeg) << 8) | (cs << 16);
            ++Q.n_valid;
          }
        }
        // LDS slot of every point.  The kernel keeps the sub-tile's points as six arrays of 256 doubles (X, and the three sums):
        // slot s lies in bank pair s mod 32 of all of them.  A wave's 8-byte reads are served 32 lanes at a time, its atomic
        // adds 16 lanes at a time, one cycle more for every further slot of a bank pair already in use: every point takes a slot
        // whose bank pair the points sharing a lane group and step with its observations use least.
        {
          static_assert(kCmMaxPts == 256, "slot colouring assumes 256 slots, eight per bank pair");
          memset(rd_cnt, 0, sizeof(rd_cnt));
          memset(at_cnt, 0, sizeof(at_cnt));
          int free_per_color[32];
          for (int c = 0; c < 32; ++c) free_per_color[c] = 0;
          const size_t sbase = Q.slot.size();
          Q.slot.resize(sbase + kCmLanes, 0);
          for (const int32_t r : porder) {
            const int32_t n = pt_count_local[r];
            const size_t o = (size_t)poff[(size_t)(r - p)];
            // cost of every bank pair at once (observation by observation: the inner loops run over contiguous counters and
            // vectorise), then the first cheapest free one in the point's rotated order
            int cost[32];
            for (int c = 0; c < 32; ++c) cost[c] = 0;
            for (int32_t i = 0; i < n; ++i) {
              const int pl = place[o + i], lane = pl & 0xFF, step = (pl >> 8) & 3;
              const int* rd = rd_cnt[(lane >> 5) * 4 + step];
              const int* at = at_cnt[(lane >> 4) * 4 + step];
              for (int c = 0; c < 16; ++c) { cost[c] += 9 * rd[c] + 6 * at[c]; cost[16 + c] += 9 * rd[16 + c] + 6 * at[c]; }
            }
            int best = -1, best_cost = 1 << 30;
            for (int cc = 0; cc < 32; ++cc) {
              const int c = (cc + (int)(r - p)) & 31;
              if (free_per_color[c] >= 8) continue;
              if (cost[c] < best_cost) { best = c; best_cost = cost[c]; if (!best_cost) break; }
            }
            if (no_color) best = (int)(r - p) & 31;
            const int slot = best + 32 * free_per_color[best]++;
            Q.slot[sbase + (size_t)(r - p)] = (uint8_t)slot;
            for (int32_t i = 0; i < n; ++i) {
              const int pl = place[o + i], lane = pl & 0xFF, step = (pl >> 8) & 3, cs = pl >> 16;
              ++rd_cnt[(lane >> 5) * 4 + step][best];
              ++at_cnt[(lane >> 4) * 4 + step][best & 15];
              const uint64_t e = (uint64_t)slot | ((uint64_t)cs << 8) | 0x8000u;
              out[lane] |= e << (16 * step);
              src_out[lane * kCmSeg + step] = pt_first_local[r] + i;
            }
          }
        }
        for (int c = 0; c < kMaxCamSlots; ++c) {
          const int rem = cnt[c] % kCmSeg;
          if (!rem) continue;
          const int lane = lane_of(seg_off[c + 1] - 1);
          const uint64_t e0 = out[lane] & 0x7FFFu;   // the segment's first pair, not valid
          for (int step = rem; step < kCmSeg; ++step) out[lane] |= e0 << (16 * step);
        }
        Q.subs.push_back(CmSub{p, q - p, seg_off[kMaxCamSlots], 0});
        ++Q.nsub[(size_t)(ti - ta)];
        p = q;
      }
      // accumulator rows of the tile's cameras: from the largest number of segments a camera has in one wave of one sub-tile
      const int nsub_t = Q.nsub[(size_t)(ti - ta)];
      uint64_t* tw = Q.idx.data() + (Q.idx.size() - (size_t)nsub_t * kCmLanes);
      int need[kMaxCamSlots], rows[kMaxCamSlots], base[kMaxCamSlots], seen[kMaxCamSlots];
      for (int c = 0; c < kMaxCamSlots; ++c) need[c] = 0;
      for (int k = 0; k < nsub_t * 4; ++k) {   // (sub-tile, wave)
        for (int c = 0; c < kMaxCamSlots; ++c) seen[c] = 0;
        for (int l = 0; l < 64; ++l) {
          const uint64_t w = tw[(size_t)k * 64 + l];
          if (w & 0x8000u) { const int cs = (int)((w >> 8) & 63); need[cs] = std::max(need[cs], ++seen[cs]); }
        }
      }
      for (int cap = 8; cap >= 1; cap >>= 1) {
        int total = 0;
        for (int c = 0; c < T.ncs; ++c) {
          int r = 1;
          while (r < need[c] && r < cap) r <<= 1;
          rows[c] = r; base[c] = total; total += r;
        }
        Q.nrows[(size_t)(ti - ta)] = total;
        if (total <= kCmMaxRows) break;
      }
      for (int c = 0; c < T.ncs; ++c) L.rowinfo[(size_t)T.cam0 + c] = base[c] | (rows[c] << 8);   // (tiles own disjoint ranges)
      for (int k = 0; k < nsub_t * 4; ++k) {
        for (int c = 0; c < kMaxCamSlots; ++c) seen[c] = 0;
        for (int l = 0; l < 64; ++l) {
          uint64_t& w = tw[(size_t)k * 64 + l];
          if (!(w & 0x8000u)) continue;
          const int cs = (int)((w >> 8) & 63);
          const uint64_t row = (uint64_t)(base[cs] + (seen[cs]++ & (rows[cs] - 1)));
          w = (w & ~((uint64_t)0x7F00u << 16)) | (row << 24);
        }
      }
    }
  });
  size_t total_sub = 0;
  for (auto& Q : parts) total_sub += Q.subs.size();
  L.subs.reserve(total_sub);
  L.idx.resize(total_sub * (size_t)kCmLanes);
  L.slot.resize(total_sub * (size_t)kCmLanes);
  L.src.resize(total_sub * (size_t)kCmLanes * kCmSeg);
  {
    size_t ti = 0, so = 0;
    for (unsigned th = 0; th < nthr; ++th) {
      Part& Q = parts[th];
      for (size_t k = 0; k < Q.nsub.size(); ++k, ++ti) {
        L.tiles[ti].sub0 = (int32_t)so;
        L.tiles[ti].nsub = Q.nsub[k];
        L.tiles[ti].nrows = Q.nrows[k];
        so += (size_t)Q.nsub[k];
      }
      std::copy(Q.idx.begin(), Q.idx.end(), L.idx.begin() + (size_t)L.subs.size() * kCmLanes);
      std::copy(Q.slot.begin(), Q.slot.end(), L.slot.begin() + (size_t)L.subs.size() * kCmLanes);
      std::copy(Q.src.begin(), Q.src.end(), L.src.begin() + (size_t)L.subs.size() * kCmLanes * kCmSeg);
      L.subs.insert(L.subs.end(), Q.subs.begin(), Q.subs.end());
      L.n_valid += Q.n_valid;
    }
    // the sub-tiles of a tile cover consecutive points: with the next one's point count in the record, a workgroup can request
    // the next sub-tile's points without waiting for its record
    for (CmTile& T : L.tiles) {
      for (int32_t k = 0; k + 1 < T.nsub; ++k) L.subs[(size_t)(T.sub0 + k)].npt_next = L.subs[(size_t)(T.sub0 + k + 1)].npt;
      if (T.nsub) { T.pt0 = L.subs[(size_t)T.sub0].pt0; T.npt0 = L.subs[(size_t)T.sub0].npt; }
    }
  }
  L.n_entries = (int64_t)total_sub * kCmLanes * kCmSeg;
}


// Diagnostics of a camera-major layout (CPU; miba_plan_product_stats and the layout test): every observation of the shard
// appears exactly once (returns false otherwise), padding, and the LDS collision figures that price the kernel's atomic adds.
struct CmStats {
  int64_t n_obs = 0, n_entries = 0, n_sub = 0, n_seg = 0;
  double pt_mult = 0.0;    // mean over (sub-tile, wave, step) of the largest number of lanes on one point
  double cam_mult = 0.0;   // mean over (sub-tile, wave) of the largest number of lanes on one (camera, accumulator copy)
  double rd_extra = 0.0;   // mean over (sub-tile, wave, step): extra LDS cycles of an 8-byte read of a point array (bank pairs hit
                           // by more than one slot, 32 lanes at a time)
  double at_extra = 0.0;   // the same for an atomic add (16 lanes at a time), slots on different addresses only
};
inline bool cm_layout_stats(const CmLayout& L, const std::vector<Tile>& tiles, const std::vector<Group>& grps, const int32_t* idx,
                            const int32_t* pt_first_local, const int32_t* pt_count_local, CmStats& S) {
  S = CmStats();
  S.n_sub = (int64_t)L.subs.size();
  int64_t n_ws = 0, n_w = 0;
  bool ok = true;
  std::vector<uint8_t> seen;
  for (size_t ti = 0; ti < tiles.size(); ++ti) {
    const Tile& T = tiles[ti];
    const CmTile& C = L.tiles[ti];
    if (T.giant) { ok = ok && C.nsub == 0; continue; }
    for (int32_t k = 0; k < C.nsub; ++k) {
      const CmSub& sb = L.subs[(size_t)(C.sub0 + k)];
      const uint64_t* w = L.idx.data() + (size_t)(C.sub0 + k) * kCmLanes;
      S.n_seg += sb.nseg;
      // expected multiset of (point, camera slot) pairs of the sub-tile's points
      std::vector<int32_t> want((size_t)sb.npt * kMaxCamSlots, 0);
      for (int32_t r = 0; r < sb.npt; ++r)
        for (int32_t s = pt_first_local[sb.pt0 + r]; s < pt_first_local[sb.pt0 + r] + pt_count_local[sb.pt0 + r]; ++s)
          want[(size_t)r * kMaxCamSlots + ((idx[s] >> 8) & 0xFF)]++;
      // slot -> point of the sub-tile (a bijection onto the slots in use)
      int inv[kCmMaxPts];
      for (int k2 = 0; k2 < kCmMaxPts; ++k2) inv[k2] = -1;
      const uint8_t* slot = L.slot.data() + (size_t)(C.sub0 + k) * kCmLanes;
      for (int32_t r = 0; r < sb.npt; ++r) { ok = ok && inv[slot[r]] < 0; inv[slot[r]] = r; }
      for (int wv = 0; wv < 4; ++wv) {
        int cam_cnt[kCmMaxRows] = {0};
        int cmax = 0;
        for (int step = 0; step < kCmSeg; ++step) {
          int pt_cnt[kCmMaxPts] = {0};
          int pmax = 0;
          int rd_load[2][32] = {{0}}, at_load[4][16] = {{0}};
          for (int l = 0; l < 64; ++l) {
            const unsigned e = (unsigned)(w[wv * 64 + l] >> (16 * step)) & 0xFFFFu;
            const unsigned e0 = (unsigned)w[wv * 64 + l] & 0xFFFFu;
            if (!(e0 & 0x8000u)) { ok = ok && w[wv * 64 + l] == 0; continue; }
            const int sl = e & 0xFF, cs = (e0 >> 8) & 63, pt = inv[sl];
            if (step == 1) {   // bits 8-14: the lane's accumulator row, one of its camera's
              const int row = (e >> 8) & 127, ri = L.rowinfo[(size_t)T.cam0 + cs];
              ok = ok && row >= (ri & 0xFF) && row < (ri & 0xFF) + (ri >> 8) && row < C.nrows && C.nrows <= kCmMaxRows;
              cmax = std::max(cmax, ++cam_cnt[row]);
            } else ok = ok && cs == (int)((e >> 8) & 63);
            ok = ok && pt >= 0;
            if (pt < 0) continue;
            if (pt_cnt[pt] == 0) { ++rd_load[l >> 5][sl & 31]; }   // (pads read too; the same slot again is a broadcast)
            if (!(e & 0x8000u)) continue;
            if (pt_cnt[pt] == 0) ++at_load[l >> 4][sl & 15];
            want[(size_t)pt * kMaxCamSlots + cs]--;
            pmax = std::max(pmax, ++pt_cnt[pt]);
            ++S.n_obs;
          }
          for (int g = 0; g < 2; ++g) { int m = 1; for (int c = 0; c < 32; ++c) m = std::max(m, rd_load[g][c]); S.rd_extra += m - 1; }
          for (int g = 0; g < 4; ++g) { int m = 1; for (int c = 0; c < 16; ++c) m = std::max(m, at_load[g][c]); S.at_extra += m - 1; }
          S.pt_mult += pmax; ++n_ws;
        }
        S.cam_mult += cmax; ++n_w;
      }
      for (int32_t v : want) ok = ok && v == 0;
      // the sub-tiles of a tile cover its points in order
      if (k + 1 < C.nsub) ok = ok && sb.npt_next == L.subs[(size_t)(C.sub0 + k + 1)].npt && L.subs[(size_t)(C.sub0 + k + 1)].pt0 == sb.pt0 + sb.npt;
    }
  }
  S.n_entries = L.n_entries;
  if (n_ws) { S.pt_mult /= (double)n_ws; S.rd_extra /= (double)n_ws; S.at_extra /= (double)n_ws; }
  if (n_w) S.cam_mult /= (double)n_w;
  return ok;
}

}  // namespace miba
