// Host code under AddressSanitizer + UBSan (the reference has no sanitizer builds; GPU sanitizers are not
// available on this pool, so the host side is what can be checked): the tiler (csrc/miba_plan.hpp, serial and threaded
// layout, giant points, filtered observations, several shards) and the BAL loader (threaded parse, binary cache).
// Built and run by tests/test_sanitizers.py; exits 0 when every check holds.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "bundleadjust/BalDataloader.h"
#include "miba_plan.hpp"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); ++fails; } } while (0)

struct Obs { std::vector<int32_t> cam, pt; std::vector<double> xy; };

static Obs make_obs(int n_cam, int n_pt, int per_pt, int n_giant, unsigned seed) {
  std::mt19937 rng(seed);
  Obs o;
  for (int p = 0; p < n_pt; ++p) {
    const int len = p < n_giant ? std::min(n_cam, 70 + 60 * p) : 2 + (int)(rng() % (2 * per_pt - 3));
    const int first = (int)(rng() % n_cam);
    for (int k = 0; k < len; ++k) {
      o.cam.push_back((first + k) % n_cam);
      o.pt.push_back(p);
      o.xy.push_back((double)(rng() % 1000) - 500.0);
      o.xy.push_back((double)(rng() % 1000) - 500.0);
    }
  }
  // file order is not point order
  for (size_t i = o.cam.size(); i > 1; --i) {
    const size_t j = rng() % i;
    std::swap(o.cam[i - 1], o.cam[j]); std::swap(o.pt[i - 1], o.pt[j]);
    std::swap(o.xy[2 * i - 2], o.xy[2 * j]); std::swap(o.xy[2 * i - 1], o.xy[2 * j + 1]);
  }
  // what createProblem filters (src/BundleAdjustment.cpp:49-60): point -1, NaN observation
  for (size_t i = 0; i < o.cam.size(); i += 97) o.pt[i] = -1;
  for (size_t i = 5; i < o.cam.size(); i += 131) o.xy[2 * i] = std::numeric_limits<double>::quiet_NaN();
  return o;
}

static void check_plan(int n_cam, int n_pt, int per_pt, int n_giant, int world, int order, unsigned seed) {
  const Obs o = make_obs(n_cam, n_pt, per_pt, n_giant, seed);
  std::vector<double> X(3 * (size_t)n_pt, 1.0);
  X[3 * (size_t)(n_pt - 1)] = std::numeric_limits<double>::quiet_NaN();   // NaN point: its observations are skipped
  miba::Plan P;
  const int rc = miba::build_plan(P, n_cam, n_pt, (int64_t)o.cam.size(), o.cam.data(), o.pt.data(), o.xy.data(), X.data(), world, order);
  CHECK(rc == 0);
  if (rc) return;
  int64_t expect = 0;
  for (size_t i = 0; i < o.cam.size(); ++i)
    if (o.pt[i] >= 0 && !std::isnan(o.xy[2 * i]) && !std::isnan(o.xy[2 * i + 1]) && o.pt[i] != n_pt - 1) ++expect;
  CHECK(P.n_valid == expect);
  // every surviving observation sits in exactly one slot, with its own camera
  std::vector<uint8_t> seen(o.cam.size(), 0);
  int64_t placed = 0;
  CHECK((int64_t)P.slot_orig.size() == P.n_slots && (int64_t)P.idx.size() == P.n_slots && (int64_t)P.ocam.size() == P.n_slots);
  for (int64_t s = 0; s < P.n_slots; ++s) {
    const int64_t i = P.slot_orig[s];
    if (i < 0) { CHECK(P.idx[s] < 0); continue; }
    CHECK(i < (int64_t)o.cam.size() && !seen[i]);
    seen[i] = 1; ++placed;
    CHECK(P.ocam[s] == o.cam[i] && P.idx[s] >= 0);
  }
  CHECK(placed == expect);
  // tiles cover the slots in order, in whole chunks, with at most kMaxCamSlots cameras unless giant
  int64_t slot = 0;
  for (const miba::Tile& t : P.tiles) {
    CHECK(t.slot0 == slot && t.nslot % miba::kChunk == 0 && t.nslot > 0);
    CHECK(t.giant || t.ncs <= miba::kMaxCamSlots);
    slot += t.nslot;
  }
  CHECK(slot == P.n_slots);
  CHECK((int)P.shard_tile.size() == world + 1 && P.shard_tile.front() == 0 && P.shard_tile.back() == (int64_t)P.tiles.size());
  for (int r = 0; r < world; ++r) CHECK(P.shard_tile[r] <= P.shard_tile[r + 1]);
}

static void check_loader(const std::string& dir) {
  const std::string path = dir + "/problem-sanitize-pre.txt";
  const int n_cam = 7, n_pt = 4000, n_obs = 3 * n_pt;
  {
    FILE* f = fopen(path.c_str(), "w");
    CHECK(f != nullptr);
    if (!f) return;
    fprintf(f, "%d %d %d\n", n_cam, n_pt, n_obs);
    for (int i = 0; i < n_obs; ++i) fprintf(f, "%d %d     %.6e %.6e\n", i % n_cam, i / 3, -1.5 * i, 0.25 * i + 1e-3);
    for (int i = 0; i < 9 * n_cam; ++i) fprintf(f, "%.16e\n", 0.001 * i - 0.02);
    for (int i = 0; i < 3 * n_pt; ++i) fprintf(f, "%.16e\n", 0.5 * i - 7.0);
    fclose(f);
  }
  auto verify = [&](const BalDataloader& d) {
    CHECK(d.getNumFrames() == n_cam && d.getNumPoints() == n_pt && d.getNumObservations() == n_obs);
    CHECK(d.getObsCam(n_obs - 1) == (n_obs - 1) % n_cam && d.getObsPoint(n_obs - 1) == (n_obs - 1) / 3);
    CHECK(d.observations[10].first == (float)(-15.0) && d.cameras[6].k2 == (float)(0.001 * 62 - 0.02));
    CHECK(std::get<2>(d.points[n_pt - 1]) == (float)(0.5 * (3 * n_pt - 1) - 7.0));
    const auto obs = d.getObservations();
    CHECK((int)obs.size() == n_obs && obs[n_obs - 1].y == (float)(0.25 * (n_obs - 1) + 1e-3));
  };
  for (const char* threads : {"1", "5"}) {
    setenv("BA_LOADER_THREADS", threads, 1);
    BalDataloader d(path);
    verify(d);
    std::vector<double> R(3 * n_cam), T(3 * n_cam), K(6 * n_cam), X(3 * (size_t)n_pt);
    d.initialize(R.data(), T.data(), K.data(), X.data());
    CHECK(K[6 * 6 + 5] == (double)d.cameras[6].k2);
  }
  setenv("BA_LOADER_CACHE", "1", 1);
  { BalDataloader d(path); verify(d); }   // writes the cache
  { BalDataloader d(path); verify(d); }   // reads it
  unsetenv("BA_LOADER_CACHE");
  bool threw = false;
  try { BalDataloader d(dir + "/no-such-file.txt"); } catch (const std::invalid_argument&) { threw = true; }
  CHECK(threw);
  // a truncated file must be rejected, not read past its end
  {
    FILE* f = fopen(path.c_str(), "w");
    if (f) { fprintf(f, "%d %d %d\n0 0 1.0 2.0\n1 0 3.0", n_cam, n_pt, n_obs); fclose(f); }
    threw = false;
    try { BalDataloader d(path); } catch (const std::exception&) { threw = true; }
    CHECK(threw);
  }
}

// the exact solve's elimination order (order_tile_graph): bands, rings and graphs without locality, nested separator sets and one
// set, arc counts up to the limit; checks what the numeric phase relies on: a permutation, contiguous groups, and every column's rows in
// its own group or a higher level
static void check_tile_order(int n, int w, bool ring, bool random, int arcs, bool nested, unsigned seed) {
  const size_t nW = ((size_t)n + 63) / 64;
  std::vector<uint64_t> adj((size_t)n * nW, 0);
  auto edge = [&](int u, int v) {
    if (u == v) return;
    adj[(size_t)u * nW + (v >> 6)] |= (uint64_t)1 << (v & 63);
    adj[(size_t)v * nW + (u >> 6)] |= (uint64_t)1 << (u & 63);
  };
  std::mt19937 rng(seed);
  if (random) {
    for (int i = 0; i < 6 * n; ++i) edge((int)(rng() % n), (int)(rng() % n));
  } else {
    for (int u = 0; u < n; ++u)
      for (int d = 1; d <= w; ++d) {
        if (u + d < n) edge(u, u + d);
        else if (ring) edge(u, (u + d) % n);
      }
  }
  miba::TileOrder O;
  const bool ok = miba::order_tile_graph(adj, n, arcs, (int64_t)1 << 40, 0, O, nested);
  CHECK(ok);
  if (!ok) return;
  std::vector<int> seen(n, 0);
  for (int v = 0; v < n; ++v) { CHECK(O.pos[v] >= 0 && O.pos[v] < n); seen[O.pos[v]]++; }
  for (int k = 0; k < n; ++k) CHECK(seen[k] == 1 && O.pos[O.grp[k]] == k);
  const int nG = (int)O.group_level.size();
  CHECK((int)O.group_start.size() == nG + 1 && O.group_start[0] == 0 && O.group_start[nG] == n);
  std::vector<int> g_of(n, -1);
  for (int g = 0; g < nG; ++g) {
    CHECK(O.group_start[g] <= O.group_start[g + 1]);
    if (g) CHECK(O.group_level[g - 1] <= O.group_level[g]);
    for (int k = O.group_start[g]; k < O.group_start[g + 1]; ++k) g_of[k] = g;
  }
  int64_t fill = 0;
  for (int k = 0; k < n; ++k) {
    fill += (int64_t)O.col_nodes[k].size();
    for (int32_t u : O.col_nodes[k]) {
      const int r = O.pos[u];
      CHECK(r > k);
      CHECK(g_of[r] == g_of[k] || O.group_level[g_of[r]] > O.group_level[g_of[k]]);
    }
  }
  CHECK(fill == O.n_off);
  std::vector<std::vector<int32_t>> steps;
  std::vector<uint8_t> first;
  miba::tile_order_steps(O, steps, first);
  size_t cols = 0;
  for (auto& ks : steps) cols += ks.size();
  CHECK(cols == (size_t)n && steps.size() == first.size());
  CHECK(miba::tile_order_cost(O) > 0.0);
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  check_tile_order(400, 6, true, false, 3, true, 1);
  check_tile_order(1200, 3, false, false, 32, true, 2);
  check_tile_order(2000, 6, true, false, 48, true, 3);
  check_tile_order(2000, 6, true, false, 64, false, 4);
  check_tile_order(257, 9, true, false, 2, true, 5);
  check_tile_order(300, 0, false, true, 4, true, 6);     // no locality: not dissected
  check_tile_order(65, 2, false, false, 8, true, 7);
  check_tile_order(7, 1, false, false, 1, true, 8);
  // serial layout (below 2^18 observations) and threaded layout, 1 and 3 shards, both point orders, giant points
  check_plan(90, 3000, 5, 3, 1, 1, 1);
  check_plan(90, 3000, 5, 3, 3, 0, 2);
  check_plan(40, 5, 3, 0, 8, 1, 3);          // more shards than tiles
  setenv("MIBA_PLAN_THREADS", "6", 1);
  check_plan(300, 70000, 5, 2, 2, 1, 4);     // > 2^18 observations: phase 2 on worker threads
  setenv("MIBA_TILE_CHUNKS", "1", 1);
  check_plan(90, 3000, 5, 3, 2, 1, 5);
  unsetenv("MIBA_TILE_CHUNKS");
  check_loader(argv[1]);
  if (fails) { fprintf(stderr, "%d check(s) failed\n", fails); return 1; }
  printf("host sanitize ok\n");
  return 0;
}
