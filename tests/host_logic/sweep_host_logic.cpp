// Sweep of the library's host-side index arithmetic and list building (oscillink_amd/csrc/host_logic.hpp -- the very
// functions osc_api.hip runs) over N x k x world, built with a plain host compiler under
// -fsanitize=address,undefined by tests/test_host_logic_sanitized.py.  Every check is an invariant the device code relies
// on; a violated one prints the case and exits non-zero, a sanitizer report aborts.
//   usage: sweep_host_logic [max_N]
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <set>

#include "../../oscillink_amd/csrc/host_logic.hpp"
#include "../../oscillink_amd/csrc/knn_rowmap.hpp"

using namespace osc::host;

static int g_fail = 0;
#define CHECK(cond, ...)                                  \
  do {                                                    \
    if (!(cond)) {                                        \
      std::fprintf(stderr, "FAIL %s:%d %s  ", __FILE__, __LINE__, #cond); \
      std::fprintf(stderr, __VA_ARGS__);                  \
      std::fprintf(stderr, "\n");                         \
      if (++g_fail > 20) std::exit(1);                    \
    }                                                     \
  } while (0)

// random symmetric graph of N rows, degree <= k, as an ELL (columns ascending) and as CSR triplets in shuffled order
struct Graph {
  int64_t N;
  int32_t width;
  std::vector<int32_t> col, deg;
  std::vector<float> w;
  std::vector<int64_t> rowptr;
  std::vector<int32_t> ccol;
  std::vector<float> ca;
};
static Graph random_graph(int64_t N, int k, std::mt19937_64& rng, bool local) {
  std::vector<std::map<int32_t, float>> adj((size_t)N);
  std::uniform_real_distribution<float> uw(0.05f, 1.0f);
  for (int64_t i = 0; i < N; ++i) {
    const int tries = k / 2 + 1;
    for (int t = 0; t < tries; ++t) {
      int64_t j = local ? std::min<int64_t>(N - 1, i + 1 + (int64_t)(rng() % 7)) : (int64_t)(rng() % (uint64_t)N);
      if (j == i || (int)adj[(size_t)i].size() >= k || (int)adj[(size_t)j].size() >= k) continue;
      const float v = uw(rng);
      adj[(size_t)i][(int32_t)j] = v;
      adj[(size_t)j][(int32_t)i] = v;
    }
  }
  Graph g;
  g.N = N;
  g.width = 1;
  for (auto& a : adj) g.width = std::max<int32_t>(g.width, (int32_t)a.size());
  g.col.assign((size_t)N * g.width, 0);
  g.w.assign((size_t)N * g.width, 0.f);
  g.deg.assign((size_t)N, 0);
  g.rowptr.assign((size_t)N + 1, 0);
  for (int64_t i = 0; i < N; ++i) {
    int e = 0;
    std::vector<std::pair<int32_t, float>> ents(adj[(size_t)i].begin(), adj[(size_t)i].end());
    for (auto& kv : ents) {
      g.col[(size_t)i * g.width + e] = kv.first;
      g.w[(size_t)i * g.width + e] = kv.second;
      ++e;
    }
    g.deg[(size_t)i] = e;
    std::shuffle(ents.begin(), ents.end(), rng);  // CSR in any column order
    for (auto& kv : ents) {
      g.ccol.push_back(kv.first);
      g.ca.push_back(kv.second);
    }
    g.rowptr[(size_t)i + 1] = (int64_t)g.ccol.size();
  }
  return g;
}

// the prefilter image's row order (knn_rowmap.hpp): every piece a bijection of ITS rows, the two directions inverse to each
// other, consecutive image rows of a scattered piece far apart, bad piece tables refused
static void check_row_map(int32_t N, std::mt19937_64& rng) {
  for (int pieces : {1, 2, 3, 11, 24}) {
    if (pieces > N) continue;
    for (int scatter = 0; scatter < 2; ++scatter) {
      std::vector<int32_t> starts;
      if (pieces == 1 || (rng() & 1)) {  // equal pieces (the streamed create's) ...
        const int32_t rows = (N + pieces - 1) / pieces;
        for (int32_t r = 0; r < N; r += std::max(1, rows)) starts.push_back(r);
      } else {  // ... or any ascending table
        std::set<int32_t> cut{0};
        while ((int)cut.size() < pieces) cut.insert((int32_t)(rng() % (uint64_t)N));
        starts.assign(cut.begin(), cut.end());
      }
      const osc::KnnRowMap m = osc::knn_row_map(N, starts.data(), (int)starts.size(), scatter != 0);
      std::vector<char> seen((size_t)N, 0);
      for (int32_t r = 0; r < N; ++r) {
        const int32_t row = osc::knn_map_lattice_row(m, N, r);
        const int j = osc::knn_map_piece(m, r);
        CHECK(row >= m.start[j] && row < m.start[j + 1], "N %d image row %d leaves its piece", N, r);
        CHECK(row >= 0 && row < N && !seen[(size_t)row], "N %d image row %d -> lattice row %d twice or out of range", N, r, row);
        if (row >= 0 && row < N) seen[(size_t)row] = 1;
        CHECK(osc::knn_map_image_row(m, N, row) == r, "N %d: the inverse of image row %d", N, r);
        if (!scatter) CHECK(row == r, "N %d: identity map moves row %d", N, r);
      }
      for (int j = 0; j < m.npieces && scatter; ++j) {
        const int32_t n = m.start[j + 1] - m.start[j];
        if (n >= 64) {  // neighbours in the image are far apart in the lattice
          const int32_t d = std::abs(osc::knn_map_lattice_row(m, N, m.start[j] + 1) - osc::knn_map_lattice_row(m, N, m.start[j]));
          CHECK(std::min(d, n - d) >= n / 4, "N %d piece %d: stride %d of %d", N, j, d, n);
        }
      }
    }
  }
  bool threw = false;
  try {
    const int32_t bad[2] = {0, N};
    (void)osc::knn_row_map(N, bad, 2, true);
  } catch (const std::invalid_argument&) {
    threw = true;
  }
  CHECK(threw, "N %d: a piece starting at N accepted", N);
}

// the threshold sample's order (knn_rowmap.hpp): a bijection of the sample indices, G consecutive indices in G different
// groups (G - 1 once the short last group is full), lattice rows ascending with the index and inside the lattice
static void check_sample_order(int32_t tiles, int32_t group_tiles, int32_t N) {
  const int32_t m = tiles * 128, gsz = group_tiles * 128, G = (tiles + group_tiles - 1) / group_tiles;
  std::vector<int32_t> group_of((size_t)m, -1);
  for (int32_t r = 0; r < m; ++r) {
    const int32_t t = osc::knn_sample_index(r, m, gsz, G);
    CHECK(t >= 0 && t < m && group_of[(size_t)t] < 0, "sample of %d tiles in groups of %d: position %d -> index %d twice or out of range", tiles, group_tiles, r, t);
    if (t >= 0 && t < m) group_of[(size_t)t] = r / gsz;
  }
  for (int32_t t = 0; t + 1 < m; ++t) {
    const int32_t span = std::min<int32_t>(G - 1, m - t);
    bool distinct = true;
    for (int32_t u = 1; u < span; ++u) distinct = distinct && group_of[(size_t)t] != group_of[(size_t)(t + u)];
    CHECK(distinct, "sample of %d tiles in groups of %d: index %d shares a group with one of the next %d", tiles, group_tiles, t, span - 1);
    CHECK(osc::knn_sample_lattice_row(t, m, N) <= osc::knn_sample_lattice_row(t + 1, m, N) && osc::knn_sample_lattice_row(t + 1, m, N) < N, "sample rows not ascending at %d", t);
  }
}

static void check_partitions(int64_t N, int32_t dcols) {
  for (int world : {1, 2, 3, 4, 5, 8, 16}) {
    int32_t prev = 0;
    bool usable = true;
    for (int r = 0; r < world; ++r) {
      auto cw = column_shard(dcols, r, world);
      CHECK(cw.first == prev && cw.second >= cw.first && cw.first % 4 == 0 && cw.second % 4 == 0, "dcols %d world %d rank %d", dcols, world, r);
      if (cw.second == cw.first) usable = false;
      prev = cw.second;
    }
    CHECK(prev == dcols / 4 * 4, "dcols %d world %d", dcols, world);
    (void)usable;
    // row blocks: owner() inverts row_lo()
    int64_t covered = 0;
    for (int r = 0; r < world; ++r) {
      const int64_t a = row_lo(N, world, r), b = row_lo(N, world, r + 1);
      CHECK(a <= b && a == covered, "N %lld world %d rank %d", (long long)N, world, r);
      covered = b;
      for (int64_t row : {a, (a + b) / 2, b - 1})
        if (row >= a && row < b) CHECK(row_owner(N, world, row) == r, "N %lld world %d row %lld", (long long)N, world, (long long)row);
    }
    CHECK(covered == N, "N %lld world %d", (long long)N, world);
  }
}

static void check_halo(const Graph& g, std::mt19937_64& rng) {
  for (int G : {1, 2, 3, 4, 8}) {
    if (g.N < G) continue;
    std::vector<std::pair<int64_t, int64_t>> chain;
    if (g.N >= 4)
      for (int t = 0; t < 5; ++t) chain.emplace_back((int64_t)(rng() % (uint64_t)g.N), (int64_t)(rng() % (uint64_t)g.N));
    std::vector<HaloLists> all;
    std::vector<int32_t> counts((size_t)G * 2 * G);
    for (int me = 0; me < G; ++me) {
      const int64_t r0 = row_lo(g.N, G, me);
      HaloLists hl = build_halo_lists(g.N, G, me, g.width, g.col.data() + (size_t)r0 * g.width, g.deg.data() + r0, chain);
      for (int q = 0; q < G; ++q) {
        counts[(size_t)me * 2 * G + q] = (int32_t)(hl.need_off[(size_t)q + 1] - hl.need_off[(size_t)q]);
        counts[(size_t)me * 2 * G + G + q] = (int32_t)(hl.give_off[(size_t)q + 1] - hl.give_off[(size_t)q]);
        for (int64_t t = hl.need_off[(size_t)q]; t < hl.need_off[(size_t)q + 1]; ++t) {
          const int32_t row = hl.need_idx[(size_t)t];
          CHECK(row_owner(g.N, G, row) == q && q != me, "need list of rank %d holds row %d not owned by %d", me, row, q);
          CHECK(t == hl.need_off[(size_t)q] || hl.need_idx[(size_t)t - 1] < row, "need list not strictly ascending");
        }
        for (int64_t t = hl.give_off[(size_t)q]; t < hl.give_off[(size_t)q + 1]; ++t)
          CHECK(row_owner(g.N, G, hl.give_idx[(size_t)t]) == me, "give list holds a foreign row");
      }
      all.push_back(std::move(hl));
    }
    // what r needs from q is exactly what q gives to r, row for row (both sorted)
    for (int r = 0; r < G; ++r)
      for (int q = 0; q < G; ++q) {
        const auto& nr = all[(size_t)r];
        const auto& gq = all[(size_t)q];
        const int64_t n0 = nr.need_off[(size_t)q], n1 = nr.need_off[(size_t)q + 1], g0 = gq.give_off[(size_t)r], g1 = gq.give_off[(size_t)r + 1];
        CHECK(n1 - n0 == g1 - g0, "N %lld G %d: rank %d needs %lld rows of %d which gives %lld", (long long)g.N, G, r, (long long)(n1 - n0), q, (long long)(g1 - g0));
        for (int64_t t = 0; t < std::min(n1 - n0, g1 - g0); ++t)
          CHECK(nr.need_idx[(size_t)(n0 + t)] == gq.give_idx[(size_t)(g0 + t)], "halo row lists of a rank pair differ");
      }
    const HaloDecision d = halo_decide(g.N, G, counts);
    CHECK(d.consistent, "N %lld G %d", (long long)g.N, G);
  }
}

static void check_pack_csr(const Graph& g, std::mt19937_64& rng) {
  PackedEll p = pack_csr(g.N, g.rowptr.data(), g.ccol.data(), g.ca.data());
  CHECK(p.width == g.width || (g.ccol.empty() && p.width == 1), "width %lld vs %d", (long long)p.width, g.width);
  for (int64_t i = 0; i < g.N; ++i) {
    CHECK(p.deg[(size_t)i] == g.deg[(size_t)i], "row %lld", (long long)i);
    for (int e = 0; e < g.deg[(size_t)i]; ++e) {
      CHECK(p.col[(size_t)i * p.width + e] == g.col[(size_t)i * g.width + e], "row %lld", (long long)i);
      CHECK(p.a[(size_t)i * p.width + e] == g.w[(size_t)i * g.width + e], "row %lld", (long long)i);
    }
  }
  if (g.ccol.empty()) return;
  // refused inputs: a diagonal entry, a duplicate, an out-of-range column, a missing transposed edge, A_ij != A_ji
  const size_t pick = (size_t)(rng() % g.ccol.size());
  int64_t row = 0;
  while (g.rowptr[(size_t)row + 1] <= (int64_t)pick) ++row;
  for (int kind = 0; kind < 5; ++kind) {
    std::vector<int32_t> c = g.ccol;
    std::vector<float> a = g.ca;
    if (kind == 0) c[pick] = (int32_t)row;
    if (kind == 1) {
      if (g.rowptr[(size_t)row + 1] - g.rowptr[(size_t)row] < 2) continue;
      c[(size_t)g.rowptr[(size_t)row]] = c[(size_t)g.rowptr[(size_t)row] + 1];
    }
    if (kind == 2) c[pick] = (int32_t)g.N;
    if (kind == 3) a[pick] = 0.f;
    if (kind == 4) a[pick] *= 1.001f;
    bool refused = false;
    try {
      (void)pack_csr(g.N, g.rowptr.data(), c.data(), a.data());
    } catch (const InvalidArg&) {
      refused = true;
    }
    CHECK(refused, "pack_csr accepted a broken adjacency (kind %d, N %lld)", kind, (long long)g.N);
  }
}

static void check_blocked(int64_t N, std::mt19937_64& rng) {
  constexpr int kGmax = 16, kWaves = 7, kPadRows = 8192;  // cg_kernels.hip: kBlkGroups, kBlkGatherWaves; blocked_view's padding
  for (int xg : {1, 2, 4, 8})
    for (int resident : {32, 64, 96, 128})
      for (int grid : {8, 64, 512, 1024}) {
        const BlockedGeom g = blocked_geometry(N, xg, grid, resident, kGmax, kWaves);
        CHECK(g.xs >= 1 && g.xs <= std::max(1, grid / 8) && g.groups >= 1 && g.groups <= kGmax && g.slices >= 1, "N %lld xg %d grid %d", (long long)N, xg, grid);
        const int64_t extent = blocked_list_extent(N, g, kWaves);  // throws if the slices do not cover the rows
        CHECK(extent <= N - 1 + 8 * kWaves && extent <= N + kPadRows, "N %lld xg %d resident %d grid %d: list copies reach row %lld", (long long)N, xg, resident, grid, (long long)extent);
      }
  (void)rng;
}

static void check_blk_place(const Graph& g) {
  constexpr int SL = 4;
  std::vector<BlkEntry> slots, over;
  for (int nb : {1, 2, 3, 7, 9, 16, 24, 32}) {
    if (nb > g.N) continue;
    const int rpb = (int)((g.N + nb - 1) / nb);
    for (int64_t i = 0; i < g.N; ++i) {
      blk_place_row(g.col.data() + (size_t)i * g.width, g.w.data() + (size_t)i * g.width, g.deg[(size_t)i], (int32_t)g.N, nb, SL, slots, over);
      // every edge exactly once, with its weight; fillers have weight 0 and point at the block's first row
      std::multiset<std::pair<int32_t, float>> placed, want;
      for (int e = 0; e < g.deg[(size_t)i]; ++e) want.emplace(g.col[(size_t)i * g.width + e], g.w[(size_t)i * g.width + e]);
      for (int q = 0; q < nb; ++q)
        for (int t = 0; t < SL; ++t) {
          const BlkEntry& en = slots[(size_t)q * SL + t];
          CHECK(en.col >= 0 && en.col < g.N, "slot column out of range");
          if (en.w != 0.f) placed.emplace(en.col, en.w);
          else CHECK(en.col == std::min<int64_t>(g.N - 1, (int64_t)q * rpb), "filler slot does not point at its block's first row");
        }
      for (auto& en : over) placed.emplace(en.col, en.w);
      CHECK(placed == want, "N %lld nb %d row %lld: edges lost or duplicated", (long long)g.N, nb, (long long)i);
      CHECK((int64_t)over.size() == std::max<int64_t>(0, (int64_t)g.deg[(size_t)i] - (int64_t)nb * SL), "overflow count");
    }
  }
}

// ---- CgXSchedule against a model of the device ---------------------------------------------------------------------
// The host loop of run_cg (osc_api.hip) is replayed here verbatim; the "device" executes the launches in order with the
// gating rule of the kernels (a gated launch of iteration it runs iff iteration it - 1 did not converge; an ungated
// one always runs) and tracks which iteration's values p, alpha and r hold.  Checked: every iteration up to the one the
// solve stopped in has its x update applied exactly once, with its own p and alpha; no other; every kernel that reads
// r finds the r it expects; the host never launches anything for an iteration it should not.
static void check_cg_schedule(int max_iters, int stop_guess, int converge_at, bool ungated, bool xdefer, bool last_form) {
  CgXSchedule xs;
  xs.xdefer = xdefer, xs.last_form = last_form, xs.ungated = ungated, xs.stop_guess = stop_guess, xs.max_iters = max_iters;
  // device state: the iteration whose p / alpha / r the arrays hold (p: 1 after the INIT pass, r: 0 = r of x0)
  int p_ver = 1, alpha_ver = 0, r_ver = 0;
  std::vector<int> x_applied((size_t)max_iters + 3, 0);
  auto converged = [&](int it) { return converge_at > 0 && it == converge_at; };
  // a launch of iteration `it` runs on the device iff ...
  auto runs = [&](int it, bool gated) { return !gated || it == 1 || !converged(it - 1); };
  const bool gated = !ungated;
  auto apply_x = [&](int it_expected) {  // x += alpha p with whatever the arrays hold
    CHECK(p_ver == alpha_ver, "x update with p of iteration %d and alpha of iteration %d", p_ver, alpha_ver);
    CHECK(p_ver == it_expected, "x update meant for iteration %d applied with p of iteration %d", it_expected, p_ver);
    ++x_applied[(size_t)p_ver];
  };
  auto enqueue_iter = [&](int it, bool speculative) {
    const CgXSchedule::IterForm f = xs.enqueue(it, speculative);
    if (it > 1 && runs(it, gated)) {  // update_p: [x += alpha p,] p = z(r) + beta p
      if (f.p_applies_x) apply_x(it - 1);
      CHECK(r_ver == it - 1, "p update of iteration %d reads r of iteration %d", it, r_ver);
      p_ver = it;
    }
    CHECK(it > 1 || !f.p_applies_x, "iteration 1 has no predecessor");
    if (runs(it, gated)) alpha_ver = it;  // matvec + reduce_alpha
    if (runs(it, gated)) {                // x-r kernel
      CHECK(r_ver == it - 1, "x-r kernel of iteration %d reads r of iteration %d", it, r_ver);
      if (f.xr == CgXSchedule::XR_WITH_X || f.xr == CgXSchedule::XR_LAST) apply_x(it);
      if (f.xr != CgXSchedule::XR_LAST) r_ver = it;
    }
    CHECK(xdefer || f.xr == CgXSchedule::XR_WITH_X, "not deferred: the x-r kernel updates x");
  };
  auto finish_x = [&](int it) {  // ungated launch on its own
    apply_x(it);
    xs.finished(it);
  };
  int iters = max_iters, enqueued = 1;
  enqueue_iter(1, false);
  for (int it = 1; it <= max_iters; ++it) {
    if (it < max_iters && it != stop_guess && enqueued == it) enqueue_iter(++enqueued, true);
    else if (xs.finish_before_wait(it)) finish_x(it);
    if (converged(it)) {
      iters = it;
      break;
    }
    if (it < max_iters && enqueued == it) {
      if (xs.restore_r(it)) {
        CHECK(r_ver == it - 1, "redoing the r update of iteration %d from r of iteration %d", it, r_ver);
        r_ver = it;
      }
      enqueue_iter(++enqueued, false);
    }
  }
  if (xs.finish_at_end(iters)) finish_x(iters);
  for (int it = 1; it <= max_iters + 1; ++it)
    CHECK(x_applied[(size_t)it] == (it <= iters ? 1 : 0),
          "max_iters %d guess %d converge_at %d ungated %d defer %d last %d: x update of iteration %d applied %d times (solve stopped in %d)",
          max_iters, stop_guess, converge_at, (int)ungated, (int)xdefer, (int)last_form, it, x_applied[(size_t)it], iters);
}

int main(int argc, char** argv) {
  for (int max_iters = 1; max_iters <= 9; ++max_iters)
    for (int guess = 0; guess <= max_iters + 2; ++guess)
      for (int conv = 0; conv <= max_iters + 1; ++conv)  // 0 / beyond max_iters: never converges
        for (int m = 0; m < 8; ++m) {
          if ((m & 1) != 0 && (m & 2) == 0) continue;  // (ungated iterations need the deferred x update: run_cg never combines these)
          check_cg_schedule(max_iters, guess, conv > max_iters ? 0 : conv, (m & 1) != 0, (m & 2) != 0, (m & 4) != 0);
        }
  const int64_t max_n = argc > 1 ? std::atoll(argv[1]) : 200000;
  std::mt19937_64 rng(12345);
  std::vector<int64_t> ns;
  for (int64_t n = 1; n <= 70; ++n) ns.push_back(n);
  for (int64_t n : {127, 128, 129, 255, 256, 257, 1000, 4095, 4096, 4097, 7167, 7168, 7169, 16384, 57344, 57345, 100000, 114688, 114689, 114700, 130000, 131072, 200000})
    if (n <= max_n) ns.push_back(n);
  for (int t = 0; t < 40; ++t) ns.push_back(71 + (int64_t)(rng() % (uint64_t)std::max<int64_t>(1, max_n - 71)));
  for (int64_t N : ns) {
    for (int32_t dcols : {4, 8, 52, 96, 128, 768, 1000, 1536}) check_partitions(N, dcols);
    check_blocked(N, rng);
    if (N >= 2) check_row_map((int32_t)N, rng);
    if (N > 20000) continue;  // graph-building checks: small and mid sizes (seconds under ASan)
    for (int k : {1, 3, 6, 16, 33, 64}) {
      if (N > 3000 && k != 6 && k != 33) continue;
      const Graph g = random_graph(N, k, rng, (k & 1) != 0);
      check_halo(g, rng);
      check_pack_csr(g, rng);
      if (N <= 3000) check_blk_place(g);
    }
  }
  for (int32_t tiles : {24, 25, 65, 87, 128, 129, 141, 183, 255, 256, 257, 651})
    for (int32_t gt : {(tiles + 127) / 128, (tiles + 127) / 128 + 1})
      check_sample_order(tiles, gt, tiles * 128 * 12 + 77);
  // xs group counts: divisors of 8; slabs in flight within 128 MiB up to kXsBudgetRows rows, beyond that (the wide blocked
  // matvec's regime: its working set is a source block) at most four groups and only for windows of >= 4 slabs
  for (int64_t N : {1000, 16384, 100000, 131072, 131073, 200000, 262144, 300000, 524288, 524289, 1000000, 3000000})
    for (int32_t ncols : {32, 96, 128, 192, 256, 384, 768, 1000, 1536, 2048}) {
      const int g = xs_groups_for(N, ncols, 8);
      const bool in_budget = (double)g * (double)N * 128.0 <= 128.0 * 1024 * 1024;
      CHECK(g == 0 || (8 % g == 0 && (in_budget || (N > kXsBudgetRows && ncols >= 128 && g <= 4 && g <= xs_groups(ncols, 8)))),
            "N %lld ncols %d groups %d", (long long)N, ncols, g);
    }
  for (double deg : {1.0, 13.8, 28.8, 59.5})
    for (int64_t N : {96000, 140000, 220000, 450000, 450001, 2000000}) {
      const int nb = blocked_block_count(deg, blocked_edges_per_block_wide(N), 32);
      CHECK(nb >= 2 && nb <= 32, "wide deg %g", deg);
    }
  for (double deg : {0.0, 1.0, 6.5, 13.8, 28.8, 59.5, 128.0})
    for (int64_t N : {1000, 100000, 140000, 140001, 1000000}) {
      const int nb = blocked_block_count(deg, blocked_edges_per_block(N), 32);
      CHECK(nb >= 2 && nb <= 32, "deg %g", deg);
    }
  if (g_fail) {
    std::fprintf(stderr, "%d check(s) failed\n", g_fail);
    return 1;
  }
  std::printf("host logic sweep ok: %zu lattice sizes up to %lld\n", ns.size(), (long long)max_n);
  return 0;
}
