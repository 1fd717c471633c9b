// Host-side index arithmetic and list building of liboscillink_hip.so, kept free of every HIP type: the functions the
// library itself runs (osc_api.hip includes this header) also compile with a plain host compiler, and
// tests/host_logic/sweep_host_logic.cpp sweeps them over N x k x world under -fsanitize=address,undefined on the CPU box
// (sanitizers belong on the CPU build; GPU AddressSanitizer is not available on the pool).
//   column_shard / row_lo / row_owner : the partitions of the sharded solves (column windows, row blocks)
//   build_halo_lists / halo_decide    : which rows of the search direction a row-sharded rank sends and receives
//   pack_csr                          : validation + ELL packing of an injected adjacency (osc_set_csr)
//   xs_groups / blocked_geometry / blocked_list_extent : launch geometry of the XCD-affine and source-blocked matvec
//   blocked_block_count               : how many source blocks the blocked matvec walks
//   blk_place_row                     : host model of k_blk_count / k_blk_fill (one row of the block-major graph copy)
//   CgXSchedule                       : which launch of a CG solve carries which iteration's x update (run_cg)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <utility>
#include <vector>

namespace osc {
namespace host {

struct InvalidArg : std::invalid_argument {
  using std::invalid_argument::invalid_argument;
};

// ---- partitions ---------------------------------------------------------------------------------------------------
// column window of `rank` in a column-sharded solve: slabs in units of 4 floats, as even as possible; empty when there
// are more ranks than 4-column groups (the caller refuses that)
inline std::pair<int32_t, int32_t> column_shard(int32_t dcols, int rank, int world) {
  const int32_t q = dcols / 4;
  const int32_t lo = (int32_t)((int64_t)q * rank / world), hi = (int32_t)((int64_t)q * (rank + 1) / world);
  return {lo * 4, hi * 4};
}
// row block [row_lo(r), row_lo(r + 1)) of rank r of G
inline int64_t row_lo(int64_t N, int G, int r) { return N * r / G; }
inline int row_owner(int64_t N, int G, int64_t row) {
  int r = (int)std::min<int64_t>(G - 1, (row * G + G - 1) / std::max<int64_t>(1, N));
  while (r > 0 && row < row_lo(N, G, r)) --r;
  while (r + 1 < G && row >= row_lo(N, G, r + 1)) ++r;
  return r;
}

// ---- halo lists of the row-sharded CG -------------------------------------------------------------------------------
// Which rows of the search direction rank `me` needs from its peers -- the off-partition column ids its ELL rows (and
// its ends of the chain's path-graph edges) reference -- and which of its own rows each peer needs.  The adjacency is
// symmetric, so "peer q needs my row i" == "my row i has a neighbour in q's block": both lists of a rank pair follow
// from each rank's OWN rows, sorted by row id on both sides.  col / deg: the ELL rows [row_lo(me), row_lo(me + 1)).
struct HaloLists {
  std::vector<int64_t> give_off, need_off;  // [G + 1] offsets of each peer's slice
  std::vector<int32_t> give_idx, need_idx;  // my rows each peer needs / the peers' rows I need (sorted per peer)
};
inline HaloLists build_halo_lists(int64_t N, int G, int me, int32_t width, const int32_t* col, const int32_t* deg,
                                  const std::vector<std::pair<int64_t, int64_t>>& chain_edges) {
  const int64_t r0 = row_lo(N, G, me), r1 = row_lo(N, G, me + 1);
  std::vector<std::vector<int32_t>> need((size_t)G), give((size_t)G);
  std::vector<char> need_mark((size_t)N, 0);
  std::vector<int> give_last((size_t)G);
  auto edge = [&](int64_t j) {  // one of my rows references row j
    if (j >= r0 && j < r1) return;
    if (!need_mark[(size_t)j]) {
      need_mark[(size_t)j] = 1;
      need[(size_t)row_owner(N, G, j)].push_back((int32_t)j);
    }
  };
  for (int64_t i = r0; i < r1; ++i) {
    std::fill(give_last.begin(), give_last.end(), 0);
    const int32_t* ci = col + (size_t)(i - r0) * width;
    for (int e = 0; e < deg[(size_t)(i - r0)]; ++e) {
      const int64_t j = ci[e];
      if (j < 0 || j >= N) throw InvalidArg("halo lists: neighbour id out of range");
      edge(j);
      if (j >= r0 && j < r1) continue;
      const int q = row_owner(N, G, j);
      if (!give_last[(size_t)q]) {
        give_last[(size_t)q] = 1;
        give[(size_t)q].push_back((int32_t)i);
      }
    }
  }
  for (const auto& ab : chain_edges) {  // path graph: consecutive chain nodes (graph.py:96-111), both directions
    for (int dir = 0; dir < 2; ++dir) {
      const int64_t i = dir ? ab.second : ab.first, j = dir ? ab.first : ab.second;
      if (i < r0 || i >= r1 || (j >= r0 && j < r1)) continue;
      edge(j);
      give[(size_t)row_owner(N, G, j)].push_back((int32_t)i);
    }
  }
  HaloLists out;
  out.give_off.assign((size_t)G + 1, 0);
  out.need_off.assign((size_t)G + 1, 0);
  for (int q = 0; q < G; ++q) {
    auto& g = give[(size_t)q];
    std::sort(g.begin(), g.end());
    g.erase(std::unique(g.begin(), g.end()), g.end());
    auto& n = need[(size_t)q];
    std::sort(n.begin(), n.end());
    out.give_idx.insert(out.give_idx.end(), g.begin(), g.end());
    out.need_idx.insert(out.need_idx.end(), n.begin(), n.end());
    out.give_off[(size_t)q + 1] = (int64_t)out.give_idx.size();
    out.need_off[(size_t)q + 1] = (int64_t)out.need_idx.size();
  }
  return out;
}
// all-gathered counts, row r = [need from 0..G-1 | give to 0..G-1] of rank r: consistency of every rank pair, the
// largest need list, and the common decision to exchange whole row blocks when some list covers > 70 % of the remote rows
struct HaloDecision {
  bool consistent = true;
  bool full = false;
  int64_t need_rows_max = 0;
};
inline HaloDecision halo_decide(int64_t N, int G, const std::vector<int32_t>& all) {
  HaloDecision d;
  for (int r = 0; r < G; ++r) {
    int64_t tot = 0;
    for (int q = 0; q < G; ++q) {
      tot += all[(size_t)r * 2 * G + q];
      if (all[(size_t)r * 2 * G + q] != all[(size_t)q * 2 * G + G + r]) d.consistent = false;  // r needs from q == q gives to r
    }
    d.need_rows_max = std::max(d.need_rows_max, tot);
    const int64_t remote = N - (row_lo(N, G, r + 1) - row_lo(N, G, r));
    if (remote > 0 && (double)tot > 0.7 * (double)remote) d.full = true;  // packing would move ~everything anyway
  }
  return d;
}

// ---- injected adjacency -> ELL (osc_set_csr) --------------------------------------------------------------------------
// The graph contract of every consumer: columns ascending within a row (the reference's argwhere order for _signature,
// the first-max tie-break of the null points), no diagonal, no duplicates, symmetric (SPD operator).  Entries <= 0 are
// not edges (graph.py:64).  Throws InvalidArg; nothing else is touched before it returns.
struct PackedEll {
  int64_t width = 1;
  std::vector<int32_t> col, deg;  // [N][width], [N]
  std::vector<float> a;           // [N][width]
};
inline PackedEll pack_csr(int64_t N, const int64_t* rowptr, const int32_t* col, const float* a) {
  if (!rowptr || rowptr[0] != 0) throw InvalidArg("osc_set_csr: rowptr[0] must be 0");
  PackedEll p;
  for (int64_t i = 0; i < N; ++i) {
    if (rowptr[i + 1] < rowptr[i]) throw InvalidArg("osc_set_csr: rowptr must be non-decreasing");
    p.width = std::max(p.width, rowptr[i + 1] - rowptr[i]);
  }
  const int64_t nnz = rowptr[N];
  if (nnz > 0 && (!col || !a)) throw InvalidArg("osc_set_csr: col / a missing");
  const size_t W = (size_t)p.width, n = (size_t)N * W;
  p.col.assign(n, 0);
  p.a.assign(n, 0.f);
  p.deg.assign((size_t)N, 0);
  std::vector<std::pair<int32_t, float>> ent;
  for (int64_t i = 0; i < N; ++i) {
    ent.clear();
    for (int64_t q = rowptr[i]; q < rowptr[i + 1]; ++q) {
      if (col[q] < 0 || col[q] >= N) throw InvalidArg("osc_set_csr: column index out of range");
      if (!(a[q] > 0.f)) continue;  // only strictly positive weights are edges (graph.py:64)
      if (col[q] == i) throw InvalidArg("osc_set_csr: diagonal entry (the lattice adjacency has a zero diagonal)");
      ent.emplace_back(col[q], a[q]);
    }
    std::sort(ent.begin(), ent.end());
    for (size_t e = 1; e < ent.size(); ++e)
      if (ent[e].first == ent[e - 1].first) throw InvalidArg("osc_set_csr: duplicate column within a row");
    for (size_t e = 0; e < ent.size(); ++e) {
      p.col[(size_t)i * W + e] = ent[e].first;
      p.a[(size_t)i * W + e] = ent[e].second;
    }
    p.deg[(size_t)i] = (int32_t)ent.size();
  }
  for (int64_t i = 0; i < N; ++i) {  // symmetry: (j, i) exists with the same weight
    const int32_t* ci = p.col.data() + (size_t)i * W;
    for (int e = 0; e < p.deg[(size_t)i]; ++e) {
      const int32_t j = ci[e];
      const int32_t* cj = p.col.data() + (size_t)j * W;
      const int32_t* hit = std::lower_bound(cj, cj + p.deg[(size_t)j], (int32_t)i);
      if (hit == cj + p.deg[(size_t)j] || *hit != i)
        throw InvalidArg("osc_set_csr: adjacency is not symmetric (missing transposed edge)");
      const float x = p.a[(size_t)i * W + e], y = p.a[(size_t)j * W + (hit - cj)];
      if (std::fabs(x - y) > 1e-6f * std::max(std::fabs(x), std::fabs(y)))
        throw InvalidArg("osc_set_csr: adjacency is not symmetric (A_ij != A_ji)");
    }
  }
  return p;
}

// ---- launch geometry of the XCD-affine / source-blocked matvec ----------------------------------------------------------
// slab groups of a window of `ncols` columns cut into 32-column slabs: gcd(8, slabs), capped
inline int xs_groups(int32_t ncols, int cap = 8) {
  const int nsl = (ncols + 31) / 32;
  const int g = (nsl % 8 == 0) ? 8 : (nsl % 4 == 0) ? 4 : (nsl % 2 == 0) ? 2 : 1;
  return std::min(g, cap);
}
// ... halved until the slabs in flight (groups x N x 128 B) fit 128 MiB of the Infinity Cache; 0 = the mode does not pay:
// below min_reduced groups (the plain slab apply loses to the general path under 4 groups -- config 5's shape 56.4 ms
// general, 54.2 at 4 groups, 57.1 at 2, 60.7 at 1 --, the blocked matvec still wins at 2: xs_plan in osc_api.hip; config
// 4's shape loses at every count)
// Round 5: beyond that budget (N > 524 288) the mode survives only under the wide blocked matvec -- what it gathers from at one
// time is a source BLOCK, not a slab, so the slabs in flight need not fit anything -- with at most four slab groups and
// windows of at least four slabs (profiles/r05_large_n_blocked.txt, per settle against the plain slab apply: 600k x 768 k 32
// 62.1 -> 45.6 ms, 700k x 384 k 16 21.8 -> 20.9, 800k x 256 16.2 -> 15.6, 1M x 384 31.5 -> 29.6, 1M x 128 10.06 -> 9.98, 1.5M x
// 256 34.1 -> 32.5; eight groups lose to four: 600k x 768 47.5 vs 45.6; 1M x 64 k 8 loses: 3.70 vs 3.64).  The caller
// (xs_plan) keeps the mode there only when the blocked matvec can run.
constexpr int64_t kXsBudgetRows = 524288;
inline int xs_groups_for(int64_t N, int32_t ncols, int cap, int min_reduced = 4) {
  const int natural = xs_groups(ncols, cap);
  int g = natural;
  const double cap_bytes = 128.0 * 1024 * 1024;
  while (g > 1 && (double)g * (double)N * 128.0 > cap_bytes) g >>= 1;
  const bool fits = (double)g * (double)N * 128.0 <= cap_bytes && !(g != natural && g < min_reduced);
  if (fits) return g;
  if (N > kXsBudgetRows && ncols >= 128) return std::min(natural, 4);
  return 0;
}
// Work decomposition of k_apply_blocked: xs workgroups per XCD take part; the XCDs form xs_groups slab groups, the
// 8 / xs_groups XCDs of a group split the rows; a gathering wave holds `groups` row groups (of 8 rows) per slice, the
// destination rows of an XCD are cut into `slices` slices, as few as the gmax row groups a wave can hold allow, evenly
// filled.
struct BlockedGeom {
  int32_t xs = 0, xs_groups = 1, groups = 1, slices = 1;
};
inline BlockedGeom blocked_geometry(int64_t N, int xs_groups_, int grid, int resident_per_xcd, int gmax, int gather_waves) {
  BlockedGeom g;
  g.xs = std::min(std::min(grid / 8, 128), std::max(1, resident_per_xcd));
  g.xs_groups = xs_groups_;
  const int64_t parts = 8 / g.xs_groups;
  const int64_t rows = (N + parts - 1) / parts, per_group = (int64_t)g.xs * gather_waves * 8;
  const int64_t nsl = (rows + per_group * gmax - 1) / (per_group * gmax);
  g.slices = (int32_t)nsl;
  g.groups = (int32_t)std::max<int64_t>(1, (rows + nsl * per_group - 1) / (nsl * per_group));
  return g;
}
// One past the largest row index (within one block's N slot rows) the list wave of k_apply_blocked copies: a row group
// that starts inside the lattice is copied whole (8 x gather_waves slot rows), one that starts at or past N is skipped.
// The block-major copy must be padded by at least (extent - N) slot rows behind its last block.  Also checks that the
// slices cover every destination row of every XCD part.
inline int64_t blocked_list_extent(int64_t N, const BlockedGeom& g, int gather_waves) {
  const int parts = 8 / g.xs_groups;
  const int64_t W8 = (int64_t)g.xs * gather_waves * 8, slice_rows = W8 * g.groups;
  int64_t extent = 0;
  for (int xp = 0; xp < parts; ++xp) {
    const int64_t rlo = N * xp / parts, rhi = N * (xp + 1) / parts;
    if (rlo + (int64_t)g.slices * slice_rows < rhi) throw InvalidArg("blocked geometry: the slices do not cover the rows");
    for (int sl = 0; sl < g.slices; ++sl)
      for (int wgx = 0; wgx < g.xs; ++wgx)
        for (int gi = 0; gi < g.groups; ++gi) {
          const int64_t row0 = rlo + sl * slice_rows + (int64_t)wgx * gather_waves * 8 + gi * W8;
          if (row0 >= N) continue;
          extent = std::max(extent, row0 + (int64_t)gather_waves * 8);
        }
  }
  return extent;
}

// ---- block-major copy of the graph ----------------------------------------------------------------------------------------
// Source blocks of the blocked matvec: as many as give a row ~e edges into each (4 slots per (row, block); an edge that
// finds its block's slot row full moves to a later block's, where it is gathered as a miss among hits, so the slot rows
// should be nearly but not quite full).  Measured optimum of e (scripts/exp/nb_sweep.py, profiles/r03_nb_shapes.txt: 11
// shapes from 60k x 768 k 64 to 260k x 768 k 64, k = 16 / 32 / 64): 2.8-3.7 up to N = 131k, 2.2-2.5 from N = 160k on,
// whatever k is and however many XCDs share a slab (the per-rank windows of a sharded config-3 solve, two to eight XCDs
// per slab at N = 100k, also run fastest at 3.3).  Config 5 (N = 200k, k = 64): 16 -> 24 blocks took the L2 misses per
// apply from 367 M to 183 M, the bytes fetched from 45.6 to 22.5 GB and the apply from 6.98 to 5.00 ms.
inline double blocked_edges_per_block(int64_t N) { return N <= 140000 ? 3.3 : 2.5; }
// ... under the wide kernel shapes (round 5: one workgroup per CU, four gather rounds in flight) fuller slot rows win at
// every size -- fewer sub-phases per slab, fewer slot rows to stage, and the displaced edges' misses travel in a deeper
// pipeline (scripts/exp/nb_sweep.py, profiles/r05_nb_sweep.txt, per AP launch): 100k x 768 k 32 8 blocks 0.554 ms / 9 0.557
// / 12 0.627; k 16 4-5 blocks; k 64 16-18; 160k and 200k x 768 k 32 9 blocks (200k: 1.220 against 1.303 at the 12 the old
// rule gives); 260k 10; 200k x 1536 k 64 16-18 blocks 4.03 ms against 4.59 at 24.
// Beyond 450k rows (round 5, profiles/r05_large_n_blocked.txt): 2.2 -- 500k x 384 k 16 7 blocks 14.27 ms per settle against 14.66
// at 5, 600k x 768 k 32 12 blocks 45.6 against 50.1 at 9 and 49.2 at 14, 1M x 384 k 16 6 blocks 29.6 against 30.0 at 5.
inline double blocked_edges_per_block_wide(int64_t N) { return N <= 140000 ? 3.5 : N <= 220000 ? 3.2 : N <= 450000 ? 2.9 : 2.2; }
inline int blocked_block_count(double mean_deg, double edges_per_block, int max_blocks) {
  const int nb = (int)std::max(2.0, std::floor(mean_deg / edges_per_block + 0.5));
  return std::min(nb, max_blocks);
}
inline int blk_of(int col, int rpb, int nb) { return std::min(nb - 1, col / rpb); }
// Host model of k_blk_count / k_blk_fill for ONE row (cols ascending, deg entries): an edge goes into the slot row of its
// own block while that has room (slots 0 .. c - 1 for the block's c <= SL own edges); an edge that finds its block full
// moves to the first later block (cyclically) whose slot row has room behind that block's own edges; what fits nowhere
// goes to `over`.  Unused slots hold {first row of the block, 0.0f}.  slots: [nb][SL] for this row.
struct BlkEntry {
  int32_t col;
  float w;
};
inline void blk_place_row(const int32_t* cols, const float* w, int deg, int32_t N, int nb, int SL, std::vector<BlkEntry>& slots,
                          std::vector<BlkEntry>& over) {
  const int rpb = (N + nb - 1) / nb;
  std::vector<int> c((size_t)nb, 0), k((size_t)nb, 0), tail((size_t)nb, 0);
  for (int e = 0; e < deg; ++e) {
    if (cols[e] < 0 || cols[e] >= N) throw InvalidArg("blk_place_row: column out of range");
    ++c[(size_t)blk_of(cols[e], rpb, nb)];
  }
  for (int q = 0; q < nb; ++q) tail[(size_t)q] = std::min(c[(size_t)q], SL);
  slots.assign((size_t)nb * SL, BlkEntry{0, 0.f});
  over.clear();
  for (int e = 0; e < deg; ++e) {
    const int b = blk_of(cols[e], rpb, nb);
    const int kb = k[(size_t)b]++;
    const BlkEntry ent{cols[e], w[e]};
    if (kb < SL) {
      slots[(size_t)b * SL + kb] = ent;
      continue;
    }
    int tb = -1, ts = 0;
    for (int step = 1; step < nb && tb < 0; ++step) {
      const int q = (b + step) % nb;
      if (tail[(size_t)q] < SL) tb = q, ts = tail[(size_t)q]++;
    }
    if (tb >= 0) slots[(size_t)tb * SL + ts] = ent;
    else over.push_back(ent);
  }
  for (int q = 0; q < nb; ++q)
    for (int t = tail[(size_t)q]; t < SL; ++t) slots[(size_t)q * SL + t] = BlkEntry{std::min(N - 1, q * rpb), 0.f};
}

// ---- where the x update of a CG iteration happens (run_cg in osc_api.hip) ---------------------------------------------
// The solve's host loop enqueues iteration it + 1 before it has read iteration it's residual, so a launch may belong to
// an iteration that never was one.  With the x update deferred (x += alpha p of iteration it is applied by iteration
// it + 1's p update, which reads p anyway) this object decides, launch by launch, which kernel carries which x update,
// so that every real iteration's update is applied exactly once, with that iteration's alpha and p, and none of an
// iteration behind the one the solve stopped in:
//   * gated solves (one GPU): every launch of iteration it carries the gate "residual(it - 1) > tol" and is a no-op
//     otherwise -- an x update that rides in a gated-off p update must be made up for at the end (alpha and p are intact
//     then: everything behind the stop is a no-op);
//   * ungated solves (a sharded solve whose stop test runs beside it): a speculative launch must not touch x at all.
// The iteration expected to be the last (stop_guess, or max_iters) finishes x itself in its x-r kernel and does not
// store the new r; if the solve goes on after all, r is stored by redoing that kernel's r part first.
struct CgXSchedule {
  bool xdefer = true, last_form = true, ungated = false;
  int stop_guess = 0, max_iters = 1;
  int x_done = 0;         // iterations whose x update is applied or rides in an enqueued launch
  int x_rides_gated = 0;  // the iteration whose x update rides in a GATED p update (0: none)
  int r_unstored = 0;     // the iteration whose x-r kernel did not store r (0: none)
  enum XrForm { XR_WITH_X = 0, XR_SKIPS_X = 1, XR_LAST = 2 };
  struct IterForm {
    bool p_applies_x;  // this iteration's p update also applies the previous iteration's x update
    XrForm xr;
  };
  // iteration `it` is being enqueued; speculative: before its predecessor's residual has been read
  IterForm enqueue(int it, bool speculative) {
    IterForm f{false, XR_WITH_X};
    if (!xdefer) return f;
    if (it > 1 && x_done < it - 1) {
      f.p_applies_x = true;
      x_done = it - 1;
      x_rides_gated = ungated ? 0 : it - 1;
    }
    const bool last = last_form && (it == stop_guess || it == max_iters) && !(ungated && speculative);
    f.xr = last ? XR_LAST : XR_SKIPS_X;
    if (last) x_done = it, r_unstored = it;
    return f;
  }
  // no successor of `it` is enqueued for now and the host has seen its predecessor unconverged: apply its x update now?
  bool finish_before_wait(int it) const { return xdefer && x_done < it; }
  // the solve goes on behind `it`: must its r be stored first (by redoing the r part of its x-r kernel)?
  bool restore_r(int it) {
    if (r_unstored != it) return false;
    r_unstored = 0;
    return true;
  }
  // the solve stopped in `iters`: apply its x update now (it rode in a gated p update that did not run)?
  bool finish_at_end(int iters) const { return xdefer && x_rides_gated == iters; }
  void finished(int it) { x_done = it, x_rides_gated = 0; }  // after the x update of `it` was launched on its own
};

}  // namespace host
}  // namespace osc
