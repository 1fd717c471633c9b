// Prefilter route "panel": fp16 similarity GEMM with the query panel register-resident (knn_gemm.hip).
#pragma once
#include <vector>

#include "common.hpp"
#include "knn.hpp"

namespace osc {

// K steps of 64 halfs the panel kernel is built for: 6 (D <= 384) or 12 (D <= 768); 0 = D not served by this route
int knn_panel_nkt(int32_t D);
// ... and of the tile core that serves 768 < D <= 4096 in single-process builds (0 = not served)
int knn_tile_nkt(int32_t D);

struct KnnPanelPlan {
  bool ok;             // the lattice is large enough for sampled thresholds (else: use the tile prefilter)
  int nkt;             // 6 or 12
  int nrg;             // row groups (32 rows of consecutive 128-row blocks) a wave's register panel holds in the MAIN sweep: 2 at K depth 6 and, round 6, in the half sweep at K depth 12 from 256 row blocks on (64-column passes); else 1
  int nrg_s;           // ... in the SAMPLE sweep: nrg at K depth 6, else 1 (its running maxima beside 384 panel registers would not fit)
  int32_t ldh;         // pitch of the fp16 images in halfs = 64 * nkt
  int32_t npad;        // rows of the query image (N rounded up to 128), zero-filled beyond N
  int32_t nrb;         // query row blocks (npad / 128)
  int32_t sample_tiles;  // 128-column tiles of the strided column sample the thresholds come from
  int32_t group_tiles;   // sample tiles per maximum (so that a row has <= 128 maxima)
  int32_t sample_groups;
  int32_t sample_rank;   // threshold = sample_rank-th largest group maximum of the sample
  int32_t S;           // column splits of the main sweep
  int32_t tiles_per_split;
  int32_t SA;          // column splits of the sample sweep
  int32_t sample_tiles_per_split;
  int32_t hit_cap;     // entries of one hit list (one per work item and wave)
  int32_t keep;        // candidates handed to the exact re-scoring
  double hit_bound;    // candidates per row the thresholds are expected to let through at most: max(5 keep, 20 rho)
  // Half sweep (single-process builds): the similarity matrix is symmetric bit for bit, so row block I visits only the
  // column tiles J >= I and every accumulator is tested against its row's AND its column's threshold.  The sweep is cut
  // into S column CHUNKS of T tiles (S, tiles_per_split = T above); chunk c is swept by the row blocks I < min(nrb,
  // (c + 1) T): nitems work items in all.  All hits are delivered to buckets of bucket_cap entries, one per group of 32
  // receiving rows (npad / 32 of them), which is all the select reads.
  bool sym;
  bool tile_core;      // D > 768: both operands through LDS (k_tile_thr) instead of the register-resident panel; half sweep only
  int tile_group_sets;  // k_tile_thr2: row sets per launch
  bool tile_wide;      // ... its main sweep with 64 x 128 wave tiles (k_tile_thr2); the fp16 image then carries one zero tile behind npad
  int32_t T;
  int32_t nitems;
  int32_t bucket_cap;
  // Row scatter of the fp16 images: image row r (r < N) holds lattice row (r * scatter) mod N -- a bijection (scatter is
  // coprime to N; 1 = identity).  Anchors often arrive grouped (documents, clusters): then the 32 rows of a wave all
  // have their ~cluster-size best columns in the same one or two column tiles, one (work item, wave) hit list takes
  // 32 x cluster size entries, overflows, and every row of the lattice falls back to the exact kernel (measured: 1000
  // clusters x 100 rows in cluster order, N = 100k: 100 000 fallback rows, build 190 ms against 39 ms for the same
  // anchors shuffled).  Scattered, neighbouring image rows are unrelated lattice rows whatever the caller's order.  Only
  // the prefilter stage works on image rows; k_panel_select hands lattice ids (rows and candidate columns) to the
  // re-scoring.  Single-process builds and sharded builds that share the half sweep (their ranks own IMAGE row blocks and
  // assemble the lists by all-reduce: osc_graph.hip); a sharded build with a full sweep per rank (OSC_KNN_PANEL_SYM=0) keeps
  // the identity, because its ranks own contiguous LATTICE row blocks.
  int32_t scatter;
  // ... and the general form every kernel reads (knn.hpp: KnnRowMap): one piece with multiplier `scatter` as planned;
  // knn_panel_set_pieces cuts the image into pieces that are permuted separately (the streamed create)
  KnnRowMap map;
};
// image row -> lattice row of a plan (rows >= N are padding and map to themselves)
inline int32_t knn_panel_row(const KnnPanelPlan& p, int32_t N, int32_t r) { return r < N ? knn_map_lattice_row(p.map, N, r) : r; }
// pieces starting at the given image rows, each holding ITS lattice rows (scattered within the piece iff the plan scatters)
void knn_panel_set_pieces(KnnPanelPlan& p, int32_t N, const int32_t* starts, int npieces);
// A/B overrides of the planner (OSC_KNN_PANEL_NRG / _RHO / _T / _RANK, read by the caller; 0 = the planner's own choice)
struct KnnPanelTune {
  int nrg = 0;     // 1: one row group per wave also at K depth 6; 2: two also at K depth 12 below 256 row blocks (half sweep)
  double rho = 0;  // one sample column in rho
  int T = 0;       // half sweep: tiles per chunk
  int rank = 0;    // threshold = rank-th largest group maximum of the sample
  int tile_wide = 1;  // 0: the tile core's main sweep with 32 x 128 wave tiles (k_tile_thr<1>)
  int tile_group_mb = 0;  // k_tile_thr2: image bytes of one group of row sets (0: 128 MB)
  int sa = 0;      // sample sweep: splits of the sample's tile groups per row-block set (0: the planner's tail model)
};
KnnPanelPlan knn_panel_plan(int32_t N, int32_t D, int32_t keep, int32_t cus, bool scatter_rows = false, bool sym = false,
                            const KnnPanelTune& tune = KnnPanelTune{});
// half sweep: the device arrays the sweep and the select share
struct KnnPanelSymDev {
  void* bucket_ent;      // uint2 [npad / 32][bucket_cap]
  int32_t* bucket_cnt;   // [npad / 32], zeroed by the caller
  int32_t* flags;        // [S], zeroed by the caller
};

// fp32 unit rows -> fp16 image of 16 * Yn with pitch plan.ldh, rows [N, npad) zero; image rows [r0, r1) only (r1 < 0: npad)
void launch_panel_image(const float* Yn, int32_t ldn, void* Yh, const KnnPanelPlan& p, int32_t N, int32_t D, hipStream_t s,
                        int32_t r0 = 0, int32_t r1 = -1);
// the streamed create's column sample: `rows` unit rows (pitch ldn), already in sample order -> the sample image
void launch_panel_sample_rows(const float* Yn_rows, int32_t ldn, void* Ys, const KnnPanelPlan& p, int32_t rows, int32_t D, hipStream_t s);
// the column sample (knn_rowmap.hpp: knn_sample_index / knn_sample_lattice_row): an even stride of LATTICE rows, dealt to the
// threshold groups in turn -- copied from the rows' places in the query image
void launch_panel_sample(const void* Yh, void* Ys, const KnnPanelPlan& p, int32_t N, hipStream_t s);
// phase A: per (query row, group of sample tiles) maximum fp16 score -> tmax [npad][sample_groups], for the query row
// blocks [rb_begin, rb_begin + rb_count)
void launch_panel_tilemax(const void* Yh, const void* Ys, const KnnPanelPlan& p, int32_t N, int rb_begin, int rb_count,
                          float* tmax, unsigned* queue, int grid, hipStream_t s);
// threshold per row = sample_rank-th largest of its tile maxima; image rows [r0, r1) only (r1 < 0: npad)
void launch_panel_tau(const float* tmax, const KnnPanelPlan& p, int32_t N, float* tau, hipStream_t s, int32_t r0 = 0, int32_t r1 = -1);
// phase B: every (row, column) with fp16 score > tau[row] (diagonal excluded) is appended to the hit list of its
// (column split, row block, wave): hit_list [(list * 4 + wave) * hit_cap + e] = 8-byte entries {local row << 27 | column,
// score bits}, hit_cnt [list * 4 + wave] (may exceed hit_cap: overflow); list = split * rb_count + (row block - rb_begin)
// shards > 1 (half sweep of a sharded build): this call sweeps the work items shard, shard + shards, ... only
// chunk_hi >= 0 (half sweep on the panel core): only the column chunks [chunk_lo, chunk_hi) -- what the streamed create
// launches as the rows of those chunks arrive; `grid` is then capped by the items of the window
void launch_panel_filter(const void* Yh, const KnnPanelPlan& p, int32_t N, int rb_begin, int rb_count, const float* tau,
                         void* hit_list, int32_t* hit_cnt, unsigned* queue, int grid, hipStream_t s,
                         const KnnPanelSymDev* sd = nullptr, int shard = 0, int shards = 1, int chunk_lo = 0, int chunk_hi = -1);
// sharded half sweep: min(cnt, cap) per bucket; a rank's buckets packed behind one another (off = exclusive scan of the
// clamped counts); the other ranks' entries appended to the buckets [b0, b0 + nb_mine) (knn_gemm.hip: k_bucket_merge)
void launch_bucket_clamp(const int32_t* cnt, int32_t nb, int32_t cap, int32_t* clamped, hipStream_t s);
void launch_bucket_pack(const void* ent, const int32_t* cnt, const int32_t* off, int32_t nb, int32_t cap, void* out, hipStream_t s);
void launch_bucket_merge(void* ent, int32_t* cnt, const int32_t* all_cnt, const int32_t* src_off, const int64_t* seg_off,
                         const void* recv, int32_t b0, int32_t nb_mine, int32_t nb_all, int32_t cap, int32_t me, int32_t ranks,
                         hipStream_t s);
// per row: `keep` candidates holding the keep best fp16 scores (unsorted, the minimum in the last slot) -> cval / cidx
// [N][keep]; rows whose candidate set is incomplete (a list overflowed) or too small (< keep) are appended to fail_rows
// and get an empty list
void launch_panel_select(const KnnPanelPlan& p, int rb_begin, int rb_count, int32_t N, const void* hit_list,
                         const int32_t* hit_cnt, float* cval, int32_t* cidx, int32_t* fail_rows, int32_t* fail_count,
                         hipStream_t s, const KnnPanelSymDev* sd = nullptr);

// Half-sweep builds: rows the first re-scoring could not prove (rows_in, nrows) are re-scored against EVERY candidate of
// their bucket and proven against tau_row instead of the list's last score; rows still undecided are appended to fail_rows
void launch_bucket_rescore(const KnnPanelPlan& p, const KnnPanelSymDev& sd, const float* Yn, int32_t ldn, int32_t N,
                           const int32_t* rows_in, int32_t nrows, const float* tau, int32_t k, float delta, float* out_val,
                           int32_t* out_idx, int32_t* fail_rows, int32_t* fail_count, hipStream_t s);

}  // namespace osc
