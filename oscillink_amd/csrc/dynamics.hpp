#pragma once
#include "common.hpp"

namespace osc {

// per-block partial (sum in fp64, max) of a float array; grid = nblocks <= 256, the host adds the partials
void launch_sum_max(const float* v, int64_t n, int nblocks, double* psum, float* pmax, hipStream_t s);
// per-block top-K (K <= 32) strictly positive entries of the per-ELL-entry array v (width entries per row) in the order
// (value desc, caller's row id asc, caller's column id asc; api_id = device row -> caller's id, or nullptr); block b
// scans the contiguous chunk b; out_val / out_idx hold nblocks x K entries (idx = -1: none); out_col[t] = col[idx]
void launch_top_select(const float* v, const int32_t* col, const int32_t* api_id, int32_t width, int64_t n, int nblocks, int K,
                       float* out_val, int64_t* out_idx, int32_t* out_col, hipStream_t s);
// BFS over the ELL lattice graph: seeds = rows with sqrt(move2 + 1e-12) >= thr (dist 0), everything else -1
void launch_bfs_seeds(const float* move2, int64_t N, float thr, int32_t* dist, hipStream_t s);
// one level: every row at distance `level` labels its unvisited neighbours level + 1 and raises *changed
void launch_bfs_level(const int32_t* col, const int32_t* deg, int32_t width, int64_t N, int32_t* dist, int32_t level,
                      int32_t* changed, hipStream_t s);

}  // namespace osc
