// Launchers of the lattice-build kernels (knn_kernels.hip).
#pragma once
#include "common.hpp"

namespace osc {

struct KnnPlan {
  int E;               // list entries per lane (capacity 32E >= k)
  int KC;              // candidate slots per (row, split) = 32E
  int row_blocks;      // ceil(N / 128)
  int S;               // column splits
  int cols_per_split;  // multiple of 128
  int rb_begin;        // first row block this process computes (multi-GPU: row-block-sharded build)
  int rb_count;        // number of row blocks this process computes
};

KnnPlan knn_plan(int32_t N, int32_t k, int32_t slots, int rb_begin = 0, int rb_count = -1);
void launch_normalize_rows(const float* Y, int32_t ldy, float* Yn, int32_t ldn, int64_t N, int32_t D, hipStream_t s);
void launch_rows_dot(const float* Yn, int32_t ldn, const float* q, float* out, int64_t N, int32_t D, hipStream_t s);
void launch_knn_topk(const KnnPlan& p, const float* Yn, int32_t ldn, int32_t N, int32_t k, float* cand_val,
                     int32_t* cand_idx, hipStream_t s);
void launch_knn_merge(const KnnPlan& p, const float* cand_val, const int32_t* cand_idx, int32_t N, int32_t k,
                      float* out_val, int32_t* out_idx, hipStream_t s);
void launch_mutual_ell(const float* kval, const int32_t* kidx, int32_t N, int32_t k, int32_t width, int32_t* ell_col,
                       float* ell_a, int32_t* deg, hipStream_t s);
void launch_cap_and_normalize(float* ell_a, float* ell_w, const int32_t* ell_col, const int32_t* deg, int32_t width,
                              int32_t N, float cap, int apply_cap, float* scale_tmp, float* sqrt_deg, hipStream_t s);

}  // namespace osc
