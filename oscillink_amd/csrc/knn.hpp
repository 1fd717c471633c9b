// Launchers of the lattice-build kernels (knn_kernels.hip).
#pragma once
#include "common.hpp"
#include "knn_rowmap.hpp"

namespace osc {

struct KnnPlan {
  int E;               // list entries per lane (capacity 32E >= keep)
  int KC;              // candidate slots per (row, split) = 32E
  int keep;            // entries each list keeps (threshold = similarity at rank keep-1)
  int row_blocks;      // 128-row blocks this launch covers
  int S;               // column splits
  int cols_per_split;  // multiple of 128
  int rb_begin;        // first row block this process computes (multi-GPU: row-block-sharded build)
  int rb_count;        // number of row blocks this process computes
  bool f16;            // prefilter variant (fp16 MFMA on the fp16 image of 16*Yn)
  const int32_t* qrows;  // optional explicit list of query rows (per-row exact fallback), device pointer
  int nq;
};

// keep <= 128 (exact) / <= 96 (f16).  slots = resident blocks on the device (load balance of the split count)
KnnPlan knn_plan(int32_t N, int32_t keep, int32_t slots, int rb_begin, int rb_count, bool f16, int splits_override = 0);  // (override: OSC_KNN_SPLITS, read by the caller)
void launch_normalize_rows(const float* Y, int32_t ldy, float* Yn, int32_t ldn, int64_t N, int32_t D, hipStream_t s);
void launch_rows_dot(const float* Yn, int32_t ldn, const float* q, float* out, int64_t N, int32_t D, hipStream_t s);
void launch_rows_cosine(const float* A, int32_t ld, const float* q, float* out, int64_t N, int32_t D, hipStream_t s);
void launch_to_f16(const float* Yn, int32_t ldn, void* Yh, int32_t ldh, int64_t N, int32_t D, hipStream_t s);
// Yop: fp32 Yn (ld floats) or the fp16 image (ld = ldh/2 float slots)
void launch_knn_topk(const KnnPlan& p, const float* Yop, int32_t ld, int32_t N, float* cand_val, int32_t* cand_idx,
                     hipStream_t s);
// small lattices (N <= 4096): dense similarity matrix Sm (N x lds_ floats of scratch) + per-row selection of the k best
void launch_knn_dense(const float* Yn, int32_t ldn, int32_t N, int32_t k, float* Sm, int32_t lds_, float* out_val,
                      int32_t* out_idx, hipStream_t s);
// any k (the k > 128 route): rows [row_begin, row_begin + rows) of the dense similarity matrix into the scratch Sm
// (rows_cap x lds_ floats, rows <= rows_cap, lds_ >= N) and a radix select of each row's k best columns; row_begin must
// be a multiple of 128.  out_val / out_idx are the full N x k lists.
void launch_knn_rows_any(const float* Yn, int32_t ldn, int32_t N, int32_t k, int32_t row_begin, int32_t rows, float* Sm,
                         int32_t lds_, float* out_val, int32_t* out_idx, hipStream_t s);
// exact fp32 lists of a FEW rows (nq <= 32; the prefilter routes' fallback): scores of the listed rows against all
// columns into Sm (nq x lds_ floats, lds_ >= N) in k_knn_rescore's arithmetic, then each row's k best.  Returns false
// when the rows are too wide for it (ldn > 1536): the caller then uses the MFMA kernel's row-list form.
bool launch_knn_few_rows(const float* Yn, int32_t ldn, int32_t N, int32_t k, const int32_t* qrows, int32_t nq, float* Sm,
                         int32_t lds_, float* out_val, int32_t* out_idx, hipStream_t s);
// rank-select the best k_out of the S*KC candidates of each row of the plan's range (or of plan.qrows)
void launch_knn_merge(const KnnPlan& p, const float* cand_val, const int32_t* cand_idx, int32_t N, int32_t k_out,
                      float* out_val, int32_t* out_idx, int clip, hipStream_t s);
void launch_knn_rescore(const KnnPlan& p, const float* Yn, int32_t ldn, int32_t D, int32_t N, const int32_t* cidx,
                        const float* cval, int32_t k, float delta, float* out_val, int32_t* out_idx, int32_t* fail_rows,
                        int32_t* fail_count, hipStream_t s, const KnnRowMap* map = nullptr, float* pair_sc = nullptr,
                        int32_t* pair_pos = nullptr);  // pair_sc / pair_pos (N x keep each): score every undirected candidate pair once (knn_kernels.hip)  // map: the plan's rows are IMAGE rows (KnnRowMap)
void launch_mutual_ell(const float* kval, const int32_t* kidx, int32_t N, int32_t k, int32_t width, int32_t* ell_col,
                       float* ell_a, int32_t* deg, hipStream_t s);
void launch_cap_and_normalize(float* ell_a, float* ell_w, const int32_t* ell_col, const int32_t* deg, int32_t width,
                              int32_t N, float cap, int apply_cap, float* scale_tmp, float* sqrt_deg, hipStream_t s);

}  // namespace osc
