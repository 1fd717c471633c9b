// Mutual-kNN lattice build on gfx950 (graph.py:8-93), never materialising the N x N similarity matrix.
//
//   k_normalize_rows : Yn = Y / (||Y_i|| + 1e-12)                      (graph.py:35)       HBM-bound, 1 pass
//   k_knn_topk<E>    : S = Yn Yn^T in exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) tiles, LDS-staged,
//                      fused with a register-resident running top-k per query row (graph.py:36-37, 59)
//   k_knn_merge      : merges the per-column-split candidate lists into the final (sim desc, idx asc)
//                      top-k (graph.py:46-49), clips at 0 (graph.py:62)
//   k_mutual_ell     : mutual test + max-symmetrise (graph.py:64-65) -> ELL rows with ascending columns
//   k_row_scale / k_apply_cap / k_sqrt_deg / k_normalize_w : row_sum_cap + normalized_laplacian (graph.py:69-93)
//
// MFMA tiling (wave64): a block = 4 waves owns 128 query rows; wave w computes rows 32w..32w+31 against a
// 128-column tile as four 32x32 accumulators (64 acc VGPRs).  With D[i][j] = sum_k A[i][k] B[k][j] and
// B[k][j] = Yn[col j][k], both operands are "row of Yn, element k": lane l supplies element k = l>>5 of row
// l&31.  The 32-deep K tile sits in LDS as [128 rows][36 floats] (pad 4 -> ds_read_b128 conflict-free);
// lane half h reads k = 8s+4h..8s+4h+3 with one ds_read_b128 and feeds component u to MFMA (s,u), so each
// MFMA pairs k = 8s+u with k = 8s+4+u.  The summation order is the same for S_ij and S_ji, hence the
// similarity matrix is bitwise symmetric, as the mutual test needs.
//
// Running top-k: after the 32x32x2 MFMA chain lane l holds column (l&31) of rows (g&3)+8(g>>2)+4(l>>5),
// g = 0..15.  So one query row lives in one register across the 32 lanes of a half-wave; its candidate list
// lives the same way (E entries per lane, capacity 32E >= k).  A (t,g) slice is compared against the row's
// threshold with one v_cmp + ballot; only on a hit does the half-wave run the exact (value desc, index asc)
// replace-worst step.  Expected hits per row are O(k log(N/k)), so the MFMA pipe stays the bound.
#include "common.hpp"
#include "knn.hpp"
#include <mutex>

namespace osc {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr float NEG = -3.0e38f;
constexpr int BM = 128, BN = 128, BK = 32, LDT = 36;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// a is a worse list entry than b  <=>  smaller similarity, or equal similarity and larger index
__device__ __forceinline__ bool worse(float av, int ai, float bv, int bi) { return av < bv || (av == bv && ai > bi); }

__global__ __launch_bounds__(256) void k_normalize_rows(const float* Y, int32_t ldy, float* Yn, int32_t ldn, int64_t N,
                                                        int32_t D) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* y = Y + row * ldy;
  float* yn = Yn + row * ldn;
  float ss = 0.f;
  // the row stays in registers between the sum and the scaling (same sums, same order: one read of Y)
  auto in_registers = [&](auto NC) {
    constexpr int NV = decltype(NC)::value;
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      v[i] = c < D ? y[c] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < D) ss = fmaf(v[i], v[i], ss);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float inv = 1.f / (sqrtf(ss) + 1e-12f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < ldn) yn[c] = c < D ? v[i] * inv : 0.f;
    }
  };
  if (D <= 1024) {
    in_registers(std::integral_constant<int, 16>{});
    return;
  }
  if (D <= 2048) {
    in_registers(std::integral_constant<int, 32>{});
    return;
  }
  for (int c = lane; c < D; c += 64) ss = fmaf(y[c], y[c], ss);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float inv = 1.f / (sqrtf(ss) + 1e-12f);
  for (int c = lane; c < ldn; c += 64) yn[c] = c < D ? y[c] * inv : 0.f;
}

// out[i] = <Yn_i, q>  (cosine to a pre-normalised query; diffusion.py:104-107)
__global__ __launch_bounds__(256) void k_rows_dot(const float* Yn, int32_t ldn, const float* q, float* out, int64_t N,
                                                  int32_t D) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* y = Yn + row * ldn;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s = fmaf(y[c], q[c], s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) out[row] = s;
}

// out[i] = <A_i / (|A_i| + 1e-12), q> in one pass over the rows (q pre-normalised): the cosine of every row of an N x D
// array to one query without a normalised copy (diffusion.py:104-107, lattice.py:530-568, graph.py:114-133)
__global__ __launch_bounds__(256) void k_rows_cosine(const float* A, int32_t ld, const float* q, float* out, int64_t N,
                                                     int32_t D) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* y = A + row * ld;
  float s = 0.f, n2 = 0.f;
  const int d4 = D & ~3;  // 16 bytes per lane over the aligned part (ld is a multiple of 4), scalar tail
  for (int c = lane * 4; c < d4; c += 256) {
    const float4 v = ld4(y + c), w = ld4(q + c);
    s = fmaf(v.x, w.x, fmaf(v.y, w.y, fmaf(v.z, w.z, fmaf(v.w, w.w, s))));
    n2 = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, n2))));
  }
  for (int c = d4 + lane; c < D; c += 64) {
    const float v = y[c];
    s = fmaf(v, q[c], s);
    n2 = fmaf(v, v, n2);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    n2 += __shfl_xor(n2, o, 64);
  }
  if (lane == 0) out[row] = s / (sqrtf(n2) + 1e-12f);
}

// Sorted insert of the candidates flagged in (m0, m1) -- one bit per lane of half 0 / half 1 -- of register `c` (a query
// row per half-wave, 32 columns starting at `cbase`) into that row's register-resident list: one candidate per half
// per trip, rank by ballot + popcount, shift by v_mov_dpp wave_shr:1, carries by v_readlane; no LDS traffic.
template <int E>
__device__ __forceinline__ void list_insert(float (&lvg)[E], int (&lig)[E], float c, int cbase, unsigned m0, unsigned m1,
                                            int h, int l31) {
  while (m0 | m1) {
    const int s0 = m0 ? (__ffs(m0) - 1) : 0, s1 = m1 ? (__ffs(m1) - 1) : 0;
    const float cv0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), s0));
    const float cv1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 32 + s1));
    const float cv = h ? cv1 : cv0;
    const int cc = cbase + (h ? s1 : s0);
    // insertion rank = number of list entries that beat the candidate
    int p0 = 0, p1 = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const bool better = lvg[e] > cv || (lvg[e] == cv && lig[e] < cc);
      const unsigned long long bm = __ballot(better);
      p0 += __popc((unsigned)bm);
      p1 += __popc((unsigned)(bm >> 32));
    }
    if (!m0) p0 = 1 << 20;
    if (!m1) p1 = 1 << 20;
    const int p = h ? p1 : p0;
    // shift ranks > p down by one (v_mov_dpp wave_shr:1; the rank-32e slot takes the carry of register e-1)
#pragma unroll
    for (int e = E - 1; e >= 0; --e) {
      const int rank = l31 + 32 * e;
      float inv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(lvg[e]), 0x138, 0xf, 0xf, false));
      int ini = __builtin_amdgcn_update_dpp(0, lig[e], 0x138, 0xf, 0xf, false);
      if (e > 0) {
        const int pv = __float_as_int(lvg[e - 1]);
        const float c0 = __int_as_float(__builtin_amdgcn_readlane(pv, 31));
        const float c1 = __int_as_float(__builtin_amdgcn_readlane(pv, 63));
        const int i0 = __builtin_amdgcn_readlane(lig[e - 1], 31);
        const int i1 = __builtin_amdgcn_readlane(lig[e - 1], 63);
        if (l31 == 0) {
          inv = h ? c1 : c0;
          ini = h ? i1 : i0;
        }
      }
      if (rank > p) {
        lvg[e] = inv;
        lig[e] = ini;
      } else if (rank == p) {
        lvg[e] = cv;
        lig[e] = cc;
      }
    }
    m0 &= m0 - 1;  // drop the candidate just handled in each half
    m1 &= m1 - 1;
  }
}

// ---------------------------------------------------------------------------------------------
// work item = (row block of 128, column split s of S).  cand_*: [N][S][32E]
//
// F16 = true is the PREFILTER variant: Yn is the fp16 image of 16*Yn (two halfs per float slot, so tile staging and
// fragment addresses are byte-identical to the fp32 variant) and each (k16-step, tile) is ONE v_mfma_f32_32x32x16_f16
// instead of four fp32 MFMAs.  Its scores only choose candidates; exact fp32 re-scoring follows (k_knn_rescore).
// qrows != nullptr: the query rows are the nq rows listed there (per-row exact fallback), else rows are identity.
template <int E, bool F16, bool QR>
__device__ __forceinline__ void knn_topk_body(const float* __restrict__ Yn, int32_t ldn, int32_t N, int32_t k, int32_t S,
                                              int32_t cols_per_split, float* cand_val, int32_t* cand_idx,
                                              int32_t rb_begin, int32_t rb_count, const int32_t* __restrict__ qrows,
                                              int32_t nq) {
  __shared__ __attribute__((aligned(16))) float lds[2 * BM * LDT];
  float* As = lds;
  float* Bs = lds + BM * LDT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  // XCD-aware item order: blocks b and b+8 share an XCD (round-robin dispatch), so XCD x = b % 8 walks row blocks
  // x, x+8, ... and, for each, all S column splits back to back: the blocks resident on an XCD at one time share
  // a few 393 KB query panels, which then stay in that XCD's 4 MB L2 across all their column tiles.
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int rloc = (jx / S) * 8 + xcd, split = jx % S;
  const int rblk = rb_begin + rloc;
  const int nrows = QR ? nq : N;  // number of query rows
  if (rloc >= rb_count || rblk * BM >= nrows) return;
  const int row0 = rblk * BM;
  const int cbeg = split * cols_per_split;
  const int cend = min(N, cbeg + cols_per_split);
  const int nkt = ldn / BK;

  // candidate lists, kept SORTED (similarity desc, index asc): rank r = (lane & 31) + 32 e of row R(g,h) lives in
  // lv[g][e] of the lanes of half h.  thr[g] = similarity at rank k-1 (NEG until k candidates were seen).
  float lv[16][E];
  int li[16][E];
  float thr[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    thr[g] = NEG;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      lv[g][e] = NEG;
      li[g][e] = 0x7fffffff;
    }
  }
  const int thr_l = (k - 1) & 31;  // lane (within a half) and register holding rank k-1
  const int thr_e = (k - 1) >> 5;

  // staging: 128 rows x 32 floats per operand tile = 1024 float4; thread owns 4 of each
  int srow[4], sc4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int f = tid + 256 * q;
    srow[q] = f >> 3;
    sc4[q] = (f & 7) * 4;
  }
  const float* a_ptr[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int pos = min(row0 + srow[q], nrows - 1);
    a_ptr[q] = Yn + (size_t)(QR ? qrows[pos] : pos) * ldn + sc4[q];
  }

  const int wrow_base = row0 + 32 * wave;  // query position of this wave's local row 0
  // global row id of the query row register g of this half-wave stands for, local row (g&3)+8(g>>2)+4h
  auto grow_at = [&](int g) -> int {
    const int pos = wrow_base + (g & 3) + 8 * (g >> 2) + 4 * h;
    if (pos >= nrows) return -1;
    if constexpr (QR) return qrows[pos];
    else return pos;
  };

  for (int ct = cbeg; ct < cend; ct += BN) {
    const float* b_ptr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b_ptr[q] = Yn + (size_t)min(ct + srow[q], N - 1) * ldn + sc4[q];

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;

    float4 ra[4], rb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ra[q] = ld4(a_ptr[q]);
      rb[q] = ld4(b_ptr[q]);
    }
    for (int kt = 0; kt < nkt; ++kt) {
      __syncthreads();  // previous tile fully consumed
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        *reinterpret_cast<float4*>(As + srow[q] * LDT + sc4[q]) = ra[q];
        *reinterpret_cast<float4*>(Bs + srow[q] * LDT + sc4[q]) = rb[q];
      }
      __syncthreads();
      if (kt + 1 < nkt) {  // issue next tile's global loads; they land under the MFMAs below
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          ra[q] = ld4(a_ptr[q] + (kt + 1) * BK);
          rb[q] = ld4(b_ptr[q] + (kt + 1) * BK);
        }
      }
      const float* ap = As + (32 * wave + l31) * LDT + 4 * h;
      const float* bp = Bs + l31 * LDT + 4 * h;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4 av = ld4(ap + 8 * s);
        float4 bv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) bv[t] = ld4(bp + 32 * t * LDT + 8 * s);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if constexpr (F16) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, av), __builtin_bit_cast(half8, bv[t]),
                                                            acc[t], 0, 0, 0);
          } else {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv[t].x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv[t].y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv[t].z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv[t].w, acc[t], 0, 0, 0);
          }
        }
      }
    }

    // ---- running top-k update for this 32 x 128 slice -------------------------------------
    const bool need_mask = QR || (ct + BN > cend) || (ct < wrow_base + 32 && ct + BN > wrow_base);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int grow = grow_at(g);  // global query row of this half
      bool touched = false;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float c = acc[t][g];
        const int ccol = ct + 32 * t + l31;
        if (need_mask && (ccol >= cend || ccol == grow)) c = NEG;  // graph.py:37 (diag = -inf) and the ragged tail
        // prefilter scores only pick candidates (ties at the list boundary are covered by the margin): strict test
        const bool pred = (F16 ? (c > thr[g]) : (c >= thr[g])) && (c > NEG);
        unsigned long long m = __ballot(pred);
        unsigned m0 = (unsigned)m, m1 = (unsigned)(m >> 32);  // wave-uniform (SGPR) candidate masks of the two halves
        if (m0 | m1) {
          touched = true;
          list_insert<E>(lv[g], li[g], c, ct + 32 * t, m0, m1, h, l31);
        }
      }
      if (touched) {  // refresh the filter threshold = similarity at rank keep-1
        float src = lv[g][0];
#pragma unroll
        for (int e = 1; e < E; ++e)
          if (thr_e == e) src = lv[g][e];
        const float t0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(src), thr_l));
        const float t1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(src), 32 + thr_l));
        thr[g] = h ? t1 : t0;
      }
    }
  }

  // ---- write this item's candidate lists ---------------------------------------------------
  constexpr int KC = 32 * E;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int grow = grow_at(g);
    if (grow >= 0) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const size_t o = ((size_t)grow * S + split) * KC + l31 + 32 * e;
        cand_val[o] = lv[g][e];
        cand_idx[o] = li[g][e];
      }
    }
  }
}

template <int E, bool F16, bool QR>
__global__ __launch_bounds__(256, 2) void k_knn_topk(const float* __restrict__ Yn, int32_t ldn, int32_t N, int32_t k,
                                                     int32_t S, int32_t cols_per_split, float* cand_val,
                                                     int32_t* cand_idx, int32_t rb_begin, int32_t rb_count,
                                                     const int32_t* __restrict__ qrows, int32_t nq) {
  knn_topk_body<E, F16, QR>(Yn, ldn, N, k, S, cols_per_split, cand_val, cand_idx, rb_begin, rb_count, qrows, nq);
}
// ---- small lattices: dense similarity matrix + per-row selection ---------------------------------------------------
// For N <= 4096 the streaming top-k above spends its time inserting the first tile's 128 columns one by one into empty
// lists (N = 1200: 374 us of a 0.5 ms build).  Here one workgroup per (row block, column tile) writes its 128 x 128
// tile of S = Yn Yn^T (same staging, fragment order and MFMA sequence as knn_topk_body, so S is bitwise symmetric and
// bit-identical to what the streaming kernel scores), and one wave per row then picks the k best by repeated argmax.
__global__ __launch_bounds__(256, 2) void k_knn_dense(const float* __restrict__ Yn, int32_t ldn, int32_t N,
                                                      float* __restrict__ Sm, int32_t lds_, int32_t row_base) {
  __shared__ __attribute__((aligned(16))) float lds[2 * BM * LDT];
  float* As = lds;
  float* Bs = lds + BM * LDT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int row0 = row_base + blockIdx.y * BM, ct = blockIdx.x * BN;  // Sm holds the rows from row_base on
  const int nkt = ldn / BK;
  int srow[4], sc4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int f = tid + 256 * q;
    srow[q] = f >> 3;
    sc4[q] = (f & 7) * 4;
  }
  const float* a_ptr[4];
  const float* b_ptr[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    a_ptr[q] = Yn + (size_t)min(row0 + srow[q], N - 1) * ldn + sc4[q];
    b_ptr[q] = Yn + (size_t)min(ct + srow[q], N - 1) * ldn + sc4[q];
  }
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
  float4 ra[4], rb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ra[q] = ld4(a_ptr[q]);
    rb[q] = ld4(b_ptr[q]);
  }
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<float4*>(As + srow[q] * LDT + sc4[q]) = ra[q];
      *reinterpret_cast<float4*>(Bs + srow[q] * LDT + sc4[q]) = rb[q];
    }
    __syncthreads();
    if (kt + 1 < nkt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ra[q] = ld4(a_ptr[q] + (kt + 1) * BK);
        rb[q] = ld4(b_ptr[q] + (kt + 1) * BK);
      }
    }
    const float* ap = As + (32 * wave + l31) * LDT + 4 * h;
    const float* bp = Bs + l31 * LDT + 4 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float4 av = ld4(ap + 8 * s);
      float4 bv[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) bv[t] = ld4(bp + 32 * t * LDT + 8 * s);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv[t].x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv[t].y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv[t].z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv[t].w, acc[t], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int row = row0 + 32 * wave + (g & 3) + 8 * (g >> 2) + 4 * h;
    if (row >= N) continue;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int col = ct + 32 * t + l31;
      if (col < N) Sm[(size_t)(row - row_base) * lds_ + col] = acc[t][g];
    }
  }
}

// one wave per row: the k best columns by (similarity desc, index asc), diagonal excluded (graph.py:37, 46-49), values
// clipped at 0 (graph.py:62).  M = registers per lane (64 M >= N).
template <int M>
__global__ __launch_bounds__(256) void k_knn_select(const float* __restrict__ Sm, int32_t lds_, int32_t N, int32_t k,
                                                    float* out_val, int32_t* out_idx) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  float v[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int c = lane + 64 * m;
    v[m] = (c < N && c != row) ? Sm[(size_t)row * lds_ + c] : NEG;
  }
  for (int r = 0; r < k; ++r) {
    float bv = v[0];
    int bm = 0;
#pragma unroll
    for (int m = 1; m < M; ++m)
      if (v[m] > bv) {  // strict: equal values keep the smaller index (smaller m)
        bv = v[m];
        bm = m;
      }
    int bi = lane + 64 * bm;
    float wv = bv;
    int wi = bi;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(wv, o, 64);
      const int oi = __shfl_xor(wi, o, 64);
      if (ov > wv || (ov == wv && oi < wi)) {
        wv = ov;
        wi = oi;
      }
    }
    if (!(wv > NEG)) break;  // nothing left (cannot happen for k <= N - 1)
    if (wi == bi) {  // this lane held the winner: retire it
#pragma unroll
      for (int m = 0; m < M; ++m)
        if (m == bm) v[m] = NEG;
    }
    if (lane == 0) {
      out_val[(size_t)row * k + r] = fmaxf(wv, 0.f);
      out_idx[(size_t)row * k + r] = wi;
    }
  }
}


// ---- any k (k > 128): rows of the dense similarity matrix + radix select ------------------------------------------
// One workgroup per row of Sm (row `row_base + blockIdx.x` of the lattice, N columns).  The k best columns in the order
// (similarity desc, index asc), diagonal excluded: four 8-bit histogram passes over the order-preserving integer image
// of the row find the key T of the k-th best entry and how many entries are strictly better; one ordered pass then
// emits the better entries and, among the entries equal to T, the smallest indices until k are out (exact ties go to
// the smaller index, graph.py:46-49).  Output lists are in index order within a row (consumers only need membership
// and values).  Values clipped at 0 (graph.py:62).
__device__ __forceinline__ uint32_t sim_key(float v) {  // ascending float order == ascending key order
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int NT>
__global__ __launch_bounds__(NT) void k_knn_select_any(const float* __restrict__ Sm, int32_t lds_, int32_t N, int32_t k,
                                                        int32_t row_base, int32_t rows, float* out_val, int32_t* out_idx,
                                                        const int32_t* __restrict__ qrows) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t s_prefix, s_need, s_wave[NT / 64][2], s_base[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((int)blockIdx.x >= rows) return;
  const int row = qrows ? qrows[blockIdx.x] : row_base + (int)blockIdx.x;  // qrows: Sm row b scores lattice row qrows[b]
  const float* srow = Sm + (size_t)blockIdx.x * lds_;
  // radix select of the k-th largest key among the N - 1 off-diagonal entries
  uint32_t prefix = 0, need = (uint32_t)k;  // entries still to be taken from the current prefix class
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const uint32_t mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int c = tid; c < N; c += NT) {
      if (c == row) continue;
      const uint32_t key = sim_key(srow[c]);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t left = need;
      int b = 255;
      for (; b > 0; --b) {
        if (hist[b] >= left) break;
        left -= hist[b];
      }
      s_prefix = prefix | ((uint32_t)b << shift);
      s_need = left;  // entries to take from bin b (the bins above it are taken whole)
    }
    __syncthreads();
    prefix = s_prefix;
    need = s_need;
    __syncthreads();
  }
  const uint32_t T = prefix;        // key of the k-th best entry
  const uint32_t take_eq = need;    // how many entries with key == T belong to the list (smallest indices first)
  const uint32_t n_gt = (uint32_t)k - take_eq;
  // ordered emission: the list comes out sorted by column index (what k_mutual_ell_sorted's binary search needs); of the
  // entries equal to T only the first take_eq in index order are emitted
  (void)n_gt;
  if (tid < 2) s_base[tid] = 0;
  __syncthreads();
  float* ov = out_val + (size_t)row * k;
  int32_t* oi = out_idx + (size_t)row * k;
  for (int c0 = 0; c0 < N; c0 += NT) {
    const int c = c0 + tid;
    float v = 0.f;
    uint32_t key = 0;
    const bool valid = c < N && c != row;
    if (valid) {
      v = srow[c];
      key = sim_key(v);
    }
    const bool gt = valid && key > T, eq = valid && key == T;
    const unsigned long long bg = __ballot(gt), be = __ballot(eq);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const uint32_t pg = (uint32_t)__popcll(bg & below), pe = (uint32_t)__popcll(be & below);
    if (lane == 0) {
      s_wave[wave][0] = (uint32_t)__popcll(bg);
      s_wave[wave][1] = (uint32_t)__popcll(be);
    }
    __syncthreads();
    uint32_t og = s_base[0], oe = s_base[1];  // better / equal entries before this thread's column
    for (int w = 0; w < wave; ++w) {
      og += s_wave[w][0];
      oe += s_wave[w][1];
    }
    og += pg;
    oe += pe;
    if (gt || (eq && oe < take_eq)) {
      const uint32_t pos = og + (oe < take_eq ? oe : take_eq);
      ov[pos] = fmaxf(v, 0.f);
      oi[pos] = c;
    }
    __syncthreads();
    if (tid == 0) {
      for (int w2 = 0; w2 < NT / 64; ++w2) {
        s_base[0] += s_wave[w2][0];
        s_base[1] += s_wave[w2][1];
      }
    }
    __syncthreads();
  }
}

// k in (64, 128]: 128 list registers per lane -> one wave per SIMD with the whole 512-entry register file
template <bool QR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_knn_topk_wide(
    const float* __restrict__ Yn, int32_t ldn, int32_t N, int32_t k, int32_t S, int32_t cols_per_split, float* cand_val,
    int32_t* cand_idx, int32_t rb_begin, int32_t rb_count, const int32_t* __restrict__ qrows, int32_t nq) {
  knn_topk_body<4, false, QR>(Yn, ldn, N, k, S, cols_per_split, cand_val, cand_idx, rb_begin, rb_count, qrows, nq);
}

// ---- prefilter kernel (fp16 MFMA), LDS-DMA staged -------------------------------------------------------------------
// Same work decomposition, fragment mapping and register-resident lists as knn_topk_body<E, true, false>, different
// staging pipeline:
//   * operand tiles (128 rows x 128 B = one 64-half K step) go global -> LDS by global_load_lds_dwordx4 (no staging
//     registers, no ds_write pass).  The LDS image is lane-linear per wave-instruction (8 rows x 128 B), so the bank
//     swizzle sits on the SOURCE address: 16-byte chunk c of row r is stored at chunk position c ^ ((r >> 1) & 7), and
//     the fragment reads apply the same xor (conflict-free for ds_read_b128's 16-lane groups, no padding).
//   * two LDS stages (64 KB per workgroup, two workgroups per CU), ONE barrier per K step; the (column tile, K step)
//     loop is flat, so the first K step of the next column tile is in flight during the last MFMAs and the list update
//     of the current one.
//   * list update: one v_max3/v_max + compare per query-row register decides whether any of its 4 x 32 columns can
//     enter the list before the four per-tile ballots are taken.
// Two shapes of the same body: <E, 4, 2> = 128 query rows, two stages, 64 KB, two workgroups per CU (default);
// <E, 8, 3> = 256 query rows (half the B-tile traffic per MFMA), three stages (two K steps of DMA in flight across
// each barrier, retired by a counted vmcnt), 144 KB, one workgroup per CU.

template <int E, int NW, int NST>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void k_knn_pref(const float* __restrict__ Yh, int32_t ldn,
                                                                       int32_t N, int32_t k, int32_t S,
                                                                       int32_t cols_per_split, float* cand_val,
                                                                       int32_t* cand_idx, int32_t rb_begin,
                                                                       int32_t rb_count) {
  constexpr int BMX = 32 * NW;                 // query rows per workgroup
  constexpr int PF_STAGE = (BMX + BN) * BK;    // floats per stage: A tile (BMX rows) + B tile (128 rows), 32 slots each
  constexpr int RB = NW / 4;                   // 128-row blocks of the plan per workgroup
  extern __shared__ __attribute__((aligned(1024))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;  // XCD-aware item order (see knn_topk_body)
  const int rloc = ((jx / S) * 8 + xcd) * RB, split = jx % S;
  const int rblk = rb_begin + rloc;
  if (rloc >= rb_count || rblk * BM >= N) return;
  const int row0 = rblk * BM;
  const int row_limit = min(N, (rb_begin + rb_count) * BM);  // rows past the plan's range belong to another pass
  const int cbeg = split * cols_per_split;
  const int cend = min(N, cbeg + cols_per_split);
  const int nkt = ldn / BK;
  const int ntile = (cend - cbeg + BN - 1) / BN;
  if (ntile <= 0) return;

  float lv[16][E];
  int li[16][E];
  float thr[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    thr[g] = NEG;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      lv[g][e] = NEG;
      li[g][e] = 0x7fffffff;
    }
  }
  const int thr_l = (k - 1) & 31, thr_e = (k - 1) >> 5;

  // LDS-DMA roles: wave w fills rows [32 w, 32 w + 32) of both tiles, 8 rows (1 KiB) per instruction.  Lane L of an
  // instruction lands at row L >> 3, chunk position L & 7, and therefore fetches source chunk (L & 7) ^ (row & 7).
  // Bank swizzle.  ds_read_b128 is served in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) over a
  // 256-byte bank row (MI355X_MICROARCH.md, LDS): with 128-byte tile rows the 16-byte slot of (row, chunk position p) is
  // (row & 1) * 8 + p, so the eight even and the eight odd rows of a group must get eight different positions for one
  // source chunk.  p = chunk ^ ((row >> 1) & 7) does that for both groups (row >> 1 mod 8 runs over 0..7 on the even
  // rows and on the odd rows of each); the former chunk ^ (row & 7) was 2-way conflicted (SQ_LDS_BANK_CONFLICT = half
  // of SQ_LDS_IDX_ACTIVE).
  const int frow = lane >> 3;                       // row inside the 8-row piece (pieces start at multiples of 8)
  // piece q of a wave starts at tile row 8 q (mod 16): (row >> 1) & 7 = ((q & 1) << 2) | (frow >> 1)
  auto fchunk_of = [&](int q) -> int { return ((lane & 7) ^ (((q & 1) << 2) | (frow >> 1))) * 4; };  // source offset in floats
  const float* a_src[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) a_src[q] = Yh + (size_t)min(row0 + 32 * wave + 8 * q + frow, N - 1) * ldn + fchunk_of(q);
  constexpr int BQ = 16 / NW;  // 8-row pieces of the B tile per wave (4 waves: 4, 8 waves: 2)

  // The DMA is issued from inline asm: through the builtin the compiler orders every later ds_read behind it with
  // s_waitcnt vmcnt(0) (it cannot see that the reads go to the other stage), which would expose the whole load latency
  // in every K step.  The wait that retires the DMA is the explicit vmcnt(0) in front of the step's barrier.
  const unsigned lds_base = (unsigned)(size_t)lds;  // LDS byte offset of the staging area (low half of the flat address)
  auto glds16 = [&](const float* src, unsigned dst_bytes) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(dst_bytes)
                 : "memory");
  };
  auto issue = [&](int stage, int ct, int kt) {  // one K step of tile `ct` into LDS stage `stage`
    const unsigned a_dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(stage * PF_STAGE + 32 * wave * BK) * 4u);
    const unsigned b_dst =
        __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(stage * PF_STAGE + BMX * BK + 8 * BQ * wave * BK) * 4u);
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(a_src[q] + kt * BK, a_dst + (unsigned)(8 * q * BK) * 4u);
#pragma unroll
    for (int q = 0; q < BQ; ++q) {
      const float* bsrc = Yh + (size_t)min(ct + 8 * BQ * wave + 8 * q + frow, N - 1) * ldn + fchunk_of(q) + kt * BK;
      glds16(bsrc, b_dst + (unsigned)(8 * q * BK) * 4u);
    }
  };
  constexpr int DMA_PER_STEP = 4 + BQ;  // LDS-DMA instructions one wave issues per K step
  // retire the DMA of the step that is read next; with three stages the step after that stays in flight across the
  // barrier (counted vmcnt: the wave's youngest DMA_PER_STEP operations) unless nothing was issued behind it
  auto dma_wait_and_barrier = [&](bool keep_one_step) {
    if (NST == 3 && keep_one_step) {
      if constexpr (DMA_PER_STEP == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  };

  const int wrow_base = row0 + 32 * wave;
  auto grow_at = [&](int g) -> int {
    const int pos = wrow_base + (g & 3) + 8 * (g >> 2) + 4 * h;
    return pos < row_limit ? pos : -1;
  };
  const int swz = (l31 >> 1) & 7;  // ((row >> 1) & 7) of every fragment row this lane reads (tile bases are multiples of 32)

  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;

  int ct = cbeg, kt = 0;       // the step being computed
  const int total = ntile * nkt;
  // (ict, ikt) = the next step to issue; the pipeline runs NST - 1 steps ahead of the compute
  int ict = cbeg, ikt = 0, issued = 0;
  auto issue_next = [&]() {
    issue(issued % NST, ict, ikt);
    ++issued;
    if (++ikt == nkt) {
      ikt = 0;
      ict += BN;
    }
  };
  for (int pre = 0; pre < NST - 1 && issued < total; ++pre) issue_next();
  dma_wait_and_barrier(issued > 1);
  for (int step = 0; step < total; ++step) {
    const int stage = step % NST;
    // prefetch into the stage whose readers finished before the barrier that ended step-1
    bool issued_now = false;
    if (issued < total) {
      issue_next();
      issued_now = true;
    }
    const float* Asw = lds + stage * PF_STAGE + (32 * wave + l31) * BK;
    const float* Bsw = lds + stage * PF_STAGE + BMX * BK + l31 * BK;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int co = (((2 * s + h) ^ swz)) * 4;
      const float4 av = ld4(Asw + co);
      float4 bv[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) bv[t] = ld4(Bsw + 32 * t * BK + co);
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, av), __builtin_bit_cast(half8, bv[t]),
                                                        acc[t], 0, 0, 0);
    }
    if (kt == nkt - 1) {  // ---- tile finished: running top-k update for this 32 x 128 slice ----
      const bool need_mask = (ct + BN > cend) || (ct < wrow_base + 32 && ct + BN > wrow_base);
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        float c4[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) c4[t] = acc[t][g];
        if (need_mask) {
          const int grow = grow_at(g);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int ccol = ct + 32 * t + l31;
            if (ccol >= cend || ccol == grow) c4[t] = NEG;  // graph.py:37 (diag = -inf) and the ragged tail
          }
        }
        const float cmax = fmaxf(fmaxf(c4[0], c4[1]), fmaxf(c4[2], c4[3]));
        if (__ballot(cmax > thr[g]) != 0ull) {  // some column of this row (either half) may enter its list
          bool touched = false;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const unsigned long long m = __ballot(c4[t] > thr[g] && c4[t] > NEG);
            const unsigned m0 = (unsigned)m, m1 = (unsigned)(m >> 32);
            if (m0 | m1) {
              touched = true;
              list_insert<E>(lv[g], li[g], c4[t], ct + 32 * t, m0, m1, h, l31);
              // later tiles of this register see the tightened threshold too
              float src = lv[g][0];
#pragma unroll
              for (int e = 1; e < E; ++e)
                if (thr_e == e) src = lv[g][e];
              const float t0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(src), thr_l));
              const float t1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(src), 32 + thr_l));
              thr[g] = h ? t1 : t0;
            }
          }
          (void)touched;
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
      kt = 0;
      ct += BN;
    } else {
      ++kt;
    }
    dma_wait_and_barrier(issued_now);  // next stage landed and every wave is done reading this one
  }

  constexpr int KC = 32 * E;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int grow = grow_at(g);
    if (grow >= 0) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const size_t o = ((size_t)grow * S + split) * KC + l31 + 32 * e;
        cand_val[o] = lv[g][e];
        cand_idx[o] = li[g][e];
      }
    }
  }
}

// fp16 image of 16*Yn for the prefilter (|Yn| <= 1, so no overflow; the scale keeps small entries out of the
// fp16 subnormal range).  Row pitch ldh halfs (multiple of 64), zero padded.
__global__ __launch_bounds__(256) void k_to_f16(const float* Yn, int32_t ldn, _Float16* Yh, int32_t ldh, int64_t N,
                                                int32_t D) {
  const int64_t row = blockIdx.x;
  for (int c = threadIdx.x; c < ldh; c += 256)
    Yh[row * ldh + c] = c < D ? (_Float16)(16.0f * Yn[row * ldn + c]) : (_Float16)0.0f;
}

// The re-scoring's selection and proof, shared by its one-launch and its two-launch (pair) form: lane l holds candidate slots
// l and l + 64 (exact score sc, lattice id; id < 0: no candidate); writes the k best by (score desc, index asc), clipped at
// 0, and queues the row for the exact kernel unless "every left-out column has fp16 score <= the list's last" proves it.
__device__ __forceinline__ void rescore_select_and_prove(const float (&sc)[2], const int (&id)[2], int nvalid, int lane, int row, int32_t KC,
                                                         int32_t k, float delta, const float* cval, float* out_val, int32_t* out_idx,
                                                         int32_t* fail_rows, int32_t* fail_count) {
  // (only the KC slots in use are visited -- both callers fill slots q < KC and nothing else: at 48 candidates 48 rounds on one
  // register instead of 128 on two, k_knn_rescore_finish 0.29 -> 0.1 ms at config 3)
  int rank[2] = {0, 0};
  const bool two = KC > 64;  // (wave-uniform)
#pragma unroll
  for (int m2 = 0; m2 < 2; ++m2) {
    const int nl = m2 == 0 ? min(KC, 64) : KC - 64;
    for (int l = 0; l < nl; ++l) {
      const float ov = __shfl(sc[m2], l, 64);
      const int oi = __shfl(id[m2], l, 64);
      if (oi < 0) continue;
      if (id[0] >= 0 && (ov > sc[0] || (ov == sc[0] && oi < id[0]))) ++rank[0];
      if (two && id[1] >= 0 && (ov > sc[1] || (ov == sc[1] && oi < id[1]))) ++rank[1];
    }
  }
  float tk = NEG;  // exact k-th best score
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    if (id[m] >= 0 && rank[m] < k) {
      out_val[(size_t)row * k + rank[m]] = fmaxf(sc[m], 0.f);
      out_idx[(size_t)row * k + rank[m]] = id[m];
    }
    if (id[m] >= 0 && rank[m] == k - 1) tk = sc[m];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tk = fmaxf(tk, __shfl_xor(tk, o, 64));
  if (lane == 0 && nvalid == KC) {  // a full list: columns were left out, prove none of them belongs
    const float vlast = cval[(size_t)row * KC + KC - 1] * (1.0f / 256.0f);
    if (!(vlast + delta < tk)) fail_rows[atomicAdd(fail_count, 1)] = row;
  }
}

// Exact fp32 re-scoring of the prefilter's candidates, one wave per row.
//   score(i,j) = sum_k Yn_i[k] Yn_j[k], lanes stride k, butterfly sum: symmetric in (i,j) bit for bit.
//   final list = best k by (score desc, index asc), clipped at 0 (graph.py:46-49, 62).
//   verify: every column outside the candidate list has prefilter score <= v_last, hence exact score
//   <= v_last/256 + delta; the row is safe iff that is < the exact k-th score.  Unsafe rows are queued for the
//   exact kernel (fail_rows / fail_count).
// NCH = float4 chunks per lane covering a row (ldn <= 256 NCH): the query row stays in registers for all its candidates
// and EU candidates are gathered together (independent sums and butterflies) -- the first version re-read the query row
// per candidate through a thrashing L1 (28 GB fetched for 14.7 GB of candidate rows) and paid one exposed gather latency
// per candidate.  Per-candidate arithmetic is unchanged (same products, same order), so scores are bit-identical.
// NCH = 0: generic loop for wider rows.
template <int NCH>
__global__ __launch_bounds__(256) void k_knn_rescore(const float* __restrict__ Yn, int32_t ldn, int32_t D, int32_t N,
                                                     int32_t row_begin, int32_t row_end, const int32_t* cidx,
                                                     const float* cval, int32_t KC, int32_t k, float delta,
                                                     float* out_val, int32_t* out_idx, int32_t* fail_rows,
                                                     int32_t* fail_count, int32_t mapped, KnnRowMap map) {
  constexpr int EU = 4;  // candidates in flight
  const int lane = threadIdx.x & 63;
  // [row_begin, row_end) are rows of the prefilter IMAGE when `mapped` (KnnRowMap: which lattice row an image row holds):
  // a rank of a sharded build re-scores the lattice rows of its image row blocks
  const int irow = row_begin + blockIdx.x * 4 + (threadIdx.x >> 6);
  if (irow >= row_end) return;
  const int row = mapped ? knn_map_lattice_row(map, N, irow) : irow;
  const float* yi = Yn + (size_t)row * ldn;
  float4 yr[NCH > 0 ? NCH : 1];
  if constexpr (NCH > 0) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c = lane * 4 + ch * 256;
      yr[ch] = c < ldn ? ld4(yi + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  auto partial = [&](int j) -> float {  // this lane's share of <Yn_row, Yn_j>
    const float* yj = Yn + (size_t)j * ldn;
    float s = 0.f;
    if constexpr (NCH > 0) {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int c = lane * 4 + ch * 256;
        if (c < ldn) {
          const float4 a = yr[ch], b = ld4(yj + c);
          s = fmaf(a.x, b.x, s);
          s = fmaf(a.y, b.y, s);
          s = fmaf(a.z, b.z, s);
          s = fmaf(a.w, b.w, s);
        }
      }
    } else {
      for (int c = lane * 4; c < ldn; c += 256) {  // rows are zero-padded to ldn (multiple of 32 floats)
        const float4 a = ld4(yi + c), b = ld4(yj + c);
        s = fmaf(a.x, b.x, s);
        s = fmaf(a.y, b.y, s);
        s = fmaf(a.z, b.z, s);
        s = fmaf(a.w, b.w, s);
      }
    }
    return s;
  };
  float sc[2] = {NEG, NEG};   // exact score of candidate slot lane + 64 m   (KC <= 128)
  int id[2] = {-1, -1};
  int nvalid = 0;
  for (int q0 = 0; q0 < KC; q0 += EU) {
    int jj[EU];
    float ss[EU];
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int q = q0 + u;
      jj[u] = q < KC ? cidx[(size_t)row * KC + q] : -1;
      if (jj[u] >= N) jj[u] = -1;  // uniform
    }
#pragma unroll
    for (int u = 0; u < EU; ++u) ss[u] = jj[u] >= 0 ? partial(jj[u]) : 0.f;
#pragma unroll
    for (int u = 0; u < EU; ++u) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ss[u] += __shfl_xor(ss[u], o, 64);
    }
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int q = q0 + u;
      if (jj[u] < 0) continue;
      ++nvalid;
      if (lane == (q & 63)) {
        if (q < 64) { sc[0] = ss[u]; id[0] = jj[u]; } else { sc[1] = ss[u]; id[1] = jj[u]; }
      }
    }
  }
  rescore_select_and_prove(sc, id, nvalid, lane, row, KC, k, delta, cval, out_val, out_idx, fail_rows, fail_count);
}

// ---- the same re-scoring with every undirected candidate pair scored ONCE (round 6) ------------------------------------------
// score(i, j) is symmetric bit for bit (above), and most candidate pairs are mutual -- j is a candidate of i and i one of j:
// the one-launch kernel gathers Yn_j for row i AND Yn_i for row j (config 3: 48 x 3 KB per row, 14.7 GB at the HBM ceiling;
// config 5: 96 x 6 KB, 118 GB, 16 of a 98 ms build).  Here row i scores candidate j itself unless j < i and i stands in j's
// list -- then row j scores the pair (for it i > j) and row i only notes where: pos[i][q] = j KC + (i's slot in j's list).  A
// lookup reads j's list (KC ints: 192 B against a 3 KB row).  Launch 1 (k_knn_rescore_pair) writes the scores it computes
// and the notes; launch 2 (k_knn_rescore_finish) gives every slot its score -- its own or the noted one -- and runs the
// one-launch kernel's selection and proof on them.  Single-process builds only: every row's list must be at hand.
template <int NCH>
__global__ __launch_bounds__(256) void k_knn_rescore_pair(const float* __restrict__ Yn, int32_t ldn, int32_t N, int32_t row_begin,
                                                          int32_t row_end, const int32_t* __restrict__ cidx, int32_t KC,
                                                          float* __restrict__ sc_g, int32_t* __restrict__ pos_g, int32_t mapped,
                                                          KnnRowMap map) {
  constexpr int EU = 4;  // candidates in flight
  const int lane = threadIdx.x & 63;
  const int irow = row_begin + blockIdx.x * 4 + (threadIdx.x >> 6);
  if (irow >= row_end) return;
  const int row = mapped ? knn_map_lattice_row(map, N, irow) : irow;
  const float* yi = Yn + (size_t)row * ldn;
  float4 yr[NCH > 0 ? NCH : 1];
  if constexpr (NCH > 0) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c = lane * 4 + ch * 256;
      yr[ch] = c < ldn ? ld4(yi + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  auto partial = [&](int j) -> float {  // this lane's share of <Yn_row, Yn_j>: k_knn_rescore's, term for term
    const float* yj = Yn + (size_t)j * ldn;
    float s = 0.f;
    if constexpr (NCH > 0) {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int c = lane * 4 + ch * 256;
        if (c < ldn) {
          const float4 a = yr[ch], b = ld4(yj + c);
          s = fmaf(a.x, b.x, s);
          s = fmaf(a.y, b.y, s);
          s = fmaf(a.z, b.z, s);
          s = fmaf(a.w, b.w, s);
        }
      }
    } else {
      for (int c = lane * 4; c < ldn; c += 256) {
        const float4 a = ld4(yi + c), b = ld4(yj + c);
        s = fmaf(a.x, b.x, s);
        s = fmaf(a.y, b.y, s);
        s = fmaf(a.z, b.z, s);
        s = fmaf(a.w, b.w, s);
      }
    }
    return s;
  };
  for (int q0 = 0; q0 < KC; q0 += EU) {
    int jj[EU], look0[EU], look1[EU], at[EU];
    float ss[EU];
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int q = q0 + u;
      jj[u] = q < KC ? cidx[(size_t)row * KC + q] : -1;
      if (jj[u] >= N) jj[u] = -1;  // uniform
    }
#pragma unroll
    for (int u = 0; u < EU; ++u) {  // the lists of the candidates below this row: all loads first
      look0[u] = look1[u] = -2;
      if (jj[u] >= 0 && jj[u] < row) {
        if (lane < KC) look0[u] = cidx[(size_t)jj[u] * KC + lane];
        if (lane + 64 < KC) look1[u] = cidx[(size_t)jj[u] * KC + lane + 64];
      }
    }
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      at[u] = -1;
      if (jj[u] >= 0 && jj[u] < row) {
        const unsigned long long b0 = __ballot(look0[u] == row), b1 = __ballot(look1[u] == row);
        if (b0 != 0ull) at[u] = __ffsll((long long)b0) - 1;
        else if (b1 != 0ull) at[u] = 64 + __ffsll((long long)b1) - 1;
      }
    }
#pragma unroll
    for (int u = 0; u < EU; ++u) ss[u] = (jj[u] >= 0 && at[u] < 0) ? partial(jj[u]) : 0.f;
#pragma unroll
    for (int u = 0; u < EU; ++u) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ss[u] += __shfl_xor(ss[u], o, 64);
    }
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int q = q0 + u;
      if (jj[u] < 0 || lane != 0) continue;
      if (at[u] >= 0) {
        pos_g[(size_t)row * KC + q] = jj[u] * KC + at[u];  // (N KC < 2^31: checked by the launcher)
      } else {
        pos_g[(size_t)row * KC + q] = -1;
        sc_g[(size_t)row * KC + q] = ss[u];
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_knn_rescore_finish(int32_t N, int32_t row_begin, int32_t row_end, const int32_t* __restrict__ cidx,
                                                            const float* __restrict__ cval, int32_t KC, int32_t k, float delta,
                                                            const float* __restrict__ sc_g, const int32_t* __restrict__ pos_g,
                                                            float* out_val, int32_t* out_idx, int32_t* fail_rows, int32_t* fail_count,
                                                            int32_t mapped, KnnRowMap map) {
  const int lane = threadIdx.x & 63;
  const int irow = row_begin + blockIdx.x * 4 + (threadIdx.x >> 6);
  if (irow >= row_end) return;
  const int row = mapped ? knn_map_lattice_row(map, N, irow) : irow;
  float sc[2] = {NEG, NEG};
  int id[2] = {-1, -1};
  int nvalid = 0;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int q = lane + 64 * m;
    bool ok = false;
    if (q < KC) {
      const int j = cidx[(size_t)row * KC + q];
      if (j >= 0 && j < N) {
        const int at = pos_g[(size_t)row * KC + q];
        id[m] = j;
        sc[m] = sc_g[at < 0 ? (size_t)row * KC + q : (size_t)at];
        ok = true;
      }
    }
    nvalid += __popcll(__ballot(ok));
  }
  rescore_select_and_prove(sc, id, nvalid, lane, row, KC, k, delta, cval, out_val, out_idx, fail_rows, fail_count);
}

// ---- exact scores of a FEW query rows against all columns (the prefilter routes' fallback when only a handful of rows
// could not be proven): a wave keeps up to four query rows in registers and streams over its share of the columns; the
// per-lane partial sums and the butterfly are those of k_knn_rescore, so the scores are bit-identical to the re-scored
// rows'.  Sm[q][j] = <Yn_qrows[q], Yn_j>; k_knn_select_any then picks each row's k best.
template <int NCH>
__global__ __launch_bounds__(256) void k_rows_scores(const float* __restrict__ Yn, int32_t ldn, int32_t N,
                                                     const int32_t* __restrict__ qrows, int32_t nq, float* __restrict__ Sm,
                                                     int32_t lds_) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  const int q0 = blockIdx.y * 4;
  float4 yq[4][NCH];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int q = min(q0 + u, nq - 1);
    const float* yi = Yn + (size_t)qrows[q] * ldn;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c = lane * 4 + ch * 256;
      yq[u][ch] = c < ldn ? ld4(yi + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  for (int j = wid; j < N; j += nw) {
    const float* yj = Yn + (size_t)j * ldn;
    float4 b[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c = lane * 4 + ch * 256;
      b[ch] = c < ldn ? ld4(yj + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float ss[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float s = 0.f;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int c = lane * 4 + ch * 256;
        if (c < ldn) {
          const float4 a = yq[u][ch];
          s = fmaf(a.x, b[ch].x, s);
          s = fmaf(a.y, b[ch].y, s);
          s = fmaf(a.z, b[ch].z, s);
          s = fmaf(a.w, b[ch].w, s);
        }
      }
      ss[u] = s;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ss[u] += __shfl_xor(ss[u], o, 64);
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (q0 + u < nq) Sm[(size_t)(q0 + u) * lds_ + j] = ss[u];
    }
  }
}

// one wave per row: rank-select the k best of the S*KC candidates -> sorted (sim desc, idx asc), clipped at 0
__global__ __launch_bounds__(256) void k_knn_merge(const float* cand_val, const int32_t* cand_idx, int32_t ncand,
                                                   int32_t row_begin, int32_t N, int32_t k, float* out_val,
                                                   int32_t* out_idx, int32_t clip, const int32_t* __restrict__ qrows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* sv = reinterpret_cast<float*>(smem) + (size_t)wave * ncand;
  int32_t* si = reinterpret_cast<int32_t*>(smem + (size_t)4 * ncand * sizeof(float)) + (size_t)wave * ncand;
  int row = row_begin + blockIdx.x * 4 + wave;  // N = end of the row range (or of the qrows list)
  const bool live = row < N;
  if (live && qrows) row = qrows[row];
  if (live) {
    for (int c = lane; c < ncand; c += 64) {
      sv[c] = cand_val[(size_t)row * ncand + c];
      si[c] = cand_idx[(size_t)row * ncand + c];
    }
  }
  __syncthreads();
  if (!live) return;
  for (int c = lane; c < ncand; c += 64) {
    const float v = sv[c];
    const int i = si[c];
    if (!(v > NEG && v < 3.0e38f)) continue;  // empty or dead slot
    int rank = 0;
    for (int d = 0; d < ncand; ++d) {
      const float dv = sv[d];
      const int di = si[d];
      if (dv < 3.0e38f && (dv > v || (dv == v && di < i))) ++rank;
    }
    if (rank < k) {
      out_val[(size_t)row * k + rank] = clip ? fmaxf(v, 0.f) : v;  // graph.py:62
      out_idx[(size_t)row * k + rank] = i;
    }
  }
}

// one wave per row: keep (i,j) iff j in topk(i), i in topk(j), both sims > 0; weight = max (graph.py:64-65)
__global__ __launch_bounds__(256) void k_mutual_ell(const float* kval, const int32_t* kidx, int32_t N, int32_t k,
                                                    int32_t width, int32_t* ell_col, float* ell_a, int32_t* deg) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  int total = 0;
  // k <= 128: two passes of 64 entries; ranks computed against all kept entries of both passes.
  // The back-edge test "is `row` in the list of j?" is done by the WAVE for one neighbour at a time: 64 lanes read j's
  // list in one coalesced load (two for k > 64) and a ballot finds the slot.  Before round 4 every lane scanned the list of
  // ITS neighbour with 16-byte loads -- 64 different lines per wave instruction, k / 4 instructions: 1024 line requests per
  // row at k = 64, 7.5 ms of config 5's build; this form makes 128.
  int jcol[2], bpos[2];
  float jw[2], myv[2];
  bool keep[2];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int e = lane + 64 * ps;
    keep[ps] = false;
    jcol[ps] = 0x7fffffff;
    jw[ps] = 0.f;
    myv[ps] = 0.f;
    bpos[ps] = -1;
    if (e < k) {
      const int j = kidx[(size_t)row * k + e];
      const float v = kval[(size_t)row * k + e];
      if (v > 0.f && j >= 0 && j < N) {
        jcol[ps] = j;
        myv[ps] = v;
      }
    }
  }
  const int npass = k > 64 ? 2 : 1;
  for (int ps = 0; ps < npass; ++ps) {
    const int ne = min(64, k - 64 * ps);
    for (int e0 = 0; e0 < ne; e0 += 4) {  // four neighbours' lists in flight
      int jj[4], c0[4], c1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        jj[u] = e0 + u < ne ? __shfl(jcol[ps], e0 + u, 64) : 0x7fffffff;
        c0[u] = c1[u] = -1;
        if (jj[u] != 0x7fffffff) {
          const int32_t* lj = kidx + (size_t)jj[u] * k;
          if (lane < k) c0[u] = lj[lane];
          if (lane + 64 < k) c1[u] = lj[lane + 64];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (jj[u] == 0x7fffffff) continue;  // (wave-uniform)
        const unsigned long long m0 = __ballot(c0[u] == row), m1 = __ballot(c1[u] == row);
        const int pos = m0 ? __ffsll((long long)m0) - 1 : m1 ? 64 + __ffsll((long long)m1) - 1 : -1;
        if (lane == e0 + u) bpos[ps] = pos;
      }
    }
  }
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    if (bpos[ps] >= 0) {  // the back edge's similarity: one gather per lane
      const float back = kval[(size_t)jcol[ps] * k + bpos[ps]];
      if (back > 0.f) {
        keep[ps] = true;
        jw[ps] = fmaxf(myv[ps], back);
      }
    }
    if (!keep[ps]) jcol[ps] = 0x7fffffff;
  }
  // (a kept column's slot = the number of kept columns below it; slots e >= k hold 0x7fffffff and are not visited: k rounds
  // instead of 256 per row)
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    int rank = 0;
    if (ps < npass) {
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        const int nl = min(64, k - 64 * qs);
        for (int l = 0; l < nl; ++l) {
          const int oc = __shfl(jcol[qs], l, 64);
          rank += (oc < jcol[ps]) ? 1 : 0;
        }
      }
    }
    if (keep[ps]) {
      ell_col[(size_t)row * width + rank] = jcol[ps];
      ell_a[(size_t)row * width + rank] = jw[ps];
    }
    total += __popcll(__ballot(keep[ps]));
  }
  if (lane == 0) deg[row] = total;
}

// ---- row_sum_cap / normalized_laplacian on the ELL graph (thread per row; rows are <= a few hundred bytes) ----
__global__ void k_row_scale(const float* ell_a, const int32_t* deg, int32_t width, int32_t N, float cap, float* scale) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= N) return;
  float s = 0.f;
  for (int e = 0; e < deg[row]; ++e) s += ell_a[(size_t)row * width + e];
  scale[row] = fminf(1.0f, cap / (s + 1e-12f));  // graph.py:77-78
}
// (one thread per ELL slot since round 6 -- a thread per row walked its row's slots one by one, 64 rows a wave instruction: 0.54 ms
// at config 5's 200 000 x 76 slots; the per-entry expressions are unchanged)
__global__ void k_apply_cap(float* ell_a, const int32_t* ell_col, const int32_t* deg, int32_t width, int32_t N,
                            const float* scale) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= (int64_t)N * width) return;
  const int row = (int)(o / width), e = (int)(o - (int64_t)row * width);
  if (e >= deg[row]) return;
  const float si = scale[row];
  const float a2 = ell_a[o] * sqrtf(si * scale[ell_col[o]]);  // graph.py:80-81
  ell_a[o] = 0.5f * (a2 + a2);                                 // graph.py:83 (symmetric input: identity)
}
__global__ void k_sqrt_deg(const float* ell_a, const int32_t* deg, int32_t width, int32_t N, float* sqrt_deg) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= N) return;
  float d = 0.f;
  for (int e = 0; e < deg[row]; ++e) d += ell_a[(size_t)row * width + e];
  sqrt_deg[row] = sqrtf(fmaxf(d, 1e-12f));  // graph.py:87-88
}
__global__ void k_normalize_w(const float* ell_a, const int32_t* ell_col, const int32_t* deg, int32_t width, int32_t N,
                              const float* sqrt_deg, float* ell_w) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (one thread per ELL slot, as k_apply_cap)
  if (o >= (int64_t)N * width) return;
  const int row = (int)(o / width), e = (int)(o - (int64_t)row * width);
  if (e >= deg[row]) return;
  const float di = 1.0f / sqrt_deg[row];
  ell_w[o] = (ell_a[o] * di) * (1.0f / sqrt_deg[ell_col[o]]);  // graph.py:89-90
}

}  // namespace

// ---- launchers --------------------------------------------------------------------------------
void launch_normalize_rows(const float* Y, int32_t ldy, float* Yn, int32_t ldn, int64_t N, int32_t D, hipStream_t s) {
  hipLaunchKernelGGL(k_normalize_rows, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, Y, ldy, Yn, ldn, N, D);
  HIP_CHECK(hipGetLastError());
}
void launch_rows_cosine(const float* A, int32_t ld, const float* q, float* out, int64_t N, int32_t D, hipStream_t s) {
  hipLaunchKernelGGL(k_rows_cosine, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, A, ld, q, out, N, D);
  HIP_CHECK(hipGetLastError());
}
void launch_rows_dot(const float* Yn, int32_t ldn, const float* q, float* out, int64_t N, int32_t D, hipStream_t s) {
  hipLaunchKernelGGL(k_rows_dot, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, Yn, ldn, q, out, N, D);
  HIP_CHECK(hipGetLastError());
}

KnnPlan knn_plan(int32_t N, int32_t keep, int32_t slots, int rb_begin, int rb_count, bool f16, int splits_override) {
  KnnPlan p{};
  p.keep = keep;
  p.E = (keep + 31) / 32;
  if (p.E == 3 && !f16) p.E = 4;  // exact variants: 1, 2 and the wide one
  p.KC = 32 * p.E;
  p.f16 = f16;
  p.qrows = nullptr;
  p.nq = 0;
  const int all_blocks = (N + BM - 1) / BM;
  p.rb_begin = rb_begin;
  p.rb_count = rb_count < 0 ? all_blocks : rb_count;
  p.row_blocks = std::max(1, p.rb_count);
  const int col_tiles = (N + BN - 1) / BN;
  // choose the column split count S: enough work items to balance `slots` resident blocks, while keeping
  // the per-item list warm-up (k log) small.
  int best_S = 1;
  double best_cost = 1e300;
  // a handful of row blocks (the per-row exact fallback of the prefilter routes): up to 64 splits, so that the sweep
  // of one 128-row block is spread over 64 workgroups instead of 16 (3.5 -> 0.9 ms at N = 100k)
  // (bounded by the merge's candidate count S x KC <= 2048: its rank-select is quadratic in it)
  const int max_S = (!f16 && p.row_blocks * 16 < slots / 2) ? std::max(16, std::min(64, 2048 / p.KC)) : 16;
  for (int S = 1; S <= std::min(col_tiles, max_S); ++S) {
    const int tiles_per = (col_tiles + S - 1) / S;
    const int S_eff = (col_tiles + tiles_per - 1) / tiles_per;
    if (S_eff != S) continue;
    const long items = (long)p.row_blocks * S;
    const long rounds = (items + slots - 1) / slots;
    const double cost = (double)rounds * tiles_per * (1.0 + (f16 ? 0.12 : 0.004) * S);  // imbalance x warm-up overhead
    if (cost < best_cost) {
      best_cost = cost;
      best_S = S;
    }
  }
  if (splits_override > 0) best_S = std::max(1, std::min(col_tiles, splits_override));
  p.S = best_S;
  p.cols_per_split = ((col_tiles + p.S - 1) / p.S) * BN;
  return p;
}

void launch_to_f16(const float* Yn, int32_t ldn, void* Yh, int32_t ldh, int64_t N, int32_t D, hipStream_t s) {
  hipLaunchKernelGGL(k_to_f16, dim3((unsigned)N), dim3(256), 0, s, Yn, ldn, reinterpret_cast<_Float16*>(Yh), ldh, N, D);
  HIP_CHECK(hipGetLastError());
}

void launch_knn_topk(const KnnPlan& p, const float* Yop, int32_t ld, int32_t N, float* cand_val, int32_t* cand_idx,
                     hipStream_t s) {
  if (p.rb_count <= 0) return;
  const dim3 grid((unsigned)(8 * ((p.rb_count + 7) / 8) * p.S)), block(256);
#define OSC_KNN_ARGS \
  Yop, ld, N, p.keep, p.S, p.cols_per_split, cand_val, cand_idx, p.rb_begin, p.rb_count, p.qrows, p.nq
#define OSC_KNN_PREF_ARGS Yop, ld, N, p.keep, p.S, p.cols_per_split, cand_val, cand_idx, p.rb_begin, p.rb_count
  if (p.f16) {
    if (p.qrows) throw std::runtime_error("the prefilter kernel has no row-list variant");
    if (p.E > 3) throw std::runtime_error("f16 prefilter supports at most 96 kept candidates");
    // (a 256-row, 8-wave, three-stage form of the same body was measured at 32.0 vs 31.0 ms at config 3 and removed)
    constexpr size_t lds = (size_t)2 * (128 + 128) * BK * 4;
    if (p.E == 1) hipLaunchKernelGGL((k_knn_pref<1, 4, 2>), grid, block, lds, s, OSC_KNN_PREF_ARGS);
    else if (p.E == 2) hipLaunchKernelGGL((k_knn_pref<2, 4, 2>), grid, block, lds, s, OSC_KNN_PREF_ARGS);
    else if (p.E == 3) hipLaunchKernelGGL((k_knn_pref<3, 4, 2>), grid, block, lds, s, OSC_KNN_PREF_ARGS);
  } else if (p.qrows) {
    if (p.E == 1) hipLaunchKernelGGL((k_knn_topk<1, false, true>), grid, block, 0, s, OSC_KNN_ARGS);
    else if (p.E == 2) hipLaunchKernelGGL((k_knn_topk<2, false, true>), grid, block, 0, s, OSC_KNN_ARGS);
    else hipLaunchKernelGGL(k_knn_topk_wide<true>, grid, block, 0, s, OSC_KNN_ARGS);
  } else {
    if (p.E == 1) hipLaunchKernelGGL((k_knn_topk<1, false, false>), grid, block, 0, s, OSC_KNN_ARGS);
    else if (p.E == 2) hipLaunchKernelGGL((k_knn_topk<2, false, false>), grid, block, 0, s, OSC_KNN_ARGS);
    else hipLaunchKernelGGL(k_knn_topk_wide<false>, grid, block, 0, s, OSC_KNN_ARGS);
  }
#undef OSC_KNN_ARGS
  HIP_CHECK(hipGetLastError());
}

void launch_knn_dense(const float* Yn, int32_t ldn, int32_t N, int32_t k, float* Sm, int32_t lds_, float* out_val,
                      int32_t* out_idx, hipStream_t s) {
  if (N > 8192) throw std::runtime_error("launch_knn_dense: N > 8192");
  const int nb = (N + BM - 1) / BM;
  hipLaunchKernelGGL(k_knn_dense, dim3(nb, nb), dim3(256), 0, s, Yn, ldn, N, Sm, lds_, 0);
  const dim3 grid((unsigned)((N + 3) / 4)), block(256);
  const int m = (N + 63) / 64;
#define OSC_SEL(MM) hipLaunchKernelGGL(k_knn_select<MM>, grid, block, 0, s, Sm, lds_, N, k, out_val, out_idx)
  if (m <= 4) OSC_SEL(4);
  else if (m <= 8) OSC_SEL(8);
  else if (m <= 16) OSC_SEL(16);
  else if (m <= 24) OSC_SEL(24);
  else if (m <= 32) OSC_SEL(32);
  else if (m <= 48) OSC_SEL(48);
  else if (m <= 64) OSC_SEL(64);
  else if (m <= 80) OSC_SEL(80);
  else if (m <= 96) OSC_SEL(96);
  else if (m <= 112) OSC_SEL(112);
  else OSC_SEL(128);
#undef OSC_SEL
  HIP_CHECK(hipGetLastError());
}

void launch_knn_rows_any(const float* Yn, int32_t ldn, int32_t N, int32_t k, int32_t row_begin, int32_t rows, float* Sm,
                         int32_t lds_, float* out_val, int32_t* out_idx, hipStream_t s) {
  if (rows <= 0) return;
  if (row_begin % BM != 0) throw std::runtime_error("launch_knn_rows_any: row_begin must be a multiple of 128");
  hipLaunchKernelGGL(k_knn_dense, dim3((N + BN - 1) / BN, (rows + BM - 1) / BM), dim3(256), 0, s, Yn, ldn, N, Sm, lds_,
                     row_begin);
  hipLaunchKernelGGL(k_knn_select_any<256>, dim3((unsigned)rows), dim3(256), 0, s, Sm, lds_, N, k, row_begin, rows, out_val,
                     out_idx, nullptr);
  HIP_CHECK(hipGetLastError());
}

bool launch_knn_few_rows(const float* Yn, int32_t ldn, int32_t N, int32_t k, const int32_t* qrows, int32_t nq, float* Sm,
                         int32_t lds_, float* out_val, int32_t* out_idx, hipStream_t s) {
  const int nch = (ldn + 255) / 256;
  if (nq <= 0 || nch > 6) return false;
  const dim3 grid(256, (unsigned)((nq + 3) / 4)), block(256);
#define OSC_RS(NN) hipLaunchKernelGGL(k_rows_scores<NN>, grid, block, 0, s, Yn, ldn, N, qrows, nq, Sm, lds_)
  if (nch <= 1) OSC_RS(1);
  else if (nch == 2) OSC_RS(2);
  else if (nch == 3) OSC_RS(3);
  else if (nch == 4) OSC_RS(4);
  else OSC_RS(6);
#undef OSC_RS
  // few rows, each a pass over all N scores: 1024 threads per row
  hipLaunchKernelGGL(k_knn_select_any<1024>, dim3((unsigned)nq), dim3(1024), 0, s, Sm, lds_, N, k, 0, nq, out_val, out_idx,
                     qrows);
  HIP_CHECK(hipGetLastError());
  return true;
}

void launch_knn_merge(const KnnPlan& p, const float* cand_val, const int32_t* cand_idx, int32_t N, int32_t k_out,
                      float* out_val, int32_t* out_idx, int clip, hipStream_t s) {
  const int ncand = p.S * p.KC;
  const size_t shmem = (size_t)4 * ncand * (sizeof(float) + sizeof(int32_t));
  int row_begin = p.rb_begin * BM, row_end = std::min(N, (p.rb_begin + p.rb_count) * BM);
  if (p.qrows) {
    row_begin = 0;
    row_end = p.nq;
  }
  if (row_end <= row_begin) return;
  if (shmem > 48 * 1024)
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_knn_merge), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)shmem));
  hipLaunchKernelGGL(k_knn_merge, dim3((unsigned)((row_end - row_begin + 3) / 4)), dim3(256), shmem, s, cand_val,
                     cand_idx, ncand, row_begin, row_end, k_out, out_val, out_idx, clip, p.qrows);
  HIP_CHECK(hipGetLastError());
}

void launch_knn_rescore(const KnnPlan& p, const float* Yn, int32_t ldn, int32_t D, int32_t N, const int32_t* cidx,
                        const float* cval, int32_t k, float delta, float* out_val, int32_t* out_idx, int32_t* fail_rows,
                        int32_t* fail_count, hipStream_t s, const KnnRowMap* map, float* pair_sc, int32_t* pair_pos) {
  const int row_begin = p.rb_begin * BM, row_end = std::min(N, (p.rb_begin + p.rb_count) * BM);
  const int32_t mapped = map != nullptr ? 1 : 0;
  const KnnRowMap rm = map != nullptr ? *map : knn_row_map(N, nullptr, 1, false);
  if (row_end <= row_begin) return;
  const dim3 grid((unsigned)((row_end - row_begin + 3) / 4)), block(256);
  const int nch = (ldn + 255) / 256;
  // pair form (every undirected candidate pair scored once): the caller's scratch arrays ask for it; it needs every row's list
  // in THIS launch (a single-process build's whole row range) and slot addresses that fit an int
  if (pair_sc != nullptr && pair_pos != nullptr && row_begin == 0 && row_end == N && (int64_t)N * p.keep < (int64_t)1 << 31) {
#define OSC_RESCORE_PAIR(NN) \
  hipLaunchKernelGGL(k_knn_rescore_pair<NN>, grid, block, 0, s, Yn, ldn, N, row_begin, row_end, cidx, p.keep, pair_sc, pair_pos, mapped, rm)
    if (nch <= 1) OSC_RESCORE_PAIR(1);
    else if (nch == 2) OSC_RESCORE_PAIR(2);
    else if (nch == 3) OSC_RESCORE_PAIR(3);
    else if (nch == 4) OSC_RESCORE_PAIR(4);
    else if (nch <= 6) OSC_RESCORE_PAIR(6);
    else if (nch <= 8) OSC_RESCORE_PAIR(8);
    else OSC_RESCORE_PAIR(0);
#undef OSC_RESCORE_PAIR
    HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_knn_rescore_finish, grid, block, 0, s, N, row_begin, row_end, cidx, cval, p.keep, k, delta, pair_sc, pair_pos,
                       out_val, out_idx, fail_rows, fail_count, mapped, rm);
    HIP_CHECK(hipGetLastError());
    return;
  }
#define OSC_RESCORE(NN)                                                                                              \
  hipLaunchKernelGGL(k_knn_rescore<NN>, grid, block, 0, s, Yn, ldn, D, N, row_begin, row_end, cidx, cval, p.keep, k, \
                     delta, out_val, out_idx, fail_rows, fail_count, mapped, rm)
  if (nch <= 1) OSC_RESCORE(1);
  else if (nch == 2) OSC_RESCORE(2);
  else if (nch == 3) OSC_RESCORE(3);
  else if (nch == 4) OSC_RESCORE(4);
  else if (nch <= 6) OSC_RESCORE(6);
  else if (nch <= 8) OSC_RESCORE(8);
  else OSC_RESCORE(0);
#undef OSC_RESCORE
  HIP_CHECK(hipGetLastError());
}

// the same for lists of any length that are sorted by column index (the k > 128 route): binary search for the back
// edge; the kept entries are already in ascending column order, so an entry's ELL slot is the count of kept entries
// before it
__global__ __launch_bounds__(256) void k_mutual_ell_sorted(const float* kval, const int32_t* kidx, int32_t N, int32_t k,
                                                           int32_t width, int32_t* ell_col, float* ell_a, int32_t* deg) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  int total = 0;
  for (int e0 = 0; e0 < k; e0 += 64) {
    const int e = e0 + lane;
    bool keep = false;
    int j = 0;
    float w = 0.f;
    if (e < k) {
      j = kidx[(size_t)row * k + e];
      const float v = kval[(size_t)row * k + e];
      if (v > 0.f && j >= 0 && j < N) {
        const int32_t* lj = kidx + (size_t)j * k;
        int lo = 0, hi = k;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (lj[mid] < row) lo = mid + 1;
          else hi = mid;
        }
        if (lo < k && lj[lo] == row) {
          const float back = kval[(size_t)j * k + lo];
          if (back > 0.f) {
            keep = true;
            w = fmaxf(v, back);
          }
        }
      }
    }
    const unsigned long long b = __ballot(keep);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (keep) {
      const int slot = total + __popcll(b & below);
      ell_col[(size_t)row * width + slot] = j;
      ell_a[(size_t)row * width + slot] = w;
    }
    total += __popcll(b);
  }
  if (lane == 0) deg[row] = total;
}

void launch_mutual_ell(const float* kval, const int32_t* kidx, int32_t N, int32_t k, int32_t width, int32_t* ell_col,
                       float* ell_a, int32_t* deg, hipStream_t s) {
  if (k > 128) {  // lists of the any-k route (sorted by column index)
    hipLaunchKernelGGL(k_mutual_ell_sorted, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, kval, kidx, N, k, width,
                       ell_col, ell_a, deg);
    HIP_CHECK(hipGetLastError());
    return;
  }
  hipLaunchKernelGGL(k_mutual_ell, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, kval, kidx, N, k, width, ell_col,
                     ell_a, deg);
  HIP_CHECK(hipGetLastError());
}

void launch_cap_and_normalize(float* ell_a, float* ell_w, const int32_t* ell_col, const int32_t* deg, int32_t width,
                              int32_t N, float cap, int apply_cap, float* scale_tmp, float* sqrt_deg, hipStream_t s) {
  const dim3 grid((unsigned)((N + 255) / 256)), block(256);
  const dim3 grid_slots((unsigned)(((int64_t)N * width + 255) / 256));  // (N x width < 2^31: the ELL arrays' own limit)
  if (apply_cap) {
    hipLaunchKernelGGL(k_row_scale, grid, block, 0, s, ell_a, deg, width, N, cap, scale_tmp);
    hipLaunchKernelGGL(k_apply_cap, grid_slots, block, 0, s, ell_a, ell_col, deg, width, N, scale_tmp);
  }
  hipLaunchKernelGGL(k_sqrt_deg, grid, block, 0, s, ell_a, deg, width, N, sqrt_deg);
  hipLaunchKernelGGL(k_normalize_w, grid_slots, block, 0, s, ell_a, ell_col, deg, width, N, sqrt_deg, ell_w);
  HIP_CHECK(hipGetLastError());
}

}  // namespace osc
