// Receipt diagnostics on the sparse lattice (receipts.py:28-83), one wave per row, edges gathered like the CG matvec.
//
// For row i with neighbours j (capped adjacency a_ij, columns ascending):
//   coh_drop_i = sum_j 0.5 lamC a_ij (||Yn_i - Yn_j||^2 - ||Un_i - Un_j||^2),  Yn = Y/(sqrt_deg+1e-12), Un = U*/(sqrt_deg+1e-12)
//   anchor_i   = lamG ||U*_i - Y_i||^2 ;  query_i = lamQ B_i ||U*_i - psi||^2                      (receipts.py:40-59)
//   null point : R_ij = lamC a_ij ||Un_i - Un_j||^2 ; the reference z-scores each DENSE row (N entries, zeros
//                included): mu = sum R / N, sigma = sqrt(sum R^2 / N - mu^2) + 1e-12; candidate = first argmax.
#include "common.hpp"
#include "receipts.hpp"

namespace osc {
namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// a * sa - b * sb with both products rounded on their own (no fma contraction): the value for edge (i, j) is then the
// exact negation of the value for (j, i), like the reference's pre-scaled rows Un = U / (sqrt_deg + 1e-12), so the two
// directions of an edge get bit-identical energies (the dynamics' flow ranking relies on those ties)
__device__ __forceinline__ float sdiff(float a, float sa, float b, float sb) {
#pragma clang fp contract(off)  // (HIP's __fmul_rn is a plain multiply and would be contracted into an fma as well)
  const float p = a * sa;
  const float q = b * sb;
  return p - q;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// NCH = float4 chunks per lane that cover a row (ld <= 256 NCH): the row's own scaled Y / U* values stay in registers for
// all its edges, and EU edges are gathered together (independent accumulators and reductions), so the loop is no
// longer one exposed gather latency per edge.  Arithmetic per edge is unchanged (same products, same fma order per
// lane, same butterfly), hence bit-identical results.  NCH = 0: generic loop for wider rows.
template <int NCH>
__global__ __launch_bounds__(256) void k_receipt_rows(const ReceiptArgs a) {
  constexpr int EU = 2;  // edges in flight
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.N) return;
  const size_t ro = (size_t)row * a.ld;
  const float inv_i = 1.0f / (a.sqrt_deg[row] + 1e-12f);
  // anchor / query terms; the row's own values are kept for the edge loop
  float an = 0.f, qu = 0.f;
  float4 yi_r[NCH > 0 ? NCH : 1], ui_r[NCH > 0 ? NCH : 1];
  if constexpr (NCH > 0) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c = lane * 4 + ch * 256;
      float4 us = make_float4(0.f, 0.f, 0.f, 0.f), y = us, ps = us;
      if (c < a.ld) {
        us = ld4(a.Ustar + ro + c);
        y = ld4(a.Y + ro + c);
        ps = ld4(a.psi + c);
        const float d0 = us.x - y.x, d1 = us.y - y.y, d2 = us.z - y.z, d3 = us.w - y.w;
        const float q0 = us.x - ps.x, q1 = us.y - ps.y, q2 = us.z - ps.z, q3 = us.w - ps.w;
        an = fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, fmaf(d3, d3, an))));
        qu = fmaf(q0, q0, fmaf(q1, q1, fmaf(q2, q2, fmaf(q3, q3, qu))));
      }
      yi_r[ch] = y;
      ui_r[ch] = us;
    }
  } else {
    for (int c = lane * 4; c < a.ld; c += 256) {  // pitch ld is a multiple of 4; pad columns are zero everywhere
      const float4 us = ld4(a.Ustar + ro + c), y = ld4(a.Y + ro + c), ps = ld4(a.psi + c);
      const float d0 = us.x - y.x, d1 = us.y - y.y, d2 = us.z - y.z, d3 = us.w - y.w;
      const float q0 = us.x - ps.x, q1 = us.y - ps.y, q2 = us.z - ps.z, q3 = us.w - ps.w;
      an = fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, fmaf(d3, d3, an))));
      qu = fmaf(q0, q0, fmaf(q1, q1, fmaf(q2, q2, fmaf(q3, q3, qu))));
    }
  }
  an = wave_sum(an);
  qu = wave_sum(qu);
  // edges
  const int deg = a.deg[row];
  float coh = 0.f;
  double s1 = 0.0, s2 = 0.0;
  float rmax = 0.f;
  int jmax = -1;
  auto edge_terms = [&](int j, float inv_j, float& dy, float& du) {  // per-lane partial sums of one edge
    const size_t jo = (size_t)j * a.ld;
    dy = 0.f;
    du = 0.f;
    if constexpr (NCH > 0) {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int c = lane * 4 + ch * 256;
        if (c < a.ld) {
          const float4 yi = yi_r[ch], ui = ui_r[ch];
          const float4 yj = ld4(a.Y + jo + c), uj = ld4(a.Ustar + jo + c);
          const float y0 = sdiff(yi.x, inv_i, yj.x, inv_j), y1 = sdiff(yi.y, inv_i, yj.y, inv_j);
          const float y2 = sdiff(yi.z, inv_i, yj.z, inv_j), y3 = sdiff(yi.w, inv_i, yj.w, inv_j);
          const float u0 = sdiff(ui.x, inv_i, uj.x, inv_j), u1 = sdiff(ui.y, inv_i, uj.y, inv_j);
          const float u2 = sdiff(ui.z, inv_i, uj.z, inv_j), u3 = sdiff(ui.w, inv_i, uj.w, inv_j);
          dy = fmaf(y0, y0, fmaf(y1, y1, fmaf(y2, y2, fmaf(y3, y3, dy))));
          du = fmaf(u0, u0, fmaf(u1, u1, fmaf(u2, u2, fmaf(u3, u3, du))));
        }
      }
    } else {
      for (int c = lane * 4; c < a.ld; c += 256) {
        const float4 yi = ld4(a.Y + ro + c), yj = ld4(a.Y + jo + c);
        const float4 ui = ld4(a.Ustar + ro + c), uj = ld4(a.Ustar + jo + c);
        const float y0 = sdiff(yi.x, inv_i, yj.x, inv_j), y1 = sdiff(yi.y, inv_i, yj.y, inv_j);
        const float y2 = sdiff(yi.z, inv_i, yj.z, inv_j), y3 = sdiff(yi.w, inv_i, yj.w, inv_j);
        const float u0 = sdiff(ui.x, inv_i, uj.x, inv_j), u1 = sdiff(ui.y, inv_i, uj.y, inv_j);
        const float u2 = sdiff(ui.z, inv_i, uj.z, inv_j), u3 = sdiff(ui.w, inv_i, uj.w, inv_j);
        dy = fmaf(y0, y0, fmaf(y1, y1, fmaf(y2, y2, fmaf(y3, y3, dy))));
        du = fmaf(u0, u0, fmaf(u1, u1, fmaf(u2, u2, fmaf(u3, u3, du))));
      }
    }
  };
  auto edge_finish = [&](int e_slot, int j, float w, float dy, float du) {  // in edge order: the reference's column order
    if (a.edge_flow != nullptr && lane == 0) {
      const double f = 0.5 * (double)a.lamC * (double)w * ((double)dy - (double)du);
      a.edge_flow[(size_t)row * a.width + e_slot] = (w > 0.f && f > 0.0) ? (float)f : 0.f;
    }
    if (w > 0.f) {
      coh += 0.5f * a.lamC * w * (dy - du);
      const float R = a.lamC * w * du;
      s1 += (double)R;
      s2 += (double)R * (double)R;
      // first argmax in the reference's column order: strict >, ties to the smaller API column id
      bool take = R > rmax;
      if (!take && R == rmax && R > 0.f && a.api_id != nullptr && jmax >= 0) take = a.api_id[j] < a.api_id[jmax];
      if (take) {
        rmax = R;
        jmax = j;
      }
    }
  };
  const int32_t* crow = a.col + (size_t)row * a.width;
  const float* arow = a.adj + (size_t)row * a.width;
  int e = 0;
  for (; e + EU <= deg; e += EU) {
    int jj[EU];
    float ww[EU], ij[EU], dy[EU], du[EU];
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      jj[u] = crow[e + u];
      ww[u] = arow[e + u];
    }
#pragma unroll
    for (int u = 0; u < EU; ++u) ij[u] = 1.0f / (a.sqrt_deg[jj[u]] + 1e-12f);
#pragma unroll
    for (int u = 0; u < EU; ++u) edge_terms(jj[u], ij[u], dy[u], du[u]);
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      dy[u] = wave_sum(dy[u]);
      du[u] = wave_sum(du[u]);
    }
#pragma unroll
    for (int u = 0; u < EU; ++u) edge_finish(e + u, jj[u], ww[u], dy[u], du[u]);
  }
  for (; e < deg; ++e) {
    const int j = crow[e];
    const float w = arow[e];
    const float inv_j = 1.0f / (a.sqrt_deg[j] + 1e-12f);
    float dy, du;
    edge_terms(j, inv_j, dy, du);
    dy = wave_sum(dy);
    du = wave_sum(du);
    edge_finish(e, j, w, dy, du);
  }
  if (lane == 0) {
    if (a.coh) a.coh[row] = coh;
    if (a.anchor) a.anchor[row] = a.lamG * an;
    if (a.query) a.query[row] = a.lamQ * a.B[row] * qu;
    if (a.null_j) {
      const double mu = s1 / (double)a.N;
      double var = s2 / (double)a.N - mu * mu;
      if (var < 0.0) var = 0.0;
      const double z = ((double)rmax - mu) / (sqrt(var) + 1e-12);
      const bool is_null = (jmax >= 0) && (rmax > 0.f) && (z > (double)a.z_th);
      a.null_j[row] = is_null ? jmax : -1;
      a.null_z[row] = (float)z;
      a.null_r[row] = rmax;
    }
  }
}


// ---- the pair form (round 6): every undirected edge's two squared distances computed ONCE ---------------------------------
// k_receipt_rows gathers the rows of both ends of every edge from both ends: 2 nnz D 4 bytes x 2 arrays, 17.7 GB at config 3,
// its whole time (2.5 ms at 6.7 TB/s).  The per-edge values are symmetric bit for bit by construction (sdiff above: the two
// directions' differences are exact negations, the squares, the per-lane fma order and the butterfly are the same), so row i
// computes only its edges to j >= i and leaves (dy, du) in an ELL-shaped scratch; k_receipt_finish then walks every row's
// edges in order, takes its own slot's pair or -- j < i -- the slot of i in row j's list (the wave reads that list with one
// coalesced load and finds the slot with a ballot, four lists in flight), and accumulates exactly as k_receipt_rows does:
// same values, same order, same outputs.  An edge whose mirror slot does not exist (an asymmetric list: cannot happen, the
// build and the injection enforce symmetry) raises a flag; the caller then runs the one-launch kernel instead.
template <int NCH>
__global__ __launch_bounds__(256) void k_receipt_pairs(const ReceiptArgs a) {
  static_assert(NCH > 0, "the pair form keeps the row in registers");
  constexpr int EU = 2;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.N) return;
  const size_t ro = (size_t)row * a.ld;
  const float inv_i = 1.0f / (a.sqrt_deg[row] + 1e-12f);
  float an = 0.f, qu = 0.f;
  float4 yi_r[NCH], ui_r[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {  // (k_receipt_rows' own-row pass, expression for expression)
    const int c = lane * 4 + ch * 256;
    float4 us = make_float4(0.f, 0.f, 0.f, 0.f), y = us, ps = us;
    if (c < a.ld) {
      us = ld4(a.Ustar + ro + c);
      y = ld4(a.Y + ro + c);
      ps = ld4(a.psi + c);
      const float d0 = us.x - y.x, d1 = us.y - y.y, d2 = us.z - y.z, d3 = us.w - y.w;
      const float q0 = us.x - ps.x, q1 = us.y - ps.y, q2 = us.z - ps.z, q3 = us.w - ps.w;
      an = fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, fmaf(d3, d3, an))));
      qu = fmaf(q0, q0, fmaf(q1, q1, fmaf(q2, q2, fmaf(q3, q3, qu))));
    }
    yi_r[ch] = y;
    ui_r[ch] = us;
  }
  an = wave_sum(an);
  qu = wave_sum(qu);
  if (lane == 0) {
    if (a.anchor) a.anchor[row] = a.lamG * an;
    if (a.query) a.query[row] = a.lamQ * a.B[row] * qu;
  }
  const int deg = a.deg[row];
  const int32_t* crow = a.col + (size_t)row * a.width;
  int e = 0;
  while (e < deg) {
    int jj[EU], es[EU], n = 0;
    while (e < deg && n < EU) {  // (wave-uniform: the next edges to a row not below this one)
      const int j = crow[e];
      if (j >= row) {
        jj[n] = j;
        es[n] = e;
        ++n;
      }
      ++e;
    }
    float ij[EU], dy[EU], du[EU];
#pragma unroll
    for (int u = 0; u < EU; ++u) ij[u] = u < n ? 1.0f / (a.sqrt_deg[jj[u]] + 1e-12f) : 0.f;
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      dy[u] = 0.f;
      du[u] = 0.f;
      if (u < n) {
        const size_t jo = (size_t)jj[u] * a.ld;
        const float inv_j = ij[u];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          const int c = lane * 4 + ch * 256;
          if (c < a.ld) {
            const float4 yi = yi_r[ch], ui = ui_r[ch];
            const float4 yj = ld4(a.Y + jo + c), uj = ld4(a.Ustar + jo + c);
            const float y0 = sdiff(yi.x, inv_i, yj.x, inv_j), y1 = sdiff(yi.y, inv_i, yj.y, inv_j);
            const float y2 = sdiff(yi.z, inv_i, yj.z, inv_j), y3 = sdiff(yi.w, inv_i, yj.w, inv_j);
            const float u0 = sdiff(ui.x, inv_i, uj.x, inv_j), u1 = sdiff(ui.y, inv_i, uj.y, inv_j);
            const float u2 = sdiff(ui.z, inv_i, uj.z, inv_j), u3 = sdiff(ui.w, inv_i, uj.w, inv_j);
            dy[u] = fmaf(y0, y0, fmaf(y1, y1, fmaf(y2, y2, fmaf(y3, y3, dy[u]))));
            du[u] = fmaf(u0, u0, fmaf(u1, u1, fmaf(u2, u2, fmaf(u3, u3, du[u]))));
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      if (u < n) {
        dy[u] = wave_sum(dy[u]);
        du[u] = wave_sum(du[u]);
        if (lane == 0) {
          a.pair_dy[(size_t)row * a.width + es[u]] = dy[u];
          a.pair_du[(size_t)row * a.width + es[u]] = du[u];
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_receipt_finish(const ReceiptArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.N) return;
  const int deg = a.deg[row];
  const int32_t* crow = a.col + (size_t)row * a.width;
  const float* arow = a.adj + (size_t)row * a.width;
  float coh = 0.f;
  double s1 = 0.0, s2 = 0.0;
  float rmax = 0.f;
  int jmax = -1;
  for (int e0 = 0; e0 < deg; e0 += 64) {
    const int e = e0 + lane;
    const bool have = e < deg;
    const int j = have ? crow[e] : -1;
    const float w = have ? arow[e] : 0.f;
    long long src = (have && j >= row) ? (long long)row * a.width + e : -1;
    // the mirror slots of this chunk's edges to rows below this one: four of those rows' lists in flight
    unsigned long long need = __ballot(have && j < row);
    while (need) {
      int ln[4], jj[4], c0[4], c1[4], dj[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ln[u] = -1;
        jj[u] = 0;
        c0[u] = c1[u] = -2;
        dj[u] = 0;
        if (need) {
          ln[u] = __ffsll((long long)need) - 1;
          need &= need - 1;
          jj[u] = __shfl(j, ln[u], 64);
          dj[u] = a.deg[jj[u]];
          const int32_t* lj = a.col + (size_t)jj[u] * a.width;
          if (lane < dj[u]) c0[u] = lj[lane];
          if (lane + 64 < dj[u]) c1[u] = lj[lane + 64];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (ln[u] < 0) continue;  // (wave-uniform)
        const unsigned long long m0 = __ballot(c0[u] == row), m1 = __ballot(c1[u] == row);
        int pos = m0 ? __ffsll((long long)m0) - 1 : m1 ? 64 + __ffsll((long long)m1) - 1 : -1;
        if (pos < 0 && dj[u] > 128) {  // (lists longer than two registers: the rest, one coalesced load at a time)
          const int32_t* lj = a.col + (size_t)jj[u] * a.width;
          for (int t0 = 128; t0 < dj[u] && pos < 0; t0 += 64) {
            const unsigned long long m = __ballot(t0 + lane < dj[u] && lj[t0 + lane] == row);
            if (m) pos = t0 + __ffsll((long long)m) - 1;
          }
        }
        if (pos < 0 && lane == 0) atomicOr(a.pair_fail, 1);
        if (lane == ln[u]) src = pos >= 0 ? (long long)jj[u] * a.width + pos : -1;
      }
    }
    const float dyv = src >= 0 ? a.pair_dy[src] : 0.f, duv = src >= 0 ? a.pair_du[src] : 0.f;
    if (a.edge_flow != nullptr && have) {
      const double f = 0.5 * (double)a.lamC * (double)w * ((double)dyv - (double)duv);
      a.edge_flow[(size_t)row * a.width + e] = (w > 0.f && f > 0.0) ? (float)f : 0.f;
    }
    const int n = min(64, deg - e0);
    for (int t = 0; t < n; ++t) {  // in edge order, every lane the same sums: k_receipt_rows' edge_finish
      const float wt = __shfl(w, t, 64), dyt = __shfl(dyv, t, 64), dut = __shfl(duv, t, 64);
      const int jt = __shfl(j, t, 64);
      if (wt > 0.f) {
        coh += 0.5f * a.lamC * wt * (dyt - dut);
        const float R = a.lamC * wt * dut;
        s1 += (double)R;
        s2 += (double)R * (double)R;
        bool take = R > rmax;
        if (!take && R == rmax && R > 0.f && a.api_id != nullptr && jmax >= 0) take = a.api_id[jt] < a.api_id[jmax];
        if (take) {
          rmax = R;
          jmax = jt;
        }
      }
    }
  }
  if (lane == 0) {
    if (a.coh) a.coh[row] = coh;
    if (a.null_j) {
      const double mu = s1 / (double)a.N;
      double var = s2 / (double)a.N - mu * mu;
      if (var < 0.0) var = 0.0;
      const double z = ((double)rmax - mu) / (sqrt(var) + 1e-12);
      const bool is_null = (jmax >= 0) && (rmax > 0.f) && (z > (double)a.z_th);
      a.null_j[row] = is_null ? jmax : -1;
      a.null_z[row] = (float)z;
      a.null_r[row] = rmax;
    }
  }
}


// ---- greedy MMR on the device (graph.py:114-133; bundle(), lattice.py:530-568) ---------------------------------------
// val_i = (1 - lambda) score_i - lambda max_{j chosen} cos(Y_i, Y_j) over the rows still alive; the next item is the
// first maximum in API row order.  One step = argmax (two stages), normalise the chosen row, one cosine pass over the
// anchors fused with the running maximum.  All in fp64 where the host version used NumPy float64.
constexpr double MMR_DEAD = -1.0e300;

__device__ __forceinline__ bool mmr_better(double v, int id, double bv, int bid) { return v > bv || (v == bv && id < bid); }

// stage 1: per-block best (value, API id, device row)
__global__ __launch_bounds__(256) void k_mmr_argmax1(const double* base, const double* maxsim, const unsigned char* alive,
                                                     const int32_t* api_id, int32_t N, double lambda, int first,
                                                     double* pval, int32_t* pid, int32_t* prow) {
  __shared__ double sv[256];
  __shared__ int sid[256], srow[256];
  double bv = MMR_DEAD;
  int bid = 0x7fffffff, brow = -1;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
    if (!alive[i]) continue;
    const double v = base[i] - (first ? 0.0 : lambda * maxsim[i]);
    const int id = api_id ? api_id[i] : i;
    if (brow < 0 || mmr_better(v, id, bv, bid)) {
      bv = v;
      bid = id;
      brow = i;
    }
  }
  sv[threadIdx.x] = bv;
  sid[threadIdx.x] = bid;
  srow[threadIdx.x] = brow;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      const int t = threadIdx.x + o;
      if (srow[t] >= 0 && (srow[threadIdx.x] < 0 || mmr_better(sv[t], sid[t], sv[threadIdx.x], sid[threadIdx.x]))) {
        sv[threadIdx.x] = sv[t];
        sid[threadIdx.x] = sid[t];
        srow[threadIdx.x] = srow[t];
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    pval[blockIdx.x] = sv[0];
    pid[blockIdx.x] = sid[0];
    prow[blockIdx.x] = srow[0];
  }
}

// stage 2 (one block): the winner; mark it dead, record it, and write its normalised anchor row into q
__global__ __launch_bounds__(256) void k_mmr_argmax2(const double* pval, const int32_t* pid, const int32_t* prow, int nb,
                                                     unsigned char* alive, int32_t* chosen_api, int step, const float* Y,
                                                     int32_t ld, int32_t D, float* q) {
  __shared__ double sv[256];
  __shared__ int sid[256], srow[256];
  __shared__ float sn[256];
  double bv = MMR_DEAD;
  int bid = 0x7fffffff, brow = -1;
  for (int b = threadIdx.x; b < nb; b += 256) {
    if (prow[b] >= 0 && (brow < 0 || mmr_better(pval[b], pid[b], bv, bid))) {
      bv = pval[b];
      bid = pid[b];
      brow = prow[b];
    }
  }
  sv[threadIdx.x] = bv;
  sid[threadIdx.x] = bid;
  srow[threadIdx.x] = brow;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      const int t = threadIdx.x + o;
      if (srow[t] >= 0 && (srow[threadIdx.x] < 0 || mmr_better(sv[t], sid[t], sv[threadIdx.x], sid[threadIdx.x]))) {
        sv[threadIdx.x] = sv[t];
        sid[threadIdx.x] = sid[t];
        srow[threadIdx.x] = srow[t];
      }
    }
    __syncthreads();
  }
  const int row = srow[0];
  if (row < 0) {
    if (threadIdx.x == 0) chosen_api[step] = -1;
    return;
  }
  if (threadIdx.x == 0) {
    alive[row] = 0;
    chosen_api[step] = sid[0];
  }
  // q = Y_row / (|Y_row| + 1e-12)
  const float* y = Y + (size_t)row * ld;
  float n2 = 0.f;
  for (int c = threadIdx.x; c < D; c += 256) n2 = fmaf(y[c], y[c], n2);
  sn[threadIdx.x] = n2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sn[threadIdx.x] += sn[threadIdx.x + o];
    __syncthreads();
  }
  const float inv = 1.0f / (sqrtf(sn[0]) + 1e-12f);
  for (int c = threadIdx.x; c < D; c += 256) q[c] = y[c] * inv;
}

// maxsim_i = max(maxsim_i, cos(Y_i, q)) for every row (first pass: = cos), one wave per row -- and, round 6, the first argmax
// stage of the NEXT step in the same launch: a wave updates its row's running maximum and, if the row is still alive, offers (1 - lambda) score - lambda maxsim; the workgroup's four offers
// go to one partial.  (The argmax is a maximum under a total order -- value, then the smaller API id -- so the shape of
// the reduction tree does not matter: same winners as k_mmr_argmax1's strided partials.)  Two launches per step instead of
// three, and no cosine pass behind the last pick.
__global__ __launch_bounds__(256) void k_mmr_update_argmax(const float* Y, int32_t ld, int32_t D, const float* q, int32_t N, int first,
                                                           double* maxsim, const double* base, const unsigned char* alive,
                                                           const int32_t* api_id, double lambda, double* pval, int32_t* pid,
                                                           int32_t* prow) {
  __shared__ double sv[4];
  __shared__ int sid[4], srow[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double v = MMR_DEAD;  // (lane 0: the best offer of this wave's rows)
  int id = 0x7fffffff, rr = -1;
  for (int row = blockIdx.x * 4 + wave; row < N; row += (int)gridDim.x * 4) {  // (at most MMR_PARTS workgroups: a few rows per wave)
    const float* y = Y + (size_t)row * ld;
    float s = 0.f, n2 = 0.f;
    const int d4 = D & ~3;
    for (int c = lane * 4; c < d4; c += 256) {
      const float4 a = ld4(y + c), w = ld4(q + c);
      s = fmaf(a.x, w.x, fmaf(a.y, w.y, fmaf(a.z, w.z, fmaf(a.w, w.w, s))));
      n2 = fmaf(a.x, a.x, fmaf(a.y, a.y, fmaf(a.z, a.z, fmaf(a.w, a.w, n2))));
    }
    for (int c = d4 + lane; c < D; c += 64) {
      const float a = y[c];
      s = fmaf(a, q[c], s);
      n2 = fmaf(a, a, n2);
    }
    s = wave_sum(s);
    n2 = wave_sum(n2);
    if (lane == 0) {
      const double cs = (double)(s / (sqrtf(n2) + 1e-12f));
      const double ms = first ? cs : fmax(maxsim[row], cs);
      maxsim[row] = ms;
      if (alive[row]) {
        const double cand = base[row] - lambda * ms;
        const int cid = api_id ? api_id[row] : row;
        if (rr < 0 || mmr_better(cand, cid, v, id)) {
          v = cand;
          id = cid;
          rr = row;
        }
      }
    }
  }
  if (lane == 0) {
    sv[wave] = v;
    sid[wave] = id;
    srow[wave] = rr;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double bv = MMR_DEAD;
    int bid = 0x7fffffff, brow = -1;
    for (int w = 0; w < 4; ++w)
      if (srow[w] >= 0 && (brow < 0 || mmr_better(sv[w], sid[w], bv, bid))) {
        bv = sv[w];
        bid = sid[w];
        brow = srow[w];
      }
    pval[blockIdx.x] = bv;
    pid[blockIdx.x] = bid;
    prow[blockIdx.x] = brow;
  }
}

}  // namespace

void launch_receipt_rows(const ReceiptArgs& a, hipStream_t s) {
  const dim3 grid((unsigned)((a.N + 3) / 4)), block(256);
  const int nch = (a.ld + 255) / 256;
  if (a.pair_dy != nullptr && a.pair_du != nullptr && a.pair_fail != nullptr && nch <= 6) {  // the pair form (two launches)
    if (nch <= 1) hipLaunchKernelGGL(k_receipt_pairs<1>, grid, block, 0, s, a);
    else if (nch == 2) hipLaunchKernelGGL(k_receipt_pairs<2>, grid, block, 0, s, a);
    else if (nch == 3) hipLaunchKernelGGL(k_receipt_pairs<3>, grid, block, 0, s, a);
    else if (nch == 4) hipLaunchKernelGGL(k_receipt_pairs<4>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(k_receipt_pairs<6>, grid, block, 0, s, a);
    hipLaunchKernelGGL(k_receipt_finish, grid, block, 0, s, a);
    HIP_CHECK(hipGetLastError());
    return;
  }
  if (nch <= 1) hipLaunchKernelGGL(k_receipt_rows<1>, grid, block, 0, s, a);
  else if (nch == 2) hipLaunchKernelGGL(k_receipt_rows<2>, grid, block, 0, s, a);
  else if (nch == 3) hipLaunchKernelGGL(k_receipt_rows<3>, grid, block, 0, s, a);
  else if (nch == 4) hipLaunchKernelGGL(k_receipt_rows<4>, grid, block, 0, s, a);
  else if (nch <= 6) hipLaunchKernelGGL(k_receipt_rows<6>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(k_receipt_rows<0>, grid, block, 0, s, a);
  HIP_CHECK(hipGetLastError());
}

// step 0: strided first stage (no running maximum yet); step s > 0: the cosine pass to the item picked in step s - 1 fused with
// the first stage (one partial per workgroup: pval / pid / prow hold max(nblocks, mmr_parts(N)) entries)
int mmr_parts(int32_t N) { return (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)N + 3) / 4, 2048)); }
void launch_mmr_step(const MmrArgs& a, int step, hipStream_t s) {
  int nb = a.nblocks;
  if (step == 0) {
    hipLaunchKernelGGL(k_mmr_argmax1, dim3(nb), dim3(256), 0, s, a.base, a.maxsim, a.alive, a.api_id, a.N, a.lambda, 1, a.pval,
                       a.pid, a.prow);
  } else {
    nb = mmr_parts(a.N);
    hipLaunchKernelGGL(k_mmr_update_argmax, dim3((unsigned)nb), dim3(256), 0, s, a.Y, a.ld, a.D, a.q, a.N, step == 1 ? 1 : 0,
                       a.maxsim, a.base, a.alive, a.api_id, a.lambda, a.pval, a.pid, a.prow);
  }
  hipLaunchKernelGGL(k_mmr_argmax2, dim3(1), dim3(256), 0, s, a.pval, a.pid, a.prow, nb, a.alive, a.chosen_api, step, a.Y,
                     a.ld, a.D, a.q);
  HIP_CHECK(hipGetLastError());
}

}  // namespace osc
