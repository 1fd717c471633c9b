// Internal row order of a lattice.  The API's row i is stored at device row inv[i]; perm[new] = old.  A locality-
// preserving order (BFS over the lattice graph) makes the operator apply's neighbour gathers hit the XCD-local L2
// (cg_kernels.hip: XCD-aware row sweep).  These kernels move state between the two orders; they run once per graph
// build / state transfer, never inside a solve.
#include "common.hpp"
#include "perm.hpp"

namespace osc {
namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// GATHER: out[i] = in[map[i]]   SCATTER: out[map[i]] = in[i]     (rows of ld floats, ld % 4 == 0)
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_move_rows(float* out, const float* in, const int32_t* map, int64_t N,
                                                   int32_t ld) {
  const int lane = threadIdx.x & 63;
  for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < N; i += (int64_t)gridDim.x * 4) {
    const int64_t src = SCATTER ? i : map[i], dst = SCATTER ? map[i] : i;
    for (int c = lane * 4; c < ld; c += 256) st4(out + dst * ld + c, ld4(in + src * ld + c));
  }
}

template <typename T, bool SCATTER>
__global__ void k_move_1d(T* out, const T* in, const int32_t* map, int64_t N) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (SCATTER) out[map[i]] = in[i];
  else out[i] = in[map[i]];
}

// new row i takes old row from[i]; column ids are relabelled through relabel[]
__global__ void k_permute_ell(const int32_t* col_in, const float* a_in, const float* w_in, const int32_t* deg_in,
                              const int32_t* from, const int32_t* relabel, int32_t width, int64_t N, int32_t* col_out,
                              float* a_out, float* w_out, int32_t* deg_out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * width) return;
  const int64_t i = t / width;
  const int e = (int)(t % width);
  const int64_t o = (int64_t)from[i];
  const int d = deg_in[o];
  if (e == 0) deg_out[i] = d;
  const int64_t src = o * width + e;
  col_out[t] = e < d ? relabel[col_in[src]] : 0;
  a_out[t] = e < d ? a_in[src] : 0.f;
  w_out[t] = e < d ? w_in[src] : 0.f;
}

// Local clustering coefficient on a sample of rows: of the neighbour pairs (a, b) of a sampled row, how many are
// themselves adjacent.  ~k/N on an unstructured graph, a few tenths on clustered anchors: decides whether the BFS
// re-order is worth its few milliseconds.  Four waves per sampled row; counts[2 s] = adjacent pairs, counts[2 s + 1] = pairs of sampled row s (the caller adds them up).
__global__ __launch_bounds__(256) void k_clustering_sample(const int32_t* col, const int32_t* deg, int32_t width,
                                                           int64_t N, int32_t nsample, unsigned long long* counts) {
  const int lane = threadIdx.x & 63;
  // (four waves per sampled row, each taking every fourth neighbour a: a wave's rounds are one dependent list load each, and a
  // row of degree 30-60 was 30-60 of them in a row -- 68 us of a 0.96 ms build at N = 12 000; same counts)
  constexpr int SPLIT = 4;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int sidx = gw / SPLIT, part = gw % SPLIT;
  if (sidx >= nsample) return;
  const int64_t row = (int64_t)sidx * N / nsample;
  const int d = deg[row];
  const int32_t* nb = col + (size_t)row * width;
  // (round 6: the wave reads neighbour a's list ONCE, coalesced, and tests every later neighbour b against it with a ballot -- a
  // lane per pair scanned a's list entry by entry for every pair: 0.08 ms at config 3, 0.53 ms at config 5's degree; same counts)
  unsigned hits = 0, pairs = 0;  // (wave-uniform)
  // the row's neighbours and their degrees in two registers (d <= 128; longer rows: through memory), so that a's list is the
  // only dependent load of a round -- and the next round's is issued before this round's tests
  const int32_t nb0 = lane < d ? nb[lane] : -1, nb1 = lane + 64 < d ? nb[lane + 64] : -1;
  const int dg0 = nb0 >= 0 ? deg[nb0] : 0, dg1 = nb1 >= 0 ? deg[nb1] : 0;
  // (wave-uniform lane numbers: v_readlane, not a trip through the LDS crossbar per pair)
  auto nb_at = [&](int i) -> int32_t {
    return i < 64 ? __builtin_amdgcn_readlane(nb0, i) : i < 128 ? __builtin_amdgcn_readlane(nb1, i - 64) : nb[i];
  };
  auto dg_at = [&](int i) -> int {
    return i < 64 ? __builtin_amdgcn_readlane(dg0, i) : i < 128 ? __builtin_amdgcn_readlane(dg1, i - 64) : deg[nb[i]];
  };
  int32_t v0 = -1, v1 = -1;
  if (part + 1 < d) {
    const int32_t* n0 = col + (size_t)nb_at(part) * width;
    const int d0 = dg_at(part);
    v0 = lane < d0 ? n0[lane] : -1;
    v1 = lane + 64 < d0 ? n0[lane + 64] : -1;
  }
  for (int ia = part; ia + 1 < d; ia += SPLIT) {
    const int da = dg_at(ia);
    const int32_t* na = col + (size_t)nb_at(ia) * width;
    const int32_t c0 = v0, c1 = v1;
    if (ia + SPLIT + 1 < d) {  // (this wave's next a, if it still has a later b)
      const int32_t* nn = col + (size_t)nb_at(ia + SPLIT) * width;
      const int dn = dg_at(ia + SPLIT);
      v0 = lane < dn ? nn[lane] : -1;
      v1 = lane + 64 < dn ? nn[lane + 64] : -1;
    }
    for (int ib = ia + 1; ib < d; ++ib) {
      const int32_t b = nb_at(ib);
      bool hit = __ballot(c0 == b || c1 == b) != 0ull;
      for (int t0 = 128; t0 < da && !hit; t0 += 64) hit = __ballot(t0 + lane < da && na[t0 + lane] == b) != 0ull;
      hits += hit ? 1u : 0u;
      ++pairs;
    }
  }
  // (the sampled row's two counts go to words of its own -- counts[2 sidx], counts[2 sidx + 1] -- and the host adds them up:
  // thousands of atomic adds to the same two words took longer than the counting)
  __shared__ unsigned s_hits[4], s_pairs[4];
  if (lane == 0) {
    s_hits[threadIdx.x >> 6] = hits;
    s_pairs[threadIdx.x >> 6] = pairs;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    counts[2 * (size_t)sidx] = (unsigned long long)(s_hits[0] + s_hits[1] + s_hits[2] + s_hits[3]);
    counts[2 * (size_t)sidx + 1] = (unsigned long long)(s_pairs[0] + s_pairs[1] + s_pairs[2] + s_pairs[3]);
  }
}

}  // namespace

void launch_clustering_sample(const int32_t* col, const int32_t* deg, int32_t width, int64_t N, int32_t nsample,
                              unsigned long long* counts, hipStream_t s) {
  hipLaunchKernelGGL(k_clustering_sample, dim3((unsigned)nsample), dim3(256), 0, s, col, deg, width, N, nsample,
                     counts);  // (four waves per sampled row: one workgroup each)
  HIP_CHECK(hipGetLastError());
}

void launch_move_rows(float* out, const float* in, const int32_t* map, int64_t N, int32_t ld, bool scatter,
                      hipStream_t s) {
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((N + 3) / 4, 4096));
  if (scatter) hipLaunchKernelGGL(k_move_rows<true>, dim3(grid), dim3(256), 0, s, out, in, map, N, ld);
  else hipLaunchKernelGGL(k_move_rows<false>, dim3(grid), dim3(256), 0, s, out, in, map, N, ld);
  HIP_CHECK(hipGetLastError());
}

void launch_move_f32(float* out, const float* in, const int32_t* map, int64_t N, bool scatter, hipStream_t s) {
  const dim3 grid((unsigned)((N + 255) / 256)), block(256);
  if (scatter) hipLaunchKernelGGL((k_move_1d<float, true>), grid, block, 0, s, out, in, map, N);
  else hipLaunchKernelGGL((k_move_1d<float, false>), grid, block, 0, s, out, in, map, N);
  HIP_CHECK(hipGetLastError());
}

void launch_permute_ell(const int32_t* col_in, const float* a_in, const float* w_in, const int32_t* deg_in,
                        const int32_t* from, const int32_t* relabel, int32_t width, int64_t N, int32_t* col_out,
                        float* a_out, float* w_out, int32_t* deg_out, hipStream_t s) {
  const int64_t n = N * width;
  hipLaunchKernelGGL(k_permute_ell, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, col_in, a_in, w_in, deg_in, from,
                     relabel, width, N, col_out, a_out, w_out, deg_out);
  HIP_CHECK(hipGetLastError());
}

}  // namespace osc
