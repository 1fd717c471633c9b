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
// re-order is worth its few milliseconds.  One wave per sampled row; counts[0] += adjacent pairs, counts[1] += pairs.
__global__ __launch_bounds__(256) void k_clustering_sample(const int32_t* col, const int32_t* deg, int32_t width,
                                                           int64_t N, int32_t nsample, unsigned long long* counts) {
  const int lane = threadIdx.x & 63;
  const int sidx = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (sidx >= nsample) return;
  const int64_t row = (int64_t)sidx * N / nsample;
  const int d = deg[row];
  const int32_t* nb = col + (size_t)row * width;
  unsigned hits = 0, pairs = 0;
  for (int t = lane; t < d * d; t += 64) {
    const int ia = t / d, ib = t % d;
    if (ia >= ib) continue;
    const int32_t a = nb[ia], b = nb[ib];
    ++pairs;
    const int da = deg[a];
    const int32_t* na = col + (size_t)a * width;
    for (int e = 0; e < da; ++e)
      if (na[e] == b) {
        ++hits;
        break;
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    hits += __shfl_xor(hits, o, 64);
    pairs += __shfl_xor(pairs, o, 64);
  }
  if (lane == 0) {
    atomicAdd(counts, (unsigned long long)hits);
    atomicAdd(counts + 1, (unsigned long long)pairs);
  }
}

}  // namespace

void launch_clustering_sample(const int32_t* col, const int32_t* deg, int32_t width, int64_t N, int32_t nsample,
                              unsigned long long* counts, hipStream_t s) {
  hipLaunchKernelGGL(k_clustering_sample, dim3((unsigned)((nsample + 3) / 4)), dim3(256), 0, s, col, deg, width, N, nsample,
                     counts);
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
