// Reductions behind the single-step dynamics snapshot (reference lattice.py:825-927): temperature / movement statistics,
// total and top structural energy flows, BFS radius of the activated nodes.  The per-node movement and the per-edge
// flows themselves come out of the receipt-rows kernel (receipt_kernels.hip) run on (U_prev, U_next).
#include "dynamics.hpp"

namespace osc {
namespace {

__global__ __launch_bounds__(256) void k_sum_max(const float* v, int64_t n, double* psum, float* pmax) {
  __shared__ double ss[256];
  __shared__ float sm[256];
  double s = 0.0;
  float m = 0.f;
  bool nan = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float x = v[i];
    s += (double)x;
    nan |= x != x;
    m = fmaxf(m, x);
  }
  ss[threadIdx.x] = s;
  sm[threadIdx.x] = nan ? __uint_as_float(0x7FC00000u) : m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      ss[threadIdx.x] += ss[threadIdx.x + o];
      const float a = sm[threadIdx.x], b = sm[threadIdx.x + o];
      sm[threadIdx.x] = (a != a || b != b) ? __uint_as_float(0x7FC00000u) : fmaxf(a, b);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    psum[blockIdx.x] = ss[0];
    pmax[blockIdx.x] = sm[0];
  }
}

// order: value descending, then the caller's (row, column) ids ascending -- the reference ranks its flows with a stable sort
// over argwhere order (lattice.py:879-881), i.e. ties go to the smaller API (i, j); on a lattice stored in an internal row
// order the ELL position would break exact ties differently, and a block's cut at K could then drop an entry the
// reference keeps
__device__ __forceinline__ bool before(float v, int64_t k, float bv, int64_t bk) { return v > bv || (v == bv && k < bk); }

__global__ __launch_bounds__(256) void k_top_select(const float* v, const int32_t* col, const int32_t* api_id, int32_t width,
                                                    int64_t n, int K, float* out_val, int64_t* out_idx, int32_t* out_col) {
  __shared__ float sv[256];
  __shared__ int64_t sk[256], si[256];
  const int64_t chunk = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  auto key_of = [&](int64_t i) -> int64_t {  // (API row, API column) of ELL entry i
    const int32_t r = (int32_t)(i / width), c = col[i];
    const int64_t ar = api_id ? api_id[r] : r, ac = api_id ? api_id[c] : c;
    return (ar << 32) | ac;
  };
  float pv = __uint_as_float(0x7F800000u);  // +inf: everything comes after the (virtual) previous pick
  int64_t pk = -1;
  for (int r = 0; r < K; ++r) {
    float bv = 0.f;
    int64_t bk = -1, bi = -1;
    bool bk_known = false;  // the (API row, API column) key costs a division and two dependent gathers: only ties need it
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
      const float x = v[i];
      if (!(x > 0.f) || x > pv) continue;
      if (x == pv && !(pk < key_of(i))) continue;  // not after the previous pick
      if (bi < 0 || x > bv) {
        bv = x;
        bi = i;
        bk_known = false;
      } else if (x == bv) {
        if (!bk_known) bk = key_of(bi), bk_known = true;
        const int64_t k = key_of(i);
        if (k < bk) bk = k, bi = i;
      }
    }
    if (bi >= 0 && !bk_known) bk = key_of(bi);
    sv[threadIdx.x] = bv;
    sk[threadIdx.x] = bk;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) {
        const float ov = sv[threadIdx.x + o];
        const int64_t ok = sk[threadIdx.x + o], oi = si[threadIdx.x + o];
        if (oi >= 0 && (si[threadIdx.x] < 0 || before(ov, ok, sv[threadIdx.x], sk[threadIdx.x]))) {
          sv[threadIdx.x] = ov;
          sk[threadIdx.x] = ok;
          si[threadIdx.x] = oi;
        }
      }
      __syncthreads();
    }
    pv = sv[0];
    pk = sk[0];
    const int64_t pi = si[0];
    __syncthreads();
    if (threadIdx.x == 0) {
      const size_t o = (size_t)blockIdx.x * K + r;
      out_val[o] = pi >= 0 ? pv : 0.f;
      out_idx[o] = pi;
      out_col[o] = pi >= 0 ? col[pi] : -1;
    }
    if (pi < 0) {  // chunk exhausted: the remaining slots are empty
      if (threadIdx.x == 0)
        for (int q = r + 1; q < K; ++q) {
          out_val[(size_t)blockIdx.x * K + q] = 0.f;
          out_idx[(size_t)blockIdx.x * K + q] = -1;
          out_col[(size_t)blockIdx.x * K + q] = -1;
        }
      return;
    }
  }
}

__global__ void k_bfs_seeds(const float* move2, int64_t N, float thr, int32_t* dist) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) dist[i] = (sqrtf(move2[i] + 1e-12f) >= thr) ? 0 : -1;
}

__global__ void k_bfs_level(const int32_t* col, const int32_t* deg, int32_t width, int64_t N, int32_t* dist,
                            int32_t level, int32_t* changed) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N || dist[i] != level) return;
  const int32_t* c = col + (size_t)i * width;
  bool any = false;
  for (int e = 0; e < deg[i]; ++e) {
    const int32_t j = c[e];
    if (dist[j] < 0) {  // benign race: every writer of this level stores the same value
      dist[j] = level + 1;
      any = true;
    }
  }
  if (any) *changed = 1;
}

}  // namespace

void launch_sum_max(const float* v, int64_t n, int nblocks, double* psum, float* pmax, hipStream_t s) {
  hipLaunchKernelGGL(k_sum_max, dim3(nblocks), dim3(256), 0, s, v, n, psum, pmax);
  HIP_CHECK(hipGetLastError());
}
void launch_top_select(const float* v, const int32_t* col, const int32_t* api_id, int32_t width, int64_t n, int nblocks, int K,
                       float* out_val, int64_t* out_idx, int32_t* out_col, hipStream_t s) {
  hipLaunchKernelGGL(k_top_select, dim3(nblocks), dim3(256), 0, s, v, col, api_id, width, n, K, out_val, out_idx, out_col);
  HIP_CHECK(hipGetLastError());
}
void launch_bfs_seeds(const float* move2, int64_t N, float thr, int32_t* dist, hipStream_t s) {
  hipLaunchKernelGGL(k_bfs_seeds, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, move2, N, thr, dist);
  HIP_CHECK(hipGetLastError());
}
void launch_bfs_level(const int32_t* col, const int32_t* deg, int32_t width, int64_t N, int32_t* dist, int32_t level,
                      int32_t* changed, hipStream_t s) {
  hipLaunchKernelGGL(k_bfs_level, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, col, deg, width, N, dist, level,
                     changed);
  HIP_CHECK(hipGetLastError());
}

}  // namespace osc
