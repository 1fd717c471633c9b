// One-launch CG for small lattices (the reference's own regime: N <= 5000, cloud/app/config.py:10).
//
// The D columns of the multi-RHS CG are independent recurrences (per-column alpha/beta, solver.py:22-36), so a
// workgroup that owns C columns can keep its slab of x, r, p, Ap (4 x N x C floats) in LDS for the whole solve:
// the operator's neighbour gathers become LDS reads, nothing but the graph is re-read per iteration, and the only
// thing workgroups share is the stop test max_c ||r_c|| (solver.py:29) -- one counter barrier per iteration.
// Arithmetic, epsilons and stop rule are those of the general path (cg_kernels.hip); column sums are reduced in fp64.
#include "common.hpp"
#include "small.hpp"

namespace osc {
namespace {

template <int C>
struct Vec {
  float v[C];
};

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

constexpr int T = 512;  // threads per workgroup: one row per thread per sweep keeps the gather latency chain short
constexpr int NW = T / 64;

// block-wide sum of C per-thread partials -> every thread gets the totals
template <int C>
__device__ __forceinline__ void block_sum(const float (&part)[C], double* red /*[NW][C]*/, double (&tot)[C]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const double s = wave_sum_d((double)part[c]);
    if (lane == 0) red[wave * C + c] = s;
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < C; ++c) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += red[w * C + c];
    tot[c] = t;
  }
}

template <int C>
__global__ __launch_bounds__(T) void k_settle_small(const SmallArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int N = a.N;
  float* x = lds;
  float* r = x + (size_t)N * C;
  float* p = r + (size_t)N * C;
  float* ap = p + (size_t)N * C;
  double* red = reinterpret_cast<double*>(ap + (size_t)N * C);  // [NW][C]
  float* s_res = reinterpret_cast<float*>(red + NW * C);         // all LDS lives in the dynamic region (16-B aligned)
  int* s_fail = reinterpret_cast<int*>(s_res + 1);
  const int tid = threadIdx.x;
  const int col0 = blockIdx.x * C;  // first column of this block (pitch ld is a multiple of 4; C divides 4 or is 8)
  const OpParams op = a.op;

  // ---- x0, rhs, r = b - A x0 ------------------------------------------------------------------
  for (int row = tid; row < N; row += T) {
#pragma unroll
    for (int c = 0; c < C; ++c) x[row * C + c] = (col0 + c < a.ld) ? a.x0[(size_t)row * a.ld + col0 + c] : 0.f;
  }
  __syncthreads();
  auto apply = [&](const float* v, int row, float (&out)[C]) {  // out = (A v)_row for this block's columns
    float acc[C], accp[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = accp[c] = 0.f;
    const int deg = a.g.deg[row];
    // transposed ELL ([entry][row]): consecutive threads (rows) read consecutive words.  Entries are fetched 8 at a
    // time (all loads in flight together) before the LDS gathers: the loop is latency-bound, not bandwidth-bound.
    for (int e0 = 0; e0 < deg; e0 += 8) {
      int jj[8];
      float ww[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool ok = e0 + u < deg;
        jj[u] = ok ? a.col_t[(size_t)(e0 + u) * N + row] : row;
        ww[u] = ok ? a.w_t[(size_t)(e0 + u) * N + row] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = fmaf(ww[u], v[jj[u] * C + c], acc[c]);
    }
    if (a.g.path_slot != nullptr) {
      const int ps = a.g.path_slot[row];
      if (ps >= 0) {
        const int pd = a.g.pdeg[ps];
        for (int e = 0; e < pd; ++e) {
          const int j = a.g.pcol[(size_t)ps * a.g.pwidth + e];
          const float w = a.g.pw[(size_t)ps * a.g.pwidth + e];
#pragma unroll
          for (int c = 0; c < C; ++c) accp[c] = fmaf(w, v[j * C + c], accp[c]);
        }
      }
    }
    const float cs = fmaf(op.cs_B, a.B[row], op.cs_const);
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = cs * v[row * C + c] - op.cW * acc[c] - op.cP * accp[c];
  };

  float part[C], part2[C];
  double rz[C], tot[C], tot2[C];
#pragma unroll
  for (int c = 0; c < C; ++c) part[c] = 0.f;
  for (int row = tid; row < N; row += T) {
    float o[C];
    apply(x, row, o);
    const float Bi = a.B[row];
    const float invMd = op.precond ? 1.f / (fmaf(op.md_B, Bi, op.md_const) + 1e-12f) : 1.f;
    const float qb = op.rbB * Bi;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float rr = 0.f;
      if (col0 + c < a.ld) {
        const size_t off = (size_t)row * a.ld + col0 + c;
        rr = (op.rbU * a.U[off] + op.rbY * a.Y[off] + qb * a.psi[col0 + c]) - o[c];
      }
      const float z = rr * invMd;
      r[row * C + c] = rr;
      p[row * C + c] = z;
      part[c] = fmaf(rr, z, part[c]);
    }
  }
  block_sum<C>(part, red, rz);  // ends with a barrier: p is complete before the first gather

  // ---- iterations -------------------------------------------------------------------------------
  int it = 1;
  for (; it <= a.max_iters; ++it) {
#pragma unroll
    for (int c = 0; c < C; ++c) part[c] = 0.f;
    for (int row = tid; row < N; row += T) {
      float o[C];
      apply(p, row, o);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        ap[row * C + c] = o[c];
        part[c] = fmaf(p[row * C + c], o[c], part[c]);
      }
    }
    block_sum<C>(part, red, tot);
    float alpha[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      alpha[c] = (float)(rz[c] / (tot[c] + 1e-18));  // solver.py:25-26
      part[c] = part2[c] = 0.f;
    }
    for (int row = tid; row < N; row += T) {
      const float invMd = op.precond ? 1.f / (fmaf(op.md_B, a.B[row], op.md_const) + 1e-12f) : 1.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int o = row * C + c;
        x[o] = fmaf(p[o], alpha[c], x[o]);
        const float rr = fmaf(-ap[o], alpha[c], r[o]);
        r[o] = rr;
        part[c] = fmaf(rr, rr, part[c]);
        part2[c] = fmaf(rr, rr * invMd, part2[c]);
      }
    }
    block_sum<C>(part, red, tot);
    block_sum<C>(part2, red, tot2);
    // ---- shared stop test: publish this block's max column residual, wait for every block -------
    if (tid == 0) {
      float mx = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {  // NaN propagates (fmaxf would drop it): solver.py:29 reports NaN for a diverged column
        const float v = (float)sqrt(tot[c]);
        mx = (v != v || mx != mx) ? __uint_as_float(0x7FC00000u) : fmaxf(mx, v);
      }
      atomicMax(a.res_bits + it, __float_as_uint(mx));
      __threadfence();
      atomicAdd(a.arrive + it, 1u);
      int fail = 0;
      const uint64_t t_start = wall_clock64();  // 100 MHz constant clock
      while (__hip_atomic_load(a.arrive + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
        __builtin_amdgcn_s_sleep(2);
        // never hang the GPU: when the grid is not co-resident (another persistent grid holds CUs) give up after 5 ms,
        // report, and let the host take the general path (the output buffer is never the caller's state: osc_api.hip)
        if (wall_clock64() - t_start > 500000ull) {
          fail = 1;
          break;
        }
      }
      *s_fail = fail;
      *s_res = __uint_as_float(atomicMax(a.res_bits + it, 0u));  // returning atomic: the value at the memory side
      if (fail) {
        atomicExch(a.status, 2u);
        if (a.host_words != nullptr) {  // tell the polling host at once (every block that gives up writes the same word)
          *reinterpret_cast<volatile uint32_t*>(a.host_words + a.max_iters + 2) = 2u;
          __threadfence_system();
        }
      }
    }
    __syncthreads();
    if (*s_fail) return;
    if (*s_res <= a.tol) break;  // solver.py:30-31, before the beta / p update
    if (it == a.max_iters) break;
    for (int row = tid; row < N; row += T) {
      const float invMd = op.precond ? 1.f / (fmaf(op.md_B, a.B[row], op.md_const) + 1e-12f) : 1.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int o = row * C + c;
        const float beta = (float)(tot2[c] / (rz[c] + 1e-18));  // solver.py:33-34
        p[o] = fmaf(p[o], beta, r[o] * invMd);
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) rz[c] = tot2[c];
    __syncthreads();
  }
  if (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;  // another block gave up
  for (int row = tid; row < N; row += T) {
#pragma unroll
    for (int c = 0; c < C; ++c)
      if (col0 + c < a.ld) a.X[(size_t)row * a.ld + col0 + c] = x[row * C + c];
  }
  if (a.host_words == nullptr) return;
  // The last workgroup to get here publishes the residuals of the iterations that ran and a "done" word into host-mapped
  // memory: the host polls that word instead of paying a copy, an event and a stream wait for a solve of 60-100 us.
  // (Every workgroup's rows of X are out before it counts itself in.)
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    *s_fail = atomicAdd(a.finish, 1u) == gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (!*s_fail) return;
  for (int i = 1 + tid; i <= it; i += T)
    *reinterpret_cast<volatile uint32_t*>(a.host_words + i) = atomicMax(a.res_bits + i, 0u);
  __threadfence_system();
  __syncthreads();
  if (tid == 0) {
    *reinterpret_cast<volatile uint32_t*>(a.host_words + a.max_iters + 2) = 0u;
    __threadfence_system();
  }
}

}  // namespace

size_t small_lds_bytes(int32_t N, int C) { return (size_t)N * C * 4 * sizeof(float) + NW * C * sizeof(double) + 64; }

// every workgroup must be resident at once (counter barrier with a timeout that falls back to the general path)
int small_pick_cols(int32_t N, int32_t ld) {
  for (int C : {4, 2, 1}) {
    const int blocks = (ld + C - 1) / C;
    // one column per workgroup leaves most of the chip idle behind long serial row loops: past N = 6000 the general
    // multi-launch path is faster (measured: N = 5000 0.18 vs 0.23 ms, 7000 0.24 vs 0.23, 9000 0.30 vs 0.23)
    if (C == 1 && N > 6000) continue;
    // co-residency: one workgroup per CU always fits (256 CUs); two per CU when each stays under half the LDS
    const size_t lds = small_lds_bytes(N, C);
    // a second workgroup per CU only pays while the per-workgroup row loops are short (measured, 384 workgroups:
    // N = 2000 x 768 0.162 vs 0.236 ms general; N = 2500 x 768 a tie; N = 5000 x 384 0.319 vs 0.240)
    const int resident = (lds <= 79 * 1024 && N <= 2200) ? 512 : 256;
    if (blocks <= resident && lds <= 150 * 1024) return C;
  }
  return 0;
}

__global__ void k_transpose_ell(const int32_t* col, const float* w, int32_t N, int32_t width, int32_t* col_t, float* w_t) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)N * width) return;
  const int row = (int)(i / width), e = (int)(i % width);
  col_t[(size_t)e * N + row] = col[i];
  w_t[(size_t)e * N + row] = w[i];
}

void launch_transpose_ell(const int32_t* col, const float* w, int32_t N, int32_t width, int32_t* col_t, float* w_t,
                          hipStream_t s) {
  const int64_t n = (int64_t)N * width;
  hipLaunchKernelGGL(k_transpose_ell, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, col, w, N, width, col_t, w_t);
  HIP_CHECK(hipGetLastError());
}

void launch_settle_small(const SmallArgs& a, int C, hipStream_t s) {
  const int blocks = (a.ld + C - 1) / C;
  const size_t shmem = small_lds_bytes(a.N, C);
#define OSC_SMALL(CC)                                                                                             \
  do {                                                                                                            \
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_settle_small<CC>),                              \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));                       \
    hipLaunchKernelGGL(k_settle_small<CC>, dim3(blocks), dim3(T), shmem, s, a);                                   \
  } while (0)
  if (C == 4) OSC_SMALL(4);
  else if (C == 2) OSC_SMALL(2);
  else OSC_SMALL(1);
#undef OSC_SMALL
  HIP_CHECK(hipGetLastError());
}

}  // namespace osc
