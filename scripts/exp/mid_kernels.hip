// Persistent CG for mid-size lattices (6000 < N, N x D <= ~5 M elements, D in {64, 128, 256}): ONE launch per solve.
//
// Between the one-launch small-lattice kernel (state in LDS, N <= 6000) and the bandwidth-bound regime, a settle is ~27
// launches of 4-20 us kernels: ~4.5 us of dispatch per launch and every N x D array streamed through the Infinity Cache
// several times per iteration.  Here each of G workgroups (one per CU, all co-resident) owns a contiguous block of rows
// for ALL columns and keeps its rows of x, r and A p in REGISTERS for the whole solve (one wave per SIMD: 512 registers
// per lane); only the search direction p lives in memory, double-buffered, because the neighbours' rows are gathered
// from it.  Per iteration: the operator apply on the own rows (gather of p, L2 / Infinity Cache), column sums through
// per-workgroup partials, and three grid barriers -- after p.Ap (alpha), after r.r / r.z (beta, stop test), after the
// new p is written (next gather).  The barrier is XCD-hierarchical (MI355X_MICROARCH.md, barrier-xcd: per-group counter,
// the group's last arriver adds to a top counter and publishes the group's generation word) with an agent-scope release
// before arriving and an agent-scope acquire after leaving; every wait is bounded (5 ms) and a timed-out solve reports
// failure: the host then takes the general path (the kernel never writes the caller's state, only its output buffer).
// Arithmetic, epsilons and stop rule are those of cg_kernels.hip (solver.py:6-37); column sums are completed in fp64.
#include "common.hpp"
#include "mid.hpp"

namespace osc {
namespace {

// 512 threads per workgroup, one workgroup per CU: 8 waves per CU keep more neighbour gathers in flight than one wave per
// SIMD did (256 threads: 373 us per settle at N = 20000, D = 128 against 279 us for the multi-launch path) and still
// leave 256 registers per thread (1024 threads: 128, and hipcc spilled the state)
constexpr int MT = 512;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// write-through (sc1) store for bytes another workgroup reads after the next grid barrier: no release fence is needed
// then (cdna_hip_programming.md Guideline 16, R1: sc1 payload, every storing wave drains vmcnt, workgroup barrier, one
// lane signals); hipcc does not count asm stores, grid_barrier()'s own s_waitcnt vmcnt(0) covers them
using v4f = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ void st4_wt(float* p, float4 v) {
  const v4f t = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}

struct MidSync {
  uint32_t* top;    // [0]: arrivals of group leaders (monotonic)
  uint32_t* grp;    // [8][32]: per-group arrival counters (one 128-byte line each)
  uint32_t* gen;    // [8][32]: per-group generation words
  uint32_t* status; // != 0: a wait timed out
};

// grid barrier number `b` (1, 2, ...).  Returns false on timeout (uniform over the workgroup).
__device__ __forceinline__ bool grid_barrier(const MidSync& s, uint32_t b, int ngroups, int group_size, int* s_fail) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the workgroup's barrier
  __syncthreads();
  if (threadIdx.x == 0) {
    int fail = 0;
    const int g = blockIdx.x & 7;
    // (the handed-off bytes -- p rows, partial sums -- were stored write-through and drained above: no release fence)
    const uint32_t arrived = __hip_atomic_fetch_add(s.grp + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const uint64_t t0 = wall_clock64();
    if (arrived == b * (uint32_t)group_size) {  // last of its group: report to the top, wait for every group, release mine
      __hip_atomic_fetch_add(s.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(s.top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < b * (uint32_t)ngroups) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 500000ull) {  // 5 ms at 100 MHz
          fail = 1;
          break;
        }
      }
      __hip_atomic_store(s.gen + g * 32, fail ? 0xFFFFFFFFu : b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      for (;;) {
        const uint32_t v = __hip_atomic_load(s.gen + g * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v == 0xFFFFFFFFu) {
          fail = 1;
          break;
        }
        if (v >= b) break;
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 500000ull) {
          fail = 1;
          break;
        }
      }
    }
    if (fail) __hip_atomic_store(s.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop this CU's stale L1 lines
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *s_fail = fail;
  }
  __syncthreads();
  return *s_fail == 0;
}

// UN = float4 units per thread (rows per workgroup = UN * 256 / Q), Q = float4 per row (D / 4)
template <int UN, int Q>
__global__ __launch_bounds__(MT) void k_settle_mid(const MidArgs a) {
  constexpr int RS = MT / Q;  // rows one unit step of the workgroup covers
  constexpr int DC = 4 * Q;   // columns
  __shared__ double red[MT][4];
  __shared__ float redf[MT];
  __shared__ int s_fail;
  const int tid = threadIdx.x;
  const int cq = tid % Q, sl = tid / Q;  // this thread's column quad and row slot
  const int G = gridDim.x;
  const int rows_per = (a.N + G - 1) / G;
  const int r0 = blockIdx.x * rows_per, r1 = min(a.N, r0 + rows_per);
  const OpParams op = a.op;
  const MidSync sync{a.sync, a.sync + 32, a.sync + 32 + 8 * 32, a.status};
  const int ngroups = min(8, G), group_size = G / ngroups;  // the host launches G as a multiple of 8 (or < 8: one each)
  uint32_t bar = 0;

  float4 x[UN], r[UN], ap[UN];
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto row_of = [&](int j) { return r0 + sl + j * RS; };

  // operator apply on the own rows: out_j = cs v_own - cW sum_e W_e v[col_e]   (cg_kernels.hip: k_spmm)
  auto apply = [&](const float* __restrict__ v, float4 (&out)[UN], const float4 (&own)[UN]) {
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      const int row = row_of(j);
      float4 acc = zero4;
      if (row < r1) {
        const int deg = a.deg[row];
        const int32_t* cr = a.col + (size_t)row * a.width;
        const float* wr = a.w + (size_t)row * a.width;
        for (int e0 = 0; e0 < deg; e0 += 4) {  // four neighbour rows in flight per thread, 16 waves per CU
          int jj[4];
          float ww[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const bool ok = e0 + u < deg;
            jj[u] = ok ? cr[e0 + u] : row;
            ww[u] = ok ? wr[e0 + u] : 0.f;
          }
          float4 g4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) g4[u] = ld4(v + (size_t)jj[u] * DC + 4 * cq);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            acc.x = fmaf(ww[u], g4[u].x, acc.x);
            acc.y = fmaf(ww[u], g4[u].y, acc.y);
            acc.z = fmaf(ww[u], g4[u].z, acc.z);
            acc.w = fmaf(ww[u], g4[u].w, acc.w);
          }
        }
        const float cs = fmaf(op.cs_B, a.B[row], op.cs_const);
        out[j] = make_float4(cs * own[j].x - op.cW * acc.x, cs * own[j].y - op.cW * acc.y, cs * own[j].z - op.cW * acc.z,
                             cs * own[j].w - op.cW * acc.w);
      } else {
        out[j] = zero4;
      }
    }
  };

  // column sums: this workgroup's per-column partial (fp32 over its rows, threads of one column quad folded in slot
  // order) -> part[block][DC]; after the barrier every workgroup adds the G partials in block order, in fp64
  auto publish = [&](const float4& s4, float* part) {
    red[tid][0] = s4.x;
    red[tid][1] = s4.y;
    red[tid][2] = s4.z;
    red[tid][3] = s4.w;
    __syncthreads();
    if (sl == 0) {
      float4 t = zero4;
      for (int k = 0; k < RS; ++k) {
        t.x += (float)red[cq + k * Q][0];
        t.y += (float)red[cq + k * Q][1];
        t.z += (float)red[cq + k * Q][2];
        t.w += (float)red[cq + k * Q][3];
      }
      st4_wt(part + (size_t)blockIdx.x * DC + 4 * cq, t);
    }
    __syncthreads();
  };
  auto collect = [&](const float* part, double (&tot)[4]) {  // every thread gets the totals of its column quad
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int g = sl; g < G; g += RS) {
      const float4 v = ld4(part + (size_t)g * DC + 4 * cq);
      s0 += (double)v.x;
      s1 += (double)v.y;
      s2 += (double)v.z;
      s3 += (double)v.w;
    }
    red[tid][0] = s0;
    red[tid][1] = s1;
    red[tid][2] = s2;
    red[tid][3] = s3;
    __syncthreads();
    tot[0] = tot[1] = tot[2] = tot[3] = 0.0;
    for (int k = 0; k < RS; ++k) {
      tot[0] += red[cq + k * Q][0];
      tot[1] += red[cq + k * Q][1];
      tot[2] += red[cq + k * Q][2];
      tot[3] += red[cq + k * Q][3];
    }
    __syncthreads();
  };

  // ---- x0, rhs, r = b - A x0, z, p = z ----------------------------------------------------------------------------
  float4 own[UN];
#pragma unroll
  for (int j = 0; j < UN; ++j) {
    const int row = row_of(j);
    x[j] = row < r1 ? ld4(a.x0 + (size_t)row * DC + 4 * cq) : zero4;
    own[j] = x[j];
  }
  apply(a.x0, ap, own);
  const float4 psi4 = ld4(a.psi + 4 * cq);
  float4 prz = zero4;
  float* P0 = a.P0;
  float* P1 = a.P1;
#pragma unroll
  for (int j = 0; j < UN; ++j) {
    const int row = row_of(j);
    if (row < r1) {
      const float Bi = a.B[row];
      const float invMd = op.precond ? 1.f / (fmaf(op.md_B, Bi, op.md_const) + 1e-12f) : 1.f;
      const float qb = op.rbB * Bi;
      const float4 u4 = ld4(a.U + (size_t)row * DC + 4 * cq), y4 = ld4(a.Y + (size_t)row * DC + 4 * cq);
      float4 rr;
      rr.x = (op.rbU * u4.x + op.rbY * y4.x + qb * psi4.x) - ap[j].x;
      rr.y = (op.rbU * u4.y + op.rbY * y4.y + qb * psi4.y) - ap[j].y;
      rr.z = (op.rbU * u4.z + op.rbY * y4.z + qb * psi4.z) - ap[j].z;
      rr.w = (op.rbU * u4.w + op.rbY * y4.w + qb * psi4.w) - ap[j].w;
      r[j] = rr;
      const float4 z = make_float4(rr.x * invMd, rr.y * invMd, rr.z * invMd, rr.w * invMd);
      st4_wt(P0 + (size_t)row * DC + 4 * cq, z);
      prz.x = fmaf(rr.x, z.x, prz.x);
      prz.y = fmaf(rr.y, z.y, prz.y);
      prz.z = fmaf(rr.z, z.z, prz.z);
      prz.w = fmaf(rr.w, z.w, prz.w);
    } else {
      r[j] = zero4;
    }
  }
  publish(prz, a.part0);
  if (!grid_barrier(sync, ++bar, ngroups, group_size, &s_fail)) return;
  double rz[4];
  collect(a.part0, rz);

  // ---- iterations ----------------------------------------------------------------------------------------------
  float* Pc = P0;  // current search direction
  float* Pn = P1;
  int it = 1;
  for (; it <= a.max_iters; ++it) {
    // own rows of p (written by this workgroup, read back through L2)
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      const int row = row_of(j);
      own[j] = row < r1 ? ld4(Pc + (size_t)row * DC + 4 * cq) : zero4;
    }
    apply(Pc, ap, own);
    float4 ppap = zero4;
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      ppap.x = fmaf(own[j].x, ap[j].x, ppap.x);
      ppap.y = fmaf(own[j].y, ap[j].y, ppap.y);
      ppap.z = fmaf(own[j].z, ap[j].z, ppap.z);
      ppap.w = fmaf(own[j].w, ap[j].w, ppap.w);
    }
    publish(ppap, a.part0);
    if (!grid_barrier(sync, ++bar, ngroups, group_size, &s_fail)) return;
    double tot[4];
    collect(a.part0, tot);
    float alpha[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) alpha[c] = (float)(rz[c] / (tot[c] + 1e-18));  // solver.py:25-26
    float4 prr = zero4;
    prz = zero4;
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      const int row = row_of(j);
      if (row < r1) {
        const float invMd = op.precond ? 1.f / (fmaf(op.md_B, a.B[row], op.md_const) + 1e-12f) : 1.f;
        x[j].x = fmaf(own[j].x, alpha[0], x[j].x);
        x[j].y = fmaf(own[j].y, alpha[1], x[j].y);
        x[j].z = fmaf(own[j].z, alpha[2], x[j].z);
        x[j].w = fmaf(own[j].w, alpha[3], x[j].w);
        float4 rr;
        rr.x = fmaf(-ap[j].x, alpha[0], r[j].x);
        rr.y = fmaf(-ap[j].y, alpha[1], r[j].y);
        rr.z = fmaf(-ap[j].z, alpha[2], r[j].z);
        rr.w = fmaf(-ap[j].w, alpha[3], r[j].w);
        r[j] = rr;
        prr.x = fmaf(rr.x, rr.x, prr.x);
        prr.y = fmaf(rr.y, rr.y, prr.y);
        prr.z = fmaf(rr.z, rr.z, prr.z);
        prr.w = fmaf(rr.w, rr.w, prr.w);
        prz.x = fmaf(rr.x, rr.x * invMd, prz.x);
        prz.y = fmaf(rr.y, rr.y * invMd, prz.y);
        prz.z = fmaf(rr.z, rr.z * invMd, prz.z);
        prz.w = fmaf(rr.w, rr.w * invMd, prz.w);
      }
    }
    publish(prr, a.part1);
    publish(prz, a.part2);
    if (!grid_barrier(sync, ++bar, ngroups, group_size, &s_fail)) return;
    double trr[4], trz[4];
    collect(a.part1, trr);
    collect(a.part2, trz);
    // shared stop test: max over all columns of ||r_c||, NaN-propagating (solver.py:29)
    float mx = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float v = (float)sqrt(trr[c]);
      mx = (v != v || mx != mx) ? __uint_as_float(0x7FC00000u) : fmaxf(mx, v);
    }
    redf[tid] = mx;
    __syncthreads();
    for (int o = MT / 2; o > 0; o >>= 1) {
      if (tid < o) {
        const float p = redf[tid], q = redf[tid + o];
        redf[tid] = (p != p || q != q) ? __uint_as_float(0x7FC00000u) : fmaxf(p, q);
      }
      __syncthreads();
    }
    const float res = redf[0];
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) a.res[it] = res;
    if (res <= a.tol) break;  // solver.py:30-31, before the beta / p update
    if (it == a.max_iters) break;
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      const int row = row_of(j);
      if (row < r1) {
        const float invMd = op.precond ? 1.f / (fmaf(op.md_B, a.B[row], op.md_const) + 1e-12f) : 1.f;
        float4 pn;
        pn.x = fmaf(own[j].x, (float)(trz[0] / (rz[0] + 1e-18)), r[j].x * invMd);  // solver.py:33-35
        pn.y = fmaf(own[j].y, (float)(trz[1] / (rz[1] + 1e-18)), r[j].y * invMd);
        pn.z = fmaf(own[j].z, (float)(trz[2] / (rz[2] + 1e-18)), r[j].z * invMd);
        pn.w = fmaf(own[j].w, (float)(trz[3] / (rz[3] + 1e-18)), r[j].w * invMd);
        st4_wt(Pn + (size_t)row * DC + 4 * cq, pn);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) rz[c] = trz[c];
    if (!grid_barrier(sync, ++bar, ngroups, group_size, &s_fail)) return;
    float* t = Pc;
    Pc = Pn;
    Pn = t;
  }
  if (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
#pragma unroll
  for (int j = 0; j < UN; ++j) {
    const int row = row_of(j);
    if (row < r1) st4(a.X + (size_t)row * DC + 4 * cq, x[j]);
  }
}

template <int Q>
void launch_q(const MidArgs& a, int un, int grid, hipStream_t s) {
#define OSC_MID(U)                                                                              \
  if (un <= U) {                                                                               \
    hipLaunchKernelGGL((k_settle_mid<U, Q>), dim3(grid), dim3(MT), 0, s, a);                    \
    return;                                                                                    \
  }
  OSC_MID(2)
  OSC_MID(4)
  OSC_MID(6)
#undef OSC_MID
  throw std::runtime_error("launch_settle_mid: rows per workgroup out of range");
}

}  // namespace

MidPlan mid_plan(int64_t N, int32_t dcols, int32_t ld, int cus, bool forced) {
  MidPlan p{};
  p.ok = false;
  if (ld != dcols || (dcols != 64 && dcols != 128 && dcols != 256) || N > (int64_t)1 << 30) return p;
  if (!forced && N <= 6000) return p;  // the LDS-resident one-launch kernel serves those
  const int Q = dcols / 4, RS = MT / Q;
  int G = std::min(cus, 256) & ~7;  // one workgroup per CU, a multiple of 8 (the barrier's groups)
  if (G < 8) return p;
  const int64_t rows_per = (N + G - 1) / G;
  const int un = (int)((rows_per + RS - 1) / RS);
  if (un > 6) return p;  // x, r, A p and the own rows of p in registers: 16 un of the 256 a thread has at 8 waves per CU
  p.ok = true;
  p.grid = G;
  p.un = un;
  p.Q = Q;
  return p;
}

void launch_settle_mid(const MidArgs& a, const MidPlan& p, hipStream_t s) {
  if (p.Q == 16) launch_q<16>(a, p.un, p.grid, s);
  else if (p.Q == 32) launch_q<32>(a, p.un, p.grid, s);
  else launch_q<64>(a, p.un, p.grid, s);
  HIP_CHECK(hipGetLastError());
}

}  // namespace osc
