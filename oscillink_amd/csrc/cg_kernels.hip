// CG hot loop of the lattice settle path, hand-written for gfx950 (wave64, HBM-bound).
//
//   k_spmm      : operator apply  out = (cs_i) x_i - cW sum_j W_ij x_j - cP sum_j Wp_ij x_j   (lattice.py:173-182, 247-255)
//                 fused with the per-column dot product the CG needs next (solver.py:21,25) and, in INIT mode,
//                 with the right-hand side, residual, Jacobi scaling and first search direction (solver.py:19-22).
//   k_update_xr : x += alpha p ; r -= alpha Ap ; column sums of r.r and r.z   (solver.py:27-29, 32-33)
//   k_update_p  : p = z + beta p with z = r / (Md + 1e-12) recomputed          (solver.py:32, 35)
//   k_reduce_*  : deterministic second stage of the column reductions -> alpha, beta, residual
//
// Layout: every N x D array is row-major with pitch ld (multiple of 4 floats) so one wave reads a row as
// 16-byte-per-lane coalesced segments.  LPR lanes cover one row's column window (LPR*4*NCH floats); a wave
// carries 64/LPR rows.  Neighbour ids/weights are read one entry per lane and broadcast with v_readlane
// (LPR == 64: wave-uniform row base -> scalar address + lane offset) or ds_bpermute (LPR < 64).
// Column sums never use atomics: each block keeps per-lane partials in registers over its grid-stride rows,
// folds its 4 waves through LDS and writes one row of part[grid][ld]; a small second kernel finishes in fp64.
#include "common.hpp"
#include <type_traits>

namespace osc {

namespace {

// max that PROPAGATES NaN (fmaxf drops it): a column that diverged to NaN / Inf must reach the stop test as NaN, like
// np.linalg.norm(...).max() of the reference (solver.py:29) -- the solve then runs to max_iters and reports res = NaN.
// The canonical NaN's bit pattern (0x7FC00000) is above every non-negative float's, so the uint atomicMax keeps it.
__device__ __forceinline__ float nanmax(float a, float b) {
  return (a != a || b != b) ? __uint_as_float(0x7FC00000u) : fmaxf(a, b);
}


__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// streaming store (written once, not re-read by this kernel): keeps the gathered operand's lines in the caches
// (measured on config 3: nontemporal streams took the settle from 8.71 to 8.37 ms)
using v4f = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ void st4_stream(float* p, float4 v) {
  const v4f t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
}
// streaming load (read once per kernel)
__device__ __forceinline__ float4 ld4_stream(const float* p) {
  const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}
// ... or an ordinary one, by a kernel-uniform flag (UpdateArgs::temporal)
__device__ __forceinline__ float4 ld4_sel(const float* p, bool temporal) { return temporal ? ld4(p) : ld4_stream(p); }
__device__ __forceinline__ void st4_sel(float* p, float4 v, bool temporal) {
  if (temporal) st4(p, v);
  else st4_stream(p, v);
}
__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 c) {
  return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}
__device__ __forceinline__ float4 mulacc4(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

// offset of (row, col) in a slab-major array of `rows` rows (32-column slabs)
__device__ __forceinline__ size_t blk_off(int64_t rows, int row, int col) {
  return ((size_t)(col >> 5) * (size_t)rows + (size_t)row) * 32 + (size_t)(col & 31);
}
// offset of the float4 at (row, col) of the search direction P of an update kernel
__device__ __forceinline__ size_t p_off(const UpdateArgs& a, int row, int col) {
  return a.pblk ? blk_off(a.pblk, row, col) : (size_t)row * a.ld + col;
}

template <int LPR>
__device__ __forceinline__ int bcast_i(int v, int sub, int t) {
  if constexpr (LPR == 64) {
    return __builtin_amdgcn_readlane(v, t);
  } else {
    return __shfl(v, sub * LPR + t, 64);
  }
}
template <int LPR>
__device__ __forceinline__ float bcast_f(float v, int sub, int t) {
  return __int_as_float(bcast_i<LPR>(__float_as_int(v), sub, t));
}

// fold per-lane column partials of the 4 waves of a block and write one row of part[grid][ld]
template <int LPR, int NCH>
__device__ __forceinline__ void block_fold(float4 (&dot)[NCH], float* red /*[4][CPW]*/, float* part, int32_t ld,
                                           int32_t c0, int32_t c1) {
  constexpr int CPW = NCH * LPR * 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, lr = lane % LPR;
  if constexpr (LPR < 64) {
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        dot[ch].x += __shfl_xor(dot[ch].x, o, 64);
        dot[ch].y += __shfl_xor(dot[ch].y, o, 64);
        dot[ch].z += __shfl_xor(dot[ch].z, o, 64);
        dot[ch].w += __shfl_xor(dot[ch].w, o, 64);
      }
    }
  }
  __syncthreads();  // red may still be read by a previous fold
  if (sub == 0) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) st4(red + wave * CPW + (ch * LPR + lr) * 4, dot[ch]);
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < CPW; idx += 256) {
    const int col = c0 + idx;
    if (col < c1) part[(size_t)blockIdx.x * ld + col] = (red[idx] + red[CPW + idx]) + (red[2 * CPW + idx] + red[3 * CPW + idx]);
  }
}

constexpr int OSC_SPMM_U1 = 2;
constexpr int OSC_SPMM_U8 = 2;  // 8 lanes per row (xs mode): 8 rows per wave, 2 neighbour rows of each in flight (4 and 8: slower)
template <int NCH>
struct Unroll {  // neighbour rows fetched per batch (all loads in flight together)
  static constexpr int U = NCH == 1 ? OSC_SPMM_U1 : (NCH <= 3 ? 4 : (NCH <= 6 ? 2 : 1));
};

// ---------------------------------------------------------------------------------------------
// UDEEP > 0: that many neighbour rows in flight per row instead (lattices stored in a local row order: their gathers hit
// the XCD's L2, so the apply is bound by how many hits a wave keeps in flight, not by misses -- measured on 1000
// clusters x 100 rows, N = 100k, D = 768: 0.595 ms per apply at 2, 0.490 at 4, 0.450 at 8; on an unstructured lattice
// the extra registers only cost occupancy: 7.81 -> 8.17 ms per settle on the plain path of config 3)
constexpr int OSC_SPMM_UDEEP = 8;
template <int LPR, int NCH, int MODE, int UDEEP = 0>
__global__ __launch_bounds__(256) void k_spmm(const SpmmArgs a) {
  constexpr int RPW = 64 / LPR;
  constexpr int CPW = NCH * LPR * 4;
  constexpr int U = UDEEP > 0 ? UDEEP : (LPR == 8 ? OSC_SPMM_U8 : Unroll<NCH>::U);
  __shared__ __attribute__((aligned(16))) float red[4 * CPW];
  if (a.gate != nullptr && *a.gate <= a.gate_tol) return;  // converged earlier: speculative launch is a no-op
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, lr = lane % LPR;
  const int32_t ld = a.ld;
  const bool xs = a.xs != 0;  // workgroup-uniform
  if (xs) {  // this workgroup's row of part[][] is summed over every column by the reduce kernels: clear the columns
             // of the slabs other XCDs own (block_fold's leading barrier orders this against the folds below)
    for (int c = a.c0 + threadIdx.x; c < a.c1; c += 256) a.part[(size_t)blockIdx.x * ld + c] = 0.f;
    if ((int)(blockIdx.x >> 3) >= a.xs) return;  // a.xs workgroups per XCD do the work
  }
  // slab loop: one pass over [c0, c1) normally; in xs mode the slabs x, x+8, ... of this workgroup's XCD
  const int xgroups = xs ? a.xs_groups : 1, xgrp = (int)(blockIdx.x & 7) % xgroups, xpart = (int)(blockIdx.x & 7) / xgroups;
  for (int32_t sc0 = xs ? a.c0 + xgrp * CPW : a.c0; sc0 < a.c1; sc0 += xs ? xgroups * CPW : (1 << 30)) {
  const int32_t sc1 = xs ? min(a.c1, sc0 + CPW) : a.c1;
  int coff[NCH];
  bool cok[NCH];
  float4 psi4[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    coff[ch] = sc0 + (ch * LPR + lr) * 4;
    cok[ch] = coff[ch] < sc1;
    psi4[ch] = f4(0.f);
    if (MODE == SPMM_INIT && cok[ch]) psi4[ch] = ld4(a.psi + coff[ch]);
  }
  float4 dot[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) dot[ch] = f4(0.f);

  // XCD-aware row order.  Workgroups b and b+8 share an XCD (round-robin dispatch): XCD x = b % 8 takes the x-th
  // eighth of the row range and its workgroups sweep that eighth together, 4*RPW rows per workgroup per step, so at any
  // moment the rows in flight on one XCD form a narrow contiguous window.  On lattices whose row order has locality
  // (clustered anchors) the neighbours of the window are inside it and their rows are re-read from that XCD's 4 MB L2
  // instead of from the fabric; on unstructured graphs the order is irrelevant.  (gridDim.x is a multiple of 8 or < 8.)
  const int64_t span = a.N - a.row0;
  int64_t rbeg, rend, rstep;
  if (xs) {  // the rows of this XCD's part, shared among its workgroups
    const int64_t parts = 8 / xgroups;
    rbeg = a.row0 + span * xpart / parts + (int64_t)(blockIdx.x >> 3) * 4 * RPW;
    rend = a.row0 + span * (xpart + 1) / parts;
    rstep = (int64_t)a.xs * 4 * RPW;
  } else if (gridDim.x >= 8 && (gridDim.x & 7) == 0) {
    const int64_t per_xcd = ((span + 7) / 8 + 4 * RPW - 1) / (4 * RPW) * (4 * RPW);
    const int64_t x0 = a.row0 + (int64_t)(blockIdx.x & 7) * per_xcd;
    rend = min(a.N, x0 + per_xcd);
    rbeg = x0 + (int64_t)(blockIdx.x >> 3) * 4 * RPW;
    rstep = (int64_t)(gridDim.x >> 3) * 4 * RPW;
  } else {
    rbeg = a.row0 + (int64_t)blockIdx.x * 4 * RPW;
    rend = a.N;
    rstep = (int64_t)gridDim.x * 4 * RPW;
  }
  for (int64_t rb = rbeg + (int64_t)wave * RPW; rb < rend; rb += rstep) {
    int row = (int)rb + sub;
    const bool rok = row < rend;
    if (!rok) row = (int)rend - 1;
    if constexpr (LPR == 64) row = __builtin_amdgcn_readfirstlane(row);
    int deg = rok ? a.g.deg[row] : 0;
    if constexpr (LPR == 64) deg = __builtin_amdgcn_readfirstlane(deg);
    int maxdeg = deg;
    if constexpr (LPR < 64) {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, o, 64));
    }
    // operand addressing: row-major, or slab-major in xs mode (one slab per pass: the slab base is loop-invariant)
    const bool xb = LPR == 8 && a.xblk != 0;
    const float* xbase = xb ? a.X + (size_t)(sc0 >> 5) * (size_t)a.xblk * 32 + lr * 4 : a.X + coff[0];
    const size_t xpitch = xb ? 32 : (size_t)ld;
    const float* xrow = xbase + (size_t)row * xpitch;
    float4 xs[NCH], acc[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      xs[ch] = cok[ch] ? ld4(xrow + (coff[ch] - coff[0])) : f4(0.f);
      acc[ch] = f4(0.f);
    }
    const int32_t* crow = a.g.col + (size_t)row * a.g.width;
    const float* wrow = a.g.w + (size_t)row * a.g.width;
    // edge lists are fetched NBF batches of LPR entries at a time (all index loads of a row in flight together: with
    // 8 lanes per row a 32-wide ELL row is four batches, and one exposed index latency instead of four)
    constexpr int NBF = LPR == 8 ? 4 : 1;
    for (int e0 = 0; e0 < maxdeg; e0 += LPR * NBF) {
      int cjb[NBF];
      float wjb[NBF];
#pragma unroll
      for (int b = 0; b < NBF; ++b) {
        const int e = e0 + b * LPR + lr;
        cjb[b] = row;
        wjb[b] = 0.f;
        if (e < deg) {
          cjb[b] = crow[e];
          wjb[b] = wrow[e];
        }
      }
#pragma unroll
      for (int b = 0; b < NBF; ++b) {
        const int cnt = min(LPR, maxdeg - e0 - b * LPR);  // <= 0: nothing left (the loop below does not run)
        const int cj = cjb[b];
        const float wj = wjb[b];
        for (int t = 0; t < cnt; t += U) {
          float4 v[U][NCH];
          float wv[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int j = bcast_i<LPR>(cj, sub, t + u);  // t+u <= LPR-1: cnt <= LPR and LPR % U == 0
            wv[u] = bcast_f<LPR>(wj, sub, t + u);
            const float* xj = xbase + (size_t)j * xpitch;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) v[u][ch] = cok[ch] ? ld4(xj + (coff[ch] - coff[0])) : f4(0.f);
          }
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) acc[ch] = fma4(wv[u], v[u][ch], acc[ch]);
        }
      }
    }
    // chain prior rows (a handful): second tiny ELL addressed through path_slot
    float4 accp[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) accp[ch] = f4(0.f);
    if (a.g.path_slot != nullptr) {
      const int ps = rok ? a.g.path_slot[row] : -1;
      if (ps >= 0) {
        const int pd = a.g.pdeg[ps];
        for (int e = 0; e < pd; ++e) {
          const int j = a.g.pcol[(size_t)ps * a.g.pwidth + e];
          const float w = a.g.pw[(size_t)ps * a.g.pwidth + e];
          const float* xj = xbase + (size_t)j * xpitch;
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch)
            if (cok[ch]) accp[ch] = fma4(w, ld4(xj + (coff[ch] - coff[0])), accp[ch]);
        }
      }
    }
    if (rok) {
      const float Bi = a.B[row];
      const float cs = fmaf(a.op.cs_B, Bi, a.op.cs_const);
      float invMd = 1.f;
      if (MODE == SPMM_INIT && a.op.precond) invMd = 1.f / (fmaf(a.op.md_B, Bi, a.op.md_const) + 1e-12f);
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        if (!cok[ch]) continue;
        float4 o;
        o.x = cs * xs[ch].x - a.op.cW * acc[ch].x - a.op.cP * accp[ch].x;
        o.y = cs * xs[ch].y - a.op.cW * acc[ch].y - a.op.cP * accp[ch].y;
        o.z = cs * xs[ch].z - a.op.cW * acc[ch].z - a.op.cP * accp[ch].z;
        o.w = cs * xs[ch].w - a.op.cW * acc[ch].w - a.op.cP * accp[ch].w;
        const size_t off = (size_t)row * ld + coff[ch];
        if (MODE == SPMM_AP) {
          st4_stream(a.OUT + off, o);
          dot[ch] = mulacc4(xs[ch], o, dot[ch]);
        } else if (MODE == SPMM_DOT) {
          dot[ch] = mulacc4(xs[ch], o, dot[ch]);
        } else {  // INIT: r = b - A x0 ; z = r / (Md + eps) ; p = z ; rz = sum r.z   (solver.py:19-22)
          // warm-started settle: the rhs term U is the gathered operand itself (x0 = U), already in xs; the stationary
          // solve has no U term at all (rbU = 0) and starts from Y, which is then the gathered operand
          const float4 u = (a.U == a.X) ? xs[ch] : (a.op.rbU != 0.f ? ld4_stream(a.U + off) : f4(0.f));
          const float4 y = (a.Y == a.X) ? xs[ch] : ld4_stream(a.Y + off);
          const float qb = a.op.rbB * Bi;
          float4 r, z;
          r.x = (a.op.rbU * u.x + a.op.rbY * y.x + qb * psi4[ch].x) - o.x;
          r.y = (a.op.rbU * u.y + a.op.rbY * y.y + qb * psi4[ch].y) - o.y;
          r.z = (a.op.rbU * u.z + a.op.rbY * y.z + qb * psi4[ch].z) - o.z;
          r.w = (a.op.rbU * u.w + a.op.rbY * y.w + qb * psi4[ch].w) - o.w;
          z = make_float4(r.x * invMd, r.y * invMd, r.z * invMd, r.w * invMd);
          if (a.OUT != a.X) st4_stream(a.OUT + off, xs[ch]);  // in-place solve (x0 is the work array): nothing to copy
          st4_stream(a.R + off, r);
          st4(a.P + (LPR == 8 && a.pblk != 0 ? blk_off(a.pblk, row, coff[ch]) : off), z);
          dot[ch] = mulacc4(r, z, dot[ch]);
        }
      }
    }
  }
  block_fold<LPR, NCH>(dot, red, a.part, ld, sc0, sc1);
  }  // slab loop
}

// ---------------------------------------------------------------------------------------------
// Operator apply, source-blocked (the CG matvec wherever the XCD-affine slab mode runs: blocked_plan in osc_api.hip).  Measured (scripts/exp/gather_bench.hip, profiles/r02_gather_bench.txt): a CU completes a random
// 128-byte row gather every 2.4 clk when the rows come from <= 3.6 MB per XCD, every 7.0 clk from a 12.8 MB slab -- the
// request rate, not the bytes, is what the plain apply pays for.  Here the neighbour rows are visited block by block:
//   * source rows are cut into nb blocks, nb such that a row has ~3.3 edges into each (4 slots per row and block);
//   * every gathering wave owns a fixed set of row groups (8 rows each, dealt round-robin) per slice of the destination
//     rows and runs  for block b: for my groups: gather the edges that point into b,  the per-row sums staying in
//     registers across the blocks (no partial results through memory); the destination rows are cut into slices so that
//     the rows in flight on an XCD fit its waves' registers.  All waves of an XCD walk the same (slab, slice, block)
//     sequence from the same start, so what they gather from at any time is a few neighbouring blocks rather than the
//     whole slab (L2 hits 25 -> 56 M, misses 58 -> 26 M per launch at config 3);
//   * loads of one wave complete in issue order, so a miss in a gathering wave's stream holds up every L2 hit behind it.
//     Hence two roles per workgroup: waves 0-6 gather (LDS reads and slab rows only), wave 7 copies the next block's edge
//     lists (block-major copy of the graph, BlockedView: always misses) into the other half of an LDS staging area by
//     LDS-DMA while the others gather; one workgroup barrier per block.
// What was tried on top and did not pay (scripts/exp/blocked_apply/README.md): pulling the next block into the L2 with
// dword-per-line loads, gated or not by a progress counter in the XCD's L2 (no change: 877-890 us either way), and
// holding leaders back at that counter (stragglers fall out of the resident set and get later still: 1.8-2.3 ms).
// Same terms per row as k_spmm; the order of summation differs where a row has more than OSC_BLK_SLOTS edges into one
// block (those move to a later block's free slots, launch_blocked_fill): results agree with k_spmm's to fp32 rounding
// of the sums (3e-8 relative on the state).  The chain prior's few rows are applied by k_chain_fix behind this kernel.
constexpr int kBlkGroups = 16;   // row groups per gathering wave (4 registers each for the sums)
constexpr int kBlkGatherWaves = 7;  // + the list wave: workgroups of 512 (3 + 1 with 17 groups: 0.71 instead of 0.66 ms at config 3)

__device__ __forceinline__ float4 ld4_at(const float* base, uint32_t byte_off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float ld1_at(const float* base, uint32_t byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ int2 ld2_at(const int2* base, uint32_t byte_off) {
  return *reinterpret_cast<const int2*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ int opaque(int v) {  // a wave-uniform value the optimiser must not fold into hoisted products
  asm volatile("" : "+s"(v));
  return v;
}

struct BlkPhase {  // sub-phase ph = (slab, slice, block): every wave of the XCD walks the same sequence
  int sc0, slice, b;
};

// the slab's column sums of one workgroup: every wave (the list wave with zeros) folds its 8 row lanes, then the waves'
// rows of 32 columns are added in wave order
template <int NW>
__device__ __forceinline__ void blk_fold(float4 dot, float (&red)[NW][32], float* part, int32_t ld, int32_t c0, int32_t c1,
                                         int wave, int lane) {
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    dot.x += __shfl_xor(dot.x, o, 64);
    dot.y += __shfl_xor(dot.y, o, 64);
    dot.z += __shfl_xor(dot.z, o, 64);
    dot.w += __shfl_xor(dot.w, o, 64);
  }
  __syncthreads();  // red may still be read by a previous fold
  if (lane < 8) st4(&red[wave][lane * 4], dot);
  __syncthreads();
  if (threadIdx.x < 32 && c0 + (int)threadIdx.x < c1) {
    float sacc = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) sacc += red[w][threadIdx.x];
    part[(size_t)blockIdx.x * ld + c0 + threadIdx.x] = sacc;
  }
}

// INIT (BlkInit): the solve's initial residual in the same launch.  X holds x0 slab-major, and where a row's sums are
// complete the epilogue forms r = rhs - A x0 with rhs = rbU x0 + rbY y + rbB B psi (y = ii.Y's row, or x0 itself: the
// warm-started settle, whose rhs state term IS x0, and the U* solve, which starts from Y and has no state term),
// z = M^-1 r, stores r row-major and z SLAB-MAJOR into ii.Z -- the first search direction, in the array the caller uses
// as P from here on --, copies x0 into the solution array where that is another one, and the column sums are those of
// r . z.  Saves the separate pass over A x0, x0, y, r, p (k_init_finish: five array passes) for one more row read or
// write here.
// PD: gather rounds a wave keeps in flight (round 5).  A wave's loads return in issue order and about a fifth of the gathers
// miss the L2, so a round of 32 lines practically always waits for one fabric round trip: with one round in flight the
// kernel is bound by bytes in flight / miss latency.  PD > 1 issues round g + PD - 1 before it consumes round g (the
// slots' weights and the gathered rows of PD rounds in registers).  WPE: waves per SIMD the register budget is set for.
// STAMP (diagnostic instantiations, OSC_BLK_STAMP=1; loop form only): every wave adds the shader cycles it spent in its
// gather rounds / epilogues / at the barrier (the list wave: fetching / at the barrier) to words of its own in the
// buffer ii.R points to -- where a launch's time goes, wave by wave (osc_profile_get slots 8-13).
template <int GM, int CW, bool INIT = false, int PD = 1, int WPE = 4, bool STAMP = false>
__global__ __launch_bounds__((CW + 1) * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_apply_blocked(const BlkArgs a,
                                                                                                                const BlkInit ii) {
  constexpr int SL = OSC_BLK_SLOTS, NT = (CW + 1) * 64;
  __shared__ __attribute__((aligned(16))) float red[CW + 1][32];
  // per gathering wave: the slots of each of its rows in the current block, and (other half) in the next one
  __shared__ __attribute__((aligned(16))) int2 stage[2][GM][CW][8 * SL];
  if constexpr (!INIT) {  // (the INIT pass is never speculative)
    if (a.gate != nullptr && *a.gate <= a.gate_tol) return;
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane >> 3, lr = lane & 7;
  const int32_t ld = a.ld;
  for (int c = a.c0 + threadIdx.x; c < a.c1; c += NT) a.part[(size_t)blockIdx.x * ld + c] = 0.f;
  if ((int)(blockIdx.x >> 3) >= a.xs) return;
  const int xcd = (int)(blockIdx.x & 7), wgx = (int)(blockIdx.x >> 3);
  const int xgroups = a.xs_groups, xgrp = xcd % xgroups, xpart = xcd / xgroups, parts = 8 / xgroups;
  const int rlo = (int)((int64_t)a.N * xpart / parts), rhi = (int)((int64_t)a.N * (xpart + 1) / parts);
  const int nb = a.nb, ng = a.groups;
  const int W8 = a.xs * CW * 8;  // rows one "deal" of groups covers (8 per gathering wave of the XCD)
  const int slice_rows = W8 * ng;
  const int nslab = a.c1 - a.c0 > xgrp * 32 ? ((a.c1 - a.c0 - xgrp * 32 + xgroups * 32 - 1) / (xgroups * 32)) : 0;
  const int per_slab = a.slices * nb, nphase = nslab * per_slab;
  auto phase = [&](int ph) {
    BlkPhase p;
    const int q = ph / per_slab, r = ph - q * per_slab;
    p.sc0 = a.c0 + (xgrp + q * xgroups) * 32;
    p.slice = r / nb;
    p.b = r - p.slice * nb;
    return p;
  };
  auto slab_base = [&](int sc0) { return a.X + (size_t)(sc0 >> 5) * (size_t)a.N * 32; };

  if (wave == CW) {
    // ---- the last wave: the edge lists --------------------------------------------------------------------------------
    // For a fixed group index g the slot rows of the workgroup's CW gathering waves are 8 * CW consecutive lattice rows, i.e.
    // one contiguous piece of the block-major copy (and of the staging area, laid out [g][wave][row][slot]): it is copied
    // global -> LDS by LDS-DMA loads of 1 KB per instruction, no registers in between, ALL pieces of a sub-phase in flight
    // together (the loads always miss; under the gathers' traffic every dependent round trip costs several microseconds).
    // A group that starts inside the lattice but runs past its end (or past the slice) copies whatever follows in the
    // array -- other rows' valid slots, or the zeroed padding behind the last block (blocked_view: >= 8 * CW rows) -- and
    // the sums of such rows are never stored; a group that starts past the end is skipped here AND by the gathering waves.
    constexpr int GROUP_BYTES = CW * 8 * SL * 8, PIECES = (GROUP_BYTES + 1023) / 1024;
    static_assert(GM * PIECES <= 60, "outstanding vector-memory operations of one wave");
    const unsigned stage_lds = (unsigned)(size_t)&stage[0][0][0][0];
    auto fetch_slots = [&](int ph) {
      const BlkPhase p = phase(ph);
      const char* sb = reinterpret_cast<const char*>(a.slots + (size_t)p.b * (size_t)a.N * SL);
      const int row0 = rlo + p.slice * slice_rows + ((wgx * CW) << 3);
      const int w8 = opaque(W8);
#pragma unroll
      for (int g = 0; g < GM; ++g) {
        const bool inside = g < ng && row0 + g * w8 < a.N;
        if (PD == 1 && !inside) continue;  // (a group that starts past the lattice: nobody reads its area)
        // PD > 1: the gather rounds carry no tests at all, so every group's area is staged -- one that does not exist from the
        // zeroed padding behind the last block ({row 0, 0.0f}: one line for the whole wave, a product with zero)
        const char* src = inside ? sb + (size_t)(uint32_t)(row0 + g * w8) * (8u * SL)
                                 : reinterpret_cast<const char*>(a.slots + (size_t)nb * (size_t)a.N * SL);
        const unsigned dst = __builtin_amdgcn_readfirstlane(stage_lds + (unsigned)(((ph & 1) * GM + g) * GROUP_BYTES));
#pragma unroll
        for (int q = 0; q < PIECES; ++q) {
          if (q * 1024 + lane * 16 < GROUP_BYTES) {
            const char* lsrc = src + lane * 16;  // (the instruction's offset advances the global AND the LDS address)
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(lsrc), "s"(dst), "n"(q * 1024)
                         : "memory");
          }
        }
      }
    };
    unsigned long long t_fetch = 0, t_bar = 0, t_all = 0, ts = 0;
    if constexpr (STAMP) t_all = ts = __builtin_amdgcn_s_memtime();
    auto lap = [&](unsigned long long& into) {
      if constexpr (STAMP) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        into += now - ts;
        ts = now;
      }
    };
    if (nphase > 0) fetch_slots(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lap(t_fetch);
    __syncthreads();
    lap(t_bar);
    for (int ph = 0; ph < nphase; ++ph) {
      if (ph + 1 < nphase) fetch_slots(ph + 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lap(t_fetch);
      __syncthreads();
      lap(t_bar);
      if ((ph + 1) % per_slab == 0) {
        const int sc0 = phase(ph).sc0;
        blk_fold<CW + 1>(f4(0.f), red, a.part, ld, sc0, min(a.c1, sc0 + 32), wave, lane);
      }
    }
    if constexpr (STAMP) {
      if (lane == 0) {
        unsigned long long* w = reinterpret_cast<unsigned long long*>(ii.R) + ((size_t)blockIdx.x * (CW + 1) + wave) * 4;
        w[0] += __builtin_amdgcn_s_memtime() - t_all;
        w[1] += t_fetch;
        w[2] += t_bar;
      }
    }
    return;
  }

  // ---- the other waves: gather (L2 hits and LDS reads) ---------------------------------------------------------------
  const uint32_t lr16 = (uint32_t)lr * 16u;
  float4 acc[GM];
  float4 dot[1] = {f4(0.f)};
  unsigned long long t_gather = 0, t_epi = 0, t_bar = 0, t_all = 0, ts = 0;
  if constexpr (STAMP) t_all = ts = __builtin_amdgcn_s_memtime();
  auto lap = [&](unsigned long long& into) {
    if constexpr (STAMP) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      into += now - ts;
      ts = now;
    }
  };
  __syncthreads();
  lap(t_bar);
  for (int ph = 0; ph < nphase; ++ph) {
    const BlkPhase p = phase(ph);
    const bool cok = p.sc0 + lr * 4 < a.c1;
    const float* xbase = slab_base(p.sc0);
    if (p.b == 0) {
#pragma unroll
      for (int g = 0; g < GM; ++g) acc[g] = f4(0.f);
    }
    const int2(*st)[CW][8 * SL] = stage[ph & 1];
    // One group per round: all slots of its 8 rows in flight together.  No tests in here: an unused slot holds {first row of
    // the block, 0.0f} (k_blk_fill, stage_slots), i.e. a gather of a line everybody has and a product with zero -- the
    // round is bound by instruction issue (four waves share a SIMD), and a compare + exec-mask + branch per slot cost
    // more than the fifth of the gathers they saved.
    const int wg_row0 = rlo + p.slice * slice_rows + ((wgx * CW) << 3);  // first row of this workgroup's group 0
    if constexpr (PD == 1) {
#pragma unroll
      for (int g = 0; g < GM; ++g) {
        if (g >= ng || wg_row0 + g * W8 >= a.N) continue;  // (same test as the list wave's: that area was not staged)
        int2 e[SL];
        float4 v[SL];
#pragma unroll
        for (int u = 0; u < SL; ++u) e[u] = st[g][wave][sub * SL + u];
#pragma unroll
        for (int u = 0; u < SL; ++u) v[u] = ld4_at(xbase, (uint32_t)e[u].x * 128u + lr16);
#pragma unroll
        for (int u = 0; u < SL; ++u) acc[g] = fma4(__int_as_float(e[u].y), v[u], acc[g]);
      }
    } else {
      // PD rounds in flight, straight-line: round g + PD - 1 is issued (slots from LDS, four row gathers per lane) before
      // round g is consumed, so the compiler's counted waits leave SL * (PD - 1) loads in flight in the steady state.
      // No test per group: the list wave staged every group's area (see there).
      typedef float f32x4 __attribute__((ext_vector_type(4)));  // (a native vector: one 128-bit operand of the asm below)
      f32x4 v[PD][SL];
      float w[PD][SL];
      auto issue = [&](int g, f32x4 (&vv)[SL], float (&ww)[SL]) {
        int2 e[SL];
#pragma unroll
        for (int u = 0; u < SL; ++u) e[u] = st[g][wave][sub * SL + u];
#pragma unroll
        for (int u = 0; u < SL; ++u)
          vv[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(xbase) + ((uint32_t)e[u].x * 128u + lr16));
#pragma unroll
        for (int u = 0; u < SL; ++u) ww[u] = __int_as_float(e[u].y);
      };
#pragma unroll
      for (int j = 0; j < PD - 1 && j < GM; ++j) issue(j, v[j % PD], w[j % PD]);
#pragma unroll
      for (int g = 0; g < GM; ++g) {
        if (g + PD - 1 < GM) issue(g + PD - 1, v[(g + PD - 1) % PD], w[(g + PD - 1) % PD]);
        // round g's rows pass through an empty asm with a memory clobber: its sums can start only here (behind the
        // issue of round g + PD - 1), and a later round's loads cannot be moved up past it
#pragma unroll
        for (int u = 0; u < SL; ++u) asm volatile("" : "+v"(v[g % PD][u])::"memory");
#pragma unroll
        for (int u = 0; u < SL; ++u) {
          const f32x4 t = v[g % PD][u];
          acc[g] = fma4(w[g % PD][u], make_float4(t.x, t.y, t.z, t.w), acc[g]);
        }
        // ... and end here (volatile asm statements keep their order): otherwise the compiler may postpone the sums of
        // all rounds and park the gathered rows in scratch memory meanwhile
        asm volatile("" : "+v"(acc[g].x), "+v"(acc[g].y), "+v"(acc[g].z), "+v"(acc[g].w));
      }
    }
    lap(t_gather);
    if (p.b == nb - 1) {  // the rows of this slice are complete: the rest of long lists, diagonal term, output, p.Ap
      const int w8 = opaque(W8);
      const int row_first = rlo + p.slice * slice_rows + ((wgx * CW + wave) << 3) + sub;
      // Own rows / gates / rest descriptors of EC groups at a time (these loads miss).  Loads and stores of a wave retire
      // in issue order, so the loads of batch k + 1 are issued BEFORE the stores of batch k: otherwise every batch would
      // also wait for the previous batch's stores to be acknowledged.
      constexpr int EC = WPE <= 2 ? 4 : INIT ? 1 : 2, NBATCH = (GM + EC - 1) / EC;  // (INIT: the y rows take the second group's registers; two groups: 72 spills)
      float4 xs[2][EC];
      float4 ys[2][INIT ? EC : 1];
      int2 rr[2][EC];
      float bv[2][EC];
      float4 psi4 = f4(0.f);
      float* zbase = nullptr;
      const uint32_t ldb = (uint32_t)ld * 4u, cb = (uint32_t)(p.sc0 + lr * 4) * 4u;
      // (INIT: the 14 dwords of `ii` cost the kernel 22 scalar-register spills into vector lanes; reading them from the
      // kernel-argument segment here, per closing sub-phase, was tried -- 12 spills, the same 700 us at config 3)
      if constexpr (INIT) {
        if (cok) psi4 = ld4(ii.psi + p.sc0 + lr * 4);
        zbase = ii.Z + (size_t)(p.sc0 >> 5) * (size_t)a.N * 32;
      }
      auto fetch = [&](int k, float4 (&x)[EC], float4 (&y)[INIT ? EC : 1], int2 (&r)[EC], float (&bb)[EC]) {
#pragma unroll
        for (int i = 0; i < EC; ++i) {
          const int g = k * EC + i, row = row_first + g * w8;
          x[i] = f4(0.f);
          r[i] = make_int2(0, 0);
          bb[i] = 0.f;
          if constexpr (INIT) y[i] = f4(0.f);
          if (g < GM && g < ng && row < rhi && cok) {
            x[i] = ld4_at(xbase, (uint32_t)row * 128u + lr16);
            r[i] = ld2_at(a.rest, (uint32_t)row * 8u);  // edges beyond SL per block: {first, count}, ascending columns
            bb[i] = ld1_at(a.B, (uint32_t)row * 4u);
            if constexpr (INIT) {  // (byte offsets of a row-major array fit 32 bits: launch_apply_blocked checks N * ld)
              if (ii.Y != nullptr) y[i] = ld4_stream(reinterpret_cast<const float*>(reinterpret_cast<const char*>(ii.Y) + ((uint32_t)row * ldb + cb)));
            }
          }
        }
      };
      fetch(0, xs[0], ys[0], rr[0], bv[0]);
#pragma unroll
      for (int k = 0; k < NBATCH; ++k) {
        if (k + 1 < NBATCH) fetch(k + 1, xs[(k + 1) & 1], ys[(k + 1) & 1], rr[(k + 1) & 1], bv[(k + 1) & 1]);
#pragma unroll
        for (int i = 0; i < EC; ++i) {
          const int g = k * EC + i, row = row_first + g * w8;
          if (g >= GM) continue;
          if (g >= ng || row >= rhi || !cok) continue;
          for (int e = 0; e < rr[k & 1][i].y; ++e) {
            const int2 en = ld2_at(a.over, (uint32_t)(rr[k & 1][i].x + e) * 8u);
            acc[g] = fma4(__int_as_float(en.y), ld4_at(xbase, (uint32_t)en.x * 128u + lr16), acc[g]);
          }
          const float cs = fmaf(a.cs_B, bv[k & 1][i], a.cs_const);
          float4 o;
          o.x = cs * xs[k & 1][i].x - a.cW * acc[g].x;
          o.y = cs * xs[k & 1][i].y - a.cW * acc[g].y;
          o.z = cs * xs[k & 1][i].z - a.cW * acc[g].z;
          o.w = cs * xs[k & 1][i].w - a.cW * acc[g].w;
          if constexpr (INIT) {
            const float4 x = xs[k & 1][i], y = ii.Y != nullptr ? ys[k & 1][i] : x;
            const float qb = ii.rbB * bv[k & 1][i];
            const float invMd = 1.f / (fmaf(ii.md_B, bv[k & 1][i], ii.md_const) + 1e-12f);  // (no preconditioner: md_B = 0, md_const = 1)
            float4 r, z;  // (k_init_finish's expressions)
            r.x = (ii.rbU * x.x + ii.rbY * y.x + qb * psi4.x) - o.x;
            r.y = (ii.rbU * x.y + ii.rbY * y.y + qb * psi4.y) - o.y;
            r.z = (ii.rbU * x.z + ii.rbY * y.z + qb * psi4.z) - o.z;
            r.w = (ii.rbU * x.w + ii.rbY * y.w + qb * psi4.w) - o.w;
            z = make_float4(r.x * invMd, r.y * invMd, r.z * invMd, r.w * invMd);
            st4_stream(reinterpret_cast<float*>(reinterpret_cast<char*>(ii.R) + ((uint32_t)row * ldb + cb)), r);
            if (ii.Xcopy != nullptr) st4_stream(reinterpret_cast<float*>(reinterpret_cast<char*>(ii.Xcopy) + ((uint32_t)row * ldb + cb)), x);
            st4(reinterpret_cast<float*>(reinterpret_cast<char*>(zbase) + ((uint32_t)row * 128u + lr16)), z);
            dot[0] = mulacc4(r, z, dot[0]);
          } else {
            st4_stream(a.OUT + ((size_t)(uint32_t)row * (uint32_t)ld + (uint32_t)(p.sc0 + lr * 4)), o);
            dot[0] = mulacc4(xs[k & 1][i], o, dot[0]);
          }
        }
      }
    }
    lap(t_epi);
    __syncthreads();
    lap(t_bar);
    if ((ph + 1) % per_slab == 0) {  // the slab's column sums
      blk_fold<CW + 1>(dot[0], red, a.part, ld, p.sc0, min(a.c1, p.sc0 + 32), wave, lane);
      dot[0] = f4(0.f);
    }
  }
  if constexpr (STAMP) {
    if (lane == 0) {
      unsigned long long* w = reinterpret_cast<unsigned long long*>(ii.R) + ((size_t)blockIdx.x * (CW + 1) + wave) * 4;
      w[0] += __builtin_amdgcn_s_memtime() - t_all;
      w[1] += t_gather;
      w[2] += t_bar;
      w[3] += t_epi;
    }
  }
}

// chain prior for k_apply_blocked (ChainFixArgs): one wave per (64 columns, chunk of path rows), one thread per column
__global__ __launch_bounds__(64) void k_chain_fix(const ChainFixArgs a) {
  if (a.gate != nullptr && *a.gate <= a.gate_tol) return;
  const int col = a.c0 + (int)blockIdx.x * 64 + (int)threadIdx.x;
  const int chunk = (int)blockIdx.y;
  const int per = (a.prows + a.chunks - 1) / a.chunks;
  const int s0 = chunk * per, s1 = min(a.prows, s0 + per);
  float dotc = 0.f;
  if (col < a.c1) {
    const float* xs = a.X + (size_t)(col >> 5) * (size_t)a.N * 32 + (col & 31);  // this column of the slab-major operand
    for (int s = s0; s < s1; ++s) {
      const int i = a.prow[s], d = a.pdeg[s];
      float acc = 0.f;
      for (int e = 0; e < d; ++e) acc = fmaf(a.pw[(size_t)s * a.pwidth + e], xs[(size_t)a.pcol[(size_t)s * a.pwidth + e] * 32], acc);
      const float o = -a.cP * acc;
      if (a.initR != nullptr) {  // behind the fused INIT pass: r and z of this row lack the chain term, and so does r . z
        const float invMd = 1.f / (fmaf(a.md_B, a.B[i], a.md_const) + 1e-12f);
        float* zp = a.initZ + (size_t)(col >> 5) * (size_t)a.N * 32 + (size_t)i * 32 + (col & 31);
        const float r0 = a.initR[(size_t)i * a.ld + col], z0 = *zp;
        const float r1 = r0 - o, z1 = r1 * invMd;
        a.initR[(size_t)i * a.ld + col] = r1;
        *zp = z1;
        dotc += r1 * z1 - r0 * z0;
        continue;
      }
      a.OUT[(size_t)i * a.ld + col] += o;
      dotc = fmaf(xs[(size_t)i * 32], o, dotc);
    }
    a.part[(size_t)(a.part_row0 + chunk) * a.ld + col] = dotc;
  }
}

// ---- build of the block-major graph copy ----------------------------------------------------
__device__ __forceinline__ int blk_of(int col, int rpb, int nb) { return min(nb - 1, col / rpb); }

// Placement of a row's edges in the block-major copy: an edge goes into the slot row of its own block while that has room
// (slots 0 .. c - 1 for the block's c <= SL own edges, ascending columns); an edge that finds its block full moves to
// the first later block (cyclically) whose slot row has room behind that block's own edges -- it is then gathered while
// another block is the resident one, i.e. as a miss among hits, but inside the regular rounds; only what fits nowhere
// (degree > SL * nb) goes to the `over` list of the epilogue.  Deterministic: everything follows the ELL row's order.
__device__ __forceinline__ int blk_unplaced(const int (&c)[OSC_MAX_SRC_BLOCKS]) {
  int excess = 0, room = 0;
#pragma unroll
  for (int q = 0; q < OSC_MAX_SRC_BLOCKS; ++q) {
    excess += max(0, c[q] - OSC_BLK_SLOTS);
    room += max(0, OSC_BLK_SLOTS - c[q]);  // (blocks q >= nb have c = SL: see the callers)
  }
  return max(0, excess - room);
}

// edges that fit no slot row, in total
__global__ void k_blk_count(const int32_t* col, const int32_t* deg, int32_t width, int32_t N, int32_t nb, int32_t rpb,
                            unsigned* over_count) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= N) return;
  int c[OSC_MAX_SRC_BLOCKS];
#pragma unroll
  for (int k = 0; k < OSC_MAX_SRC_BLOCKS; ++k) c[k] = k < nb ? 0 : OSC_BLK_SLOTS;
  const int d = deg[row];
  for (int e = 0; e < d; ++e) {
    const int b = blk_of(col[(size_t)row * width + e], rpb, nb);
#pragma unroll
    for (int k = 0; k < OSC_MAX_SRC_BLOCKS; ++k) c[k] += (k == b);
  }
  const int over = blk_unplaced(c);
  if (over) atomicAdd(over_count, (unsigned)over);
}

// *over_count must be zero on entry (it hands out the ranges of `over`); every slot is written
__global__ void k_blk_fill(const int32_t* col, const float* w, const int32_t* deg, int32_t width, int32_t N, int32_t nb,
                           int32_t rpb, int2* slots, int2* rest, int2* over, unsigned* over_count) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= N) return;
  int c[OSC_MAX_SRC_BLOCKS], k[OSC_MAX_SRC_BLOCKS], tail[OSC_MAX_SRC_BLOCKS];
#pragma unroll
  for (int q = 0; q < OSC_MAX_SRC_BLOCKS; ++q) c[q] = q < nb ? 0 : OSC_BLK_SLOTS, k[q] = 0;
  const int d = deg[row];
  for (int e = 0; e < d; ++e) {
    const int b = blk_of(col[(size_t)row * width + e], rpb, nb);
#pragma unroll
    for (int q = 0; q < OSC_MAX_SRC_BLOCKS; ++q) c[q] += (q == b);
  }
#pragma unroll
  for (int q = 0; q < OSC_MAX_SRC_BLOCKS; ++q) tail[q] = min(c[q], OSC_BLK_SLOTS);  // first slot behind the own edges
  const int total = blk_unplaced(c);
  unsigned at = total ? atomicAdd(over_count, (unsigned)total) : 0u;
  rest[row] = make_int2((int)at, total);
  for (int e = 0; e < d; ++e) {
    const int j = col[(size_t)row * width + e];
    const int b = blk_of(j, rpb, nb);
    int kb = 0;
#pragma unroll
    for (int q = 0; q < OSC_MAX_SRC_BLOCKS; ++q)
      if (q == b) kb = k[q]++;
    const int2 ent = make_int2(j, __float_as_int(w[(size_t)row * width + e]));
    if (kb < OSC_BLK_SLOTS) {
      slots[((size_t)b * N + row) * OSC_BLK_SLOTS + kb] = ent;
      continue;
    }
    int tb = -1, ts = 0;  // first block after b (cyclically) with room
    for (int step = 1; step < nb && tb < 0; ++step) {
      const int q = (b + step) % nb;
#pragma unroll
      for (int t = 0; t < OSC_MAX_SRC_BLOCKS; ++t)
        if (t == q && tail[t] < OSC_BLK_SLOTS) tb = q, ts = tail[t]++;
    }
    if (tb >= 0) slots[((size_t)tb * N + row) * OSC_BLK_SLOTS + ts] = ent;
    else over[at++] = ent;
  }
  // unused slots: a row every wave gathering from block q has in its caches anyway, weight zero
#pragma unroll
  for (int q = 0; q < OSC_MAX_SRC_BLOCKS; ++q)
    if (q < nb)
      for (int t = tail[q]; t < OSC_BLK_SLOTS; ++t)
        slots[((size_t)q * N + row) * OSC_BLK_SLOTS + t] = make_int2(min(N - 1, q * rpb), 0);
}

// ---- initial residual around the blocked matvec (InitFinishArgs) ---------------------------------------------------
// DIFF: dst = src - minus (the receipt's U - U*, formed on the way into the slabs: no row-major copy of the difference)
template <int LPR, int NCH, bool DIFF = false>
__global__ __launch_bounds__(256) void k_rows_to_slab(const float* src, float* dst, int64_t N, int32_t ld, int32_t c0,
                                                      int32_t c1, const float* minus = nullptr) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, lr = lane % LPR;
  for (int64_t rb = ((int64_t)blockIdx.x * 4 + wave) * RPW; rb < N; rb += (int64_t)gridDim.x * 4 * RPW) {
    const int row = (int)rb + sub;
    if (row >= N) continue;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int col = c0 + (ch * LPR + lr) * 4;
      if (col < c1) {
        float4 v = ld4_stream(src + (size_t)row * ld + col);
        if constexpr (DIFF) {
          const float4 w = ld4_stream(minus + (size_t)row * ld + col);
          v = make_float4(v.x - w.x, v.y - w.y, v.z - w.z, v.w - w.w);
        }
        st4(dst + blk_off(N, row, col), v);
      }
    }
  }
}

template <int LPR, int NCH>
__global__ __launch_bounds__(256) void k_init_finish(const InitFinishArgs a) {
  constexpr int RPW = 64 / LPR;
  constexpr int CPW = NCH * LPR * 4;
  __shared__ __attribute__((aligned(16))) float red[4 * CPW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, lr = lane % LPR;
  const int32_t ld = a.ld;
  int coff[NCH];
  bool cok[NCH];
  float4 psi4[NCH], rz[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    coff[ch] = a.c0 + (ch * LPR + lr) * 4;
    cok[ch] = coff[ch] < a.c1;
    psi4[ch] = cok[ch] ? ld4(a.psi + coff[ch]) : f4(0.f);
    rz[ch] = f4(0.f);
  }
  for (int64_t rb = ((int64_t)blockIdx.x * 4 + wave) * RPW; rb < a.N; rb += (int64_t)gridDim.x * 4 * RPW) {
    const int row = (int)rb + sub;
    if (row >= a.N) continue;
    const float Bi = a.B[row];
    float invMd = 1.f;
    if (a.op.precond) invMd = 1.f / (fmaf(a.op.md_B, Bi, a.op.md_const) + 1e-12f);
    const float qb = a.op.rbB * Bi;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      if (!cok[ch]) continue;
      const size_t off = (size_t)row * ld + coff[ch];
      const float4 o = ld4_stream(a.AP + off);
      const float4 xs = ld4_stream(a.X0 + off);
      const float4 u = (a.U == a.X0) ? xs : (a.op.rbU != 0.f ? ld4_stream(a.U + off) : f4(0.f));
      const float4 y = (a.Y == a.X0) ? xs : ld4_stream(a.Y + off);
      float4 r, z;
      r.x = (a.op.rbU * u.x + a.op.rbY * y.x + qb * psi4[ch].x) - o.x;
      r.y = (a.op.rbU * u.y + a.op.rbY * y.y + qb * psi4[ch].y) - o.y;
      r.z = (a.op.rbU * u.z + a.op.rbY * y.z + qb * psi4[ch].z) - o.z;
      r.w = (a.op.rbU * u.w + a.op.rbY * y.w + qb * psi4[ch].w) - o.w;
      z = make_float4(r.x * invMd, r.y * invMd, r.z * invMd, r.w * invMd);
      if (a.X != a.X0) st4_stream(a.X + off, xs);
      st4_stream(a.R + off, r);
      st4(a.P + blk_off(a.pblk, row, coff[ch]), z);
      rz[ch] = mulacc4(r, z, rz[ch]);
    }
  }
  block_fold<LPR, NCH>(rz, red, a.part, ld, a.c0, a.c1);
}

// ---------------------------------------------------------------------------------------------
// WITHX = false: the x update of this iteration is left to the next iteration's k_update_p (or to k_update_x behind the
// last one): this kernel then moves r and Ap only (run_cg).
// STORE_R = false (with WITHX): the form for an iteration expected to be the solve's last -- x is finished here and the new
// r only feeds the two column sums; should the solve go on after all, the host has the r update redone with a store.
template <int LPR, int NCH, bool WITHX, bool STORE_R = true>
__global__ __launch_bounds__(256) void k_update_xr(const UpdateArgs a) {
  constexpr int RPW = 64 / LPR;
  constexpr int CPW = NCH * LPR * 4;
  __shared__ __attribute__((aligned(16))) float red[4 * CPW];
  if (a.gate != nullptr && *a.gate <= a.gate_tol) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, lr = lane % LPR;
  const int32_t ld = a.ld;
  const bool tmp = a.temporal != 0;
  int coff[NCH];
  bool cok[NCH];
  float4 al[NCH], rr[NCH], rz[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    coff[ch] = a.c0 + (ch * LPR + lr) * 4;
    cok[ch] = coff[ch] < a.c1;
    al[ch] = cok[ch] ? ld4(a.alpha + coff[ch]) : f4(0.f);
    rr[ch] = f4(0.f);
    rz[ch] = f4(0.f);
  }
  const int64_t rend = a.N;  // streaming kernels: plain grid-stride over the row range
  for (int64_t rb = a.row0 + ((int64_t)blockIdx.x * 4 + wave) * RPW; rb < rend; rb += (int64_t)gridDim.x * 4 * RPW) {
    const int row = (int)rb + sub;
    if (row >= rend) continue;
    float invMd = 1.f;
    if (a.op.precond) invMd = 1.f / (fmaf(a.op.md_B, a.B[row], a.op.md_const) + 1e-12f);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      if (!cok[ch]) continue;
      const size_t off = (size_t)row * ld + coff[ch];
      float4 r = ld4_sel(a.R + off, tmp);
      const float4 ap = ld4_sel(a.AP + off, tmp);
      if constexpr (WITHX) {
        float4 x = ld4_sel(a.X + off, tmp);
        const float4 p = ld4_sel(a.P + p_off(a, row, coff[ch]), tmp);
        x.x = fmaf(p.x, al[ch].x, x.x); x.y = fmaf(p.y, al[ch].y, x.y);
        x.z = fmaf(p.z, al[ch].z, x.z); x.w = fmaf(p.w, al[ch].w, x.w);
        st4_sel(a.X + off, x, tmp);
      }
      r.x = fmaf(-ap.x, al[ch].x, r.x); r.y = fmaf(-ap.y, al[ch].y, r.y);
      r.z = fmaf(-ap.z, al[ch].z, r.z); r.w = fmaf(-ap.w, al[ch].w, r.w);
      if constexpr (STORE_R) st4_sel(a.R + off, r, tmp);
      rr[ch] = mulacc4(r, r, rr[ch]);
      const float4 z = make_float4(r.x * invMd, r.y * invMd, r.z * invMd, r.w * invMd);
      rz[ch] = mulacc4(r, z, rz[ch]);
    }
  }
  block_fold<LPR, NCH>(rr, red, a.part_rr, ld, a.c0, a.c1);
  block_fold<LPR, NCH>(rz, red, a.part_rz, ld, a.c0, a.c1);
}

// WITHX: also the PREVIOUS iteration's x update, x += alpha p with the p this kernel is about to replace (alpha is still
// that iteration's: the next reduce_alpha comes behind the matvec).  WITHX and UPD_P = false: only that (behind the last
// iteration of a solve).
template <int LPR, int NCH, bool WITHX, bool UPD_P = true>
__global__ __launch_bounds__(256) void k_update_p(const UpdateArgs a) {
  constexpr int RPW = 64 / LPR;
  if (a.gate != nullptr && *a.gate <= a.gate_tol) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, lr = lane % LPR;
  const int32_t ld = a.ld;
  const bool tmp = a.temporal != 0;
  int coff[NCH];
  bool cok[NCH];
  float4 be[NCH], al[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    coff[ch] = a.c0 + (ch * LPR + lr) * 4;
    cok[ch] = coff[ch] < a.c1;
    be[ch] = (cok[ch] && UPD_P) ? ld4(a.beta + coff[ch]) : f4(0.f);
    al[ch] = (cok[ch] && WITHX) ? ld4(a.alpha + coff[ch]) : f4(0.f);
  }
  const int64_t rend = a.N;  // streaming kernels: plain grid-stride over the row range
  for (int64_t rb = a.row0 + ((int64_t)blockIdx.x * 4 + wave) * RPW; rb < rend; rb += (int64_t)gridDim.x * 4 * RPW) {
    const int row = (int)rb + sub;
    if (row >= rend) continue;
    float invMd = 1.f;
    if (UPD_P && a.op.precond) invMd = 1.f / (fmaf(a.op.md_B, a.B[row], a.op.md_const) + 1e-12f);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      if (!cok[ch]) continue;
      const size_t off = (size_t)row * ld + coff[ch];
      const size_t poff = p_off(a, row, coff[ch]);
      float4 p = ld4_sel(a.P + poff, tmp);
      if constexpr (WITHX) {
        float4 x = ld4_sel(a.X + off, tmp);
        x.x = fmaf(p.x, al[ch].x, x.x); x.y = fmaf(p.y, al[ch].y, x.y);
        x.z = fmaf(p.z, al[ch].z, x.z); x.w = fmaf(p.w, al[ch].w, x.w);
        st4_sel(a.X + off, x, tmp);
      }
      if constexpr (UPD_P) {
        const float4 r = ld4_sel(a.R + off, tmp);
        p.x = fmaf(p.x, be[ch].x, r.x * invMd); p.y = fmaf(p.y, be[ch].y, r.y * invMd);
        p.z = fmaf(p.z, be[ch].z, r.z * invMd); p.w = fmaf(p.w, be[ch].w, r.w * invMd);
        st4(a.P + poff, p);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// second stage of the column reductions: 64 columns per block, 16 row-groups, fp64 accumulation
template <int NIN>
__device__ __forceinline__ bool reduce_cols(const float* p0, const float* p1, int nb, int32_t ld, int32_t c0,
                                            int32_t c1, double (&tot)[NIN], int& col) {
  __shared__ double sh[NIN][16][64];
  const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
  col = c0 + blockIdx.x * 64 + cl;
  const bool ok = col < c1;
  double s[NIN];
#pragma unroll
  for (int i = 0; i < NIN; ++i) s[i] = 0.0;
  if (ok) {
    // 8 independent loads in flight per input: the loop is latency-bound, not bandwidth-bound
    int b = g;
    for (; b + 7 * 16 < nb; b += 8 * 16) {
      float v0[8], v1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v0[u] = p0[(size_t)(b + 16 * u) * ld + col];
        if (NIN > 1) v1[u] = p1[(size_t)(b + 16 * u) * ld + col];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s[0] += (double)v0[u];
        if (NIN > 1) s[NIN - 1] += (double)v1[u];
      }
    }
    for (; b < nb; b += 16) {
      s[0] += (double)p0[(size_t)b * ld + col];
      if (NIN > 1) s[NIN - 1] += (double)p1[(size_t)b * ld + col];
    }
  }
#pragma unroll
  for (int i = 0; i < NIN; ++i) sh[i][g][cl] = s[i];
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += sh[i][k][cl];
      tot[i] = t;
    }
  }
  return ok && g == 0;
}

__global__ __launch_bounds__(1024) void k_reduce_init(const float* part, int nb, int32_t ld, int32_t c0, int32_t c1,
                                                      double* rz) {
  double t[1];
  int col;
  if (reduce_cols<1>(part, nullptr, nb, ld, c0, c1, t, col)) rz[col] = t[0];
}

__global__ __launch_bounds__(1024) void k_reduce_alpha(const float* part, int nb, int32_t ld, int32_t c0, int32_t c1,
                                                       const double* rz, float* alpha, Gate gt) {
  if (gt.p != nullptr && *gt.p <= gt.tol) return;
  double t[1];
  int col;
  if (reduce_cols<1>(part, nullptr, nb, ld, c0, c1, t, col))
    alpha[col] = (float)(rz[col] / (t[0] + 1e-18));  // solver.py:25-26
}

// host_slot != nullptr: the last workgroup to finish publishes the iteration's residual straight into host-mapped
// memory (the host polls that word instead of paying a 4-byte copy + event + event wait per iteration); done_ctr is
// this iteration's arrival counter (zeroed with the residual slots).
__global__ __launch_bounds__(1024) void k_reduce_beta(const float* part_rr, const float* part_rz, int nb, int32_t ld,
                                                      int32_t c0, int32_t c1, double* rz, float* beta,
                                                      uint32_t* res_bits, Gate gt, uint32_t* done_ctr,
                                                      float* host_slot) {
  if (gt.p != nullptr && *gt.p <= gt.tol) return;
  double t[2];
  int col;
  const bool w = reduce_cols<2>(part_rr, part_rz, nb, ld, c0, c1, t, col);
  float resc = 0.f;
  if (w) {
    resc = (float)sqrt(t[0]);                            // ||r_c||_2        (solver.py:29)
    beta[col] = (float)(t[1] / (rz[col] + 1e-18));       // solver.py:33-34
    rz[col] = t[1];
  }
  if ((threadIdx.x >> 6) == 0) {  // wave 0 holds the 64 column residuals of this block
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) resc = nanmax(resc, __shfl_xor(resc, o, 64));
    if ((threadIdx.x & 63) == 0) {
      atomicMax(res_bits, __float_as_uint(resc));  // non-negative floats order as uints
      if (host_slot != nullptr) {
        __threadfence();
        if (atomicAdd(done_ctr, 1u) == gridDim.x - 1) {      // every workgroup's maximum is in
          const uint32_t bits = atomicMax(res_bits, 0u);     // read it back through the same (coherent) path
          *reinterpret_cast<volatile uint32_t*>(host_slot) = bits;
          __threadfence_system();
        }
      }
    }
  }
}

__global__ __launch_bounds__(1024) void k_reduce_sum(const float* part, int nb, int32_t ld, int32_t c0, int32_t c1,
                                                     double* out_cols, Gate gt) {
  if (gt.p != nullptr && *gt.p <= gt.tol) return;
  double t[1];
  int col;
  if (reduce_cols<1>(part, nullptr, nb, ld, c0, c1, t, col)) out_cols[col] = t[0];
}

// finish steps of the row-sharded CG (sums already complete across ranks)
__global__ void k_finish_init(const double* sums, int32_t c0, int32_t c1, double* rz) {
  const int col = c0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (col < c1) rz[col] = sums[col];
}
__global__ void k_finish_alpha(const double* sums, int32_t c0, int32_t c1, const double* rz, float* alpha, Gate gt) {
  if (gt.p != nullptr && *gt.p <= gt.tol) return;
  const int col = c0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (col < c1) alpha[col] = (float)(rz[col] / (sums[col] + 1e-18));
}
__global__ __launch_bounds__(256) void k_finish_beta(const double* srr, const double* srz, int32_t c0, int32_t c1,
                                                     double* rz, float* beta, uint32_t* res_bits, Gate gt) {
  if (gt.p != nullptr && *gt.p <= gt.tol) return;
  const int col = c0 + blockIdx.x * blockDim.x + threadIdx.x;
  float resc = 0.f;
  if (col < c1) {
    resc = (float)sqrt(srr[col]);
    beta[col] = (float)(srz[col] / (rz[col] + 1e-18));
    rz[col] = srz[col];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) resc = nanmax(resc, __shfl_xor(resc, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(res_bits, __float_as_uint(resc));
}

__global__ __launch_bounds__(256) void k_axpby(float* out, const float* a, float ca, const float* b, float cb,
                                               int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 x = ld4(a + i * 4), y = ld4(b + i * 4);
    st4(out + i * 4, make_float4(ca * x.x + cb * y.x, ca * x.y + cb * y.y, ca * x.z + cb * y.z, ca * x.w + cb * y.w));
  }
}

// ---- dispatch -------------------------------------------------------------------------------
struct Shape {
  int lpr, nch;
};
// lanes per row x float4 chunks per lane.  Exact-fit shapes for 96 / 192 / 384 columns (the per-rank windows of D = 768
// on 8 / 4 / 2 GPUs: <8,3>, <16,3>, <32,3>) were measured and are no faster than the padded ones below (the apply is
// bound by the gathered bytes, not by idle lanes), so they are not instantiated.
inline Shape pick_shape(int ncols) {
  if (ncols <= 16) return {4, 1};  // single right-hand sides (diffusion gates: pitch 4) and very narrow lattices
  if (ncols <= 64) return {16, 1};
  if (ncols <= 128) return {32, 1};
  int nch = (ncols + 255) / 256;
  if (nch == 5) nch = 6;
  if (nch == 7) nch = 8;
  if (nch > 8) throw std::runtime_error("column window wider than 2048; split it");
  return {64, nch};
}

#define OSC_SHAPE_SWITCH(sh, CALL)                                   \
  do {                                                               \
    if ((sh).lpr == 4) { CALL(4, 1); }                               \
    else if ((sh).lpr == 16) { CALL(16, 1); }                        \
    else if ((sh).lpr == 32) { CALL(32, 1); }                        \
    else switch ((sh).nch) {                                         \
      case 1: CALL(64, 1); break;                                    \
      case 2: CALL(64, 2); break;                                    \
      case 3: CALL(64, 3); break;                                    \
      case 4: CALL(64, 4); break;                                    \
      case 6: CALL(64, 6); break;                                    \
      default: CALL(64, 8); break;                                   \
    }                                                                \
  } while (0)

}  // namespace

int spmm_grid(int64_t N, int32_t ncols) {
  const Shape sh = pick_shape(ncols);
  const int rpw = 64 / sh.lpr;
  const int64_t need = (N + 4 * rpw - 1) / (4 * rpw);
  return (int)std::max<int64_t>(1, std::min<int64_t>(need, 1024));
}

static void blocked_check(int32_t N, int32_t width, int32_t nb) {
  if (nb < 1 || nb > OSC_MAX_SRC_BLOCKS) throw std::runtime_error("blocked graph copy: too many source blocks");
  if ((int64_t)N * width >= ((int64_t)1 << 28) || (int64_t)N >= ((int64_t)1 << 24))  // 32-bit byte offsets in the apply
    throw std::runtime_error("blocked graph copy: lattice too large");
}
void launch_blocked_count(const int32_t* col, const int32_t* deg, int32_t width, int32_t N, int32_t nb,
                          unsigned* over_count, hipStream_t s) {
  blocked_check(N, width, nb);
  hipLaunchKernelGGL(k_blk_count, dim3((N + 255) / 256), dim3(256), 0, s, col, deg, width, N, nb, (N + nb - 1) / nb,
                     over_count);
  HIP_CHECK(hipGetLastError());
}
void launch_blocked_fill(const int32_t* col, const float* w, const int32_t* deg, int32_t width, int32_t N, int32_t nb,
                         int2* slots, int2* rest, int2* over, unsigned* over_count, hipStream_t s) {
  blocked_check(N, width, nb);
  hipLaunchKernelGGL(k_blk_fill, dim3((N + 255) / 256), dim3(256), 0, s, col, w, deg, width, N, nb, (N + nb - 1) / nb,
                     slots, rest, over, over_count);
  HIP_CHECK(hipGetLastError());
}
// The shapes of k_apply_blocked the library holds: {row groups per gathering wave, gathering waves per workgroup, gather
// rounds in flight per wave, waves per SIMD}.  0: rounds 2-4 (two 8-wave workgroups per CU, one round in flight, tests per
// group).  1-3 (round 5, "wide"): ONE 8-wave workgroup per CU at two waves per SIMD, four rounds in flight, test-free
// straight-line rounds -- so the group count is a template constant and there are six of them, 8 to 28 groups
// (blocked_shape_for picks the smallest that holds the lattice's groups).
struct BlkShape {
  int gm, cw, pd, wpe;
};
constexpr BlkShape kBlkShapes[] = {{kBlkGroups, kBlkGatherWaves, 1, 4}, {8, 7, 4, 2},  {12, 7, 4, 2}, {16, 7, 4, 2},
                                   {20, 7, 4, 2},                       {24, 7, 4, 2}, {28, 7, 4, 2}};
constexpr int kBlkShapeCount = (int)(sizeof(kBlkShapes) / sizeof(kBlkShapes[0]));
#define OSC_BLK_SHAPE_SWITCH(v, CALL) \
  switch (v) {                        \
    case 0: CALL(kBlkGroups, kBlkGatherWaves, 1, 4); break; \
    case 1: CALL(8, 7, 4, 2); break;  \
    case 2: CALL(12, 7, 4, 2); break; \
    case 3: CALL(16, 7, 4, 2); break; \
    case 4: CALL(20, 7, 4, 2); break; \
    case 5: CALL(24, 7, 4, 2); break; \
    case 6: CALL(28, 7, 4, 2); break; \
    default: throw std::runtime_error("blocked apply: unknown kernel shape"); \
  }
// the cycle-stamping instantiations (diagnostics): the round-4 shape and the widest one
#define OSC_BLK_STAMP_SWITCH(v, CALL) \
  switch (v) {                        \
    case 0: CALL(kBlkGroups, kBlkGatherWaves, 1, 4); break; \
    case 6: CALL(28, 7, 4, 2); break; \
    default: throw std::runtime_error("blocked apply: no stamping instantiation of this kernel shape (OSC_BLK_VARIANT=0 or 6)"); \
  }
static const BlkShape& blk_shape(int variant) {
  if (variant < 0 || variant >= kBlkShapeCount) throw std::runtime_error("blocked apply: unknown kernel shape");
  return kBlkShapes[variant];
}
int blocked_variants() { return kBlkShapeCount; }
int blocked_groups_max(int variant) { return blk_shape(variant).gm; }
int blocked_gather_waves(int variant) { return blk_shape(variant).cw; }
void launch_rows_to_slab(const float* src, float* dst, int64_t N, int32_t ld, int32_t c0, int32_t c1, int grid, hipStream_t s,
                         const float* sub) {
  for (int32_t s0 = c0; s0 < c1; s0 += 2048) {  // at most 2048 columns per launch, like the other elementwise kernels
    const int32_t s1 = std::min(c1, s0 + 2048);
    const Shape sh = pick_shape(s1 - s0);
    if (sub != nullptr) {
#define CALL(L, C) hipLaunchKernelGGL((k_rows_to_slab<L, C, true>), dim3(grid), dim3(256), 0, s, src, dst, N, ld, s0, s1, sub)
      OSC_SHAPE_SWITCH(sh, CALL);
#undef CALL
      continue;
    }
#define CALL(L, C) hipLaunchKernelGGL((k_rows_to_slab<L, C, false>), dim3(grid), dim3(256), 0, s, src, dst, N, ld, s0, s1, nullptr)
    OSC_SHAPE_SWITCH(sh, CALL);
#undef CALL
  }
  HIP_CHECK(hipGetLastError());
}
void launch_init_finish(const InitFinishArgs& a, int grid, hipStream_t s) {
  if (a.c1 - a.c0 > 2048) throw std::runtime_error("init_finish: column window wider than 2048");
  const Shape sh = pick_shape(a.c1 - a.c0);
#define CALL(L, C) hipLaunchKernelGGL((k_init_finish<L, C>), dim3(grid), dim3(256), 0, s, a)
  OSC_SHAPE_SWITCH(sh, CALL);
#undef CALL
  HIP_CHECK(hipGetLastError());
}
int chain_fix_chunks(int32_t prows) { return std::max(1, std::min(OSC_CHAIN_FIX_MAX_CHUNKS, (prows + 7) / 8)); }
void launch_chain_fix(const ChainFixArgs& a, hipStream_t s) {
  if (a.prows < 1 || a.prows > OSC_CHAIN_FIX_MAX_ROWS || a.chunks < 1 || a.chunks > OSC_CHAIN_FIX_MAX_CHUNKS || a.c1 <= a.c0)
    throw std::runtime_error("chain fix-up: unsupported arguments");
  hipLaunchKernelGGL(k_chain_fix, dim3((a.c1 - a.c0 + 63) / 64, a.chunks), dim3(64), 0, s, a);
  HIP_CHECK(hipGetLastError());
}
int blocked_resident_per_cu(int variant) {
  int n = 0;
#define CALL(G, W, P, E) HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_apply_blocked<G, W, false, P, E>, (W + 1) * 64, 0))
  OSC_BLK_SHAPE_SWITCH(variant, CALL);
#undef CALL
  return n;
}

void launch_apply_blocked(const BlkArgs& a, int grid, hipStream_t s, const BlkInit* init, int variant, unsigned long long* stamps) {
  const BlkShape& sh = blk_shape(variant);
  if (grid < 8 || (grid & 7) != 0 || a.xs < 1 || a.xs > grid / 8 || a.xs_groups < 1 || 8 % a.xs_groups != 0 || a.groups < 1 ||
      a.groups > sh.gm || a.slices < 1 || a.nb < 1 || a.nb > OSC_MAX_SRC_BLOCKS || !a.slots || !a.rest ||
      (int64_t)a.N * a.ld * 4 >= ((int64_t)1 << 32) || (a.c0 & 31) != 0)
    throw std::runtime_error("blocked apply: unsupported arguments");
  if (init != nullptr) {
    if (!init->R || !init->Z || !init->psi || init->Z == a.X) throw std::runtime_error("blocked apply: bad INIT arguments");
#define CALL(G, W, P, E) hipLaunchKernelGGL((k_apply_blocked<G, W, true, P, E>), dim3(grid), dim3((W + 1) * 64), 0, s, a, *init)
    OSC_BLK_SHAPE_SWITCH(variant, CALL);
#undef CALL
  } else if (stamps != nullptr) {  // diagnostic: [grid][waves][4] cycle counters, added to launch after launch
    BlkInit st{};
    st.R = reinterpret_cast<float*>(stamps);
#define CALL(G, W, P, E) hipLaunchKernelGGL((k_apply_blocked<G, W, false, P, E, true>), dim3(grid), dim3((W + 1) * 64), 0, s, a, st)
    OSC_BLK_STAMP_SWITCH(variant, CALL);
#undef CALL
  } else {
#define CALL(G, W, P, E) hipLaunchKernelGGL((k_apply_blocked<G, W, false, P, E>), dim3(grid), dim3((W + 1) * 64), 0, s, a, BlkInit{})
    OSC_BLK_SHAPE_SWITCH(variant, CALL);
#undef CALL
  }
  HIP_CHECK(hipGetLastError());
}

void launch_spmm(int mode, const SpmmArgs& a, int grid, hipStream_t s) {
  if (a.xs != 0) {  // XCD-affine 32-column slabs: 8 lanes per row
    if (grid < 8 || (grid & 7) != 0 || a.xs > grid / 8 || a.xs_groups < 1 || 8 % a.xs_groups != 0)
      throw std::runtime_error("xs mode needs a grid that is a multiple of 8 and 1, 2, 4 or 8 slab groups");
    if (mode == SPMM_AP) hipLaunchKernelGGL((k_spmm<8, 1, SPMM_AP>), dim3(grid), dim3(256), 0, s, a);
    else if (mode == SPMM_INIT) hipLaunchKernelGGL((k_spmm<8, 1, SPMM_INIT>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_spmm<8, 1, SPMM_DOT>), dim3(grid), dim3(256), 0, s, a);
    HIP_CHECK(hipGetLastError());
    return;
  }
  const Shape sh = pick_shape(a.c1 - a.c0);
  if (a.deep != 0 && sh.nch == 1 && (sh.lpr == 32 || sh.lpr == 64) && mode != SPMM_DOT) {  // local row order: more hits in flight
    if (sh.lpr == 32) {
      if (mode == SPMM_AP) hipLaunchKernelGGL((k_spmm<32, 1, SPMM_AP, OSC_SPMM_UDEEP>), dim3(grid), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((k_spmm<32, 1, SPMM_INIT, OSC_SPMM_UDEEP>), dim3(grid), dim3(256), 0, s, a);
    } else {
      if (mode == SPMM_AP) hipLaunchKernelGGL((k_spmm<64, 1, SPMM_AP, OSC_SPMM_UDEEP>), dim3(grid), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((k_spmm<64, 1, SPMM_INIT, OSC_SPMM_UDEEP>), dim3(grid), dim3(256), 0, s, a);
    }
    HIP_CHECK(hipGetLastError());
    return;
  }
#define CALL(L, C)                                                                                   \
  do {                                                                                               \
    if (mode == SPMM_AP) hipLaunchKernelGGL((k_spmm<L, C, SPMM_AP>), dim3(grid), dim3(256), 0, s, a); \
    else if (mode == SPMM_INIT) hipLaunchKernelGGL((k_spmm<L, C, SPMM_INIT>), dim3(grid), dim3(256), 0, s, a); \
    else hipLaunchKernelGGL((k_spmm<L, C, SPMM_DOT>), dim3(grid), dim3(256), 0, s, a);               \
  } while (0)
  OSC_SHAPE_SWITCH(sh, CALL);
#undef CALL
  HIP_CHECK(hipGetLastError());
}

void launch_update_xr(const UpdateArgs& a, int grid, hipStream_t s) {
  const Shape sh = pick_shape(a.c1 - a.c0);
#define CALL(L, C) hipLaunchKernelGGL((k_update_xr<L, C, true>), dim3(grid), dim3(256), 0, s, a)
#define CALL_NOX(L, C) hipLaunchKernelGGL((k_update_xr<L, C, false>), dim3(grid), dim3(256), 0, s, a)
#define CALL_LAST(L, C) hipLaunchKernelGGL((k_update_xr<L, C, true, false>), dim3(grid), dim3(256), 0, s, a)
  if (a.xmode & OSC_XMODE_XR_LAST) {
    OSC_SHAPE_SWITCH(sh, CALL_LAST);
  } else if (a.xmode & OSC_XMODE_XR_SKIPS_X) {
    OSC_SHAPE_SWITCH(sh, CALL_NOX);
  } else {
    OSC_SHAPE_SWITCH(sh, CALL);
  }
#undef CALL
#undef CALL_NOX
#undef CALL_LAST
  HIP_CHECK(hipGetLastError());
}

void launch_update_p(const UpdateArgs& a, int grid, hipStream_t s) {
  const Shape sh = pick_shape(a.c1 - a.c0);
#define CALL(L, C) hipLaunchKernelGGL((k_update_p<L, C, false>), dim3(grid), dim3(256), 0, s, a)
#define CALL_X(L, C) hipLaunchKernelGGL((k_update_p<L, C, true>), dim3(grid), dim3(256), 0, s, a)
  if (a.xmode & OSC_XMODE_P_APPLIES_X) {
    OSC_SHAPE_SWITCH(sh, CALL_X);
  } else {
    OSC_SHAPE_SWITCH(sh, CALL);
  }
#undef CALL
#undef CALL_X
  HIP_CHECK(hipGetLastError());
}

void launch_update_x(const UpdateArgs& a, int grid, hipStream_t s) {  // x += alpha p
  const Shape sh = pick_shape(a.c1 - a.c0);
#define CALL(L, C) hipLaunchKernelGGL((k_update_p<L, C, true, false>), dim3(grid), dim3(256), 0, s, a)
  OSC_SHAPE_SWITCH(sh, CALL);
#undef CALL
  HIP_CHECK(hipGetLastError());
}

static inline int red_grid(int32_t c0, int32_t c1) { return (c1 - c0 + 63) / 64; }

void launch_reduce_init(const float* part, int nb, int32_t ld, int32_t c0, int32_t c1, double* rz, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_init, dim3(red_grid(c0, c1)), dim3(1024), 0, s, part, nb, ld, c0, c1, rz);
  HIP_CHECK(hipGetLastError());
}
void launch_reduce_alpha(const float* part, int nb, int32_t ld, int32_t c0, int32_t c1, const double* rz, float* alpha,
                         Gate g, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_alpha, dim3(red_grid(c0, c1)), dim3(1024), 0, s, part, nb, ld, c0, c1, rz, alpha, g);
  HIP_CHECK(hipGetLastError());
}
void launch_reduce_beta(const float* part_rr, const float* part_rz, int nb, int32_t ld, int32_t c0, int32_t c1,
                        double* rz, float* beta, uint32_t* res_bits_slot, Gate g, hipStream_t s, uint32_t* done_ctr,
                        float* host_slot) {
  hipLaunchKernelGGL(k_reduce_beta, dim3(red_grid(c0, c1)), dim3(1024), 0, s, part_rr, part_rz, nb, ld, c0, c1, rz,
                     beta, res_bits_slot, g, done_ctr, host_slot);
  HIP_CHECK(hipGetLastError());
}
namespace {
// one word device -> host-mapped memory (the globally reduced residual of a sharded solve, behind its all-reduce)
__global__ void k_publish_word(const uint32_t* src, uint32_t* host_slot) {
  *reinterpret_cast<volatile uint32_t*>(host_slot) = *reinterpret_cast<const volatile uint32_t*>(src);
  __threadfence_system();
}
}  // namespace
void launch_publish_word(const uint32_t* src, uint32_t* host_slot, hipStream_t s) {
  hipLaunchKernelGGL(k_publish_word, dim3(1), dim3(1), 0, s, src, host_slot);
  HIP_CHECK(hipGetLastError());
}
void launch_reduce_sum(const float* part, int nb, int32_t ld, int32_t c0, int32_t c1, double* out_cols, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_sum, dim3(red_grid(c0, c1)), dim3(1024), 0, s, part, nb, ld, c0, c1, out_cols,
                     Gate{nullptr, 0.f});
  HIP_CHECK(hipGetLastError());
}
void launch_reduce_sum_gated(const float* part, int nb, int32_t ld, int32_t c0, int32_t c1, double* out_cols, Gate g,
                             hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_sum, dim3(red_grid(c0, c1)), dim3(1024), 0, s, part, nb, ld, c0, c1, out_cols, g);
  HIP_CHECK(hipGetLastError());
}
void launch_finish_init(const double* sums, int32_t c0, int32_t c1, double* rz, hipStream_t s) {
  hipLaunchKernelGGL(k_finish_init, dim3((c1 - c0 + 255) / 256), dim3(256), 0, s, sums, c0, c1, rz);
  HIP_CHECK(hipGetLastError());
}
void launch_finish_alpha(const double* sums, int32_t c0, int32_t c1, const double* rz, float* alpha, Gate g,
                         hipStream_t s) {
  hipLaunchKernelGGL(k_finish_alpha, dim3((c1 - c0 + 255) / 256), dim3(256), 0, s, sums, c0, c1, rz, alpha, g);
  HIP_CHECK(hipGetLastError());
}
void launch_finish_beta(const double* sums_rr, const double* sums_rz, int32_t c0, int32_t c1, double* rz, float* beta,
                        uint32_t* res_bits_slot, Gate g, hipStream_t s) {
  hipLaunchKernelGGL(k_finish_beta, dim3((c1 - c0 + 255) / 256), dim3(256), 0, s, sums_rr, sums_rz, c0, c1, rz, beta,
                     res_bits_slot, g);
  HIP_CHECK(hipGetLastError());
}
void launch_axpby(float* out, const float* a, float ca, const float* b, float cb, int64_t n, hipStream_t s) {
  const int64_t n4 = n / 4;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n4 + 255) / 256, 2048));
  hipLaunchKernelGGL(k_axpby, dim3(grid), dim3(256), 0, s, out, a, ca, b, cb, n4);
  HIP_CHECK(hipGetLastError());
}

}  // namespace osc
