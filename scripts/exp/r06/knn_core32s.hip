// Experiment (round 6, not part of the product): knn_core.hip's D = 768 panel GEMM with a HAND-PLACED K loop.
// In k_panel / knn_core.hip a k16 slice is [4 fragment reads][4 MFMAs back to back][2 LDS-DMA pieces of 5 instructions]
// between scheduling fences: the 24 free issue cycles behind each MFMA stay empty and everything else is issued between
// the groups, where only the last MFMA's tail covers it.  Here every instruction of the loop is inline asm in the
// order it is to issue (MI355X_MICROARCH.md: <= 5 single-issue instructions hide per v_mfma_f32_32x32x16 gap):
//   slice u of a pair:  [wait cur 0,1] MFMA0  RD nxt0, RD nxt1   MFMA1  M0 <- piece destination
//                       [wait cur 2,3] MFMA2  RD nxt2, RD nxt3   MFMA3  global_load_lds piece u
//   the workgroup barrier of a pair sits BEFORE its last slice (whose fragments are in registers by then: lgkmcnt(0)),
//   so the first slice of the next pair is read behind the barrier under that slice's MFMAs instead of in a bubble;
//   one DMA piece per slice (8 per pair, as before), M0 written by ONE s_add (no save / restore, no nop: an MFMA sits
//   between the write and the load).
// Build: hipcc -O3 --offload-arch=gfx950 knn_core32s.hip -o knn_core32s
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cmath>
#include <algorithm>
#include <type_traits>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int D = 768, NKT = D / 64, NK16 = D / 16;
constexpr unsigned STAGE = 16384;
#ifndef VARIANT
#define VARIANT 0  // 1: reads of the next slice all behind MFMA0 / MFMA1 (earlier), 2: no early barrier (barrier behind the pair, bubble kept)
#endif

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

#define RD(DST, ABASE, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ABASE), "n"(OFF))
#define LGKM(N_) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N_))
#define VMC(N_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory")
#define SETM0(KT, STG, Q) asm volatile("s_add_i32 m0, %0, %1" ::"s"(fill_base), "n"((unsigned)((STG) * STAGE + (Q) * 1024 - (KT) * 128)) : "scc")
#define DMA(SRC, KT) asm volatile("global_load_lds_dwordx4 %0, off offset:%1" ::"v"(SRC), "n"((KT) * 128) : "memory")
// measurement-only knockouts (wrong results by construction): -DNODMA no pieces inside the K loop (stale stages),
// -DNOBAR no workgroup barrier in the K loop, -DNORD no fragment reads inside the K loop (stale fragments), -DNOM0 pieces
// without their M0 writes (all land on one stage)
#ifdef NODMA
#define LOOP_DMA(SRC, KT) do {} while (0)
#define LOOP_SETM0(KT, STG, Q) do {} while (0)
#else
#define LOOP_DMA(SRC, KT) DMA(SRC, KT)
#ifdef NOM0
#define LOOP_SETM0(KT, STG, Q) do {} while (0)
#else
#define LOOP_SETM0(KT, STG, Q) SETM0(KT, STG, Q)
#endif
#endif
#ifdef NOBAR
#define LOOP_BARRIER() do {} while (0)
#else
#define LOOP_BARRIER() __builtin_amdgcn_s_barrier()
#endif

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_core(const _Float16* __restrict__ Yh, int N, float* __restrict__ rowmax, unsigned* queue, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(1024))) float lds[];
  __shared__ int s_rb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int nblocks = (N + 127) / 128, ntile = nblocks;
  const int frow = lane >> 3;
  const int swz = (l31 >> 1) & 7;
  const unsigned lds_base = (unsigned)(size_t)lds + 2048u;
  const unsigned fill_base = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(32 * wave * 128));
  // fragment read bases (absolute LDS byte addresses) of k16 slice s: row l31, chunk (2 s + h) ^ swz; stages 0-2 from the
  // low base, 3-5 from the high one (the 16-bit offset field reaches 64 KB)
  unsigned rlo[4], rhi[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    rlo[s] = lds_base + (unsigned)(l31 * 128 + (((2 * s + h) ^ swz) * 16));
    rhi[s] = rlo[s] + 3 * STAGE;
  }
  for (;;) {
    if (tid == 0) s_rb = (int)atomicAdd(queue, 1u);
    __syncthreads();
    const int rb = s_rb;
    __syncthreads();
    if (rb >= nblocks) break;
    const int row = rb * 128 + 32 * wave + l31;
    half8 areg[NK16];
#pragma unroll
    for (int i = 0; i < NK16; ++i) areg[i] = *(const half8*)(Yh + (size_t)row * D + i * 16 + h * 8);
    float cmax[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) cmax[g] = -3.0e38f;
    const _Float16* bsrc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bsrc[q] = Yh + (size_t)(32 * wave + 8 * q + frow) * D + ((lane & 7) ^ (((q & 1) << 2) | (frow >> 1))) * 8;
    const size_t tile_stride = (size_t)128 * D;
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        SETM0(st, st, q);
        asm volatile("s_nop 0");
        DMA(bsrc[q], st);
      }
    VMC(0);
    __syncthreads();
#ifdef STAMP
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
    auto tile = [&](auto LAST, f32x16(&acc)[4]) {
      constexpr bool last_tile = decltype(LAST)::value;
      const _Float16* nsrc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) nsrc[q] = bsrc[q] + tile_stride;
      v4f fa[4], fb[4];
#define RD_AT(DST, ST, SL, T)                                                  \
  do {                                                                          \
    if constexpr ((ST) < 3) RD(DST, rlo[SL], (ST) * STAGE + (T) * 4096);        \
    else RD(DST, rhi[SL], ((ST) - 3) * STAGE + (T) * 4096);                     \
  } while (0)
      RD_AT(fa[0], 0, 0, 0);
      RD_AT(fa[1], 0, 0, 1);
      RD_AT(fa[2], 0, 0, 2);
      RD_AT(fa[3], 0, 0, 3);
      static_for<0, NKT / 2>([&](auto PR) {
        constexpr int pr = decltype(PR)::value;
        constexpr bool next_tile = 2 * pr + 4 >= NKT;
        constexpr bool fetch = !(next_tile && last_tile);
        static_for<0, 8>([&](auto UU) {
          constexpr int u = decltype(UU)::value;
          constexpr int kt = 2 * pr + (u >> 2), sl = u & 3;
          constexpr bool has_next = !(pr == NKT / 2 - 1 && u == 7);
          constexpr int nu = (u + 1) & 7, npr = pr + ((u + 1) >> 3);
          constexpr int nst = (2 * npr + (nu >> 2)) % 6, nsl = nu & 3;
          v4f(&cur)[4] = (u & 1) ? fb : fa;
          v4f(&nxt)[4] = (u & 1) ? fa : fb;
#ifdef NORD
#define RDN(T) do {} while (0)
#else
#define RDN(T)                                      \
  do {                                              \
    if constexpr (has_next) RD_AT(nxt[T], nst, nsl, T); \
  } while (0)
#endif
#define MFMA(T)                                                                                                             \
  do {                                                                                                                      \
    if constexpr (pr == 0 && u == 0)                                                                                        \
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc[T]) : "a"(areg[kt * 4 + sl]), "v"(cur[T]));          \
    else                                                                                                                    \
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[T]) : "a"(areg[kt * 4 + sl]), "v"(cur[T]));          \
  } while (0)
          constexpr bool early_bar = VARIANT != 2;
          if constexpr (u == 7 && early_bar) {
            LGKM(0);  // the last slice's fragments are in registers: this wave is done with the pair's stages
            if constexpr (fetch) VMC(7); else VMC(0);  // (its own pieces of the next pair have landed)
            LOOP_BARRIER();
          } else {
            LGKM(2);
          }
          MFMA(0);
          if constexpr (u != 7 || early_bar) {
            RDN(0);
            RDN(1);
            if constexpr (VARIANT == 1) { RDN(2); }
          }
          MFMA(1);
          if constexpr (VARIANT == 1 && (u != 7 || early_bar)) RDN(3);
          constexpr int fk = (2 * pr + 4 + (u >> 2)) % NKT, fst = (2 * pr + 4 + (u >> 2)) % 6, q = u & 3;
          if constexpr (fetch) LOOP_SETM0(fk, fst, q);
          if constexpr (!(u == 7 && early_bar)) {
            if constexpr (VARIANT == 1) LGKM(4); else LGKM(2);
            if constexpr (!has_next || (u == 7 && !early_bar)) LGKM(0);
          }
          MFMA(2);
          if constexpr (VARIANT != 1 && (u != 7 || early_bar)) {
            RDN(2);
            RDN(3);
          }
          MFMA(3);
          if constexpr (fetch) {
            if constexpr (next_tile) LOOP_DMA(nsrc[q], fk); else LOOP_DMA(bsrc[q], fk);
          }
          if constexpr (u == 7 && !early_bar) {
            if constexpr (fetch) VMC(8); else VMC(0);
            LOOP_BARRIER();
            if constexpr (has_next) {
              RDN(0); RDN(1); RDN(2); RDN(3);
            }
          }
        });
      });
#pragma unroll
      for (int q = 0; q < 4; ++q) bsrc[q] = nsrc[q];
    };
    for (int ct = 0; ct < ntile; ++ct) {
      f32x16 acc[4];
      if (ct + 1 == ntile) tile(std::true_type{}, acc);
      else tile(std::false_type{}, acc);
      asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        float m = -3.0e38f;
#pragma unroll
        for (int t = 0; t < 4; ++t) m = fmaxf(m, acc[t][g] < 200.f ? acc[t][g] : -3.0e38f);
        cmax[g] = fmaxf(cmax[g], m);
      }
    }
#ifdef STAMP
    if (tid == 0) {  // shader cycles and 100 MHz ticks of this row block's column sweep (diagnostic build only)
      stamps[2 * rb] = __builtin_amdgcn_s_memtime() - c0;
      stamps[2 * rb + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
#endif
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      float m = cmax[g];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      const int r = rb * 128 + 32 * wave + (g & 3) + 8 * (g >> 2) + 4 * h;
      if (l31 == 0 && r < N) rowmax[r] = m;
    }
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 100000;
  const int grid = argc > 2 ? atoi(argv[2]) : 256;
  const int reps = argc > 3 ? atoi(argv[3]) : 3;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  const int Npad = (N + 127) / 128 * 128;
  std::vector<_Float16> Y((size_t)Npad * D, (_Float16)0.f);
  std::vector<float> rowf(D);
  for (int i = 0; i < N; ++i) {
    double n2 = 0;
    for (int c = 0; c < D; ++c) { rowf[c] = nd(rng); n2 += (double)rowf[c] * rowf[c]; }
    const float inv = 1.0f / (float)std::sqrt(n2);
    for (int c = 0; c < D; ++c) Y[(size_t)i * D + c] = (_Float16)(rowf[c] * inv * 16.f);
  }
  _Float16* dY; float* dmax; unsigned* dq; unsigned long long* dst;
  CK(hipMalloc(&dst, (size_t)(N / 128 + 2) * 16));
  CK(hipMalloc(&dY, Y.size() * 2)); CK(hipMalloc(&dmax, (size_t)N * 4)); CK(hipMalloc(&dq, 4));
  CK(hipMemcpy(dY, Y.data(), Y.size() * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t sh = (size_t)6 * STAGE + 2048;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_core), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
  auto launch = [&]() {
    CK(hipMemsetAsync(dq, 0, 4, 0));
    hipLaunchKernelGGL(k_core, dim3(grid), dim3(256), sh, 0, dY, N, dmax, dq, dst);
    CK(hipGetLastError());
  };
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> hm(N);
  CK(hipMemcpy(hm.data(), dmax, (size_t)N * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int t = 0; t < 16; ++t) {
    const int i = (int)(((size_t)t * 12347 + (t & 3) * 32) % N);
    float best = -1e30f;
    for (int j = 0; j < N; ++j) {
      if (j == i) continue;
      float s = 0;
      for (int c = 0; c < D; ++c) s += (float)Y[(size_t)i * D + c] * (float)Y[(size_t)j * D + c];
      best = std::fmax(best, s);
    }
    worst = std::fmax(worst, std::fabs(best - hm[i]));
  }
#ifdef STAMP
  {
    const int nb = (N + 127) / 128;
    std::vector<unsigned long long> st((size_t)nb * 2);
    CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk, cyc;
    for (int b = 0; b < nb; ++b) { clk.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 0.1); cyc.push_back((double)st[2 * b] / nb); }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    printf("  in-kernel clock (median over row blocks) %.3f GHz; shader cycles per column tile: median %.0f, p10 %.0f, p90 %.0f (192 MFMAs = 6144)\n",
           clk[nb / 2], cyc[nb / 2], cyc[nb / 10], cyc[nb * 9 / 10]);
  }
#endif
  const double flop = 2.0 * N * (double)N * D;
  printf("core32s VARIANT=%d N=%d grid=%d : %.3f ms per sweep, %.1f TFLOP/s (%.1f %% of 2.5 PF), max |err| on 16 rows %.3e\n", VARIANT, N, grid,
         ms / reps, flop / (ms / reps * 1e-3) / 1e12, 100.0 * flop / (ms / reps * 1e-3) / 2.5e15, worst);
  return 0;
}
