// Experiment (round 6, not part of the product): the D = 768 panel GEMM of the kNN prefilter with TWO row groups per
// wave and 64-column stages -- VERDICT r05 item 1(a).
//   workgroup = 4 waves, one per SIMD, 256 query rows (wave w: rows 32 w .. 32 w + 31 of two consecutive 128-row blocks);
//   A: 2 x 32 x 768 fp16 panel per wave = 384 registers (the first APAN k16 slices of a row group in AGPRs, the rest in VGPRs);
//   B: 64 columns x 64 halfs per K step (8 KB) by global_load_lds_dwordx4, ring of 12 stages (stage = K step of the
//      64-column unit), KPB K steps per barrier, the group LA groups ahead is fetched during a group;
//   per k16 slice: 2 fragment reads feed 4 MFMAs; per K step and wave: 16 MFMAs, 8 ds_read_b128, 2 DMA pieces
//   (k_panel<12,1,1>: 16 MFMAs, 16 reads, 4 pieces).
//   epilogue stand-in: running row maximum.
// Build: hipcc -O3 --offload-arch=gfx950 -DKPB=4 -DLA=2 knn_core2.hip -o knn_core2
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cmath>
#include <type_traits>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

#ifndef KPB
#define KPB 4   // K steps per barrier
#endif
#ifndef LA
#define LA 2    // groups fetched ahead
#endif
#ifndef APAN
#define APAN 32  // k16 slices of a row group's panel kept in AGPRs (the other 48 - APAN in VGPRs)
#endif
constexpr int D = 768, NKT = D / 64, NK16 = D / 16;
constexpr int NG = NKT / KPB;
static_assert(NKT % KPB == 0 && LA >= 1 && LA <= NG - 1, "ring plan");
constexpr unsigned STAGE = 8192;

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

#define PIECE(SRC, KT, Q)                                                                                          \
  do {                                                                                                             \
    unsigned keep_;                                                                                                \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%3\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_)                                                                                   \
                 : "v"(SRC), "s"(fill_base + (unsigned)((KT) * STAGE + (Q) * 1024) - (unsigned)((KT) * 128)), "n"((KT) * 128) \
                 : "memory");                                                                                      \
  } while (0)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_core2(const _Float16* __restrict__ Yh, int N, float* __restrict__ rowmax, unsigned* queue) {
  extern __shared__ __attribute__((aligned(1024))) float lds[];  // 12 stages x [64 rows][32 float slots] (+2 KB lead)
  __shared__ int s_rb;
  __shared__ float s_max[4 * 32 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int nblocks = (N + 127) / 128, nsets = (nblocks + 1) / 2, nunits = nblocks * 2;
  const int frow = lane >> 3;
  const int swz = (l31 >> 1) & 7;
  const unsigned lds_base = (unsigned)(size_t)lds + 2048u;
  const unsigned fill_base = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(16 * wave * 128));  // + stage * 8192 + q * 1024
  unsigned rd[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) rd[s] = (unsigned)(l31 * 128 + (((2 * s + h) ^ swz) * 16));
  const char* ldsc = reinterpret_cast<const char*>(lds) + 2048;
  for (;;) {
    if (tid == 0) s_rb = (int)atomicAdd(queue, 1u);
    __syncthreads();
    const int set = s_rb;
    __syncthreads();
    if (set >= nsets) break;
    int rbv[2];
    rbv[0] = 2 * set;
    rbv[1] = min(2 * set + 1, nblocks - 1);
    half8 areg[2][NK16];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = rbv[r] * 128 + 32 * wave + l31;
#pragma unroll
      for (int i = 0; i < NK16; ++i) areg[r][i] = *(const half8*)(Yh + (size_t)row * D + i * 16 + h * 8);
    }
    // the running maxima wait in LDS (the product's thresholds do too): 32 more live registers made hipcc spill 38
    float* const cmax = s_max + (wave * 32) * 64 + lane;  // [r * 16 + g][lane]
#pragma unroll
    for (int i = 0; i < 32; ++i) cmax[i * 64] = -3.0e38f;
    // source of piece q of this wave's share of a 64-column unit: row 16 wave + 8 q + frow of the unit, swizzled chunk
    const _Float16* bsrc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
      bsrc[q] = Yh + (size_t)(16 * wave + 8 * q + frow) * D + ((lane & 7) ^ (((q & 1) << 2) | (frow >> 1))) * 8;
    const size_t unit_stride = (size_t)64 * D;
    // prologue: groups 0 .. LA - 1 of unit 0
    static_for<0, LA * KPB>([&](auto KT) {
      constexpr int kt = decltype(KT)::value;
      PIECE(bsrc[0], kt, 0);
      PIECE(bsrc[1], kt, 1);
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int cu = 0; cu < nunits; ++cu) {
      const bool last_unit = cu + 1 == nunits;
      f32x16 acc[2][2];
      const _Float16* nsrc[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) nsrc[q] = bsrc[q] + unit_stride;
      static_for<0, NG>([&](auto GG) {
        constexpr int g = decltype(GG)::value;
        constexpr bool next_unit = g + LA >= NG;
        constexpr int fg = (g + LA) % NG;  // group fetched during this one
        const bool fetch = !(next_unit && last_unit);
        v4f fa[2], fb[2];
        auto read_frags = [&](int st, int sl, v4f(&bv)[2]) {
#pragma unroll
          for (int t = 0; t < 2; ++t) bv[t] = *(const v4f*)(ldsc + rd[sl] + st * STAGE + t * 4096);
        };
        read_frags(g * KPB, 0, fa);
        static_for<0, 4 * KPB>([&](auto UU) {
          constexpr int u = decltype(UU)::value;
          constexpr int kt = g * KPB + (u >> 2), sl = u & 3;
          v4f(&cur)[2] = (u & 1) ? fb : fa;
          v4f(&nxt)[2] = (u & 1) ? fa : fb;
          if constexpr (u + 1 < 4 * KPB) read_frags(g * KPB + ((u + 1) >> 2), (u + 1) & 3, nxt);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              constexpr int ai = kt * 4 + sl;
              if constexpr (ai < APAN) {
                if constexpr (g == 0 && u == 0)
                  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc[r][t]) : "a"(areg[r][ai]), "v"(cur[t]));
                else
                  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[r][t]) : "a"(areg[r][ai]), "v"(cur[t]));
              } else {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[r][t]) : "v"(areg[r][ai]), "v"(cur[t]));
              }
            }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr ((u & 1) == 1) {  // one K step's two pieces behind every fourth slice; here: one piece per two slices
            constexpr int pi = u >> 1;   // 0 .. 2 KPB - 1: piece index within the group
            constexpr int fk = fg * KPB + (pi >> 1), q = pi & 1;
            if (fetch) {
              if constexpr (next_unit) { PIECE(nsrc[q], fk, q); } else { PIECE(bsrc[q], fk, q); }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        });
        if (fetch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA - 1) * KPB * 2) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      });
#pragma unroll
      for (int q = 0; q < 2; ++q) bsrc[q] = nsrc[q];
      asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int g = 0; g < 16; ++g) {  // (the diagonal, 256 in this scaling, is left out by value: the check is the best OTHER column)
          const float a0 = acc[r][0][g] < 200.f ? acc[r][0][g] : -3.0e38f, a1 = acc[r][1][g] < 200.f ? acc[r][1][g] : -3.0e38f;
          cmax[(r * 16 + g) * 64] = fmaxf(cmax[(r * 16 + g) * 64], fmaxf(a0, a1));
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        float m = cmax[(r * 16 + g) * 64];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        const int row = rbv[r] * 128 + 32 * wave + (g & 3) + 8 * (g >> 2) + 4 * h;
        if (l31 == 0 && row < N && (r == 0 || 2 * set + 1 < nblocks)) rowmax[row] = m;
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
  const int Npad = (N + 255) / 256 * 256;
  std::vector<_Float16> Y((size_t)Npad * D, (_Float16)0.f);
  std::vector<float> rowf(D);
  for (int i = 0; i < N; ++i) {
    double n2 = 0;
    for (int c = 0; c < D; ++c) { rowf[c] = nd(rng); n2 += (double)rowf[c] * rowf[c]; }
    const float inv = 1.0f / (float)std::sqrt(n2);
    for (int c = 0; c < D; ++c) Y[(size_t)i * D + c] = (_Float16)(rowf[c] * inv * 16.f);
  }
  _Float16* dY; float* dmax; unsigned* dq;
  CK(hipMalloc(&dY, Y.size() * 2 + 65536)); CK(hipMalloc(&dmax, (size_t)Npad * 4)); CK(hipMalloc(&dq, 4));
  CK(hipMemcpy(dY, Y.data(), Y.size() * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t sh = (size_t)12 * STAGE + 2048;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_core2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
  auto launch = [&]() {
    CK(hipMemsetAsync(dq, 0, 4, 0));
    hipLaunchKernelGGL(k_core2, dim3(grid), dim3(256), sh, 0, dY, N, dmax, dq);
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
  for (int t = 0; t < 8; ++t) {
    const int i = (int)(((size_t)t * 12347 + (t & 1) * 128) % N);
    float best = -1e30f;
    for (int j = 0; j < N; ++j) {
      if (j == i) continue;
      float s = 0;
      for (int c = 0; c < D; ++c) s += (float)Y[(size_t)i * D + c] * (float)Y[(size_t)j * D + c];
      best = std::fmax(best, s);
    }
    worst = std::fmax(worst, std::fabs(best - hm[i]));
  }
  const double flop = 2.0 * N * (double)N * D;
  printf("core2 KPB=%d LA=%d APAN=%d N=%d grid=%d : %.3f ms per sweep, %.1f TFLOP/s (%.1f %% of 2.5 PF), max |err| on 8 rows %.3e\n", KPB, LA, APAN, N,
         grid, ms / reps, flop / (ms / reps * 1e-3) / 1e12, 100.0 * flop / (ms / reps * 1e-3) / 2.5e15, worst);
  return 0;
}
