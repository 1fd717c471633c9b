// Experiment (not part of the product; round 3): the panel-shaped fp16 similarity GEMM for
// 768 < D <= 1536, K split over wave pairs.
//   workgroup = 4 waves = 2 pairs; pair p (waves 2p, 2p + 1) owns 32 query rows, wave `kh` of a pair holds K half kh of
//   their panel in registers (32 rows x 768 halfs = 48 half8 = 192 registers, as the D <= 768 product kernel);
//   B: 128 columns x 64 halfs per K step (16 KB) by global_load_lds_dwordx4 into a 6-stage ring; a ROUND j consumes the
//      two stages that hold K steps j (read by the kh = 0 waves) and 12 + j (kh = 1 waves): 12 rounds per column tile,
//      one barrier per round, the pair of stages two rounds ahead fetched during a round (8 pieces per wave, counted vmcnt);
//   tile end: the waves of a pair exchange half of their partial accumulators through LDS (8 KB written + 8 KB read per
//      wave), each ends with the complete sums of 32 rows x 64 columns; epilogue stand-in: running row maximum.
// Against the D <= 768 shape this moves twice the LDS-DMA bytes per MFMA (64 instead of 128 query rows share a B tile).
// Measures what MFMA rate the shape reaches.  Build: hipcc -O3 --offload-arch=gfx950 knn_core_ksplit.hip -o knn_core_ksplit
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int D = 1536, NKT = D / 64, NKH = NKT / 2, NK16H = D / 32;  // 24 K steps, 12 per wave, 48 k16 slices per wave
constexpr unsigned LEAD = 4096, STAGE = 16384, RING = 6;

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

#define PIECE(SRC, KT, STG, Q)                                                                                     \
  do {                                                                                                             \
    unsigned keep_;                                                                                                \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%3\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_)                                                                                    \
                 : "v"(SRC), "s"(fill_base + (unsigned)((STG) * STAGE + (Q) * 1024) - (unsigned)((KT) * 128)), "n"((KT) * 128) \
                 : "memory");                                                                                      \
  } while (0)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_core_ks(const _Float16* __restrict__ Yh, int N, float* __restrict__ rowmax /*[2][Npad]*/, int Npad, unsigned* queue) {
  extern __shared__ __attribute__((aligned(1024))) float lds[];  // lead | 6 stages x [128 rows][32 float slots] | exchange 4 x 8 KB
  __shared__ int s_it;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int pair = wave >> 1, kh = wave & 1;
  const int nsets = (N + 63) / 64, ntile = (N + 127) / 128;
  const int frow = lane >> 3;
  const int swz = (l31 >> 1) & 7;
  const unsigned lds_base = (unsigned)(size_t)lds + LEAD;
  const unsigned fill_base = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(32 * wave * 128));
  unsigned rd[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) rd[s] = (unsigned)(l31 * 128 + (((2 * s + h) ^ swz) * 16));
  const char* ldsc = reinterpret_cast<const char*>(lds) + LEAD;
  float* xbuf = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + LEAD + RING * STAGE);  // [wave][t2][g4][lane][4]
  for (;;) {
    if (tid == 0) s_it = (int)atomicAdd(queue, 1u);
    __syncthreads();
    const int it = s_it;
    __syncthreads();
    if (it >= nsets) break;
    const int row = it * 64 + 32 * pair + l31;
    half8 areg[NK16H];
#pragma unroll
    for (int i = 0; i < NK16H; ++i) areg[i] = *(const half8*)(Yh + (size_t)row * D + kh * (D / 2) + i * 16 + h * 8);
    float cmax[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) cmax[g] = -3.0e38f;
    const _Float16* bsrc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bsrc[q] = Yh + (size_t)(32 * wave + 8 * q + frow) * D + ((lane & 7) ^ (((q & 1) << 2) | (frow >> 1))) * 8;
    const size_t tile_stride = (size_t)128 * D;
    // prologue: rounds 0 and 1 (K steps 0, 12, 1, 13 -> stages 0..3)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      PIECE(bsrc[q], 0, 0, q);
      PIECE(bsrc[q], 12, 1, q);
      PIECE(bsrc[q], 1, 2, q);
      PIECE(bsrc[q], 13, 3, q);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int ct = 0; ct < ntile; ++ct) {
      const bool last_tile = ct + 1 == ntile;
      f32x16 acc[4];
      const _Float16* nsrc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) nsrc[q] = bsrc[q] + tile_stride;
      static_for<0, NKH>([&](auto PR) {
        constexpr int pr = decltype(PR)::value;
        constexpr bool next_tile = pr + 2 >= NKH;
        const bool fetch = !(next_tile && last_tile);
        v4f fa[4], fb[4];
        const int myst = (2 * pr) % RING + kh;  // this wave's stage of the round (2 pr even, + kh never wraps inside a pair)
        auto read_frags = [&](int sl, v4f(&bv)[4]) {
#pragma unroll
          for (int t = 0; t < 4; ++t) bv[t] = *(const v4f*)(ldsc + rd[sl] + myst * STAGE + t * 4096);
        };
        read_frags(0, fa);
        static_for<0, 4>([&](auto UU) {
          constexpr int u = decltype(UU)::value;
          v4f(&cur)[4] = (u & 1) ? fb : fa;
          v4f(&nxt)[4] = (u & 1) ? fa : fb;
          if constexpr (u + 1 < 4) read_frags(u + 1, nxt);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if constexpr (pr == 0 && u == 0)
              asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc[t]) : "a"(areg[pr * 4 + u]), "v"(cur[t]));
            else
              asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[t]) : "a"(areg[pr * 4 + u]), "v"(cur[t]));
          }
          __builtin_amdgcn_sched_barrier(0);
          if (fetch) {  // two DMA pieces behind each of the four MFMA groups: the pair of stages two rounds ahead
            constexpr int fj = (pr + 2) % NKH, fst = (2 * (pr + 2)) % RING;
            constexpr int half_ = u >> 1, qa = 2 * (u & 1);  // groups 0, 1: K step fj; groups 2, 3: K step 12 + fj
            constexpr int fk = fj + 12 * half_, stg = fst + half_;
            if constexpr (next_tile) {
              PIECE(nsrc[qa], fk, stg, qa);
              PIECE(nsrc[qa + 1], fk, stg, qa + 1);
            } else {
              PIECE(bsrc[qa], fk, stg, qa);
              PIECE(bsrc[qa + 1], fk, stg, qa + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        if (fetch) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      });
#pragma unroll
      for (int q = 0; q < 4; ++q) bsrc[q] = nsrc[q];
      asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
      // exchange: this wave keeps column subtiles 2 kh, 2 kh + 1 and hands the other two to its partner
      float* mine = xbuf + (size_t)wave * 2048;
      const float* theirs = xbuf + (size_t)(wave ^ 1) * 2048;
      auto send = [&](auto KH) {  // (compile-time subtile indices: a runtime index would put the accumulators in scratch)
        constexpr int k_ = decltype(KH)::value;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          constexpr int base = 2 * (1 - k_);
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const f32x16& a_ = t2 == 0 ? acc[base] : acc[base + 1];
            v4f v = {a_[4 * g4], a_[4 * g4 + 1], a_[4 * g4 + 2], a_[4 * g4 + 3]};
            *reinterpret_cast<v4f*>(mine + ((t2 * 4 + g4) * 64 + lane) * 4) = v;
          }
        }
      };
      auto recv = [&](auto KH) {
        constexpr int k_ = decltype(KH)::value;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          constexpr int base = 2 * k_;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const f32x16& a_ = t2 == 0 ? acc[base] : acc[base + 1];
            const v4f v = *reinterpret_cast<const v4f*>(theirs + ((t2 * 4 + g4) * 64 + lane) * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) cmax[4 * g4 + c] = fmaxf(cmax[4 * g4 + c], a_[4 * g4 + c] + v[c]);
          }
        }
      };
      if (kh == 0) send(std::integral_constant<int, 0>{});
      else send(std::integral_constant<int, 1>{});
      __syncthreads();
      if (kh == 0) recv(std::integral_constant<int, 0>{});
      else recv(std::integral_constant<int, 1>{});
      // (the next tile's first exchange write is 12 barriers away: no second barrier needed here)
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      float m = cmax[g];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      const int r = it * 64 + 32 * pair + (g & 3) + 8 * (g >> 2) + 4 * h;
      if (l31 == 0 && r < N) rowmax[(size_t)kh * Npad + r] = m;
    }
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 100000;
  const int grid = argc > 2 ? atoi(argv[2]) : 256;
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
  _Float16* dY; float* dmax; unsigned* dq;
  CK(hipMalloc(&dY, Y.size() * 2)); CK(hipMalloc(&dmax, (size_t)2 * Npad * 4)); CK(hipMalloc(&dq, 4));
  CK(hipMemcpy(dY, Y.data(), Y.size() * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t sh = LEAD + (size_t)RING * STAGE + 32768;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_core_ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
  auto launch = [&]() {
    CK(hipMemsetAsync(dq, 0, 4, 0));
    hipLaunchKernelGGL(k_core_ks, dim3(grid), dim3(256), sh, 0, dY, N, dmax, Npad, dq);
    CK(hipGetLastError());
  };
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 3;
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> hm((size_t)2 * Npad);
  CK(hipMemcpy(hm.data(), dmax, hm.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int t = 0; t < 6; ++t) {
    const int i = (int)(((size_t)t * 12347 + 5) % N);
    float best = -1e30f;
    for (int j = 0; j < N; ++j) {
      float s = 0;
      for (int c = 0; c < D; ++c) s += (float)Y[(size_t)i * D + c] * (float)Y[(size_t)j * D + c];
      best = std::fmax(best, s);
    }
    const float got = std::fmax(hm[i], hm[(size_t)Npad + i]);
    worst = std::fmax(worst, std::fabs(best - got));
  }
  const double flop = 2.0 * N * (double)N * D;
  printf("K-split panel core: N=%d D=%d grid=%d : %.3f ms per sweep, %.1f TFLOP/s (%.1f %% of 2.5 PF), max |err| on 6 rows %.3e\n", N, D,
         grid, ms / reps, flop / (ms / reps * 1e-3) / 1e12, 100.0 * flop / (ms / reps * 1e-3) / 2.5e15, worst);
  return 0;
}
