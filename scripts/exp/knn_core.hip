// Experiment (not part of the product): the fp16 similarity GEMM of the kNN prefilter in a different shape --
// query panel register-resident, column tiles streamed through a deep LDS-DMA ring, one wave per SIMD.
//   workgroup = 4 waves = 128 query rows (wave w: rows 32 w .. 32 w + 31), persistent over row blocks;
//   A: the wave's 32 x D fp16 panel lives in registers for the whole column sweep (D = 768: 48 half8 per lane);
//   B: 128 columns x 64 halfs per K step (16 KB) by global_load_lds_dwordx4 into an NSTG-stage ring (NSTG - 1 steps
//      in flight; counted vmcnt), swizzled for ds_read_b128's 16-lane groups;
//   epilogue stand-in: running row maximum (the real kernel keeps top-k lists); output = per-row max similarity.
// Measures what MFMA rate this shape reaches without the list update.
// Build: hipcc -O3 --offload-arch=gfx950 knn_core.hip -o knn_core
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

constexpr int D = 768, NKT = D / 64, NK16 = D / 16;

template <int NSTG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_core(const _Float16* __restrict__ Yh, int N, float* __restrict__ rowmax, unsigned* queue) {
  extern __shared__ __attribute__((aligned(1024))) float lds[];  // NSTG stages x [128 rows][32 float slots]
  __shared__ int s_rb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int nblocks = (N + 127) / 128, ntile = (N + 127) / 128;
  const unsigned lds_base = (unsigned)(size_t)lds;
  const int frow = lane >> 3;
  auto fchunk_of = [&](int q) -> int { return ((lane & 7) ^ (((q & 1) << 2) | (frow >> 1))) * 8; };  // halfs
  auto glds16 = [&](const _Float16* src, unsigned dst_bytes) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst_bytes) : "memory");
  };
  const int swz = (l31 >> 1) & 7;
  for (;;) {
    if (tid == 0) s_rb = (int)atomicAdd(queue, 1u);
    __syncthreads();
    const int rb = s_rb;
    __syncthreads();
    if (rb >= nblocks) break;
    const int row = min(rb * 128 + 32 * wave + l31, N - 1);
    half8 areg[NK16];
#pragma unroll
    for (int i = 0; i < NK16; ++i) areg[i] = *(const half8*)(Yh + (size_t)row * D + i * 16 + h * 8);
    float cmax[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) cmax[g] = -3.0e38f;

    // B ring: step = (tile, kt); the wave fills rows [32 wave, 32 wave + 32) of a stage, 8 rows per piece
    const int total = ntile * NKT;
    int ict = 0, ikt = 0, issued = 0;
    auto issue_next = [&]() {
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((issued % NSTG) * 128 * 32 + 32 * wave * 32) * 4u);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const _Float16* src = Yh + (size_t)min(ict * 128 + 32 * wave + 8 * q + frow, N - 1) * D + fchunk_of(q) + ikt * 64;
        glds16(src, dst + (unsigned)(8 * q * 32) * 4u);
      }
      ++issued;
      if (++ikt == NKT) { ikt = 0; ++ict; }
    };
    for (int pre = 0; pre < NSTG - 1 && issued < total; ++pre) issue_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int step = 0;
    for (int ct = 0; ct < ntile; ++ct) {
      f32x16 acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt, ++step) {
        const bool more = issued < total;
        if (more) issue_next();
        const float* Bsw = lds + (step % NSTG) * 128 * 32 + l31 * 32;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int co = ((2 * s + h) ^ swz) * 4;
          v4f bv[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) bv[t] = *(const v4f*)(Bsw + 32 * t * 32 + co);
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[kt * 4 + s], __builtin_bit_cast(half8, bv[t]), acc[t], 0, 0, 0);
        }
        // the stage read next (step + 1) was issued NSTG - 2 steps before this step's own issue: all but the youngest
        // (NSTG - 2) * 4 pieces must have landed
        if (more) {
          if constexpr (NSTG == 8) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
          else if constexpr (NSTG == 6) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
          else if constexpr (NSTG == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
      }
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        float m = fmaxf(fmaxf(acc[0][g], acc[1][g]), fmaxf(acc[2][g], acc[3][g]));
        cmax[g] = fmaxf(cmax[g], m);
      }
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      float m = cmax[g];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      const int r = rb * 128 + 32 * wave + (g & 3) + 8 * (g >> 2) + 4 * h;
      if (l31 == 0 && r < N) rowmax[r] = m;
    }
    __syncthreads();  // every wave is done with the ring before the next row block refills it
  }
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 100000;
  const int nstg = argc > 2 ? atoi(argv[2]) : 8;
  const int grid = argc > 3 ? atoi(argv[3]) : 256;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<_Float16> Y((size_t)N * D);
  std::vector<float> rowf(D);
  for (int i = 0; i < N; ++i) {
    double n2 = 0;
    for (int c = 0; c < D; ++c) { rowf[c] = nd(rng); n2 += (double)rowf[c] * rowf[c]; }
    const float inv = 1.0f / (float)std::sqrt(n2);
    for (int c = 0; c < D; ++c) Y[(size_t)i * D + c] = (_Float16)(rowf[c] * inv * 16.f);
  }
  _Float16* dY; float* dmax; unsigned* dq;
  CK(hipMalloc(&dY, Y.size() * 2)); CK(hipMalloc(&dmax, (size_t)N * 4)); CK(hipMalloc(&dq, 4));
  CK(hipMemcpy(dY, Y.data(), Y.size() * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() {
    CK(hipMemsetAsync(dq, 0, 4, 0));
    const size_t sh = (size_t)nstg * 128 * 32 * 4;
#define L(S) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_core<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); \
               hipLaunchKernelGGL((k_core<S>), dim3(grid), dim3(256), sh, 0, dY, N, dmax, dq); }
    if (nstg == 8) L(8) else if (nstg == 6) L(6) else if (nstg == 4) L(4) else L(2)
    CK(hipGetLastError());
  };
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 3;
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> hm(N);
  CK(hipMemcpy(hm.data(), dmax, (size_t)N * 4, hipMemcpyDeviceToHost));
  // check a few rows: max over all columns (the diagonal, 256 = 16 * 16 in this scaling, dominates)
  double worst = 0;
  for (int t = 0; t < 8; ++t) {
    const int i = (int)(((size_t)t * 12347) % N);
    float best = -1e30f;
    for (int j = 0; j < N; ++j) {
      float s = 0;
      for (int c = 0; c < D; ++c) s += (float)Y[(size_t)i * D + c] * (float)Y[(size_t)j * D + c];
      best = std::fmax(best, s);
    }
    worst = std::fmax(worst, std::fabs(best - hm[i]));
  }
  const double flop = 2.0 * N * (double)N * D;
  printf("N=%d nstg=%d grid=%d : %.3f ms per sweep, %.1f TFLOP/s (%.1f %% of 2.5 PF), max |err| on 8 rows %.3e\n", N, nstg, grid,
         ms / reps, flop / (ms / reps * 1e-3) / 1e12, 100.0 * flop / (ms / reps * 1e-3) / 2.5e15, worst);
  return 0;
}
