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

#ifndef SCHED
#define SCHED 1
#endif
constexpr int D = 768, NKT = D / 64, NK16 = D / 16;

// Lean variant: every address of the K-step loop is (register set up once per tile) + (compile-time immediate).
//   ring of 4 stages, 12 K steps per tile: the stage of step kt is kt % 4 and the stage filled during it (kt + 3) % 4,
//   both compile-time in the unrolled loop; B piece sources = per-tile row bases + kt * 128 B immediate offsets;
//   fragment reads = four per-lane bases (one per k16 slice, the swizzle is an xor) + immediate (stage, column subtile).
//   Rows are padded to a multiple of 128 by the host: no clamps.
template <int NSTG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_core(const _Float16* __restrict__ Yh, int N, float* __restrict__ rowmax, unsigned* queue) {
  static_assert(NSTG == 4, "template value kept from the first version; the ring has 6 stages");
  extern __shared__ __attribute__((aligned(1024))) float lds[];  // 4 stages x [128 rows][32 float slots]
  __shared__ int s_rb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int nblocks = (N + 127) / 128, ntile = nblocks;
  const int frow = lane >> 3;
  const int swz = (l31 >> 1) & 7;
  // LDS byte offsets
  // NB: the immediate offset of global_load_lds is added to the LDS destination as well as to the global address, so
  // m0 carries (destination - offset); the ring starts 2 KB into the dynamic region to keep that positive.
  const unsigned lds_base = (unsigned)(size_t)lds + 2048u;
  const unsigned fill_base = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(32 * wave * 128));  // + stage * 16384 + q * 1024
  unsigned rd[4];  // fragment read base of slice s: row l31, chunk (2 s + h) ^ swz   (+ stage * 16384 + t * 4096)
#pragma unroll
  for (int s = 0; s < 4; ++s) rd[s] = (unsigned)(l31 * 128 + (((2 * s + h) ^ swz) * 16));
  const char* ldsc = reinterpret_cast<const char*>(lds) + 2048;
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
    // source of piece q of this wave's share of a B tile: row 32 wave + 8 q + frow of the tile, swizzled chunk
    const _Float16* bsrc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bsrc[q] = Yh + (size_t)(32 * wave + 8 * q + frow) * D + ((lane & 7) ^ (((q & 1) << 2) | (frow >> 1))) * 8;
    const size_t tile_stride = (size_t)128 * D;  // halfs between column tiles
    auto piece = [&](const _Float16* src, int kt, int stage, int q) {
      // m0 = LDS destination; the K offset rides in the instruction's immediate
      asm volatile("s_add_u32 m0, %1, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%3"
                   :: "v"(src), "s"(fill_base), "n"(0), "n"(0) : "memory");
      (void)kt; (void)stage; (void)q;
    };
    (void)piece;
#define PIECE(SRC, KT, STAGE, Q)                                                                                   \
    do {                                                                                                           \
      unsigned keep_;                                                                                              \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%3\n\ts_mov_b32 m0, %0" \
                   : "=&s"(keep_)                                                                                 \
                   : "v"(SRC), "s"(fill_base + (unsigned)((STAGE) * 16384 + (Q) * 1024) - (unsigned)((KT) * 128)), "n"((KT) * 128) \
                   : "memory");                                                                                    \
    } while (0)
    // ring of 6 stages, K steps handled in pairs (one barrier per 32 MFMAs): pair pr of a tile reads stages
    // (2 pr) % 6 and (2 pr + 1) % 6 and fetches pair pr + 2 (of this tile or the next) two pairs ahead
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int q = 0; q < 4; ++q) { PIECE(bsrc[q], st, st, q); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int ct = 0; ct < ntile; ++ct) {
      const bool last_tile = ct + 1 == ntile;
      f32x16 acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
      const _Float16* nsrc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) nsrc[q] = bsrc[q] + tile_stride;
#pragma unroll
      for (int pr = 0; pr < NKT / 2; ++pr) {
        const bool next_tile = 2 * pr + 4 >= NKT;
        const bool fetch = !(next_tile && last_tile);
        v4f fa[4], fb[4];
        auto read_frags = [&](int st, int sl, v4f (&bv)[4]) {
#pragma unroll
          for (int t = 0; t < 4; ++t) bv[t] = *(const v4f*)(ldsc + rd[sl] + st * 16384 + t * 4096);
        };
        read_frags((2 * pr) % 6, 0, fa);
#pragma unroll
        for (int u = 0; u < 8; ++u) {  // eight k16 slices: two K steps
          const int kt = 2 * pr + (u >> 2), sl = u & 3;
          v4f(&cur)[4] = (u & 1) ? fb : fa;
          v4f(&nxt)[4] = (u & 1) ? fa : fb;
          if (u + 1 < 8) read_frags((2 * pr + ((u + 1) >> 2)) % 6, (u + 1) & 3, nxt);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[kt * 4 + sl], __builtin_bit_cast(half8, cur[t]), acc[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (u < 4 && fetch) {  // two DMA pieces behind each of the first four MFMA groups
            const int fk = (2 * pr + 4 + (u >> 1)) % NKT, fst = (2 * pr + 4 + (u >> 1)) % 6;
#pragma unroll
            for (int q = 2 * (u & 1); q < 2 * (u & 1) + 2; ++q) {
              if (next_tile) { PIECE(nsrc[q], fk, fst, q); } else { PIECE(bsrc[q], fk, fst, q); }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (fetch) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the pair fetched during this pair stays in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) bsrc[q] = nsrc[q];
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        float m = -3.0e38f;
#pragma unroll
        for (int t = 0; t < 4; ++t) m = fmaxf(m, acc[t][g] < 200.f ? acc[t][g] : -3.0e38f);
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
  CK(hipMalloc(&dY, Y.size() * 2)); CK(hipMalloc(&dmax, (size_t)N * 4)); CK(hipMalloc(&dq, 4));
  CK(hipMemcpy(dY, Y.data(), Y.size() * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() {
    CK(hipMemsetAsync(dq, 0, 4, 0));
    const size_t sh = (size_t)6 * 128 * 32 * 4 + 2048;
#define L(S) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_core<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); \
               hipLaunchKernelGGL((k_core<S>), dim3(grid), dim3(256), sh, 0, dY, N, dmax, dq); }
    L(4)
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
      if (j == i) continue;
      float s = 0;
      for (int c = 0; c < D; ++c) s += (float)Y[(size_t)i * D + c] * (float)Y[(size_t)j * D + c];
      best = std::fmax(best, s);
    }
    worst = std::fmax(worst, std::fabs(best - hm[i]));
  }
  const double flop = 2.0 * N * (double)N * D;
  printf("core32 N=%d nstg=%d grid=%d : %.3f ms per sweep, %.1f TFLOP/s (%.1f %% of 2.5 PF), max |err| on 8 rows %.3e\n", N, nstg, grid,
         ms / reps, flop / (ms / reps * 1e-3) / 1e12, 100.0 * flop / (ms / reps * 1e-3) / 2.5e15, worst);
  return 0;
}
