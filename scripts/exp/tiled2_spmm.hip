// Experiment (not part of the product): source-panel-blocked SpMM with LDS-resident accumulators.
// See tiled_spmm.hip for the idea; here the accumulators of a 512-row tile live in LDS (64 KB per workgroup), the graph
// is stored as one fixed-width ELL per source panel (colP/wP [P][N][Wp], dead entries have w = 0 and col = own row),
// and every wave owns 64 rows of the tile for the whole panel walk, so no workgroup barrier is needed.
// Build: hipcc -O3 --offload-arch=gfx950 tiled2_spmm.hip -o tiled2_spmm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cmath>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int live_count8(float w) {  // 1 + highest lane-in-row (0..7) holding a live entry, wave-uniform
  unsigned long long m = __ballot(w != 0.f);
  m |= m >> 32; m |= m >> 16; m |= m >> 8;
  const unsigned b = (unsigned)m & 0xffu;
  return b ? 32 - __builtin_clz(b) : 0;
}

// W = 32 columns per slab: 8 lanes per row, 8 rows per wave step.  NB = batches of 8 entries per (row, panel): Wp = 8*NB.
template <int NB, int NW /*waves per block*/, int G /*row groups per wave*/, bool BLK, int SYNC>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_tiled2(const int* __restrict__ colP, const float* __restrict__ wP,
                                                    const float* __restrict__ X, float* __restrict__ Y, int N, int D,
                                                    int P, unsigned* bar) {
  constexpr int W = 32, LPR = 8, RPW = 8, TILE = NW * G * RPW, WP = 8 * NB;
  extern __shared__ __attribute__((aligned(16))) float accs[];  // [TILE][32]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane >> 3, lr = lane & 7;
  const int nslab = D / W, ntiles = (N + TILE - 1) / TILE;
  constexpr bool XAFF = true;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, nb = gridDim.x >> 3;
  // XAFF: XCD x owns slabs x, x+8, ... and all row tiles.  else: every slab in turn, XCD x owns tiles x, x+8, ...
  unsigned epoch = 0;
  auto xcd_sync = [&]() {  // soft barrier of the workgroups of this XCD: a pacing hint, never needed for correctness
    __syncthreads();
    if (threadIdx.x == 0) {
      ++epoch;
      __hip_atomic_fetch_add(bar + xcd * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = epoch * (unsigned)nb;
      for (int spin = 0; spin < 2000; ++spin) {
        if (__hip_atomic_load(bar + xcd * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) break;
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
  };
  const int rounds = XAFF ? (ntiles + nb - 1) / nb : (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  for (int s = XAFF ? xcd : 0; s < nslab; s += XAFF ? 8 : 1) {
    const size_t RS = BLK ? W : D;
    const float* Xs = X + (BLK ? (size_t)s * N * W : (size_t)s * W) + lr * 4;
    float* Ys = Y + (BLK ? (size_t)s * N * W : (size_t)s * W) + lr * 4;
    for (int rd = 0; rd < rounds; ++rd) {
      const int t = XAFF ? jb + rd * nb : blockIdx.x + rd * gridDim.x;  // t >= ntiles: no rows, but keeps pace
      if (SYNC == 1) xcd_sync();
      const int base = t * TILE + wave * G * RPW + sub;
      float* myacc = accs + (size_t)(wave * G * RPW + sub) * W + lr * 4;
#pragma unroll
      for (int g = 0; g < G; ++g) *(v4*)(myacc + g * RPW * W) = (v4){0.f, 0.f, 0.f, 0.f};
      for (int p = 0; p < P; ++p) {
        if (SYNC == 2) xcd_sync();
        const int* cp = colP + (size_t)p * N * WP;
        const float* wp = wP + (size_t)p * N * WP;
        int cj[G][NB];
        float wj[G][NB];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int row = min(base + g * RPW, N - 1);
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            cj[g][b] = cp[(size_t)row * WP + b * 8 + lr];
            wj[g][b] = base + g * RPW < N ? wp[(size_t)row * WP + b * 8 + lr] : 0.f;
          }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
          v4 acc = *(v4*)(myacc + g * RPW * W);
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const int cnt = live_count8(wj[g][b]);
            for (int u = 0; u < cnt; u += 4) {
              v4 v[4];
              float wv[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int j = __shfl(cj[g][b], (sub << 3) + u + q, 64);
                wv[q] = __shfl(wj[g][b], (sub << 3) + u + q, 64);
                v[q] = *(const v4*)(Xs + (size_t)j * RS);
              }
#pragma unroll
              for (int q = 0; q < 4; ++q) acc += wv[q] * v[q];
            }
          }
          *(v4*)(myacc + g * RPW * W) = acc;
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int row = base + g * RPW;
        if (row < N) {
          const v4 xs = *(const v4*)(Xs + (size_t)row * RS);
          __builtin_nontemporal_store(1.5f * xs - *(v4*)(myacc + g * RPW * W), (v4*)(Ys + (size_t)row * RS));
        }
      }
    }
  }
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 100000, D = argc > 2 ? atoi(argv[2]) : 768, ellw = 32;
  std::mt19937 rng(1);
  std::vector<int> col((size_t)ellw * N), deg(N);
  std::vector<float> wg((size_t)ellw * N);
  for (int i = 0; i < N; ++i) {
    deg[i] = 20 + rng() % 13;
    std::vector<int> c(deg[i]);
    for (auto& v : c) v = (int)(rng() % N);
    std::sort(c.begin(), c.end());
    for (int e = 0; e < ellw; ++e) {
      const bool live = e < deg[i];
      col[(size_t)i * ellw + e] = live ? c[e] : i;
      wg[(size_t)i * ellw + e] = live ? 0.03f * (1 + (rng() % 7)) : 0.f;
    }
  }
  std::vector<float> X((size_t)N * D);
  for (auto& v : X) v = (float)((int)(rng() % 2001) - 1000) * 1e-3f;
  float *dX, *dY;
  CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dY, X.size() * 4));
  CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> Yb(X.size());
  unsigned* dbar; CK(hipMalloc(&dbar, 8 * 32 * 4));
  auto run = [&](int P, int NB, int variant, int grid, bool xaff, int sync) {
    const int psz = (N + P - 1) / P, WP = 8 * NB;
    std::vector<int> cP((size_t)P * N * WP);
    std::vector<float> wPh((size_t)P * N * WP, 0.f);
    int over = 0, maxc = 0;
    for (int i = 0; i < N; ++i) {
      std::vector<int> fill(P, 0);
      for (int p = 0; p < P; ++p) for (int e = 0; e < WP; ++e) cP[((size_t)p * N + i) * WP + e] = i;
      for (int e = 0; e < deg[i]; ++e) {
        const int p = col[(size_t)i * ellw + e] / psz;
        if (fill[p] >= WP) { ++over; continue; }
        cP[((size_t)p * N + i) * WP + fill[p]] = col[(size_t)i * ellw + e];
        wPh[((size_t)p * N + i) * WP + fill[p]] = wg[(size_t)i * ellw + e];
        maxc = std::max(maxc, ++fill[p]);
      }
    }
    int* dcP; float* dwP;
    CK(hipMalloc(&dcP, cP.size() * 4)); CK(hipMalloc(&dwP, wPh.size() * 4));
    CK(hipMemcpy(dcP, cP.data(), cP.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwP, wPh.data(), wPh.size() * 4, hipMemcpyHostToDevice));
    if (xaff) {
      for (int sl = 0; sl < D / 32; ++sl) for (int i = 0; i < N; ++i) for (int c = 0; c < 32; ++c) Yb[((size_t)sl * N + i) * 32 + c] = X[(size_t)i * D + sl * 32 + c];
      CK(hipMemcpy(dX, Yb.data(), X.size() * 4, hipMemcpyHostToDevice));
    } else CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dY, 0, X.size() * 4));
    auto launch = [&]() {
#define L(NBB, NWW, GG) { const size_t sh = (size_t)NWW * GG * 8 * 32 * 4; \
      CK(hipMemsetAsync(dbar, 0, 8 * 32 * 4, 0)); \
      if (xaff) { if (sync == 0) hipLaunchKernelGGL((k_tiled2<NBB, NWW, GG, true, 0>), dim3(grid), dim3(NWW * 64), sh, 0, dcP, dwP, dX, dY, N, D, P, dbar); \
      else if (sync == 1) hipLaunchKernelGGL((k_tiled2<NBB, NWW, GG, true, 1>), dim3(grid), dim3(NWW * 64), sh, 0, dcP, dwP, dX, dY, N, D, P, dbar); \
      else hipLaunchKernelGGL((k_tiled2<NBB, NWW, GG, true, 2>), dim3(grid), dim3(NWW * 64), sh, 0, dcP, dwP, dX, dY, N, D, P, dbar); } else { \
      if (sync == 0) hipLaunchKernelGGL((k_tiled2<NBB, NWW, GG, false, 0>), dim3(grid), dim3(NWW * 64), sh, 0, dcP, dwP, dX, dY, N, D, P, dbar); \
      else if (sync == 1) hipLaunchKernelGGL((k_tiled2<NBB, NWW, GG, false, 1>), dim3(grid), dim3(NWW * 64), sh, 0, dcP, dwP, dX, dY, N, D, P, dbar); \
      else hipLaunchKernelGGL((k_tiled2<NBB, NWW, GG, false, 2>), dim3(grid), dim3(NWW * 64), sh, 0, dcP, dwP, dX, dY, N, D, P, dbar); } }
      if (variant == 0) { if (NB == 1) L(1, 8, 8) else if (NB == 2) L(2, 8, 8) else L(3, 8, 8) }        // 512 rows, 64 KB
      else if (variant == 1) { if (NB == 1) L(1, 8, 4) else if (NB == 2) L(2, 8, 4) else L(3, 8, 4) }   // 256 rows, 32 KB
      else { if (NB == 1) L(1, 16, 8) else if (NB == 2) L(2, 16, 8) else L(3, 16, 8) }                  // 1024 rows, 128 KB
    };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipGetLastError());
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(Yb.data(), dY, Yb.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    if (over == 0)
      for (int t = 0; t < 400; ++t) {
        const int i = (int)(((size_t)t * 7919) % N), c = (t * 13) % D;
        double ref = 1.5 * X[(size_t)i * D + c];
        for (int e = 0; e < deg[i]; ++e) ref -= (double)wg[(size_t)i * ellw + e] * X[(size_t)col[(size_t)i * ellw + e] * D + c];
        maxerr = std::fmax(maxerr, std::fabs(ref - Yb[xaff ? ((size_t)(c / 32) * N + i) * 32 + c % 32 : (size_t)i * D + c]));
      }
    printf("blk=%d P=%2d Wp=%2d variant=%d sync=%d grid=%4d : %.3f ms per apply (maxerr %.2e, dropped %d, max in-panel deg %d)\n", (int)xaff, P, WP, variant, sync, grid, ms / reps, maxerr, over, maxc);
    fflush(stdout);
    CK(hipFree(dcP)); CK(hipFree(dwP));
  };
  if (argc > 8) { run(atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]) != 0, atoi(argv[8])); return 0; }
  for (int blk : {1, 0}) for (int sync : {0, 1, 2}) {
    run(4, 3, 0, 512, blk, sync); run(6, 2, 0, 512, blk, sync);
  }
  return 0;
}
