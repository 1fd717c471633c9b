// Experiment (not part of the product): output-stationary, source-panel-blocked SpMM for the CG operator apply.
// XCD x owns column slabs x, x+8, ... (W columns each, row-major state).  A workgroup keeps the accumulators of a tile
// of rows in registers and walks the source rows panel by panel (P panels of N/P rows, W*4 bytes per row: a panel is
// sized to sit in the XCD's 4 MB L2).  All workgroups of an XCD walk the panels in the same order at about the same
// pace, so a panel's lines are fetched from the fabric about once per round and then hit in L2.
// Build: hipcc -O3 --offload-arch=gfx950 tiled_spmm.hip -o tiled_spmm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cmath>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));
constexpr int PS = 16;  // bytes of panel offsets per row

template <int W, int K>
__global__ __launch_bounds__(256) void k_tiled(const int* __restrict__ col, const float* __restrict__ wgt,
                                               const unsigned char* __restrict__ poff, const float* __restrict__ X,
                                               float* __restrict__ Y, int N, int D, int P, int ellw) {
  constexpr int LPR = W / 4, RPW = 64 / LPR, TILE = 4 * RPW * K;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, lr = lane % LPR;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, nb = gridDim.x >> 3;
  const int nslab = D / W, ntiles = (N + TILE - 1) / TILE;
  for (int s = xcd; s < nslab; s += 8) {
    const float* Xs = X + (size_t)s * W + lr * 4;
    float* Ys = Y + (size_t)s * W + lr * 4;
    for (int t = jb; t < ntiles; t += nb) {
      const int base = t * TILE + wave * RPW * K + sub;
      v4 acc[K];
#pragma unroll
      for (int kk = 0; kk < K; ++kk) acc[kk] = (v4){0.f, 0.f, 0.f, 0.f};
      for (int p = 0; p < P; ++p) {
        int o0[K], n[K], cj[K];
        float wj[K];
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {  // stage A: panel offsets of the K rows of this lane group
          const int row = min(base + kk * RPW, N - 1);
          o0[kk] = poff[(size_t)row * PS + p];
          n[kk] = base + kk * RPW < N ? poff[(size_t)row * PS + p + 1] - o0[kk] : 0;
        }
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {  // stage B: first LPR edges of every row
          const int row = min(base + kk * RPW, N - 1);
          cj[kk] = row;
          wj[kk] = 0.f;
          if (lr < n[kk]) {
            cj[kk] = col[(size_t)row * ellw + o0[kk] + lr];
            wj[kk] = wgt[(size_t)row * ellw + o0[kk] + lr];
          }
        }
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {  // stage C: gathers
          const int c2 = min(n[kk], LPR);
          for (int u = 0; u < c2; u += 2) {
            const int j0 = __shfl(cj[kk], sub * LPR + u, 64), j1 = __shfl(cj[kk], sub * LPR + ((u + 1) & (LPR - 1)), 64);
            const float w0 = __shfl(wj[kk], sub * LPR + u, 64);
            float w1 = __shfl(wj[kk], sub * LPR + ((u + 1) & (LPR - 1)), 64);
            if (u + 1 >= c2) w1 = 0.f;
            const v4 v0 = *(const v4*)(Xs + (size_t)j0 * D);
            const v4 v1 = *(const v4*)(Xs + (size_t)j1 * D);
            acc[kk] += w0 * v0;
            acc[kk] += w1 * v1;
          }
          if (n[kk] > LPR) {  // rare: more than LPR edges of one row in one panel
            const int row = min(base + kk * RPW, N - 1);
            for (int e = LPR; e < n[kk]; ++e) {
              const int j = col[(size_t)row * ellw + o0[kk] + e];
              const float wv = wgt[(size_t)row * ellw + o0[kk] + e];
              acc[kk] += wv * *(const v4*)(Xs + (size_t)j * D);
            }
          }
        }
      }
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        const int row = base + kk * RPW;
        if (row < N) {
          const v4 xs = *(const v4*)(Xs + (size_t)row * D);
          __builtin_nontemporal_store(1.5f * xs - acc[kk], (v4*)(Ys + (size_t)row * D));
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
  int* dcol; float *dw, *dX, *dY; unsigned char* dpoff;
  CK(hipMalloc(&dcol, col.size() * 4)); CK(hipMalloc(&dw, wg.size() * 4)); CK(hipMalloc(&dpoff, (size_t)N * PS));
  CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dY, X.size() * 4));
  CK(hipMemcpy(dcol, col.data(), col.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, wg.data(), wg.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> Yb(X.size());
  std::vector<unsigned char> poff((size_t)N * PS);
  auto run = [&](int W, int K, int P, int nbx) {
    const int psz = (N + P - 1) / P;
    for (int i = 0; i < N; ++i) {
      int e = 0;
      for (int p = 0; p <= P; ++p) {
        while (e < deg[i] && col[(size_t)i * ellw + e] < p * psz) ++e;
        poff[(size_t)i * PS + p] = (unsigned char)(p == P ? deg[i] : e);
      }
    }
    CK(hipMemcpy(dpoff, poff.data(), poff.size(), hipMemcpyHostToDevice));
    CK(hipMemset(dY, 0, X.size() * 4));
    const int grid = 8 * nbx;
    auto launch = [&]() {
#define L(WW, KK) hipLaunchKernelGGL((k_tiled<WW, KK>), dim3(grid), dim3(256), 0, 0, dcol, dw, dpoff, dX, dY, N, D, P, ellw)
      if (W == 32 && K == 8) L(32, 8); else if (W == 32 && K == 16) L(32, 16); else if (W == 32 && K == 24) L(32, 24);
      else if (W == 64 && K == 8) L(64, 8); else if (W == 64 && K == 16) L(64, 16);
      else if (W == 32 && K == 4) L(32, 4); else { printf("no such variant\n"); exit(1); }
    };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(Yb.data(), dY, Yb.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int t = 0; t < 400; ++t) {
      const int i = (int)(((size_t)t * 7919) % N), c = (t * 13) % D;
      double ref = 1.5 * X[(size_t)i * D + c];
      for (int e = 0; e < deg[i]; ++e) ref -= (double)wg[(size_t)i * ellw + e] * X[(size_t)col[(size_t)i * ellw + e] * D + c];
      maxerr = std::fmax(maxerr, std::fabs(ref - Yb[(size_t)i * D + c]));
    }
    printf("W=%3d K=%2d P=%2d grid=%4d : %.3f ms per apply (maxerr %.2e)\n", W, K, P, grid, ms / reps, maxerr);
    fflush(stdout);
  };
  if (argc > 6) { run(atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6])); return 0; }
  for (int nbx : {32, 64, 128})
    for (int W : {32, 64})
      for (int K : {8, 16})
        for (int P : {1, 4, 8, 12}) run(W, K, P, nbx);
  return 0;
}
