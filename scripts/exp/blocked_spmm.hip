// Experiment (not part of the product): is a slab-major ("blocked") state layout [D/W][N][W] with one column slab per
// XCD faster for the CG operator apply than the row-major 128-column slabs?  The gathered slab (N*W*4 bytes) then fits
// the XCD's 4 MB L2 for W = 8, so gathers stop crossing the fabric; the price is one index sweep per slab and short
// (W*4-byte) gather pieces.  Build: hipcc -O3 --offload-arch=gfx950 blocked_spmm.hip -o blocked_spmm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef float v4 __attribute__((ext_vector_type(4)));

template <int W, bool NT, int LAYOUT>
__global__ __launch_bounds__(256) void k_blocked(const int* __restrict__ colT, const float* __restrict__ wT,
                                                 const int* __restrict__ deg, const float* __restrict__ X,
                                                 float* __restrict__ Y, int N, int nslab, int ellw, int ldx) {
  constexpr int LPR = W / 4, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, lr = lane % LPR;
  const int xcd = LAYOUT == 2 ? 0 : (blockIdx.x & 7), jb = LAYOUT == 2 ? blockIdx.x : (blockIdx.x >> 3),
            nb = LAYOUT == 2 ? gridDim.x : (gridDim.x >> 3);
  const size_t RS = LAYOUT == 0 ? W : (size_t)ldx;  // row stride
  for (int s = xcd; s < nslab; s += (LAYOUT == 2 ? 1 : 8)) {
    const float* Xs = LAYOUT == 0 ? X + (size_t)s * N * W : X + (size_t)s * W;
    float* Ys = LAYOUT == 0 ? Y + (size_t)s * N * W : Y + (size_t)s * W;
    for (int rb = (jb * 4 + wave) * RPW; rb < N; rb += nb * 4 * RPW) {
      const int row = rb + sub;
      if (row >= N) continue;
      const int d = deg[row];
      const v4 xs = *(const v4*)(Xs + (size_t)row * RS + lr * 4);
      v4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int e = 0; e < d; e += 4) {
        int c[4];
        float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const size_t o = (size_t)(e + u) * N + row;
          if (NT) { c[u] = __builtin_nontemporal_load(colT + o); w[u] = __builtin_nontemporal_load(wT + o); }
          else { c[u] = colT[o]; w[u] = wT[o]; }
        }
        v4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const v4*)(Xs + (size_t)c[u] * RS + lr * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += w[u] * v[u];
      }
      v4 out = 1.5f * xs - acc;
      __builtin_nontemporal_store(out, (v4*)(Ys + (size_t)row * RS + lr * 4));
    }
  }
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 100000, D = argc > 2 ? atoi(argv[2]) : 768, ellw = 32;
  const int nbx = argc > 3 ? atoi(argv[3]) : 256;
  const int LD = argc > 4 ? atoi(argv[4]) : D;
  std::mt19937 rng(1);
  std::vector<int> colT((size_t)ellw * N), deg(N);
  std::vector<float> wT((size_t)ellw * N);
  for (int i = 0; i < N; ++i) {
    deg[i] = 20 + rng() % 13;  // 20..32, mean 26
    for (int e = 0; e < ellw; ++e) {
      const bool live = e < deg[i];
      colT[(size_t)e * N + i] = live ? (int)(rng() % N) : i;
      wT[(size_t)e * N + i] = live ? 0.03f * (1 + (rng() % 7)) : 0.f;
    }
  }
  std::vector<float> X((size_t)N * LD);
  for (auto& v : X) v = (float)((int)(rng() % 2001) - 1000) * 1e-3f;
  int *dcol, *ddeg; float *dw, *dX, *dY;
  CK(hipMalloc(&dcol, colT.size() * 4)); CK(hipMalloc(&dw, wT.size() * 4)); CK(hipMalloc(&ddeg, N * 4));
  CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dY, X.size() * 4));
  CK(hipMemcpy(dcol, colT.data(), colT.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, wT.data(), wT.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(ddeg, deg.data(), N * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> Xb(X.size()), Yb(X.size());
  auto run = [&](int W, bool nt, int layout) {
    const int nslab = D / W;
    for (int s = 0; s < nslab; ++s)
      for (int i = 0; i < N; ++i)
        for (int c = 0; c < W; ++c) Xb[layout == 0 ? ((size_t)s * N + i) * W + c : (size_t)i * LD + s * W + c] = X[(size_t)i * LD + s * W + c];
    CK(hipMemcpy(dX, Xb.data(), Xb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dY, 0, Xb.size() * 4));
    const int grid = 8 * nbx;
    auto launch = [&]() {
#define L(WW, NTT, LL) hipLaunchKernelGGL((k_blocked<WW, NTT, LL>), dim3(grid), dim3(256), 0, 0, dcol, dw, ddeg, dX, dY, N, nslab, ellw, LD)
#define LW(WW) { if (layout == 0) { if (nt) L(WW, true, 0); else L(WW, false, 0); } else if (layout == 1) { if (nt) L(WW, true, 1); else L(WW, false, 1); } else { if (nt) L(WW, true, 2); else L(WW, false, 2); } }
      if (W == 16) LW(16) else if (W == 32) LW(32) else if (W == 64) LW(64) else LW(128)
    };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(Yb.data(), dY, Yb.size() * 4, hipMemcpyDeviceToHost));
    // spot check
    double maxerr = 0;
    for (int t = 0; t < 200; ++t) {
      const int i = (int)(((size_t)t * 7919) % N), c = (t * 13) % D;
      double ref = 1.5 * X[(size_t)i * LD + c];
      for (int e = 0; e < deg[i]; ++e) ref -= (double)wT[(size_t)e * N + i] * X[(size_t)colT[(size_t)e * N + i] * LD + c];
      const int s = c / W;
      maxerr = std::fmax(maxerr, std::fabs(ref - Yb[layout == 0 ? ((size_t)s * N + i) * W + c % W : (size_t)i * LD + c]));
    }
    printf("layout=%d W=%3d nt=%d grid=%d : %.3f ms per apply (maxerr %.2e)\n", layout, W, (int)nt, grid, ms / reps, maxerr);
    fflush(stdout);
  };
  printf("LD=%d\n", LD);
  for (int layout : {1, 2}) for (int W : {32, 64, 128}) run(W, false, layout);
  return 0;
}
