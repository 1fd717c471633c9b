// Experiment (not part of the product): panel-synchronised operator apply with register-resident accumulators.
// XCD x owns the 32-column slabs x, x+8, ... (slab-major operand [D/32][N][32]).  For one slab the rows are cut into
// `rounds` equal parts; in a round every lane group (8 lanes = one 128-byte line per row) of every workgroup of the XCD
// keeps the accumulators of up to R rows in registers, and ALL workgroups of the XCD walk the source rows panel by
// panel (P panels of N/P rows = N/P x 128 B, sized to sit in the XCD's 4 MB L2) behind a soft per-XCD barrier, so at
// any moment the whole XCD gathers from one L2-resident panel.  Edge lists are the product's column-sorted ELL plus
// one byte per (row, panel) boundary; no per-panel copies of the graph.
// Build: hipcc -O3 --offload-arch=gfx950 resident_spmm.hip -o resident_spmm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cmath>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

// R: row slots per lane group; RB: rows whose edges are in flight together; WPE: waves per SIMD the kernel is built for.
// Slots: the rows in the order of their per-panel edge counts (n_0, n_1, ...), so the eight rows of a wave step and the RB
// rows of a batch have (nearly) the same count in every panel and the ragged tail of the edge loop is short.  Chunk
// (= 8*RB consecutive slots) c goes to batch t = c / (rounds * nW), round (c / nW) % rounds, wave c % nW of the XCD: all
// waves work on neighbouring chunks at the same time and every wave gets an even sample of the count distribution.
template <int R, int RB, int WPE, int SYNC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void k_res(const int* __restrict__ colS, const float* __restrict__ wS, const u64* __restrict__ poffS, const int* __restrict__ rowS,
           const float* __restrict__ X, float* __restrict__ Y, int N, int D, int P, int ellw, int rounds, int nbatch /* <= R/RB */,
           int nslots, unsigned* bar, int slack) {
  static_assert(R % RB == 0, "R must be a multiple of RB");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane >> 3, lr = lane & 7;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, nb = gridDim.x >> 3;
  const int nslab = D / 32, nW = nb * 4;
  unsigned epoch = 0;
  auto xcd_sync = [&]() {  // soft barrier of the workgroups of this XCD: pacing only, never needed for correctness
    if (SYNC == 0) return;
    __syncthreads();
    if (threadIdx.x == 0) {
      ++epoch;
      __hip_atomic_fetch_add(bar + xcd * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = epoch * (unsigned)nb - (unsigned)slack;
      for (int spin = 0; spin < 4000; ++spin) {
        if (__hip_atomic_load(bar + xcd * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) break;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
  };
  const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc((void*)colS, 0, nslots * ellw * 4 + 64, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wS, 0, nslots * ellw * 4 + 64, 0x00020000);
  const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc((void*)poffS, 0, nslots * 8, 0x00020000);
  for (int s = xcd; s < nslab; s += 8) {
    // the slab as a buffer resource: 32-bit byte offsets, out-of-range offsets (dead edge slots) read as zero
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(X + (size_t)s * N * 32), 0, N * 128, 0x00020000);
    float* Ys = Y + (size_t)s * N * 32;
    for (int rd = 0; rd < rounds; ++rd) {
      v4 acc[R];
#pragma unroll
      for (int kk = 0; kk < R; ++kk) acc[kk] = (v4){0.f, 0.f, 0.f, 0.f};
      for (int p = 0; p < P; ++p) {
        xcd_sync();
#pragma unroll
        for (int kb = 0; kb < R; kb += RB) {
          if (kb / RB >= nbatch) break;
          __builtin_amdgcn_sched_barrier(0);
          int cb = (((kb / RB) * rounds + rd) * nW + jb * 4 + wave) * (8 * RB) + sub;
          asm volatile("" : "+v"(cb));  // keeps the compiler from hoisting every batch's index loads to the top
          int n[RB], cj[RB][2];
          float wj[RB][2];
          int any_n = 0;
          u64 po[RB];
#pragma unroll
          for (int q = 0; q < RB; ++q) po[q] = __builtin_bit_cast(u64, __builtin_amdgcn_raw_buffer_load_b64(pr, (cb + q * 8) * 8, 0, 0));  // past the end: 0
#pragma unroll
          for (int q = 0; q < RB; ++q) {  // unconditional loads (the arrays are padded by 16 entries), masked afterwards
            const int slot = cb + q * 8;
            const int o0 = (int)((po[q] >> (8 * p)) & 255u);
            n[q] = (int)((po[q] >> (8 * p + 8)) & 255u) - o0;
            const int eo = (slot * ellw + o0 + lr) * 4;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              cj[q][b] = __builtin_amdgcn_raw_buffer_load_b32(cr, eo + b * 32, 0, 0);
              wj[q][b] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(wr, eo + b * 32, 0, 0));
            }
          }
#pragma unroll
          for (int q = 0; q < RB; ++q) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              const bool live = b * 8 + lr < n[q];
              cj[q][b] = live ? cj[q][b] * 128 : 0x7fffff00;
              wj[q][b] = live ? wj[q][b] : 0.f;
            }
            any_n = max(any_n, n[q]);
          }
          any_n = min(any_n, 16);
          // edge u of every row of the batch in flight together
          auto fetch = [&](int u, v4 (&v)[RB], float (&wv)[RB]) {
            const int src = ((sub << 3) + (u & 7)) << 2;
#pragma unroll
            for (int q = 0; q < RB; ++q) {
              const int cs = u < 8 ? cj[q][0] : cj[q][1];
              const float ws = u < 8 ? wj[q][0] : wj[q][1];
              const int j = __builtin_amdgcn_ds_bpermute(src, cs) + lr * 16;
              wv[q] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(ws)));
              v[q] = __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(xr, j, 0, 0));
            }
          };
          // two edges of every row of the batch in flight: edge u+1 is issued before edge u is consumed
          v4 vc[RB];
          float wc[RB];
          fetch(0, vc, wc);
          for (int u = 0; __any(u < any_n); ++u) {
            v4 vn[RB];
            float wn[RB];
            fetch(u + 1 < 16 ? u + 1 : 15, vn, wn);  // past the row's end: dead slot (zero weight, out-of-range offset)
#pragma unroll
            for (int q = 0; q < RB; ++q) acc[kb + q] += wc[q] * vc[q];
#pragma unroll
            for (int q = 0; q < RB; ++q) { vc[q] = vn[q]; wc[q] = wn[q]; }
          }
#pragma unroll
          for (int q = 0; q < RB; ++q) {  // rare: more than 16 edges of one row in one panel
            if (n[q] > 16) {
              const int slot = cb + q * 8;
              const u64 po = poffS[slot];
              const unsigned eo = (unsigned)slot * (unsigned)ellw + (unsigned)((po >> (8 * p)) & 255u);
              for (int e = 16; e < n[q]; ++e) {
                const int j = colS[eo + e] * 128 + lr * 16;
                acc[kb + q] += wS[eo + e] * __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(xr, j, 0, 0));
              }
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < R; ++kk) {
        if (kk / RB >= nbatch) break;
        const int slot = (((kk / RB) * rounds + rd) * nW + jb * 4 + wave) * (8 * RB) + sub + (kk % RB) * 8;
        if (slot < nslots) {
          const int row = rowS[slot];
          const unsigned o = (unsigned)row * 32 + lr * 4;
          const v4 xs = __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(xr, o * 4, 0, 0));
          __builtin_nontemporal_store(1.5f * xs - acc[kk], (v4*)(Ys + o));
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
  std::vector<float> X((size_t)N * D), Xb((size_t)N * D), Yb((size_t)N * D);
  for (auto& v : X) v = (float)((int)(rng() % 2001) - 1000) * 1e-3f;
  for (int sl = 0; sl < D / 32; ++sl) for (int i = 0; i < N; ++i) for (int c = 0; c < 32; ++c) Xb[((size_t)sl * N + i) * 32 + c] = X[(size_t)i * D + sl * 32 + c];
  int *dcol, *drow; float *dw, *dX, *dY; u64* dpoff; unsigned* dbar;
  CK(hipMalloc(&dcol, col.size() * 4 + 256)); CK(hipMalloc(&dw, wg.size() * 4 + 256)); CK(hipMalloc(&dpoff, (size_t)N * 8)); CK(hipMalloc(&drow, (size_t)N * 4));
  CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dY, X.size() * 4)); CK(hipMalloc(&dbar, 8 * 32 * 4));
  CK(hipMemcpy(dX, Xb.data(), X.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<u64> poff(N), poffS(N);
  std::vector<int> rowS(N), colS((size_t)ellw * N);
  std::vector<float> wSh((size_t)ellw * N);
  auto run = [&](int R, int P, int nbx, int sync, int sorted, int slack = 0) {
    if (P > 7) { printf("P <= 7\n"); return; }
    const int psz = (N + P - 1) / P;
    for (int i = 0; i < N; ++i) {
      int e = 0;
      u64 pk = 0;
      for (int p = 0; p <= P; ++p) {
        while (e < deg[i] && col[(size_t)i * ellw + e] < p * psz) ++e;
        pk |= (u64)(p == P ? deg[i] : e) << (8 * p);
      }
      poff[i] = pk;
      rowS[i] = i;
    }
    if (sorted) {  // cumulative boundaries compare like the count vectors (n_0, n_0 + n_1, ...) lexicographically
      auto key = [&](int i) { u64 k = 0; for (int p = 1; p <= P; ++p) k = (k << 8) | ((poff[i] >> (8 * p)) & 255u); return k; };
      std::stable_sort(rowS.begin(), rowS.end(), [&](int a2, int b2) { return key(a2) < key(b2); });
    }
    for (int sl = 0; sl < N; ++sl) {
      poffS[sl] = poff[rowS[sl]];
      for (int e = 0; e < ellw; ++e) { colS[(size_t)sl * ellw + e] = col[(size_t)rowS[sl] * ellw + e]; wSh[(size_t)sl * ellw + e] = wg[(size_t)rowS[sl] * ellw + e]; }
    }
    CK(hipMemcpy(dpoff, poffS.data(), poffS.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(drow, rowS.data(), rowS.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcol, colS.data(), colS.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, wSh.data(), wSh.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dY, 0, X.size() * 4));
    const int grid = 8 * nbx;
    int Rt = R % 100, RB = R == 8 ? 4 : R == 108 ? 8 : R == 12 ? 4 : R == 112 ? 6 : R == 16 ? 4 : R == 18 ? 6 : R == 24 ? 4 : R == 124 ? 8 : 4;
    const int rounds = (N + nbx * 32 * Rt - 1) / (nbx * 32 * Rt);
    const int nbatch = ((N + rounds - 1) / rounds + nbx * 32 * RB - 1) / (nbx * 32 * RB);  // chunks per wave per round
    auto launch = [&]() {
      CK(hipMemsetAsync(dbar, 0, 8 * 32 * 4, 0));
#define L(RR, RBB, WW) { if (sync) hipLaunchKernelGGL((k_res<RR, RBB, WW, 1>), dim3(grid), dim3(256), 0, 0, dcol, dw, dpoff, drow, dX, dY, N, D, P, ellw, rounds, nbatch, N, dbar, slack); \
                         else hipLaunchKernelGGL((k_res<RR, RBB, WW, 0>), dim3(grid), dim3(256), 0, 0, dcol, dw, dpoff, drow, dX, dY, N, D, P, ellw, rounds, nbatch, N, dbar, slack); }
      if (R == 8) L(8, 4, 4) else if (R == 108) L(8, 8, 4) else if (R == 12) L(12, 4, 4) else if (R == 112) L(12, 6, 3)
      else if (R == 16) L(16, 4, 3) else if (R == 18) L(18, 6, 3) else if (R == 24) L(24, 4, 2) else if (R == 124) L(24, 8, 2)
      else if (R == 4) L(4, 4, 4) else { printf("no such variant\n"); exit(1); }
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
    for (int t = 0; t < 4000; ++t) {
      const int i = t < 64 ? N - 1 - t : (int)(((size_t)t * 7919) % N), c = (t * 13) % D;
      double ref = 1.5 * X[(size_t)i * D + c];
      for (int e = 0; e < deg[i]; ++e) ref -= (double)wg[(size_t)i * ellw + e] * X[(size_t)col[(size_t)i * ellw + e] * D + c];
      maxerr = std::fmax(maxerr, std::fabs(ref - Yb[((size_t)(c / 32) * N + i) * 32 + c % 32]));
    }
    printf("R=%3d RB=%d P=%d nbx=%3d sync=%d sorted=%d slack=%d rounds=%d nbatch=%d : %.3f ms per apply (maxerr %.2e)\n", R, RB, P, nbx, sync, sorted, slack, rounds, nbatch, ms / reps, maxerr);
    fflush(stdout);
  };
  if (argc > 7) { run(atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7])); return 0; }
  run(8, 4, 128, 0, 1); run(8, 1, 128, 0, 1);
  for (int P : {3, 4}) for (int slack : {0, 8, 16, 32, 64}) {
    run(8, P, 128, 1, 1, slack);
    run(16, P, 96, 1, 1, slack * 3 / 4);
    run(24, P, 64, 1, 1, slack / 2);
  }
  return 0;
}
