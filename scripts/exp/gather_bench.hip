// Microbenchmark behind DESIGN.md section 3: how fast can a CU gather random rows, as a function of (a) the footprint the
// rows come from (inside an XCD's 4 MB L2 / inside the Infinity Cache / HBM), (b) bytes per row (32 / 64 / 128),
// (c) loads in flight per wave and waves per CU.  XCD-affine like the operator apply: workgroup b gathers from region
// b % 8.  Row ids come from an LCG (no index loads), so this is the ceiling of the gather itself.
//   hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip && ./gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int LPR, int U>
__global__ __launch_bounds__(256) void k_gather(const float* base, unsigned rows, int iters, float* out) {
  constexpr int ROWF = LPR * 4;
  const int lane = threadIdx.x & 63, sub = lane / LPR, lr = lane % LPR;
  const float* region = base + (size_t)(blockIdx.x & 7) * rows * ROWF + lr * 4;
  unsigned st = (blockIdx.x * 256u + (threadIdx.x >> 6) * 64u + sub) * 2654435761u + 12345u;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      st = st * 1664525u + 1013904223u;
      const unsigned row = (unsigned)(((unsigned long long)st * rows) >> 32);
      v[u] = *reinterpret_cast<const float4*>(region + (size_t)row * ROWF);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x == 123.456f) out[0] = acc.y + acc.z + acc.w;
}

// gathers from an L2-sized region with a streaming read beside them (one 128-byte line per `every` gathered rows, read
// once, from a 1 GB array): what the edge lists / outputs of a blocked apply do to the resident block
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_gather_stream(const float* base, unsigned rows, int iters, int every,
                                                       const float* stream, float* out) {
  const int lane = threadIdx.x & 63, sub = lane / 8, lr = lane % 8;
  const float* region = base + (size_t)(blockIdx.x & 7) * rows * 32 + lr * 4;
  unsigned st = (blockIdx.x * 256u + (threadIdx.x >> 6) * 64u + sub) * 2654435761u + 12345u;
  float4 acc = make_float4(0, 0, 0, 0);
  // this wave's streaming cursor: consecutive 1 KB pieces (8 lines per wave-load), waves interleaved
  size_t cur = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 256 + lane * 4;
  const size_t stride = (size_t)gridDim.x * 4 * 256;
  for (int it = 0; it < iters; ++it) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      st = st * 1664525u + 1013904223u;
      const unsigned row = (unsigned)(((unsigned long long)st * rows) >> 32);
      v[u] = *reinterpret_cast<const float4*>(region + (size_t)row * 32);
    }
    if ((it * U) % every < U) {  // 8 stream lines per wave per `every` wave-steps of 8 rows: 1 line per `every` rows
      using v4f = __attribute__((ext_vector_type(4))) float;
      const v4f* p = reinterpret_cast<const v4f*>(stream + (cur & (((size_t)1 << 28) - 1)));
      const v4f t = NT ? __builtin_nontemporal_load(p) : *p;
      acc.x += t.x + t.y + t.z + t.w;
      cur += stride;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x == 123.456f) out[0] = acc.y + acc.z + acc.w;
}

template <bool NT>
void run_stream(const float* buf, unsigned rows, int every, const float* stream, float* out) {
  const int grid = 1024, iters = 2000;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k_gather_stream<4, NT>), dim3(grid), dim3(256), 0, 0, buf, rows, iters / 4, every, stream, out);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k_gather_stream<4, NT>), dim3(grid), dim3(256), 0, 0, buf, rows, iters, every, stream, out);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double gathers = (double)grid * 4 * 8 * (double)iters * 4;
  printf("  +stream 1 line per %2d rows%s: %7.3f ms  %.1f clk/row\n", every, NT ? " (nontemporal)" : "", ms,
         (ms * 1e-3 * 2.4e9 * 256) / gathers);
}

template <int LPR, int U>
double run(const float* buf, unsigned rows, int wg_per_cu, int iters, float* out) {
  const int grid = 256 * wg_per_cu;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k_gather<LPR, U>), dim3(grid), dim3(256), 0, 0, buf, rows, iters / 4, out);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k_gather<LPR, U>), dim3(grid), dim3(256), 0, 0, buf, rows, iters, out);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double gathers = (double)grid * 4 * (64 / LPR) * (double)iters * U;
  const double lines_per_clk_cu = gathers / (ms * 1e-3 * 2.4e9 * 256);
  printf("  %3d B/row  U=%d  %2d waves/CU: %7.3f ms  %6.2f G rows/s  %7.1f GB/s  %.3f rows/clk/CU (%.1f clk/row)\n", LPR * 16, U,
         wg_per_cu * 4, ms, gathers / ms * 1e-6, gathers * LPR * 16 / ms * 1e-6, lines_per_clk_cu, 1.0 / lines_per_clk_cu);
  return ms;
}

int main() {
  const size_t cap = (size_t)2 << 30;
  float *buf, *out;
  CK(hipMalloc(&buf, cap)); CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 0, cap));
  const double mbs[] = {1.0, 2.0, 2.56, 3.2, 3.6, 6.4, 12.8, 32.0, 200.0};
  for (double mb : mbs) {
    printf("footprint per XCD %.1f MB (x8 regions)\n", mb);
    const size_t bytes = (size_t)(mb * 1024 * 1024);
    for (int wg : {4, 8}) {
      run<8, 2>(buf, bytes / 128, wg, 4000, out);
      run<8, 4>(buf, bytes / 128, wg, 2000, out);
      run<8, 8>(buf, bytes / 128, wg, 1000, out);
    }
    if (mb < 5.0) {
      const float* stream = buf + ((size_t)1 << 28);  // second GB of the buffer
      for (int every : {16, 8, 4}) {
        run_stream<false>(buf, bytes / 128, every, stream, out);
        run_stream<true>(buf, bytes / 128, every, stream, out);
      }
    }
    run<4, 4>(buf, bytes / 64, 4, 2000, out);
    run<4, 8>(buf, bytes / 64, 8, 1000, out);
    run<2, 4>(buf, bytes / 32, 4, 2000, out);
    run<2, 8>(buf, bytes / 32, 8, 1000, out);
    run<16, 4>(buf, bytes / 256, 4, 2000, out);
    run<64, 4>(buf, bytes / 1024, 4, 2000, out);
  }
  return 0;
}
