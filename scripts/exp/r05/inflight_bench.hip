// Round 5: how many 128-byte lines does a CU keep in flight?  Random 128-byte row gathers (8 lanes per row, no index
// loads, XCD-affine regions like the operator apply), swept over waves per CU x loads in flight per wave, from an
// L2-resident region, an Infinity-Cache-resident one and HBM.  If the rate stops rising with the lines a CU's waves
// WANT in flight, the CU's vector-memory path has reached its own limit of lines in flight (rate x latency), and a
// kernel cannot buy throughput with deeper software pipelines -- the question behind VERDICT r04 item 1.
//   hipcc --offload-arch=gfx950 -O3 -o inflight_bench inflight_bench.hip && ./inflight_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U>
__global__ void k_gather(const float* base, unsigned rows, int iters, float* out, unsigned long long* clk) {
  const int lane = threadIdx.x & 63, sub = lane >> 3, lr = lane & 7;
  const float* region = base + (size_t)(blockIdx.x & 7) * rows * 32 + lr * 4;
  unsigned st = (blockIdx.x * 1024u + (threadIdx.x >> 6) * 64u + sub) * 2654435761u + 12345u;
  float4 acc = make_float4(0, 0, 0, 0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      st = st * 1664525u + 1013904223u;
      const unsigned row = (unsigned)(((unsigned long long)st * rows) >> 32);
      v[u] = *reinterpret_cast<const float4*>(region + (size_t)row * 32);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 8) clk[0] = t1 - t0;  // shader cycles of one wave's loop
  if (acc.x == 123.456f) out[0] = acc.y + acc.z + acc.w;
}

template <int U>
void run(const float* buf, unsigned rows, int waves_per_cu, float* out, unsigned long long* clk) {
  // waves_per_cu < 4: one block of 64 * w threads per CU; otherwise 256-thread blocks
  const int threads = waves_per_cu < 4 ? 64 * waves_per_cu : 256;
  const int grid = waves_per_cu < 4 ? 256 : 256 * (waves_per_cu / 4);
  const int iters = 24000 / (U * (waves_per_cu > 8 ? 2 : 1));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k_gather<U>), dim3(grid), dim3(threads), 0, 0, buf, rows, iters / 4, out, clk);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k_gather<U>), dim3(grid), dim3(threads), 0, 0, buf, rows, iters, out, clk);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  unsigned long long c = 0;
  CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
  const double rows_total = (double)grid * (threads / 64) * 8 * (double)iters * U;
  const double ghz = (double)c / (ms * 1e6);  // the wave's loop spans ~ the launch
  const double clk_per_row_cu = (double)c * 256 / rows_total;
  const int want = waves_per_cu * U * 8;
  // Little: latency = lines in flight / rate, while the wanted lines are really in flight
  printf("  %2d waves/CU x U=%d = %4d lines wanted: %7.3f ms  %6.2f TB/s  %.2f clk/row/CU (clock %.2f GHz)  wanted/rate = %5.0f clk\n",
         waves_per_cu, U, want, ms, rows_total * 128 / ms * 1e-9, clk_per_row_cu, ghz, want * clk_per_row_cu);
}

int main() {
  const size_t cap = (size_t)2 << 30;
  float *buf, *out;
  unsigned long long* clk;
  CK(hipMalloc(&buf, cap)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&clk, 8));
  CK(hipMemset(buf, 0, cap));
  const double mbs[] = {3.2, 12.8, 200.0};
  for (double mb : mbs) {
    printf("footprint per XCD %.1f MB (x8 regions)\n", mb);
    const unsigned rows = (unsigned)(mb * 1024 * 1024 / 128);
    for (int w : {1, 2, 4, 8, 16, 32}) {
      run<1>(buf, rows, w, out, clk);
      run<2>(buf, rows, w, out, clk);
      run<4>(buf, rows, w, out, clk);
      run<8>(buf, rows, w, out, clk);
      if (w <= 4) run<16>(buf, rows, w, out, clk);
    }
  }
  return 0;
}
