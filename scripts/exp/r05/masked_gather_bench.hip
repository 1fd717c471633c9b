// Round 5: does a vector-memory instruction with inactive lanes cost the CU's memory path less?  Random 128-byte row gathers
// as in inflight_bench.hip (8 lanes per row, 16 waves per CU, 4 loads in flight per wave), with only the first `active` of a
// wave's 8 lane groups taking part in each load (EXEC-masked).  If the time per wave-instruction falls with the active lane
// groups, a per-row variable number of gather rounds (masked lanes instead of padding slots) costs the memory path only
// the real edges.
//   hipcc --offload-arch=gfx950 -O3 -o masked_gather_bench masked_gather_bench.hip && ./masked_gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U>
__global__ __launch_bounds__(256) void k_gather(const float* base, unsigned rows, int iters, int active, float* out) {
  const int lane = threadIdx.x & 63, sub = lane >> 3, lr = lane & 7;
  const float* region = base + (size_t)(blockIdx.x & 7) * rows * 32 + lr * 4;
  unsigned st = (blockIdx.x * 1024u + (threadIdx.x >> 6) * 64u + sub) * 2654435761u + 12345u;
  float4 acc = make_float4(0, 0, 0, 0);
  const bool on = sub < active;
  for (int it = 0; it < iters; ++it) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      st = st * 1664525u + 1013904223u;
      const unsigned row = (unsigned)(((unsigned long long)st * rows) >> 32);
      v[u] = make_float4(0, 0, 0, 0);
      if (on) v[u] = *reinterpret_cast<const float4*>(region + (size_t)row * 32);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x == 123.456f) out[0] = acc.y + acc.z + acc.w;
}

int main() {
  const size_t cap = (size_t)2 << 30;
  float *buf, *out;
  CK(hipMalloc(&buf, cap)); CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 0, cap));
  for (double mb : {3.2, 12.8}) {
    const unsigned rows = (unsigned)(mb * 1024 * 1024 / 128);
    printf("footprint per XCD %.1f MB\n", mb);
    for (int active : {8, 6, 4, 2, 1}) {
      const int grid = 256 * 4, iters = 6000;
      hipEvent_t a, b;
      CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      hipLaunchKernelGGL((k_gather<4>), dim3(grid), dim3(256), 0, 0, buf, rows, iters / 4, active, out);
      CK(hipEventRecord(a));
      hipLaunchKernelGGL((k_gather<4>), dim3(grid), dim3(256), 0, 0, buf, rows, iters, active, out);
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      const double insts = (double)grid * 4 * (double)iters * 4;          // wave-instructions
      const double rows_total = insts * active;
      printf("  %d of 8 lane groups active: %7.3f ms  %.2f G wave-instructions/s  %6.2f TB/s of rows  (%.1f ns per instruction and CU)\n",
             active, ms, insts / ms * 1e-6, rows_total * 128 / ms * 1e-9, ms * 1e6 / (insts / 256));
    }
  }
  return 0;
}
