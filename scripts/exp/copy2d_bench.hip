// hipMemcpy2DAsync (device to device, pitched) against a plain copy kernel for the column window of a sharded state:
// N rows x w floats out of a pitch of ld floats.   hipcc --offload-arch=gfx950 -O3 copy2d_bench.hip -o copy2d_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_copy(float* dst, const float* src, int N, int ld, int c0, int w4) {
  const long n = (long)N * w4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / w4; const int c = (int)(i - r * w4);
    reinterpret_cast<float4*>(dst + r * ld + c0)[c] = reinterpret_cast<const float4*>(src + r * ld + c0)[c];
  }
}
int main() {
  const int N = 100000, ld = 768;
  float *a, *b;
  CK(hipMalloc(&a, (size_t)N * ld * 4)); CK(hipMalloc(&b, (size_t)N * ld * 4));
  CK(hipMemset(a, 0, (size_t)N * ld * 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int w : {96, 192, 384, 768}) {
    for (int mode = 0; mode < 2; ++mode) {
      double best = 1e9;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 10; ++i) {
          if (mode == 0) CK(hipMemcpy2DAsync(b, (size_t)ld * 4, a, (size_t)ld * 4, (size_t)w * 4, N, hipMemcpyDeviceToDevice, s));
          else hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, s, b, a, N, ld, 0, w / 4);
        }
        CK(hipStreamSynchronize(s));
        best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 10);
      }
      printf("w=%d %s: %.1f us per copy (%.2f TB/s)\n", w, mode ? "kernel" : "hipMemcpy2DAsync", best, 2.0 * N * w * 4 / best * 1e-6);
    }
  }
  return 0;
}
