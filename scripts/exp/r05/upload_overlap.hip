// Round 5, request path: can a lattice build start on the rows that have arrived while the rest of Y is still on the bus?
// Pageable host array of N x D floats (what a caller's NumPy array is) -> device, (a) in one hipMemcpyAsync, (b) in row chunks
// on a copy stream with an event per chunk, (c) as (b) with a compute stream that waits for each chunk's event and then runs
// a kernel of `work_us` microseconds (a stand-in for the per-chunk share of the build).  Prints wall times: (c) close to
// max(upload, compute) + one chunk means the copy engine and the kernels overlap, close to the sum means they do not.
// build: hipcc --offload-arch=gfx950 -O3 -o upload_overlap upload_overlap.hip ; usage: upload_overlap [N D chunks work_us]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      std::exit(1);                                                                \
    }                                                                              \
  } while (0)

__global__ void k_spin(float* out, long long cycles) {
  const long long t0 = wall_clock64();
  float v = (float)threadIdx.x;
  while (wall_clock64() - t0 < cycles) v = v * 1.0001f + 0.5f;
  if (v == 12345.678f) out[0] = v;
}
// reads the chunk (so the kernel really depends on the data having arrived)
__global__ void k_touch(const float* p, size_t n, float* out) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s == 12345.678f) out[0] = s;
}

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  const long long N = argc > 1 ? std::atoll(argv[1]) : 100000, D = argc > 2 ? std::atoll(argv[2]) : 768;
  const int chunks = argc > 3 ? std::atoi(argv[3]) : 8;
  const double work_us = argc > 4 ? std::atof(argv[4]) : 1000.0;
  const size_t bytes = (size_t)N * D * 4;
  float* host = static_cast<float*>(std::malloc(bytes));
  for (size_t i = 0; i < (size_t)N * D; ++i) host[i] = (float)(i & 1023) * 0.001f;
  float *dev = nullptr, *out = nullptr;
  CK(hipMalloc(&dev, bytes));
  CK(hipMalloc(&out, 64));
  hipStream_t up, comp;
  CK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&comp, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev((size_t)chunks);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  int clock_khz = 0;
  CK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeWallClockRate, 0));
  const long long spin = (long long)(work_us * 1e-6 * clock_khz * 1e3);
  auto rows_of = [&](int c) { return N * (c + 1) / chunks - N * c / chunks; };
  for (int rep = 0; rep < 4; ++rep) {
    double t0 = now_ms();
    CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, up));
    CK(hipStreamSynchronize(up));
    const double one = now_ms() - t0;
    t0 = now_ms();
    for (int c = 0; c < chunks; ++c) {
      const size_t off = (size_t)(N * c / chunks) * D;
      CK(hipMemcpyAsync(dev + off, host + off, (size_t)rows_of(c) * D * 4, hipMemcpyHostToDevice, up));
      CK(hipEventRecord(ev[(size_t)c], up));
    }
    CK(hipStreamSynchronize(up));
    const double chunked = now_ms() - t0;
    t0 = now_ms();
    for (int c = 0; c < chunks; ++c) k_spin<<<256, 256, 0, comp>>>(out, spin);
    CK(hipStreamSynchronize(comp));
    const double comp_only = now_ms() - t0;
    t0 = now_ms();
    for (int c = 0; c < chunks; ++c) {
      const size_t off = (size_t)(N * c / chunks) * D;
      CK(hipMemcpyAsync(dev + off, host + off, (size_t)rows_of(c) * D * 4, hipMemcpyHostToDevice, up));
      CK(hipEventRecord(ev[(size_t)c], up));
      CK(hipStreamWaitEvent(comp, ev[(size_t)c], 0));
      k_touch<<<1024, 256, 0, comp>>>(dev + off, (size_t)rows_of(c) * D, out);
      k_spin<<<256, 256, 0, comp>>>(out, spin);
    }
    CK(hipStreamSynchronize(comp));
    CK(hipStreamSynchronize(up));
    const double both = now_ms() - t0;
    std::printf("N=%lld D=%lld (%.0f MB) chunks=%d: one copy %.2f ms (%.1f GB/s), chunked %.2f ms, compute alone %.2f ms, chunked + compute %.2f ms\n",
                N, D, bytes / 1048576.0, chunks, one, bytes / one * 1e-6, chunked, comp_only, both);
  }
  return 0;
}
