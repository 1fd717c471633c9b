// LDS-DMA semantics check (gfx950): global_load_lds_dwordx4 writes lane-linear (16 B per lane) at M0 + offset, from
// vaddr + offset (the immediate applies to BOTH addresses), any 16-byte-aligned base, inactive lanes write nothing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const int* src, int* out, int base_off, int nlanes) {
  __shared__ __attribute__((aligned(16))) int lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = -1;
  __syncthreads();
  const unsigned lds_base = (unsigned)(size_t)&lds[0] + (unsigned)base_off;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base);
  const int lane = threadIdx.x;
  if (lane < nlanes) {
    const char* lsrc = (const char*)src + lane * 16;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lsrc), "s"(dst), "n"(1024) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += 64) out[i] = lds[i];
}
int main() {
  int *src, *out; hipMalloc(&src, 4096 * 4); hipMalloc(&out, 4096 * 4);
  std::vector<int> h(4096); for (int i = 0; i < 4096; ++i) h[i] = i;
  hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  for (int t = 0; t < 3; ++t) {
    int base = t == 0 ? 0 : (t == 1 ? 2688 : 384), nl = t == 2 ? 40 : 64;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out, base, nl);
    hipMemcpy(h.data(), out, 4096 * 4, hipMemcpyDeviceToHost);
    int first = -1, cnt = 0, bad = 0;
    for (int i = 0; i < 4096; ++i) if (h[i] != -1) { if (first < 0) first = i; ++cnt; if (h[i] != i - first + 256) ++bad; }
    if (t == 0) { for (int i = first; i < first + 24; ++i) printf("%d ", h[i]); printf("...\n"); for (int i = first + 64; i < first + 72; ++i) printf("%d ", h[i]); printf("\n"); }
    printf("base %d lanes %d: first written dword %d (expect %d), dwords written %d (expect %d), out-of-sequence %d\n", base, nl, first, (base + 1024) / 4, cnt, nl * 4, bad);
  }
  return 0;
}
