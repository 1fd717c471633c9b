// Breadth-first row order of the lattice graph ON THE DEVICE, identical to the host's queue BFS (osc_api.hip: bfs_order):
// components in the order of their smallest row id, each walked breadth-first from that row, a node's neighbours taken
// in ELL slot order.  The host version needs the ELL on the host (12.8 MB at N = 100k, k = 32) and 3-4 ms of pointer
// chasing; it was 6-8 ms of a clustered lattice's 21-27 ms creation.
//
// A queue BFS appends a node when the EARLIEST node of the queue that has it as a neighbour reaches it, at that node's
// slot.  Level by level that is: the next frontier holds the unvisited neighbours of the current one, ordered by
// (position of the claiming parent in the current frontier, slot) where the claiming parent is the one with the smallest
// such pair -- an atomicMin on a key, a count of claimed children per parent, a prefix sum, a placement; no sort per
// level.  All components advance together: the level-0 frontier is the component roots in ascending order (connected
// components by min-label propagation with pointer jumping), so every frontier stays grouped by component in root order,
// and the final order is ONE radix sort of (root, level, position in the level's frontier).
#include <hipcub/hipcub.hpp>

#include "perm.hpp"

namespace osc {
namespace {

constexpr int32_t kUnset = 0x7fffffff;

__global__ void k_cc_init(int32_t* label, int32_t N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) label[i] = i;
}
// label[v] <- min over v and its neighbours; then two pointer jumps (label[v] is always a row of v's component)
__global__ void k_cc_round(const int32_t* col, const int32_t* deg, int32_t width, int32_t N, int32_t* label, int32_t* changed) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= N) return;
  const int32_t old = label[v];
  int32_t m = old;
  const int32_t* c = col + (size_t)v * width;
  for (int e = 0; e < deg[v]; ++e) m = min(m, label[c[e]]);
  m = min(m, label[m]);
  m = min(m, label[m]);
  if (m < old) {
    atomicMin(&label[v], m);
    *changed = 1;
  }
}
__global__ void k_root_flags(const int32_t* label, int32_t N, int32_t* flag) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < N) flag[v] = label[v] == v ? 1 : 0;
}
// level 0: the roots in ascending order
__global__ void k_place_roots(const int32_t* flag, const int32_t* scan, int32_t N, int32_t* frontier, int32_t* lvl, int32_t* posl,
                              int32_t* key, int32_t* cnt) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= N) return;
  key[v] = kUnset;
  if (flag[v]) {
    frontier[scan[v]] = v;
    lvl[v] = 0;
    posl[v] = scan[v];
  } else {
    lvl[v] = -1;
    posl[v] = 0;
  }
  if (v == N - 1) cnt[0] = scan[v] + flag[v];
}
// frontier node p claims its unvisited neighbours: key = smallest (parent position, slot)
__global__ void k_claim(const int32_t* col, const int32_t* deg, int32_t width, const int32_t* frontier, const int32_t* cnt_l,
                        const int32_t* lvl, int32_t* key) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= *cnt_l) return;
  const int u = frontier[p];
  const int32_t* c = col + (size_t)u * width;
  for (int e = 0; e < deg[u]; ++e) {
    const int v = c[e];
    if (lvl[v] < 0) atomicMin(&key[v], p * width + e);
  }
}
// children each frontier node won (0 beyond the frontier: the prefix sum runs over N entries)
__global__ void k_children(const int32_t* col, const int32_t* deg, int32_t width, const int32_t* frontier, const int32_t* cnt_l,
                           const int32_t* lvl, const int32_t* key, int32_t N, int32_t* nchild) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  int n = 0;
  if (p < *cnt_l) {
    const int u = frontier[p];
    const int32_t* c = col + (size_t)u * width;
    for (int e = 0; e < deg[u]; ++e) {
      const int v = c[e];
      n += (lvl[v] < 0 && key[v] == p * width + e) ? 1 : 0;
    }
  }
  nchild[p] = n;
}
// the next frontier: a parent's children in slot order behind the children of the parents before it
__global__ void k_place(const int32_t* col, const int32_t* deg, int32_t width, const int32_t* frontier, const int32_t* cnt_l,
                        const int32_t* nchild, const int32_t* base, int32_t N, int32_t level, int32_t* lvl, int32_t* posl,
                        const int32_t* key, int32_t* next, int32_t* cnt_next) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  if (p == N - 1) *cnt_next = base[p] + nchild[p];
  if (p >= *cnt_l || nchild[p] == 0) return;
  const int u = frontier[p];
  const int32_t* c = col + (size_t)u * width;
  int q = base[p];
  for (int e = 0; e < deg[u]; ++e) {
    const int v = c[e];
    if (lvl[v] < 0 && key[v] == p * width + e) {  // (lvl[v] is set below by this thread only: v has ONE claiming pair)
      next[q] = v;
      posl[v] = q;
      lvl[v] = level + 1;
      ++q;
    }
  }
}
__global__ void k_sort_keys(const int32_t* label, const int32_t* lvl, const int32_t* posl, int32_t N, unsigned long long* keys,
                            int32_t* vals) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= N) return;
  keys[v] = ((unsigned long long)(unsigned)label[v] << 39) | ((unsigned long long)(unsigned)lvl[v] << 25) | (unsigned long long)(unsigned)posl[v];
  vals[v] = v;
}

}  // namespace

// perm_out[new] = old (device, N entries).  Returns false when the graph is too deep for the key layout (> 16383 levels:
// a chain-like graph) or N >= 2^25 -- the caller then walks it on the host.
bool device_bfs_order(const int32_t* col, const int32_t* deg, int32_t width, int32_t N, int32_t* perm_out, hipStream_t s) {
  if (N < 1 || N >= (1 << 25) || (int64_t)N * width >= ((int64_t)1 << 31)) return false;
  const unsigned nblk = (unsigned)((N + 255) / 256);
  DevBuf<int32_t> label, lvl, posl, key, fa, fb, nchild, base, cnt, vals_in;
  DevBuf<unsigned long long> keys_in, keys_out;
  constexpr int kMaxLevels = 16383;
  constexpr int kCheck = 8;  // levels between two looks at the frontier size
  label.alloc((size_t)N);
  lvl.alloc((size_t)N);
  posl.alloc((size_t)N);
  key.alloc((size_t)N);
  fa.alloc((size_t)N);
  fb.alloc((size_t)N);
  nchild.alloc((size_t)N);
  base.alloc((size_t)N);
  cnt.alloc((size_t)kMaxLevels + kCheck + 2);
  size_t tb_scan = 0, tb_sort = 0;
  HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb_scan, nchild.p, base.p, N, s));
  keys_in.alloc((size_t)N);
  keys_out.alloc((size_t)N);
  vals_in.alloc((size_t)N);
  HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb_sort, keys_in.p, keys_out.p, vals_in.p, perm_out, N, 0, 64, s));
  DevBuf<char> tmp;
  tmp.alloc(std::max(tb_scan, tb_sort) + 16);

  // ---- connected components: min-label propagation -----------------------------------------------------------------
  hipLaunchKernelGGL(k_cc_init, dim3(nblk), dim3(256), 0, s, label.p, N);
  int32_t* changed = cnt.p;  // (slot 0 is free until the levels start)
  for (int round = 0;; round += 4) {
    HIP_CHECK(hipMemsetAsync(changed, 0, 4, s));
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(k_cc_round, dim3(nblk), dim3(256), 0, s, col, deg, width, N, label.p, changed);
    int32_t hc = 0;
    HIP_CHECK(hipMemcpyAsync(&hc, changed, 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (!hc) break;
    if (round > 4 * kMaxLevels) return false;
  }
  // ---- level 0: the roots, ascending ---------------------------------------------------------------------------------
  hipLaunchKernelGGL(k_root_flags, dim3(nblk), dim3(256), 0, s, label.p, N, nchild.p);
  HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb_scan, nchild.p, base.p, N, s));
  hipLaunchKernelGGL(k_place_roots, dim3(nblk), dim3(256), 0, s, nchild.p, base.p, N, fa.p, lvl.p, posl.p, key.p, cnt.p);
  // ---- the levels ----------------------------------------------------------------------------------------------------
  int32_t* cur = fa.p;
  int32_t* nxt = fb.p;
  bool done = false;
  for (int l = 0; l < kMaxLevels && !done; l += kCheck) {
    for (int i = 0; i < kCheck; ++i) {
      const int lev = l + i;
      hipLaunchKernelGGL(k_claim, dim3(nblk), dim3(256), 0, s, col, deg, width, cur, cnt.p + lev, lvl.p, key.p);
      hipLaunchKernelGGL(k_children, dim3(nblk), dim3(256), 0, s, col, deg, width, cur, cnt.p + lev, lvl.p, key.p, N, nchild.p);
      HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb_scan, nchild.p, base.p, N, s));
      hipLaunchKernelGGL(k_place, dim3(nblk), dim3(256), 0, s, col, deg, width, cur, cnt.p + lev, nchild.p, base.p, N, lev, lvl.p,
                         posl.p, key.p, nxt, cnt.p + lev + 1);
      std::swap(cur, nxt);
    }
    int32_t hn = 0;
    HIP_CHECK(hipMemcpyAsync(&hn, cnt.p + l + kCheck, 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    done = hn == 0;
  }
  if (!done) return false;
  // ---- one sort by (root, level, position in the level's frontier) -------------------------------------------------------
  hipLaunchKernelGGL(k_sort_keys, dim3(nblk), dim3(256), 0, s, label.p, lvl.p, posl.p, N, keys_in.p, vals_in.p);
  HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb_sort, keys_in.p, keys_out.p, vals_in.p, perm_out, N, 0, 64, s));
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(s));  // the temporaries go back to the pool at scope exit
  return true;
}

}  // namespace osc
