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
// The prefix sums and the sort are this file's own (exclusive_scan_i32, radix_sort_pairs below): no library underneath.
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
// key = root | level | position, packed into as few bits as N and the level count need (fewer sort passes)
__global__ void k_sort_keys(const int32_t* label, const int32_t* lvl, const int32_t* posl, int32_t N, int row_bits, int level_bits,
                            unsigned long long* keys, int32_t* vals) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= N) return;
  keys[v] = ((unsigned long long)(unsigned)label[v] << (row_bits + level_bits)) | ((unsigned long long)(unsigned)lvl[v] << row_bits) |
            (unsigned long long)(unsigned)posl[v];
  vals[v] = v;
}

// ---- exclusive prefix sum of n int32 (three launches: block sums, their scan by one workgroup, the blocks' own scans) ----
constexpr int kScanThreads = 256, kScanItems = 4, kScanTile = kScanThreads * kScanItems;

// exclusive scan of one value per thread over the workgroup (wave shuffles + one LDS hop); *total = the workgroup's sum
template <int NT>
__device__ __forceinline__ int32_t block_exclusive_scan(int32_t v, int32_t* total) {
  __shared__ int32_t wave_sum[NT / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();  // (wave_sum may still be read by a previous call)
  if (lane == 63) wave_sum[wave] = inc;
  __syncthreads();
  int32_t before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    const int32_t t = wave_sum[w];
    before += w < wave ? t : 0;
    all += t;
  }
  *total = all;
  return before + inc - v;
}
__global__ __launch_bounds__(kScanThreads) void k_scan_block_sums(const int32_t* in, int64_t n, int32_t* sums) {
  const int64_t i0 = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int32_t v = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) v += i0 + j < n ? in[i0 + j] : 0;
  int32_t total;
  (void)block_exclusive_scan<kScanThreads>(v, &total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
// one workgroup: sums[0 .. nb) -> their exclusive scan, in place (chunks of 1024 with a running carry)
__global__ __launch_bounds__(1024) void k_scan_sums(int32_t* sums, int32_t nb) {
  int32_t carry = 0;
  for (int32_t c0 = 0; c0 < nb; c0 += 1024) {
    const int32_t i = c0 + (int32_t)threadIdx.x;
    const int32_t v = i < nb ? sums[i] : 0;
    int32_t total;
    const int32_t ex = block_exclusive_scan<1024>(v, &total);
    if (i < nb) sums[i] = carry + ex;
    carry += total;
  }
}
__global__ __launch_bounds__(kScanThreads) void k_scan_apply(const int32_t* in, int64_t n, const int32_t* sums, int32_t* out) {
  const int64_t i0 = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int32_t x[kScanItems], v = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    x[j] = i0 + j < n ? in[i0 + j] : 0;
    v += x[j];
  }
  int32_t total;
  int32_t run = sums[blockIdx.x] + block_exclusive_scan<kScanThreads>(v, &total);
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    if (i0 + j < n) out[i0 + j] = run;
    run += x[j];
  }
}
}  // namespace
size_t scan_blocks(int64_t n) { return (size_t)((n + kScanTile - 1) / kScanTile); }
// out may alias in; sums: scan_blocks(n) int32 of scratch
void exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* sums, hipStream_t s) {
  if (n <= 0) return;
  const unsigned nb = (unsigned)scan_blocks(n);
  hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(kScanThreads), 0, s, in, n, sums);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, sums, (int32_t)nb);
  hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(kScanThreads), 0, s, in, n, sums, out);
  HIP_CHECK(hipGetLastError());
}
namespace {

// ---- stable LSD radix sort of (64-bit key, int32 value) pairs, 8 bits per pass -----------------------------------------
// One wave per tile of 1024 pairs.  Histogram pass: hist[digit][tile]; its exclusive scan gives every (digit, tile) its
// first output slot; the scatter pass walks the tile in order, 64 pairs per step: the lanes holding a step's equal digits
// find each other with eight ballots, take consecutive slots behind the tile's running count of that digit.
constexpr int kSortTile = 1024;
__global__ __launch_bounds__(64) void k_sort_hist(const unsigned long long* keys, int32_t n, int shift, int32_t ntiles, int32_t* hist) {
  __shared__ int32_t h[256];
  for (int d = threadIdx.x; d < 256; d += 64) h[d] = 0;
  __syncthreads();
  const int32_t i0 = (int32_t)blockIdx.x * kSortTile;
  for (int r = 0; r < kSortTile / 64; ++r) {
    const int32_t i = i0 + r * 64 + (int32_t)threadIdx.x;
    if (i < n) atomicAdd(&h[(int)((keys[i] >> shift) & 255ull)], 1);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < 256; d += 64) hist[(size_t)d * ntiles + blockIdx.x] = h[d];
}
__global__ __launch_bounds__(64) void k_sort_scatter(const unsigned long long* keys, const int32_t* vals, int32_t n, int shift,
                                                     int32_t ntiles, const int32_t* first, unsigned long long* keys_out,
                                                     int32_t* vals_out) {
  __shared__ int32_t run[256];  // next free slot of each digit for this tile
  for (int d = threadIdx.x; d < 256; d += 64) run[d] = first[(size_t)d * ntiles + blockIdx.x];
  __syncthreads();
  const int lane = threadIdx.x;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int32_t i0 = (int32_t)blockIdx.x * kSortTile;
  for (int r = 0; r < kSortTile / 64; ++r) {
    const int32_t i = i0 + r * 64 + lane;
    const bool live = i < n;
    const unsigned long long key = live ? keys[i] : 0ull;
    const int32_t val = live ? vals[i] : 0;
    const int d = (int)((key >> shift) & 255ull);
    unsigned long long peers = __ballot(live);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long set = __ballot((d >> b) & 1);
      peers &= ((d >> b) & 1) ? set : ~set;
    }
    const int before = __popcll(peers & lt);
    const int32_t base = live ? run[d] : 0;
    __syncthreads();  // every lane has read the running counts of this step
    if (live) {
      keys_out[base + before] = key;
      vals_out[base + before] = val;
      if (before == 0) run[d] = base + __popcll(peers);
    }
    __syncthreads();
  }
}
// sorts by the key bits [0, key_bits); the sorted values end in vals_out (the keys in keys_a or keys_b: nobody needs them).
// keys_b, vals_tmp: n entries of scratch each;
// hist: 256 * tiles int32 (+ the scan's block sums behind it)
void radix_sort_pairs(unsigned long long* keys_a, unsigned long long* keys_b, int32_t* vals_in, int32_t* vals_tmp, int32_t* vals_out,
                      int32_t n, int key_bits, int32_t* hist, int32_t* sums, hipStream_t s) {
  const int32_t ntiles = (n + kSortTile - 1) / kSortTile;
  const int passes = std::max(1, (key_bits + 7) / 8);
  unsigned long long *kin = keys_a, *kout = keys_b;
  // the values end in vals_out after the LAST pass: an odd number of passes goes in -> out, an even one in -> tmp -> out
  int32_t* vin = vals_in;
  for (int p = 0; p < passes; ++p) {
    int32_t* vout = ((passes - 1 - p) & 1) ? vals_tmp : vals_out;
    hipLaunchKernelGGL(k_sort_hist, dim3(ntiles), dim3(64), 0, s, kin, n, 8 * p, ntiles, hist);
    exclusive_scan_i32(hist, hist, (int64_t)256 * ntiles, sums, s);
    hipLaunchKernelGGL(k_sort_scatter, dim3(ntiles), dim3(64), 0, s, kin, vin, n, 8 * p, ntiles, hist, kout, vout);
    std::swap(kin, kout);
    vin = vout;
  }
}
int bits_for(int64_t count) {  // bits that hold the values 0 .. count - 1
  int b = 1;
  while (((int64_t)1 << b) < count) ++b;
  return b;
}

}  // namespace

// perm_out[new] = old (device, N entries).  Returns false when the graph is too deep for the key layout (> 16383 levels:
// a chain-like graph) or N >= 2^25 -- the caller then walks it on the host.
bool device_bfs_order(const int32_t* col, const int32_t* deg, int32_t width, int32_t N, int32_t* perm_out, hipStream_t s) {
  if (N < 1 || N >= (1 << 25) || (int64_t)N * width >= ((int64_t)1 << 31)) return false;
  const unsigned nblk = (unsigned)((N + 255) / 256);
  DevBuf<int32_t> label, lvl, posl, key, fa, fb, nchild, base, cnt, vals_in, vals_tmp, hist, sums;
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
  keys_in.alloc((size_t)N);
  keys_out.alloc((size_t)N);
  vals_in.alloc((size_t)N);
  vals_tmp.alloc((size_t)N);
  const size_t ntiles = ((size_t)N + kSortTile - 1) / kSortTile;
  hist.alloc(256 * ntiles);
  sums.alloc(std::max(scan_blocks(N), scan_blocks((int64_t)256 * (int64_t)ntiles)) + 1);

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
  exclusive_scan_i32(nchild.p, base.p, N, sums.p, s);
  hipLaunchKernelGGL(k_place_roots, dim3(nblk), dim3(256), 0, s, nchild.p, base.p, N, fa.p, lvl.p, posl.p, key.p, cnt.p);
  // ---- the levels ----------------------------------------------------------------------------------------------------
  int32_t* cur = fa.p;
  int32_t* nxt = fb.p;
  bool done = false;
  int levels = 0;  // levels enqueued (the last kCheck of them may be empty)
  for (int l = 0; l < kMaxLevels && !done; l += kCheck) {
    for (int i = 0; i < kCheck; ++i) {
      const int lev = l + i;
      hipLaunchKernelGGL(k_claim, dim3(nblk), dim3(256), 0, s, col, deg, width, cur, cnt.p + lev, lvl.p, key.p);
      hipLaunchKernelGGL(k_children, dim3(nblk), dim3(256), 0, s, col, deg, width, cur, cnt.p + lev, lvl.p, key.p, N, nchild.p);
      exclusive_scan_i32(nchild.p, base.p, N, sums.p, s);
      hipLaunchKernelGGL(k_place, dim3(nblk), dim3(256), 0, s, col, deg, width, cur, cnt.p + lev, nchild.p, base.p, N, lev, lvl.p,
                         posl.p, key.p, nxt, cnt.p + lev + 1);
      std::swap(cur, nxt);
    }
    levels = l + kCheck + 1;
    int32_t hn = 0;
    HIP_CHECK(hipMemcpyAsync(&hn, cnt.p + l + kCheck, 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    done = hn == 0;
  }
  if (!done) return false;
  // ---- one sort by (root, level, position in the level's frontier) -------------------------------------------------------
  const int row_bits = bits_for(N), level_bits = bits_for(levels + 1);
  hipLaunchKernelGGL(k_sort_keys, dim3(nblk), dim3(256), 0, s, label.p, lvl.p, posl.p, N, row_bits, level_bits, keys_in.p, vals_in.p);
  radix_sort_pairs(keys_in.p, keys_out.p, vals_in.p, vals_tmp.p, perm_out, N, 2 * row_bits + level_bits, hist.p, sums.p, s);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(s));  // the temporaries go back to the pool at scope exit
  return true;
}

}  // namespace osc
