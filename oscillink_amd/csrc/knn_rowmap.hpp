// Image row order of the lattice build's fp16 prefilter: plain C++ (host and device), no HIP types -- tests/host_logic
// sweeps it with g++.
#pragma once
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <utility>

namespace osc {

// Row order of the prefilter's fp16 IMAGE (knn_gemm.hpp: KnnPanelPlan): image rows are cut into pieces, and piece j holds
// the lattice rows [start[j], start[j + 1]) in a multiplicative order of its own: image row start[j] + t holds lattice row
// start[j] + (t a[j] mod n_j), a[j] coprime to the piece's length n_j.  One piece is the round-4 scatter r -> r a mod N;
// several pieces are what a lattice whose anchors arrive piece by piece needs (osc_graph.hip: the streamed create) -- the
// image rows of a piece are complete as soon as ITS lattice rows are on the device.
constexpr int KNN_MAP_MAX = 24;
struct KnnRowMap {
  int32_t npieces;
  int32_t start[KNN_MAP_MAX + 1];  // start[npieces] = N
  int32_t a[KNN_MAP_MAX];          // multiplier per piece (1 = identity)
  int32_t inv[KNN_MAP_MAX];        // its inverse modulo the piece's length
};
#ifdef __HIPCC__
#define OSC_HD __host__ __device__
#else
#define OSC_HD
#endif
OSC_HD inline int knn_map_piece(const KnnRowMap& m, int32_t r) {
  int j = 0;
  while (j + 1 < m.npieces && r >= m.start[j + 1]) ++j;
  return j;
}
OSC_HD inline int32_t knn_map_lattice_row(const KnnRowMap& m, int32_t N, int32_t r) {  // image row -> lattice row (r < N)
  (void)N;
  const int j = knn_map_piece(m, r);
  const int32_t base = m.start[j], n = m.start[j + 1] - base;
  return base + (int32_t)(((int64_t)(r - base) * m.a[j]) % n);
}
OSC_HD inline int32_t knn_map_image_row(const KnnRowMap& m, int32_t N, int32_t row) {  // lattice row -> image row
  (void)N;
  const int j = knn_map_piece(m, row);
  const int32_t base = m.start[j], n = m.start[j + 1] - base;
  return base + (int32_t)(((int64_t)(row - base) * m.inv[j]) % n);
}

// The threshold sample (knn_gemm.hpp: KnnPanelPlan): m sample rows in G groups of gsz consecutive rows (the last group may be
// shorter); a row's threshold is the r-th largest of its maxima over the groups.  Sample index t stands for lattice row
// floor(t N / m) -- an even stride in LATTICE order whatever order the anchors arrive in -- and the indices are dealt to the
// groups in turn (t -> group t mod G while every group has room), so the sampled members of any run of consecutive lattice
// rows (anchors that arrive cluster by cluster: a row's cluster mates) sit in as many different groups as there are.
// This returns the sample index held at position r of the sample image.
OSC_HD inline int32_t knn_sample_index(int32_t r, int32_t m, int32_t gsz, int32_t G) {
  const int32_t g = r / gsz, slot = r - g * gsz;
  const int32_t last = m - (G - 1) * gsz;  // rows of the last group (1 .. gsz)
  if (slot < last) return slot * G + g;           // rounds that still reach every group
  return last * G + (slot - last) * (G - 1) + g;  // ... and those that skip the full last one (g < G - 1 here)
}
OSC_HD inline int32_t knn_sample_lattice_row(int32_t t, int32_t m, int32_t N) {
  const int64_t row = (int64_t)t * N / m;
  return (int32_t)(row < (int64_t)N - 1 ? row : (int64_t)N - 1);
}

namespace rowmap_detail {
inline int64_t gcd64(int64_t a, int64_t b) {
  while (b) {
    const int64_t t = a % b;
    a = b;
    b = t;
  }
  return a;
}
// ~ n / golden ratio, made coprime to n: consecutive image rows are far-apart lattice rows (1: n too small to permute)
inline int32_t scatter_multiplier(int64_t n) {
  if (n <= 2) return 1;
  int64_t a = (int64_t)((double)n * 0.6180339887498949) | 1;
  while (a < n && gcd64(a, n) != 1) a += 2;
  return a < n ? (int32_t)a : 1;
}
inline int32_t inverse_mod(int64_t a, int64_t n) {  // a coprime to n (extended Euclid); 1 for the identity
  if (a == 1 || n <= 1) return 1;
  int64_t t = 0, nt = 1, r = n, nr = a % n;
  while (nr != 0) {
    const int64_t q = r / nr;
    std::swap(t, nt);
    nt -= q * t;
    std::swap(r, nr);
    nr -= q * r;
  }
  return (int32_t)(((t % n) + n) % n);
}
}  // namespace rowmap_detail

// pieces starting at the given rows (ascending, first 0, all < N; at most KNN_MAP_MAX; nullptr: one piece), each scattered
// within itself or not
inline KnnRowMap knn_row_map(int32_t N, const int32_t* starts, int npieces, bool scatter) {
  KnnRowMap m{};
  if (npieces < 1 || npieces > KNN_MAP_MAX) throw std::invalid_argument("knn_row_map: 1 to 24 pieces");
  m.npieces = npieces;
  for (int j = 0; j < npieces; ++j) {
    m.start[j] = starts != nullptr ? starts[j] : 0;
    if ((j == 0 && m.start[j] != 0) || (j > 0 && (m.start[j] <= m.start[j - 1] || m.start[j] >= N)))
      throw std::invalid_argument("knn_row_map: piece starts must ascend from 0 and stay below N");
  }
  m.start[npieces] = std::max(1, N);
  for (int j = 0; j < npieces; ++j) {
    const int32_t n = m.start[j + 1] - m.start[j];
    m.a[j] = scatter ? rowmap_detail::scatter_multiplier(n) : 1;
    m.inv[j] = rowmap_detail::inverse_mod(m.a[j], n);
  }
  return m;
}

}  // namespace osc
