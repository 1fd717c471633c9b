#pragma once
#include "common.hpp"

namespace osc {

// scatter == false: out[i] = in[map[i]] ; scatter == true: out[map[i]] = in[i]
void launch_move_rows(float* out, const float* in, const int32_t* map, int64_t N, int32_t ld, bool scatter,
                      hipStream_t s);
void launch_move_f32(float* out, const float* in, const int32_t* map, int64_t N, bool scatter, hipStream_t s);
void launch_permute_ell(const int32_t* col_in, const float* a_in, const float* w_in, const int32_t* deg_in,
                        const int32_t* from, const int32_t* relabel, int32_t width, int64_t N, int32_t* col_out,
                        float* a_out, float* w_out, int32_t* deg_out, hipStream_t s);

// The breadth-first row order of the lattice graph computed on the device (bfs_order.hip): perm_out[new] = old, identical to
// the host's queue BFS (components by smallest row id, neighbours in slot order).  false: graph too deep / too large for
// the device form -- walk it on the host.
bool device_bfs_order(const int32_t* col, const int32_t* deg, int32_t width, int32_t N, int32_t* perm_out, hipStream_t s);

// exclusive prefix sum of n int32 on the device (bfs_order.hip; out may alias in; sums: scan_blocks(n) int32 of scratch)
size_t scan_blocks(int64_t n);
void exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* sums, hipStream_t s);

// counts: 2 * nsample words, written (not added to): [2 s] adjacent neighbour pairs of sampled row s, [2 s + 1] its pairs
void launch_clustering_sample(const int32_t* col, const int32_t* deg, int32_t width, int64_t N, int32_t nsample,
                              unsigned long long* counts, hipStream_t s);

}  // namespace osc
