#pragma once
#include "common.hpp"

namespace osc {

struct MidArgs {
  // lattice graph (ELL, row-major) -- no chain prior on this path
  const int32_t* col;
  const float* w;
  const int32_t* deg;
  int32_t width;
  OpParams op;
  const float* x0;   // initial guess, N x D (never written)
  float* X;          // solution out, N x D (must not alias x0 / U / Y)
  const float* U;    // rhs terms
  const float* Y;
  const float* B;
  const float* psi;
  float* P0;         // search direction, double-buffered (N x D each)
  float* P1;
  float* part0;      // [grid][D] column partial sums (p.Ap; first: r.z)
  float* part1;      // r.r
  float* part2;      // r.z
  float* res;        // [max_iters + 2] residual per iteration (slot it), zeroed by the host
  uint32_t* sync;    // barrier words: [32] top, [8][32] group counters, [8][32] group generations -- zeroed by the host
  uint32_t* status;  // 0 ok, 2 = a barrier wait timed out
  int32_t N, max_iters;
  float tol;
};
constexpr size_t MID_SYNC_WORDS = 32 + 8 * 32 + 8 * 32;

struct MidPlan {
  bool ok;
  int grid, un, Q;
};
// whether the lattice fits the persistent mid-size path on a device with `cus` CUs (D in {64, 128, 256} with an unpadded
// pitch, N > 6000, at most 20 float4 units of each state array per thread)
MidPlan mid_plan(int64_t N, int32_t dcols, int32_t ld, int cus, bool forced);
void launch_settle_mid(const MidArgs& a, const MidPlan& p, hipStream_t s);

}  // namespace osc
