#pragma once
#include "common.hpp"

namespace osc {

struct SmallArgs {
  GraphView g;
  OpParams op;
  const int32_t* col_t;  // transposed ELL [width][N] of g.col / g.w
  const float* w_t;
  const float* x0;   // initial guess, N x ld
  float* X;          // solution out, N x ld
  const float* U;    // rhs terms
  const float* Y;
  const float* B;
  const float* psi;
  uint32_t* res_bits;  // [max_iters + 2] residual per iteration (float bits, atomicMax), zeroed by the host
  uint32_t* arrive;    // [max_iters + 2] arrival counters, zeroed by the host
  uint32_t* status;    // 0 ok, 2 = barrier timeout
  uint32_t* finish;    // nullptr, or a zeroed counter: workgroups that have stored their rows of X
  // nullptr, or host-mapped words (device address): [1 .. iterations run] residual bits, [max_iters + 2] "done" (the
  // host sets it to a pending pattern; the kernel writes 0 behind the residuals, or 2 when its barrier gave up)
  uint32_t* host_words;
  int32_t N, ld, max_iters;
  float tol;
};

// columns per workgroup (0 = the lattice does not fit the one-launch path)
int small_pick_cols(int32_t N, int32_t ld);
size_t small_lds_bytes(int32_t N, int C);
void launch_settle_small(const SmallArgs& a, int C, hipStream_t s);
void launch_transpose_ell(const int32_t* col, const float* w, int32_t N, int32_t width, int32_t* col_t, float* w_t,
                          hipStream_t s);

}  // namespace osc
