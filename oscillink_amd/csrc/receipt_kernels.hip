// Receipt diagnostics on the sparse lattice (receipts.py:28-83), one wave per row, edges gathered like the CG matvec.
//
// For row i with neighbours j (capped adjacency a_ij, columns ascending):
//   coh_drop_i = sum_j 0.5 lamC a_ij (||Yn_i - Yn_j||^2 - ||Un_i - Un_j||^2),  Yn = Y/(sqrt_deg+1e-12), Un = U*/(sqrt_deg+1e-12)
//   anchor_i   = lamG ||U*_i - Y_i||^2 ;  query_i = lamQ B_i ||U*_i - psi||^2                      (receipts.py:40-59)
//   null point : R_ij = lamC a_ij ||Un_i - Un_j||^2 ; the reference z-scores each DENSE row (N entries, zeros
//                included): mu = sum R / N, sigma = sqrt(sum R^2 / N - mu^2) + 1e-12; candidate = first argmax.
#include "common.hpp"
#include "receipts.hpp"

namespace osc {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void k_receipt_rows(const ReceiptArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.N) return;
  const size_t ro = (size_t)row * a.ld;
  const float inv_i = 1.0f / (a.sqrt_deg[row] + 1e-12f);
  // anchor / query terms
  float an = 0.f, qu = 0.f;
  for (int c = lane; c < a.D; c += 64) {
    const float us = a.Ustar[ro + c];
    const float dy = us - a.Y[ro + c], dq = us - a.psi[c];
    an = fmaf(dy, dy, an);
    qu = fmaf(dq, dq, qu);
  }
  an = wave_sum(an);
  qu = wave_sum(qu);
  // edges
  const int deg = a.deg[row];
  float coh = 0.f;
  double s1 = 0.0, s2 = 0.0;
  float rmax = 0.f;
  int jmax = -1;
  for (int e = 0; e < deg; ++e) {
    const int j = a.col[(size_t)row * a.width + e];
    const float w = a.adj[(size_t)row * a.width + e];
    const size_t jo = (size_t)j * a.ld;
    const float inv_j = 1.0f / (a.sqrt_deg[j] + 1e-12f);
    float dy = 0.f, du = 0.f;
    for (int c = lane; c < a.D; c += 64) {
      const float y = a.Y[ro + c] * inv_i - a.Y[jo + c] * inv_j;
      const float u = a.Ustar[ro + c] * inv_i - a.Ustar[jo + c] * inv_j;
      dy = fmaf(y, y, dy);
      du = fmaf(u, u, du);
    }
    dy = wave_sum(dy);
    du = wave_sum(du);
    if (w > 0.f) {
      coh += 0.5f * a.lamC * w * (dy - du);
      const float R = a.lamC * w * du;
      s1 += (double)R;
      s2 += (double)R * (double)R;
      if (R > rmax) {  // strict: ties keep the smallest column (first argmax)
        rmax = R;
        jmax = j;
      }
    }
  }
  if (lane == 0) {
    if (a.coh) a.coh[row] = coh;
    if (a.anchor) a.anchor[row] = a.lamG * an;
    if (a.query) a.query[row] = a.lamQ * a.B[row] * qu;
    if (a.null_j) {
      const double mu = s1 / (double)a.N;
      double var = s2 / (double)a.N - mu * mu;
      if (var < 0.0) var = 0.0;
      const double z = ((double)rmax - mu) / (sqrt(var) + 1e-12);
      const bool is_null = (jmax >= 0) && (rmax > 0.f) && (z > (double)a.z_th);
      a.null_j[row] = is_null ? jmax : -1;
      a.null_z[row] = (float)z;
      a.null_r[row] = rmax;
    }
  }
}

}  // namespace

void launch_receipt_rows(const ReceiptArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_receipt_rows, dim3((unsigned)((a.N + 3) / 4)), dim3(256), 0, s, a);
  HIP_CHECK(hipGetLastError());
}

}  // namespace osc
