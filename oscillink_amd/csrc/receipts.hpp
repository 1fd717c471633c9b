#pragma once
#include "common.hpp"

namespace osc {

struct ReceiptArgs {
  const float* Y;
  const float* Ustar;
  const float* psi;
  const float* B;
  const float* sqrt_deg;
  const int32_t* col;
  const float* adj;  // capped adjacency A_ij
  const int32_t* deg;
  int32_t width;
  float lamG, lamC, lamQ, z_th;
  float* coh;     // may be null
  float* anchor;  // may be null
  float* query;   // may be null
  int32_t* null_j;  // may be null (then null_z / null_r unused)
  float* null_z;
  float* null_r;
  int32_t N, D, ld;
  const int32_t* api_id;  // device row -> API row id (nullptr = identity); ties of the argmax go to the smaller API id
};

void launch_receipt_rows(const ReceiptArgs& a, hipStream_t s);

}  // namespace osc
