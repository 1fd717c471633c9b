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
  // dynamics (lattice.py:862-883): with Y := U_prev and Ustar := U_next the per-edge structural energy drop
  // max(0, 0.5 lamC a_ij (||Up_i - Up_j||^2 - ||Un_i - Un_j||^2)) goes to edge_flow[row * width + e] (ELL-shaped, slots
  // past the degree untouched); nullptr = not wanted
  float* edge_flow = nullptr;
  // the pair form (receipt_kernels.hip: k_receipt_pairs + k_receipt_finish): ELL-shaped scratch [N * width] for the two squared
  // distances of every edge, computed from its lower end only, and a word the finish raises if an edge has no mirror slot
  // (then the outputs are not valid and the caller runs the one-launch kernel).  All three set = use the pair form.
  float* pair_dy = nullptr;
  float* pair_du = nullptr;
  int32_t* pair_fail = nullptr;
};

void launch_receipt_rows(const ReceiptArgs& a, hipStream_t s);

// greedy MMR (graph.py:114-133) on the device: one call per selection step
struct MmrArgs {
  const float* Y;           // anchors, N x ld
  const double* base;       // (1 - lambda) * score, device row order
  double* maxsim;           // running max cosine to the chosen items
  unsigned char* alive;     // 1 = still a candidate
  const int32_t* api_id;    // device row -> API row id (nullptr = identity): ties go to the smaller API id
  float* q;                 // [D] normalised anchor row of the item chosen in this step
  double* pval;             // [nblocks] stage-1 partials
  int32_t* pid;
  int32_t* prow;
  int32_t* chosen_api;      // [k] chosen items (API ids), -1 = none left
  int32_t N, D, ld, nblocks;
  double lambda;
};
void launch_mmr_step(const MmrArgs& a, int step, hipStream_t s);
int mmr_parts(int32_t N);  // partials of the fused cosine + argmax pass (pval / pid / prow must hold max(nblocks, this))

}  // namespace osc
