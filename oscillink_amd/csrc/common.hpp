// Shared declarations for liboscillink_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

namespace osc {

struct HipError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

inline void hip_check(hipError_t e, const char* what, const char* file, int line) {
  if (e != hipSuccess) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    throw HipError(buf);
  }
}
#define HIP_CHECK(x) ::osc::hip_check((x), #x, __FILE__, __LINE__)

// Device memory comes from a per-process caching allocator (osc_api.hip): hipMalloc / hipFree cost 0.1-1 ms each and
// hipFree synchronises the device, which dominated the create + destroy time of small lattices (one lattice per request
// in the reference's service).  Freed blocks are parked per device in size classes and handed out again; a block is
// parked only after the freeing handle's stream has drained (alloc_ctx.stream), so no other handle can receive memory
// that still has work in flight.  Contents are NOT zeroed -- exactly like hipMalloc.  OSC_POOL_MB caps the parked bytes
// per device (default 16384, 0 = no caching).
struct AllocCtx {
  int device = 0;
  hipStream_t stream = nullptr;  // the stream the calling handle enqueues on (nullptr: nothing can be in flight)
};
AllocCtx& alloc_ctx();  // thread-local, set by every API entry point
void* pool_alloc(size_t bytes, size_t* cap_bytes);
void pool_free(void* p, size_t cap_bytes);

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  size_t cap = 0;  // bytes of the underlying block (size class)
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) pool_free(p, cap);
    p = nullptr;
    n = 0;
    cap = 0;
  }
  void alloc(size_t count) {
    if (count == n && p) return;
    release();
    if (count) p = reinterpret_cast<T*>(pool_alloc(count * sizeof(T), &cap));
    n = count;
  }
  void swap(DevBuf& o) {
    std::swap(p, o.p);
    std::swap(n, o.n);
    std::swap(cap, o.cap);
  }
};

// ---- operator description ------------------------------------------------------------------
// out_i = (cs_const + cs_B * B_i) x_i - cW * sum_j W_ij x_j - cP * sum_j Wp_ij x_j
// Jacobi diagonal  Md_i = md_const + md_B * B_i   (lattice.py:187-192, 257-259: lamC is NOT in it)
// rhs              b_i  = rbU U_i + rbY Y_i + rbB B_i psi
struct OpParams {
  float cs_const, cs_B, cW, cP;
  float md_const, md_B;
  int precond;
  float rbU, rbY, rbB;
};

// ELL view of the lattice graph (degree <= width; columns ascending within a row)
struct GraphView {
  const int32_t* col;   // [N][width]
  const float* w;       // [N][width]   normalised weights W_ij
  const int32_t* deg;   // [N]
  int32_t width;
  // chain prior: path_slot[i] = -1 or row index into the (tiny) path ELL
  const int32_t* path_slot;  // nullptr when no chain / lamP == 0
  const int32_t* pcol;       // [P][pwidth]
  const float* pw;           // [P][pwidth]  normalised path weights
  const int32_t* pdeg;       // [P]
  int32_t pwidth;
};

// The lattice graph a second time, split by SOURCE-row block (k_spmm_blocked): block b holds the edges whose neighbour row
// lies in [b * rows_per_block, (b + 1) * rows_per_block), rows_per_block = ceil(N / nb).  Fixed-width: every (block, row)
// owns OSC_BLK_SLOTS slots {neighbour row, bits of W_ij}, filled in the order of the ELL row, unused slots {first row of
// the block, 0.0f} (gathered and multiplied by zero: no tests in the gather loop); an
// edge that finds the slot row of its block full sits in a later block's free slots instead (k_blk_fill), and what fits
// nowhere is in over[rest[row].x .. + rest[row].y), added after the blocks (so such rows' sums are formed in a different
// order than k_spmm's: same terms, last-bit differences).
constexpr int OSC_MAX_SRC_BLOCKS = 32;  // (register arrays of this size in k_blk_count / k_blk_fill)
constexpr int OSC_BLK_SLOTS = 4;
struct BlockedView {
  const int2* slots;    // [nb][N][OSC_BLK_SLOTS]
  const int2* rest;     // [N] {first, count} into over
  const int2* over;
  int32_t nb;           // source blocks; 0 = no blocked copy
};

// arguments of the source-blocked CG matvec (k_apply_blocked): out = (cs_const + cs_B B_i) x_i - cW sum_j W_ij x_j
struct BlkArgs {
  const float* X;   // operand, slab-major [ld / 32][N][32]
  float* OUT;       // result, row-major with pitch ld
  const float* B;   // [N]
  float* part;      // [grid][ld] column partial sums of x . out
  const int2* slots;
  const int2* rest;
  const int2* over;
  const float* gate;  // see SpmmArgs
  float gate_tol;
  float cs_const, cs_B, cW;
  int32_t N, ld, c0, c1;
  int32_t xs, xs_groups;  // workgroups per XCD that take part, slab groups (as in SpmmArgs)
  int32_t nb, groups, slices;  // source blocks; row groups (of 8 rows) per gathering wave and slice; dest-row slices
};

// k_apply_blocked as the solve's INIT pass (r = rhs - A x0, z, r . z in the matvec's epilogue): see the kernel
struct BlkInit {
  const float* Y;    // rhs term, row-major; nullptr: it is x0 (BlkArgs::X)
  float* Xcopy;      // nullptr, or the solution array x0 is copied to (row-major)
  float* R;          // residual out, row-major
  float* Z;          // z = M^-1 r out, SLAB-major like BlkArgs::X (must not be that array)
  const float* psi;  // [ld]
  float rbU, rbY, rbB, md_B, md_const;  // no preconditioner: md_B = 0, md_const = 1
};

enum SpmmMode { SPMM_AP = 0, SPMM_INIT = 1, SPMM_DOT = 2 };

struct SpmmArgs {
  GraphView g;
  OpParams op;
  const float* X;     // operand (gathered)
  float* OUT;         // AP: A x ; INIT: x copy (x0)     (may alias nothing else)
  float* R;           // INIT: residual out
  float* P;           // INIT: search direction out
  const float* U;     // INIT rhs
  const float* Y;     // INIT rhs
  const float* B;     // [N]
  const float* psi;   // [ld]
  float* part;        // [grid][ld] column partial sums of the mode's dot product
  int64_t N;          // end of the row range this launch works on (rows [row0, N))
  int64_t row0;       // first row (0 except in row-sharded multi-GPU runs)
  int32_t ld;         // row pitch in floats (multiple of 4)
  int32_t c0, c1;     // column window [c0, c1), multiples of 4
  // speculative-launch gate: the kernel is a no-op when *gate <= gate_tol (the CG already converged; the host
  // enqueues one iteration ahead of its residual read-back).  nullptr = always run.
  const float* gate;
  float gate_tol;
  // xs > 0: XCD-affine narrow slabs.  The window [c0, c1) is cut into 32-column slabs (one 128-byte line per row).
  // The eight XCDs (x = blockIdx % 8) form xs_groups slab groups of 8 / xs_groups XCDs each: group g = x % xs_groups
  // owns slabs g, g + xs_groups, ... and its XCDs split the rows among themselves (part x / xs_groups); the first xs
  // workgroups of each XCD do the work.  The slab an XCD gathers from (N x 128 B) then competes for that XCD's 4 MB
  // L2 alone instead of with the other slabs of the window.  xs_groups = gcd(8, number of slabs): always balanced.
  int32_t xs;
  int32_t xs_groups;
  // Slab-major ("blocked") layout of the CG search direction, xs mode only: element (row, col) of an array stored
  // that way sits at ((col / 32) * rows + row) * 32 + col % 32, so the 32-column slab an XCD gathers from is one
  // contiguous N x 128 B range (even spread over the L2 channels; no 3 KB row stride).  xblk != 0: the operand X is
  // stored that way (value = rows per slab = N); pblk != 0: INIT writes its P output that way.
  int64_t xblk, pblk;
  // != 0: the rows are stored in a local order (maybe_reorder): launch the variant with more gathers in flight per row
  int32_t deep;
};

struct UpdateArgs {
  float* X;
  float* R;
  float* P;
  const float* AP;
  const float* B;
  const float* alpha;  // [ld]
  const float* beta;   // [ld]
  float* part_rr;      // [grid][ld]
  float* part_rz;      // [grid][ld]
  OpParams op;
  int64_t N;      // rows [row0, N)
  int64_t row0;
  int32_t ld, c0, c1;
  const float* gate;  // see SpmmArgs
  float gate_tol;
  int64_t pblk;  // != 0: P is stored slab-major with this many rows per slab (see SpmmArgs)
  // != 0: the solve's five N x window arrays fit the Infinity Cache together (a per-rank window of a sharded solve, a
  // mid-size lattice): X, R, AP are then read and written with ordinary loads / stores so that the next kernel finds them
  // there; 0: streamed nontemporally (they would only push the gathered operand out).
  int32_t temporal;
  // where the x update happens (run_cg): 0 = in k_update_xr, next to the r update (x, r, p, Ap read; x, r written);
  // OSC_XMODE_XR_SKIPS_X: not there; OSC_XMODE_P_APPLIES_X: k_update_p applies the PREVIOUS iteration's while it has p in
  // hand (launch_update_x behind the last iteration) -- one array pass less per iteration; OSC_XMODE_XR_LAST: k_update_xr
  // in the form for the expected last iteration (x finished there, the new r not stored)
  int32_t xmode;
};
constexpr int32_t OSC_XMODE_XR_SKIPS_X = 1, OSC_XMODE_P_APPLIES_X = 2, OSC_XMODE_XR_LAST = 4;

struct Gate {
  const float* p;
  float tol;
};

// launchers implemented in cg_kernels.hip
int spmm_grid(int64_t N, int32_t ncols);
// source-blocked copy of an ELL graph (BlockedView): count the overflow entries (-> *over_count, device), then fill
void launch_blocked_count(const int32_t* col, const int32_t* deg, int32_t width, int32_t N, int32_t nb,
                          unsigned* over_count, hipStream_t s);
void launch_blocked_fill(const int32_t* col, const float* w, const int32_t* deg, int32_t width, int32_t N, int32_t nb,
                         int2* slots, int2* rest, int2* over, unsigned* over_count, hipStream_t s);
// row groups per wave the blocked apply holds in registers, and the workgroups per CU it needs resident
int blocked_variants();                 // kernel shapes of the blocked matvec (OSC_BLK_VARIANT)
int blocked_groups_max(int variant);
int blocked_gather_waves(int variant);
// Chain prior beside the blocked matvec: out_i -= cP sum_j Wp_ij x_j for the (few) rows of the chain's path graph, and
// the matching terms of the p . Ap column sums into the rows [part_row0, part_row0 + chunks) of `part`.
constexpr int OSC_CHAIN_FIX_MAX_ROWS = 4096;
constexpr int OSC_CHAIN_FIX_MAX_CHUNKS = 64;
struct ChainFixArgs {
  const float* X;        // operand, slab-major
  float* OUT;            // row-major, pitch ld: already holds the result without the chain term
  float* part;
  const int32_t* prow;   // [prows] lattice row of path row s
  const int32_t* pcol;   // [prows][pwidth]
  const float* pw;
  const int32_t* pdeg;
  const float* gate;
  float gate_tol, cP;
  int32_t prows, pwidth, N, ld, c0, c1, part_row0, chunks;
  // behind k_apply_blocked's INIT pass (BlkInit): patch r (row-major), z (slab-major) and the r . z partials of the path
  // rows instead of OUT / x . out; nullptr otherwise
  float* initR;
  float* initZ;
  const float* B;
  float md_B, md_const;  // no preconditioner: 0, 1
};
// initial residual of a solve around the blocked matvec: rows_to_slab copies the columns [c0, c1) of a row-major array into
// the slab-major layout the blocked matvec gathers from; init_finish turns AP = A x0 into r = b - A x0, z = r / (Md + eps),
// p = z (slab-major), the x0 copy (when the solve is not in place) and the r . z column partials -- the tail of
// k_spmm<.., INIT>, same expressions (solver.py:19-22)
struct InitFinishArgs {
  const float* AP;   // A x0, row-major
  const float* X0;   // x0
  float* X;          // solution array (== X0: in place, nothing to copy)
  float* R;
  float* P;          // slab-major, `pblk` rows per slab
  const float* U;    // rhs terms (either may alias X0)
  const float* Y;
  const float* B;
  const float* psi;
  float* part;       // [grid][ld] r . z partials
  OpParams op;
  int64_t N, pblk;
  int32_t ld, c0, c1;
};
void launch_rows_to_slab(const float* src, float* dst, int64_t N, int32_t ld, int32_t c0, int32_t c1, int grid, hipStream_t s,
                         const float* sub = nullptr);  // sub: dst = src - sub
void launch_init_finish(const InitFinishArgs& a, int grid, hipStream_t s);
int chain_fix_chunks(int32_t prows);
void launch_chain_fix(const ChainFixArgs& a, hipStream_t s);
void launch_apply_blocked(const BlkArgs& a, int grid, hipStream_t s, const BlkInit* init = nullptr, int variant = 0,
                          unsigned long long* stamps = nullptr);
int blocked_resident_per_cu(int variant);
void launch_spmm(int mode, const SpmmArgs& a, int grid, hipStream_t s);
void launch_update_xr(const UpdateArgs& a, int grid, hipStream_t s);
void launch_update_p(const UpdateArgs& a, int grid, hipStream_t s);
void launch_update_x(const UpdateArgs& a, int grid, hipStream_t s);  // x += alpha p (UpdateArgs::xmode)
// column reductions over `nb` partial rows
void launch_reduce_init(const float* part, int nb, int32_t ld, int32_t c0, int32_t c1, double* rz, hipStream_t s);
void launch_reduce_alpha(const float* part, int nb, int32_t ld, int32_t c0, int32_t c1, const double* rz,
                         float* alpha, Gate g, hipStream_t s);
void launch_reduce_beta(const float* part_rr, const float* part_rz, int nb, int32_t ld, int32_t c0, int32_t c1,
                        double* rz, float* beta, uint32_t* res_bits_slot, Gate g, hipStream_t s,
                        uint32_t* done_ctr = nullptr, float* host_slot = nullptr);
void launch_publish_word(const uint32_t* src, uint32_t* host_slot, hipStream_t s);
void launch_reduce_sum(const float* part, int nb, int32_t ld, int32_t c0, int32_t c1, double* out_cols,
                       hipStream_t s);
void launch_axpby(float* out, const float* a, float ca, const float* b, float cb, int64_t n, hipStream_t s);
// row-sharded CG: the column sums are completed across ranks (all-reduce of fp64 sums) before these finish steps
void launch_finish_init(const double* sums, int32_t c0, int32_t c1, double* rz, hipStream_t s);
void launch_finish_alpha(const double* sums, int32_t c0, int32_t c1, const double* rz, float* alpha, Gate g,
                         hipStream_t s);
void launch_finish_beta(const double* sums_rr, const double* sums_rz, int32_t c0, int32_t c1, double* rz, float* beta,
                        uint32_t* res_bits_slot, Gate g, hipStream_t s);
void launch_reduce_sum_gated(const float* part, int nb, int32_t ld, int32_t c0, int32_t c1, double* out_cols, Gate g,
                             hipStream_t s);

}  // namespace osc
