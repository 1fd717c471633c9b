// Internal header of liboscillink_hip.so's host side: the handle (struct osc_lattice), the error types the C ABI maps to
// status codes, and the functions the translation units share.  Round 5 cut osc_api.hip (3 500 lines) along its seams:
//   osc_runtime.hip : process-wide pools (streams, device and pinned host memory, staging buffers, control blocks),
//                     host <-> device transfers, profiling events, per-handle scratch
//   osc_graph.hip   : lattice build orchestration (graph.py:8-93 on the device), chain prior, internal row order
//   osc_solve.hip   : operator parameters, apply plans, the CG drivers (one GPU, column windows, row-sharded + halo lists)
//   osc_api.hip     : environment switches and the extern "C" entry points of include/oscillink_hip.h
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/oscillink_hip.h"
#include "common.hpp"
#include "host_logic.hpp"
#include "comm.hpp"
#include "knn.hpp"
#include "knn_gemm.hpp"
#include "receipts.hpp"
#include "perm.hpp"
#include "dynamics.hpp"
#include "small.hpp"

using namespace osc;


using namespace osc;

struct ProfSlot {
  hipEvent_t a, b;
  int which;
  int iter;  // CG iteration the launch belongs to (0 = not part of a CG loop); speculative no-ops are dropped
};

struct Invalid : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct StateError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct Unsupported : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// per device like the streams
struct CtrlBlock {
  float* res_host = nullptr;
  size_t res_host_n = 0;
  std::vector<hipEvent_t> events;
};

hipStream_t acquire_stream(int device);
void release_stream(int device, hipStream_t s);
double now_ms();

struct osc_lattice {
  int device = 0;
  hipStream_t stream = nullptr;
  int64_t N = 0;
  int32_t D = 0, ld = 0;
  // state (N x ld, row-major)
  DevBuf<float> Y, U, X, R, P, AP, Ustar;
  bool have_ustar = false;
  DevBuf<float> Uprev;  // state before the last settle (dynamics snapshot, lattice.py:825-927); allocated on first use
  bool have_uprev = false;
  DevBuf<float> B, psi;
  float lamG = 1.0f, lamC = 0.5f, lamQ = 4.0f;
  // graph (ELL)
  int32_t k_eff = 0;
  float row_cap = 1.0f;
  int deterministic = 0;
  int64_t seed = -1;
  bool have_graph = false;
  int32_t width = 0;
  DevBuf<int32_t> ell_col, deg;
  DevBuf<float> ell_a, ell_w, sqrt_deg;
  DevBuf<float> knn_val;
  DevBuf<int32_t> knn_idx;
  int32_t knn_k = 0;
  int32_t knn_fallback_rows = 0;  // rows of the last build the prefilter could not prove and the exact kernel redid
  bool knn_prefilter = false;     // the last build used the fp16 prefilter
  bool knn_panel = false;         // ... in its register-resident-panel shape (knn_gemm.hip)
  double build_ms = 0.0;
  int64_t nnz = 0;
  int32_t max_deg = 0;
  // internal row order (empty = identity): API row i lives at device row inv_h[i]; perm_h[new] = old
  int reorder = -1;        // OSC_REORDER: 0 never, 1 always, unset = auto (when the graph is clustered enough to pay)
  double clustering = 0.0;  // sampled local clustering coefficient of the last graph
  bool reordered = false;
  std::vector<int32_t> perm_h, inv_h;
  DevBuf<int32_t> perm_d, inv_d;
  // chain prior (kept in API ids on the host so it can be re-installed after a re-order)
  std::vector<int32_t> chain_nodes;
  std::vector<float> chain_w;
  bool chain_present = false;
  float lamP = 0.0f;
  int32_t prows = 0, pwidth = 0;
  DevBuf<int32_t> path_slot, pcol, pdeg, prow;  // prow: lattice row of path row s (inverse of path_slot)
  DevBuf<float> pw;
  // CG scratch
  int grid_cap = 1024;
  DevBuf<float> vec_q, vec_n;  // query / per-row result scratch of the cosine calls
  int32_t dcols = 0;      // D rounded up to 4: the columns the kernels work on (ld >= dcols is the row pitch)
  int32_t spmm_slab = 0;  // 0 = whole window per launch
  int spmm_xs = -1;        // XCD-affine narrow slabs: -1 auto, 0 off, 1 on (OSC_SPMM_XS)
  bool p_blocked = true;   // slab-major search direction in xs mode (OSC_P_BLOCKED=0 keeps it row-major)
  int xs_nb = 0;           // workgroups per XCD in that mode; 0 = automatic (OSC_XS_NB)
  int xs_groups_cap = 8;   // upper bound on the slab groups (= slabs in flight) of that mode (OSC_XS_GROUPS)
  int xs_min_cols = 32;    // narrowest column window the mode is used for (96 until round 3 -- with the
                           // blocked matvec under it, one- and two-slab windows win too: 100k x 64 k 16 0.505 -> 0.425 ms per
                           // settle, 100k x 32 0.352 -> 0.309, 200k x 64 k 32 1.43 -> 0.97, 60k x 64 k 32 0.438 -> 0.387)
  int xs_min_rows = 6144, xs_min_rows_narrow = 0;  // smallest lattice the mode is used for: windows of >= 256 columns / narrower ones (0: by width, xs_plan)
  int xs_groups_min = 2;   // fewest slab groups the mode is kept for when the natural count had to be reduced
  DevBuf<float> part0, part1, alpha, beta;
  DevBuf<double> rz, colsum;
  DevBuf<uint32_t> res_bits;  // residual slots of the row-sharded solve
  // Zeroed control words of the solves (residual slots, arrival counters): a ring of segments, one per solve, cleared all
  // at once when it wraps -- hipMemsetAsync costs ~15 us of HOST time per call on this stack, during which the device
  // sits idle at the start of a solve (7 % of a settle at N = 20000, D = 128; a quarter of one at N = 80)
  DevBuf<uint32_t> ctrl_ring;
  size_t ctrl_seg = 0;   // words per segment
  int ctrl_next = 0;     // next free segment
  bool small_path = true;             // OSC_SMALL_PATH=0 disables the one-launch CG for small lattices
  bool fake_window = false;  // OSC_FAKE_COL_SHARD under a one-rank communicator (measurement hook; reported by osc_comm_info)
  // build-route switches (read_env): every OSC_* variable the library reads per handle is read in ONE place, at
  // osc_create and again at osc_rebuild_graph (INTEGRATION.md has the table)
  int knn_mode = 0;            // OSC_KNN_MODE: 0 automatic, 1 exact, 2 tile prefilter, 3 panel prefilter
  int knn_fake_shards = 0;     // OSC_KNN_FAKE_SHARDS
  int knn_splits = 0;          // OSC_KNN_SPLITS (tile / exact routes: column splits)
  bool knn_scatter = true;     // OSC_KNN_PANEL_SCATTER
  bool knn_sym = true;         // OSC_KNN_PANEL_SYM
  bool create_stream = true;   // OSC_CREATE_STREAM: osc_create hands the anchors to the build piece by piece (osc_graph.hip)
  int32_t create_piece_mb = 24;  // OSC_CREATE_PIECE_MB: anchors per piece, at least (experiments)
  bool create_piece_mb_set = false;
  int32_t create_min_mb = 64;  // OSC_CREATE_MIN_MB: anchors below this travel whole (the streamed create's fixed costs)
  bool receipt_pair = true;  // OSC_RECEIPT_PAIR=0: the receipt's per-edge pass from both ends of every edge (one launch)
  int32_t create_two_mb = 256;  // OSC_CREATE_TWO_PIECE_MB: anchors from this size on may travel in TWO pieces (three below it)
  bool create_force_retry = false;  // OSC_CREATE_FORCE_RETRY (test hook): a streamed build always hands over to the whole-array one
  int32_t create_pieces = 0;   // pieces the last build received its anchors in (0: they were on the device before it started)
  int knn_rescore_pair = 1;  // OSC_KNN_RESCORE_PAIR: 0 = every candidate pair scored from both ends, 1 = once where rows have >= 384 columns, 2 = once at any width (experiments)
  int32_t knn_sweep = 0;       // main sweep of the last build's thresholds-and-hits prefilter: 0 none (another route), 1 every column tile per row block, 2 the half sweep (ONE per build, shared by the ranks of a sharded build)
  bool knn_force_exchange = false;  // OSC_KNN_FORCE_EXCHANGE=1 (test hook): run the sharded half sweep's collectives under a ONE-rank communicator too
  KnnPanelTune knn_tune{};     // OSC_KNN_PANEL_NRG / _RHO / _T / _RANK
  int halo_force = 0;          // OSC_HALO: 1 full, 2 lists
  bool bfs_host = false;       // OSC_BFS_HOST=1: the breadth-first row order is walked on the host (A/B, tests)
  int fake_col_r = 0, fake_col_w = 0;  // OSC_FAKE_COL_SHARD "r/w"
  int predicted_iters[3] = {0, 0, 0};  // iterations the last general-path solve of each kind (CgBuffers::kind) took (0 = unknown)
  bool x_defer = true;                // the x update rides in the next iteration's p update (run_cg; OSC_X_DEFER=0: beside the r update)
  bool x_last_form = true;            // ... and the expected last iteration finishes x itself without storing r (OSC_X_DEFER=2: off)
  DevBuf<int32_t> ell_col_t;          // transposed ELL for the one-launch path (built on first use per graph)
  DevBuf<float> ell_w_t;
  bool ell_t_ready = false;
  // block-major copy of the graph for the source-blocked CG matvec (k_spmm_blocked), built on first use per graph
  DevBuf<int2> blk_slots, blk_rest, blk_over;
  int blk_nb = 0;          // blocks of the copy held (0 = none / stale)
  int spmm_blocked = -1;   // -1 by lattice size, 0 off, > 0 = that many source blocks (OSC_SPMM_BLOCKED)
  double blk_mb = 2.0;     // smallest slab (N x 128 B, MiB) the blocked apply is chosen for
  double blk_edges = 0.0;  // edges of a row per source block the block count aims at; 0 = by lattice size: 3.3 / 2.5
  mutable int blk_resident[8] = {-1, -1, -1, -1, -1, -1, -1, -1};  // workgroups per XCD each shape of the blocked apply gets resident (queried once)
  int blk_variant = -1;    // kernel shape of the blocked matvec (cg_kernels.hip: kBlkShapes); -1 = by geometry (blocked_shape_for), OSC_BLK_VARIANT forces one
  int blk_shape_last = 0;  // the shape the last general-path solve's blocked matvec ran with
  int blk_wide_min_rows = 0;  // smallest lattice the wide shapes are chosen for (OSC_BLK_WIDE_MIN_ROWS; 0 = default)
  bool blk_stamp = false;  // OSC_BLK_STAMP=1: while profiling is on, the AP applies run the cycle-stamping instantiation
  DevBuf<unsigned long long> blk_stamps;  // [grid][waves per workgroup][4] (osc_profile_get slots 8-13)
  int64_t blk_stamp_launches = 0;
  int blk_stamp_grid = 0;
  int blk_last = 0;        // source blocks the last general-path solve's matvec used (0 = plain apply)
  double temporal_mb = 200.0;  // largest solve (5 arrays x N x window) whose update kernels use ordinary instead of nontemporal accesses
  bool spmm_deep = true;   // re-ordered lattices: the operator apply with 8 gathers in flight per row (OSC_SPMM_DEEP=0: the usual 2)
  bool blk_init = true;    // the initial residual goes through the blocked matvec as well (OSC_BLK_INIT=0: plain INIT apply)
  bool blk_init_fused = true;  // ... and is formed in that launch's epilogue where it can be (OSC_BLK_INIT=2: separate finish pass)
  int64_t blk_applies = 0; // blocked matvecs enqueued since creation
  int64_t small_solves = 0;
  float* res_host = nullptr;  // pinned, host-mapped mirror of res_bits for the per-iteration read-back
  float* res_host_dev = nullptr;  // the device's address of it
  size_t res_host_n = 0;
  bool mapped_residual = true;  // residuals published into host-mapped memory (false: copy + event per iteration)
  std::vector<hipEvent_t> iter_events;
  // sharded solves: the stop test's all-reduce runs on a second stream beside the next iteration's p update and matvec
  // (run_cg); step_events[it] = "iteration it's local residual is out" (OSC_COMM_OVERLAP=0: all-reduce in the solve's stream)
  hipStream_t comm_stream = nullptr;
  std::vector<hipEvent_t> step_events;
  int comm_overlap = -1;  // 1 / 0: always / never; -1: from four ranks on (run_cg)
  bool comm_stream_busy = false;  // a solve left work on comm_stream (at most a speculative iteration's all-reduce + publish)
  std::vector<float> history;
  // column shard (multi-GPU, column-sharded CG); single GPU: [0, ld)
  int32_t c0 = 0, c1 = 0;
  std::unique_ptr<Comm> comm;  // RCCL (one process per GPU) or the in-process loopback (comm.hpp)
  int rank = 0, world = 1;
  bool u_sharded = false;  // U holds only this rank's columns (after a sharded settle)
  int shard_mode = 0;      // 0 = column-sharded CG (default), 1 = row-sharded CG (north-star wording; OSC_SHARD=row)
  int fake_row_shards = 0; // test hook (OSC_ROW_FAKE_SHARDS=V): V row shards on this one GPU, collectives local
  DevBuf<double> sums;     // [2][ld] completed column sums of the row-sharded CG
  DevBuf<float> comm_buf;
  // halo plan of the row-sharded CG (built on first use per graph / chain / communicator: graph_epoch)
  uint64_t graph_epoch = 1;
  struct HaloPlan {
    uint64_t epoch = 0;                      // graph_epoch it was built for (0 = none)
    bool full = false;                       // halo ~ everything: exchange whole row blocks instead (all-gather)
    std::vector<int64_t> give_off, need_off; // [world + 1] offsets of each peer's slice in give_idx / need_idx
    DevBuf<int32_t> give_idx, need_idx;      // my rows each peer needs (sorted) / the peers' rows I need (sorted)
    DevBuf<float> send, recv;                // packed rows
    int64_t need_rows = 0, give_rows = 0;    // this rank
    int64_t need_rows_max = 0;               // max over ranks
  } halo;
  // profiling
  bool prof_on = false;
  std::vector<ProfSlot> prof_pending;
  std::vector<hipEvent_t> prof_pool;
  int64_t prof_count[5] = {0, 0, 0, 0, 0};
  double prof_ms[5] = {0, 0, 0, 0, 0};
  std::string err;

  ~osc_lattice() {
    for (auto& s : prof_pending) {
      (void)hipEventDestroy(s.a);
      (void)hipEventDestroy(s.b);
    }
    for (auto e : prof_pool) (void)hipEventDestroy(e);
    if (comm_stream && comm_stream_busy) (void)hipStreamSynchronize(comm_stream);
    for (auto e : step_events) (void)hipEventDestroy(e);
    park_ctrl();
    release_stream(device, comm_stream);
    release_stream(device, stream);
  }
  void park_ctrl();
};

using L = osc_lattice;

constexpr size_t kStageBytes = (size_t)32 << 20;  // one pinned staging buffer (two per StagePair)
struct StagePair {
  void* buf[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
};

struct CgBuffers {  // the arrays one solve works on (all N x ld)
  const float* x0;  // gathered in INIT
  float* X;
  float* R;
  float* P;
  float* AP;
  const float* rhsU;
  const float* rhsY;
  const float* B;
  const float* psi;
  int32_t ld, c0, c1;
  // When X aliases x0 / rhsU (the in-place warm-started settle), a path that cannot guarantee it completes -- the
  // one-launch small kernel may give up at its barrier -- writes here instead and reports it in CgResult::sol, so a
  // failed attempt never leaves the caller's state partly advanced.  nullptr: X is never aliased.
  float* Xalt = nullptr;
  int kind = 0;  // 0 settle, 1 U*, 2 single right-hand side: repeated solves of one kind take the same iteration count
};

struct CgResult {
  int iters;
  float res;
  float* sol = nullptr;  // the buffer that holds the solution (b.X, or b.Xalt)
};

struct RowShard {
  int64_t r0, r1;
};

hipEvent_t prof_event(L& h);
void prof_drain(L& h, bool nothrow = false);
void use_device(L& h);
void sync(L& h);
void upload_rows(L& h, float* dst, const float* src);
StagePair acquire_stage(int device);
void release_stage(int device, const StagePair& sp);
void* host_pool_alloc(size_t bytes);
bool host_pool_free(void* p);
bool host_pool_owns(const void* p, size_t bytes);
void parallel_copy(char* dst, const char* src, size_t bytes, int threads);
void download_contiguous(L& h, char* dst, const char* src, size_t bytes);
void download_rows(L& h, float* dst, const float* src);
void to_api_order(const L& h, float* v);
void download_api_order(L& h, float* dst, const float* src);
void drain_comm_stream(L& h);
uint32_t* ctrl_segment(L& h, size_t words);
void ensure_ctrl(L& h, size_t slots);
void ensure_cg_scratch(L& h, int max_iters);
int cg_grid(const L& h);
int blocked_quad_form(L& h, const OpParams& op, const float* x_rows, float* scratch_slab, float* scratch_out, bool with_path,
                      const float* x_sub = nullptr);  // x_sub: the form of x_rows - x_sub
GraphView graph_view(L& h, bool with_path);
void graph_counts(L& h);
void alloc_ell(L& h, int32_t width);
bool permuted(const L& h);
void install_chain(L& l);
void move_state(L& l, const int32_t* from_d, const int32_t* relabel_d);
void drop_order(L& l);
void apply_order(L& l, const std::vector<int32_t>& perm);
std::vector<int32_t> bfs_order(L& l);
void maybe_reorder(L& l);
void exchange_buckets(L& h, const KnnPanelPlan& pp, const KnnPanelSymDev& sd, int rb_per);
void build_graph(L& h, const float* host_Y = nullptr);
bool path_active(const L& h);
OpParams settle_op(const L& h, float dt, int precond);
OpParams ustar_op(const L& h);
int32_t auto_slab(const L& h, int32_t ncols);
int xs_groups(int32_t ncols, int cap = 8);
int xs_groups_for(const L& h, int32_t ncols);
int xs_plan(const L& h, int32_t ncols, int grid);
int blocked_resident(const L& h, int shape);
int blocked_shape_for(const L& h, int xg, int grid);
int blocked_plan(const L& h, bool with_path);
BlockedView blocked_view(L& h, int nb);
void spmm_slabbed(L& h, int mode, SpmmArgs sa, int grid, int iter = 0);
bool run_cg_small(L& h, const OpParams& op, const CgBuffers& b, bool with_path, int max_iters, float tol,
                  CgResult& out);
CgResult run_cg(L& h, const OpParams& op, const CgBuffers& b, bool with_path, int max_iters, float tol);
void gather_columns(L& h, float* arr);
std::vector<RowShard> row_shards(const L& h);
void exchange_rows(L& h, float* arr, int32_t ld);
void allreduce_sums(L& h, double* buf, size_t n);
void build_halo_plan(L& h);
void halo_exchange(L& h, float* arr, int32_t ld);
CgResult run_cg_rows(L& h, const OpParams& op, const CgBuffers& b, bool with_path, int max_iters, float tol);
bool row_mode(const L& h);
bool env_num(const char* name, int& out);
void read_env_solver(L& h);
void read_env_build(L& h);
void read_env(L& h);
void require_graph(L& h);

struct ProfScope {
  L& h;
  ProfSlot s{};
  bool on;
  ProfScope(L& h_, int which, int iter = 0) : h(h_), on(h_.prof_on) {
    if (on) {
      s.which = which;
      s.iter = iter;
      s.a = prof_event(h);
      s.b = prof_event(h);
      HIP_CHECK(hipEventRecord(s.a, h.stream));
    }
  }
  ~ProfScope() {
    if (on) {
      (void)hipEventRecord(s.b, h.stream);
      h.prof_pending.push_back(s);
      if (h.prof_pending.size() > 8192) prof_drain(h, true);
    }
  }
};
