// C ABI of liboscillink_hip.so (include/oscillink_hip.h): handle management, lattice build orchestration,
// the CG driver and receipts.  All device work of a handle goes to the handle's own HIP stream.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
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

namespace {

thread_local std::string g_create_error;

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

// hipStreamCreate costs 1.5-4 ms on this stack: streams of destroyed handles are parked per device and reused.
// (The only process-wide state of the library; guarded by a mutex, holds no lattice data.)
std::mutex g_pool_mu;
std::map<int, std::vector<hipStream_t>> g_stream_pool;

hipStream_t acquire_stream(int device) {
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto& v = g_stream_pool[device];
    if (!v.empty()) {
      hipStream_t s = v.back();
      v.pop_back();
      return s;
    }
  }
  hipStream_t s = nullptr;
  HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  return s;
}
void release_stream(int device, hipStream_t s) {
  if (!s) return;
  std::lock_guard<std::mutex> lk(g_pool_mu);
  auto& v = g_stream_pool[device];
  if (v.size() < 64) v.push_back(s);
  else (void)hipStreamDestroy(s);
}

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- caching device allocator (see common.hpp) ------------------------------------------------------------------
struct DevPool {
  std::multimap<size_t, void*> parked;  // size class -> block
  size_t parked_bytes = 0;
};
std::mutex g_mem_mu;
std::map<int, DevPool> g_mem_pool;
size_t pool_limit_bytes() {
  static const size_t lim = [] {
    const char* e = getenv("OSC_POOL_MB");
    const long long mb = e ? atoll(e) : 16384;
    return (size_t)std::max<long long>(0, mb) << 20;
  }();
  return lim;
}
size_t size_class(size_t bytes) {  // <= 12.5 % over-allocation, so equal shapes and near-equal ones share blocks
  if (bytes <= 4096) return 4096;
  size_t p2 = (size_t)1 << (63 - __builtin_clzll((unsigned long long)bytes));
  const size_t step = std::max<size_t>(p2 / 8, 4096);
  return (bytes + step - 1) / step * step;
}

// CG control block: pinned residual mirror + per-iteration events (hipHostMalloc ~0.3 ms, 66 x hipEventCreate); parked
// per device like the streams
struct CtrlBlock {
  float* res_host = nullptr;
  size_t res_host_n = 0;
  std::vector<hipEvent_t> events;
};
std::map<int, std::vector<CtrlBlock>> g_ctrl_pool;

}  // namespace

namespace osc {
AllocCtx& alloc_ctx() {
  static thread_local AllocCtx c;
  return c;
}
void* pool_alloc(size_t bytes, size_t* cap_bytes) {
  const size_t cls = size_class(bytes);
  const int dev = alloc_ctx().device;
  if (pool_limit_bytes() > 0) {
    std::lock_guard<std::mutex> lk(g_mem_mu);
    DevPool& dp = g_mem_pool[dev];
    auto it = dp.parked.find(cls);
    if (it != dp.parked.end()) {
      void* p = it->second;
      dp.parked.erase(it);
      dp.parked_bytes -= cls;
      *cap_bytes = cls;
      return p;
    }
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, cls);
  if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {  // give the parked blocks back and retry once
    (void)hipGetLastError();
    std::vector<void*> drop;
    {
      std::lock_guard<std::mutex> lk(g_mem_mu);
      DevPool& dp = g_mem_pool[dev];
      for (auto& kv : dp.parked) drop.push_back(kv.second);
      dp.parked.clear();
      dp.parked_bytes = 0;
    }
    for (void* q : drop) (void)hipFree(q);
    e = hipMalloc(&p, cls);
  }
  hip_check(e, "hipMalloc", __FILE__, __LINE__);
  *cap_bytes = cls;
  return p;
}
void pool_free(void* p, size_t cap_bytes) {
  if (!p) return;
  const AllocCtx& c = alloc_ctx();
  if (pool_limit_bytes() > 0 && cap_bytes > 0) {
    if (c.stream) (void)hipStreamSynchronize(c.stream);  // nothing of this handle may still touch the block
    std::lock_guard<std::mutex> lk(g_mem_mu);
    DevPool& dp = g_mem_pool[c.device];
    if (dp.parked_bytes + cap_bytes <= pool_limit_bytes()) {
      dp.parked.emplace(cap_bytes, p);
      dp.parked_bytes += cap_bytes;
      return;
    }
  }
  (void)hipFree(p);
}
}  // namespace osc

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

void osc_lattice::park_ctrl() {
  if (!res_host && iter_events.empty()) return;
  CtrlBlock cb;
  cb.res_host = res_host;
  cb.res_host_n = res_host_n;
  cb.events.swap(iter_events);
  res_host = nullptr;
  res_host_n = 0;
  std::lock_guard<std::mutex> lk(g_pool_mu);
  auto& v = g_ctrl_pool[device];
  if (v.size() < 64) {
    v.push_back(std::move(cb));
    return;
  }
  for (auto e : cb.events) (void)hipEventDestroy(e);
  if (cb.res_host) (void)hipHostFree(cb.res_host);
}

namespace {

using L = osc_lattice;

hipEvent_t prof_event(L& h) {
  if (!h.prof_pool.empty()) {
    hipEvent_t e = h.prof_pool.back();
    h.prof_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  HIP_CHECK(hipEventCreate(&e));
  return e;
}
// never throws when `nothrow` (the ProfScope destructor drains on overflow, and destructors must not throw): samples
// whose events cannot be read are dropped
void prof_drain(L& h, bool nothrow = false) {
  for (auto& s : h.prof_pending) {
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(s.b);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, s.a, s.b);
    if (e != hipSuccess) {
      if (!nothrow) hip_check(e, "profile event read-back", __FILE__, __LINE__);
      (void)hipGetLastError();
      s.which = -1;
    }
    if (s.which >= 0) {
      h.prof_count[s.which] += 1;
      h.prof_ms[s.which] += ms;
    }
    h.prof_pool.push_back(s.a);
    h.prof_pool.push_back(s.b);
  }
  h.prof_pending.clear();
}
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

void use_device(L& h) { HIP_CHECK(hipSetDevice(h.device)); }
void sync(L& h) { HIP_CHECK(hipStreamSynchronize(h.stream)); }

void upload_rows(L& h, float* dst, const float* src) {  // N x D host -> N x ld device
  if (h.ld == h.D) {  // no row padding: one contiguous copy (much faster than the strided form from pageable memory)
    HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)h.N * h.D * 4, hipMemcpyHostToDevice, h.stream));
    return;
  }
  // padded pitch: a strided copy from pageable host memory is several times slower than a contiguous one, so large
  // arrays go contiguous into a scratch array (R / P are free between solves) and are re-pitched on the device
  float* stage = (dst == h.R.p) ? h.P.p : h.R.p;
  if ((int64_t)h.N * h.D >= ((int64_t)1 << 20) && stage != nullptr && stage != dst) {
    HIP_CHECK(hipMemcpyAsync(stage, src, (size_t)h.N * h.D * 4, hipMemcpyHostToDevice, h.stream));
    HIP_CHECK(hipMemcpy2DAsync(dst, (size_t)h.ld * 4, stage, (size_t)h.D * 4, (size_t)h.D * 4, (size_t)h.N,
                               hipMemcpyDeviceToDevice, h.stream));
    return;
  }
  HIP_CHECK(hipMemcpy2DAsync(dst, (size_t)h.ld * 4, src, (size_t)h.D * 4, (size_t)h.D * 4, (size_t)h.N,
                             hipMemcpyHostToDevice, h.stream));
}
// ---- large device -> host transfers: pinned staging, chunked, the DMA of chunk c + 1 beside the host copy of chunk c ----
// A device-to-host copy into PAGEABLE memory (what a caller's NumPy array is) runs at 6-10 GB/s through the runtime's own
// staging: reading the 307 MB state of config 3 back took 30-50 ms for a 5 ms solve.  Two pinned buffers per process and
// device (parked like the streams; 2 x 32 MiB) take the DMA at PCIe rate while a few host threads copy the previous chunk
// into the caller's array -- whose pages are usually untouched, so the copy is also what faults them in, and that is what
// the threads are for.  OSC_PINNED_DL=0 keeps the plain copy.
constexpr size_t kStageBytes = (size_t)32 << 20;
struct StagePair {
  void* buf[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
};
std::map<int, std::vector<StagePair>> g_stage_pool;  // guarded by g_pool_mu

StagePair acquire_stage(int device) {
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto& v = g_stage_pool[device];
    if (!v.empty()) {
      StagePair sp = v.back();
      v.pop_back();
      return sp;
    }
  }
  StagePair sp;
  for (int i = 0; i < 2; ++i) {
    HIP_CHECK(hipHostMalloc(&sp.buf[i], kStageBytes, hipHostMallocDefault));
    HIP_CHECK(hipEventCreateWithFlags(&sp.ev[i], hipEventDisableTiming));
  }
  return sp;
}
void release_stage(int device, const StagePair& sp) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  auto& v = g_stage_pool[device];
  if (v.size() < 4) {
    v.push_back(sp);
    return;
  }
  for (int i = 0; i < 2; ++i) {
    (void)hipHostFree(sp.buf[i]);
    (void)hipEventDestroy(sp.ev[i]);
  }
}

// Pinned host arrays for results (osc_host_alloc): the Python layer hands them out as the NumPy arrays `lat.U`, `lat.Y` and
// solve_Ustar() return, so a read-back is ONE DMA at PCIe rate with no host copy and no page faults behind it.  Pinning
// is slow (tens of ms for 300 MB), so freed arrays are parked per size class and handed out again; at most kHostParkBytes
// stay parked.
constexpr size_t kHostParkBytes = (size_t)4 << 30;
struct HostBlock {
  void* p;
  size_t bytes;
};
std::vector<HostBlock> g_host_parked;       // guarded by g_pool_mu
std::map<void*, size_t> g_host_live;        // blocks handed out (pointer -> capacity)
size_t g_host_parked_bytes = 0;

void* host_pool_alloc(size_t bytes) {
  const size_t cap = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_host_parked.size(); ++i)
      if (g_host_parked[i].bytes >= cap && g_host_parked[i].bytes <= cap + cap / 8) {
        const HostBlock b = g_host_parked[i];
        g_host_parked.erase(g_host_parked.begin() + (long)i);
        g_host_parked_bytes -= b.bytes;
        g_host_live[b.p] = b.bytes;
        return b.p;
      }
  }
  void* p = nullptr;
  if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  std::lock_guard<std::mutex> lk(g_pool_mu);
  g_host_live[p] = cap;
  return p;
}
bool host_pool_free(void* p) {
  std::vector<void*> release;  // unpinned OUTSIDE the lock: hipHostFree can take milliseconds, and g_pool_mu also guards the
  {                            // stream and staging pools of every handle
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_host_live.find(p);
    if (it == g_host_live.end()) return false;
    const size_t cap = it->second;
    g_host_live.erase(it);
    if (cap <= kHostParkBytes) {  // park it; the blocks parked longest make room (a workload's current size class stays)
      while (g_host_parked_bytes + cap > kHostParkBytes && !g_host_parked.empty()) {
        release.push_back(g_host_parked.front().p);
        g_host_parked_bytes -= g_host_parked.front().bytes;
        g_host_parked.erase(g_host_parked.begin());
      }
      g_host_parked.push_back(HostBlock{p, cap});
      g_host_parked_bytes += cap;
    } else {
      release.push_back(p);
    }
  }
  for (void* q : release) (void)hipHostFree(q);
  return true;
}
bool host_pool_owns(const void* p, size_t bytes) {  // [p, p + bytes) lies inside a block this pool handed out
  std::lock_guard<std::mutex> lk(g_pool_mu);
  auto it = g_host_live.upper_bound(const_cast<void*>(p));
  if (it == g_host_live.begin()) return false;
  --it;
  const char* b = static_cast<const char*>(it->first);
  return static_cast<const char*>(p) >= b && static_cast<const char*>(p) + bytes <= b + it->second;
}

void parallel_copy(char* dst, const char* src, size_t bytes, int threads) {
  if (threads <= 1 || bytes < ((size_t)4 << 20)) {
    std::memcpy(dst, src, bytes);
    return;
  }
  const size_t per = ((bytes / (size_t)threads) + 4095) & ~(size_t)4095;
  std::vector<std::thread> ts;
  for (int t = 1; t < threads; ++t) {
    const size_t off = per * (size_t)t;
    if (off >= bytes) break;
    ts.emplace_back([=] { std::memcpy(dst + off, src + off, std::min(per, bytes - off)); });
  }
  std::memcpy(dst, src, std::min(per, bytes));
  for (auto& t : ts) t.join();
}

// contiguous device array -> host array, returns when the host array is complete
void download_contiguous(L& h, char* dst, const char* src, size_t bytes) {
  static const bool pinned = [] { const char* e = getenv("OSC_PINNED_DL"); return !(e && atoi(e) == 0); }();
  if (!pinned || bytes < 2 * kStageBytes || host_pool_owns(dst, bytes)) {  // (a pinned destination takes the DMA directly)
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h.stream));
    sync(h);
    return;
  }
  static const int threads = [] {
    const char* e = getenv("OSC_COPY_THREADS");
    const int hw = (int)std::thread::hardware_concurrency();
    return e ? std::max(1, atoi(e)) : std::max(1, std::min(8, hw / 2));
  }();
  const StagePair sp = acquire_stage(h.device);
  try {
    const size_t nchunks = (bytes + kStageBytes - 1) / kStageBytes;
    auto issue = [&](size_t c) {
      const size_t off = c * kStageBytes;
      HIP_CHECK(hipMemcpyAsync(sp.buf[c & 1], src + off, std::min(kStageBytes, bytes - off), hipMemcpyDeviceToHost, h.stream));
      HIP_CHECK(hipEventRecord(sp.ev[c & 1], h.stream));
    };
    issue(0);
    for (size_t c = 0; c < nchunks; ++c) {
      if (c + 1 < nchunks) issue(c + 1);  // (its buffer was emptied by the host copy of chunk c - 1)
      HIP_CHECK(hipEventSynchronize(sp.ev[c & 1]));
      const size_t off = c * kStageBytes;
      parallel_copy(dst + off, static_cast<const char*>(sp.buf[c & 1]), std::min(kStageBytes, bytes - off), threads);
    }
  } catch (...) {
    (void)hipStreamSynchronize(h.stream);
    release_stage(h.device, sp);
    throw;
  }
  release_stage(h.device, sp);
}

// N x ld device array -> N x D host array; returns when the host array is complete
void download_rows(L& h, float* dst, const float* src) {
  if (h.ld == h.D) {
    download_contiguous(h, reinterpret_cast<char*>(dst), reinterpret_cast<const char*>(src), (size_t)h.N * h.D * 4);
    return;
  }
  float* stage = (src == h.R.p) ? h.P.p : h.R.p;
  if ((int64_t)h.N * h.D >= ((int64_t)1 << 20) && stage != nullptr && stage != src) {
    HIP_CHECK(hipMemcpy2DAsync(stage, (size_t)h.D * 4, src, (size_t)h.ld * 4, (size_t)h.D * 4, (size_t)h.N,
                               hipMemcpyDeviceToDevice, h.stream));
    download_contiguous(h, reinterpret_cast<char*>(dst), reinterpret_cast<const char*>(stage), (size_t)h.N * h.D * 4);
    return;
  }
  HIP_CHECK(hipMemcpy2DAsync(dst, (size_t)h.D * 4, src, (size_t)h.ld * 4, (size_t)h.D * 4, (size_t)h.N,
                             hipMemcpyDeviceToHost, h.stream));
  sync(h);
}

// per-row host vector that came back in device row order -> API row order (in place)
void to_api_order(const L& h, float* v) {
  if (h.perm_h.empty() || !v) return;
  std::vector<float> t(v, v + h.N);
  for (int64_t i = 0; i < h.N; ++i) v[h.perm_h[(size_t)i]] = t[(size_t)i];
}

// N x D device array (device row order) -> host array in API row order; AP is scratch between solves
void download_api_order(L& h, float* dst, const float* src) {
  if (!h.perm_h.empty()) {
    launch_move_rows(h.AP.p, src, h.perm_d.p, h.N, h.ld, true, h.stream);  // AP[perm[i]] = src[i]
    src = h.AP.p;
  }
  download_rows(h, dst, src);
  sync(h);
}

// residual slots + arrival counters on the device, their host-mapped mirror (the device publishes each iteration's
// residual into it; also the read-back buffer of the one-launch path) and the per-iteration events
// the second stream of an overlapped sharded solve (run_cg) writes residual slots and their host mirror: it must be idle
// before those are cleared, resized or handed to another path
void drain_comm_stream(L& h) {
  if (!h.comm_stream_busy) return;
  HIP_CHECK(hipStreamSynchronize(h.comm_stream));
  h.comm_stream_busy = false;
}

constexpr int OSC_CTRL_RING = 32;
// `words` zeroed control words for one solve (valid until OSC_CTRL_RING further solves have taken theirs)
uint32_t* ctrl_segment(L& h, size_t words) {
  drain_comm_stream(h);  // (the second stream of a sharded solve may still write the previous solve's words)
  const size_t seg = (words + 63) / 64 * 64;
  if (h.ctrl_seg < seg || h.ctrl_next >= OSC_CTRL_RING || h.ctrl_ring.p == nullptr) {
    if (h.ctrl_seg < seg) {
      if (h.ctrl_ring.p != nullptr) sync(h);  // (launches of earlier solves may still read their gates from the old ring)
      h.ctrl_seg = seg;
      h.ctrl_ring.alloc(seg * OSC_CTRL_RING);
    }
    HIP_CHECK(hipMemsetAsync(h.ctrl_ring.p, 0, h.ctrl_seg * OSC_CTRL_RING * 4, h.stream));  // behind every earlier solve's launches
    h.ctrl_next = 0;
  }
  return h.ctrl_ring.p + (size_t)(h.ctrl_next++) * h.ctrl_seg;
}

void ensure_ctrl(L& h, size_t slots) {
  drain_comm_stream(h);
  if (h.res_bits.n < 2 * slots) h.res_bits.alloc(2 * slots);  // [residual bits | arrival counters]
  if (!h.res_host && h.iter_events.empty()) {  // a parked control block of a destroyed handle, if any
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto& v = g_ctrl_pool[h.device];
    if (!v.empty()) {
      h.res_host = v.back().res_host;
      h.res_host_n = v.back().res_host_n;
      h.iter_events.swap(v.back().events);
      v.pop_back();
    }
  }
  if (h.res_host_n < slots) {
    if (h.res_host) (void)hipHostFree(h.res_host);
    h.res_host = nullptr;
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h.res_host), slots * 4, hipHostMallocMapped));
    h.res_host_n = slots;
  }
  HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&h.res_host_dev), h.res_host, 0));
  while (h.iter_events.size() < slots) {
    hipEvent_t e;
    HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    h.iter_events.push_back(e);
  }
}

void ensure_cg_scratch(L& h, int max_iters) {
  const size_t pn = (size_t)(h.grid_cap + OSC_CHAIN_FIX_MAX_CHUNKS) * h.ld;  // + the chain fix-up's rows beside the blocked apply
  h.part0.alloc(pn);
  h.part1.alloc(pn);
  h.alpha.alloc(h.ld);
  h.beta.alloc(h.ld);
  h.rz.alloc(h.ld);
  h.colsum.alloc(h.ld);
  // sized for solve_Ustar's default 64 iterations from the start: a settle(12) followed by a U* solve must not pay
  // for re-allocating the residual slots, their pinned mirror and the per-iteration events
  ensure_ctrl(h, (size_t)std::max(max_iters, 64) + 2);
}

// one grid for every CG kernel of a handle, so all column partial buffers have the same number of rows
int cg_grid(const L& h) {
  int64_t g = std::max<int64_t>(1, std::min<int64_t>((h.N + 3) / 4, h.grid_cap));
  if (g >= 8) g &= ~(int64_t)7;  // multiple of 8: the operator apply maps workgroups to XCDs by blockIdx % 8
  return (int)g;
}

GraphView graph_view(L& h, bool with_path) {
  GraphView g{};
  g.col = h.ell_col.p;
  g.w = h.ell_w.p;
  g.deg = h.deg.p;
  g.width = h.width;
  if (with_path) {
    g.path_slot = h.path_slot.p;
    g.pcol = h.pcol.p;
    g.pw = h.pw.p;
    g.pdeg = h.pdeg.p;
    g.pwidth = h.pwidth;
  }
  return g;
}

// ---- graph build --------------------------------------------------------------------------------
void graph_counts(L& h) {
  std::vector<int32_t> d((size_t)h.N);
  HIP_CHECK(hipMemcpyAsync(d.data(), h.deg.p, (size_t)h.N * 4, hipMemcpyDeviceToHost, h.stream));
  sync(h);
  int64_t nnz = 0;
  int32_t mx = 0;
  for (auto v : d) {
    nnz += v;
    mx = std::max(mx, v);
  }
  h.nnz = nnz;
  h.max_deg = mx;
}

void alloc_ell(L& h, int32_t width) {
  h.ell_t_ready = false;
  h.blk_nb = 0;
  ++h.graph_epoch;
  h.width = std::max<int32_t>(1, width);
  const size_t n = (size_t)h.N * h.width;
  h.ell_col.alloc(n);
  h.ell_a.alloc(n);
  h.ell_w.alloc(n);
  h.deg.alloc((size_t)h.N);
  h.sqrt_deg.alloc((size_t)h.N);
  HIP_CHECK(hipMemsetAsync(h.ell_col.p, 0, n * 4, h.stream));
  HIP_CHECK(hipMemsetAsync(h.ell_a.p, 0, n * 4, h.stream));
  HIP_CHECK(hipMemsetAsync(h.ell_w.p, 0, n * 4, h.stream));
  HIP_CHECK(hipMemsetAsync(h.deg.p, 0, (size_t)h.N * 4, h.stream));
}

bool permuted(const L& h) { return !h.perm_h.empty(); }

// path Laplacian structures from the stored chain (graph.py:96-111), in the handle's current row order
void install_chain(L& l) {
  ++l.graph_epoch;
  if (!l.chain_present) return;
  const int32_t len = (int32_t)l.chain_nodes.size();
  auto id = [&](int32_t v) { return permuted(l) ? l.inv_h[(size_t)v] : v; };
  // path adjacency, duplicate edges keep the max weight (graph.py:102-109)
  std::map<std::pair<int32_t, int32_t>, float> adj;
  for (int t = 0; t + 1 < len; ++t) {
    const int32_t i = id(l.chain_nodes[(size_t)t]), j = id(l.chain_nodes[(size_t)t + 1]);
    const float w = l.chain_w.empty() ? 1.0f : l.chain_w[(size_t)t];
    auto put = [&](int32_t r, int32_t c) {
      auto it = adj.find({r, c});
      if (it == adj.end()) adj[{r, c}] = std::max(0.0f, w);
      else it->second = std::max(it->second, w);
    };
    put(i, j);
    put(j, i);
  }
  // normalized_laplacian(A_path) (graph.py:86-93): only rows that own an entry differ from identity
  std::map<int32_t, float> dsum;
  for (auto& kv : adj) dsum[kv.first.first] += kv.second;
  std::map<int32_t, int32_t> slot;
  for (auto& kv : dsum) slot.emplace(kv.first, (int32_t)slot.size());
  std::map<int32_t, int32_t> cnt;
  int32_t pwidth = 1;
  for (auto& kv : adj) pwidth = std::max(pwidth, ++cnt[kv.first.first]);
  const int32_t prows = (int32_t)slot.size();
  std::vector<int32_t> hslot((size_t)l.N, -1), hcol((size_t)prows * pwidth, 0), hdeg((size_t)prows, 0);
  std::vector<float> hw((size_t)prows * pwidth, 0.f);
  auto sd = [&](int32_t r) {
    auto it = dsum.find(r);
    return std::sqrt(std::max(it == dsum.end() ? 0.0f : it->second, 1e-12f));
  };
  std::vector<int32_t> hprow((size_t)prows, 0);
  for (auto& kv : slot) hslot[(size_t)kv.first] = kv.second, hprow[(size_t)kv.second] = kv.first;
  for (auto& kv : adj) {
    const int32_t r = kv.first.first, c = kv.first.second, sl = slot[r];
    const int32_t e = hdeg[(size_t)sl]++;
    hcol[(size_t)sl * pwidth + e] = c;
    hw[(size_t)sl * pwidth + e] = (kv.second * (1.0f / sd(r))) * (1.0f / sd(c));
  }
  l.path_slot.alloc((size_t)l.N);
  l.pcol.alloc(hcol.size());
  l.pw.alloc(hw.size());
  l.pdeg.alloc(hdeg.size());
  l.prow.alloc(hprow.size());
  HIP_CHECK(hipMemcpyAsync(l.prow.p, hprow.data(), hprow.size() * 4, hipMemcpyHostToDevice, l.stream));
  HIP_CHECK(hipMemcpyAsync(l.path_slot.p, hslot.data(), hslot.size() * 4, hipMemcpyHostToDevice, l.stream));
  HIP_CHECK(hipMemcpyAsync(l.pcol.p, hcol.data(), hcol.size() * 4, hipMemcpyHostToDevice, l.stream));
  HIP_CHECK(hipMemcpyAsync(l.pw.p, hw.data(), hw.size() * 4, hipMemcpyHostToDevice, l.stream));
  HIP_CHECK(hipMemcpyAsync(l.pdeg.p, hdeg.data(), hdeg.size() * 4, hipMemcpyHostToDevice, l.stream));
  sync(l);
  l.prows = prows;
  l.pwidth = pwidth;
}

// move every row-indexed device array between two row orders: new row i takes old row from[i]; ids -> relabel[id]
void move_state(L& l, const int32_t* from_d, const int32_t* relabel_d) {
  const size_t n = (size_t)l.N * l.ld;
  for (DevBuf<float>* b : {&l.Y, &l.U}) {  // AP is scratch between solves
    launch_move_rows(l.AP.p, b->p, from_d, l.N, l.ld, false, l.stream);
    HIP_CHECK(hipMemcpyAsync(b->p, l.AP.p, n * 4, hipMemcpyDeviceToDevice, l.stream));
  }
  DevBuf<float> t1;
  t1.alloc((size_t)l.N);
  for (DevBuf<float>* b : {&l.B, &l.sqrt_deg}) {
    launch_move_f32(t1.p, b->p, from_d, l.N, false, l.stream);
    HIP_CHECK(hipMemcpyAsync(b->p, t1.p, (size_t)l.N * 4, hipMemcpyDeviceToDevice, l.stream));
  }
  const size_t ne = (size_t)l.N * l.width;
  DevBuf<int32_t> col2, deg2;
  DevBuf<float> a2, w2;
  col2.alloc(ne);
  a2.alloc(ne);
  w2.alloc(ne);
  deg2.alloc((size_t)l.N);
  launch_permute_ell(l.ell_col.p, l.ell_a.p, l.ell_w.p, l.deg.p, from_d, relabel_d, l.width, l.N, col2.p, a2.p, w2.p,
                     deg2.p, l.stream);
  sync(l);
  l.ell_col.swap(col2);
  l.ell_a.swap(a2);
  l.ell_w.swap(w2);
  l.deg.swap(deg2);
  l.ell_t_ready = false;
  l.blk_nb = 0;
  l.have_ustar = false;
  l.u_sharded = false;
  ++l.graph_epoch;
}

void drop_order(L& l) {  // back to the API's row order
  if (!permuted(l)) return;
  move_state(l, l.inv_d.p, l.perm_d.p);
  l.perm_h.clear();
  l.inv_h.clear();
  install_chain(l);
}

void apply_order(L& l, const std::vector<int32_t>& perm) {  // perm[new] = old ; state must be in API order
  l.perm_h = perm;
  l.inv_h.assign((size_t)l.N, 0);
  for (int64_t i = 0; i < l.N; ++i) l.inv_h[(size_t)perm[(size_t)i]] = (int32_t)i;
  l.perm_d.alloc((size_t)l.N);
  l.inv_d.alloc((size_t)l.N);
  HIP_CHECK(hipMemcpyAsync(l.perm_d.p, l.perm_h.data(), (size_t)l.N * 4, hipMemcpyHostToDevice, l.stream));
  HIP_CHECK(hipMemcpyAsync(l.inv_d.p, l.inv_h.data(), (size_t)l.N * 4, hipMemcpyHostToDevice, l.stream));
  move_state(l, l.perm_d.p, l.inv_d.p);
  install_chain(l);
}

// breadth-first order over the lattice graph (components in order of their smallest node): neighbours end up
// within a narrow band of rows, which is what the XCD-local L2 of the operator apply can hold
std::vector<int32_t> bfs_order(L& l) {
  const size_t ne = (size_t)l.N * l.width;
  std::vector<int32_t> col(ne), deg((size_t)l.N);
  HIP_CHECK(hipMemcpyAsync(col.data(), l.ell_col.p, ne * 4, hipMemcpyDeviceToHost, l.stream));
  HIP_CHECK(hipMemcpyAsync(deg.data(), l.deg.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
  sync(l);
  std::vector<int32_t> order;
  order.reserve((size_t)l.N);
  std::vector<char> seen((size_t)l.N, 0);
  for (int64_t start = 0; start < l.N; ++start) {
    if (seen[(size_t)start]) continue;
    seen[(size_t)start] = 1;
    size_t head = order.size();
    order.push_back((int32_t)start);
    while (head < order.size()) {
      const int32_t u = order[head++];
      const int32_t* cu = col.data() + (size_t)u * l.width;
      for (int e = 0; e < deg[(size_t)u]; ++e) {
        const int32_t v = cu[e];
        if (!seen[(size_t)v]) {
          seen[(size_t)v] = 1;
          order.push_back(v);
        }
      }
    }
  }
  return order;
}

// Re-order the rows when it pays: the BFS + state move cost a few ms at N = 100k and buy ~1.5x on the operator apply
// of a clustered lattice, nothing on an unstructured one.  Auto mode decides on a sampled clustering coefficient.
void maybe_reorder(L& l) {
  l.reordered = false;
  l.clustering = 0.0;
  // under a communicator only the row-sharded CG re-orders (every rank holds the same graph and takes the same
  // deterministic decision and order; the halo lists shrink with locality); the column-sharded default keeps API order
  if (l.reorder == 0 || (l.comm != nullptr && l.shard_mode != 1) || l.N < 2) return;
  if (l.reorder < 0) {
    if (l.N < 8192 || l.nnz == 0) return;  // small lattices run out of LDS / L2 anyway
    DevBuf<unsigned long long> cnt;
    cnt.alloc(2);
    HIP_CHECK(hipMemsetAsync(cnt.p, 0, 16, l.stream));
    launch_clustering_sample(l.ell_col.p, l.deg.p, l.width, l.N, 1024, cnt.p, l.stream);
    unsigned long long hc[2] = {0, 0};
    HIP_CHECK(hipMemcpyAsync(hc, cnt.p, 16, hipMemcpyDeviceToHost, l.stream));
    sync(l);
    l.clustering = hc[1] ? (double)hc[0] / (double)hc[1] : 0.0;
    if (l.clustering < 0.05) return;
  }
  // the order itself: on the device (bfs_order.hip; the same order as the host walk below it, OSC_BFS_HOST=1 forces that)
  bool on_device = false;
  if (!l.bfs_host) {
    DevBuf<int32_t> perm;
    perm.alloc((size_t)l.N);
    if (device_bfs_order(l.ell_col.p, l.deg.p, l.width, (int32_t)l.N, perm.p, l.stream)) {
      std::vector<int32_t> ph((size_t)l.N);
      HIP_CHECK(hipMemcpyAsync(ph.data(), perm.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
      sync(l);
      apply_order(l, ph);
      on_device = true;
    }
  }
  if (!on_device) apply_order(l, bfs_order(l));
  l.reordered = true;
}

// Sharded half sweep: every rank holds partial buckets of ALL rows; rank q needs the other ranks' entries of the buckets of
// ITS row blocks, [4 rb_per q, 4 rb_per (q + 1)).  Raw counts all-gathered (they also carry overflow: a count above the
// capacity stays above it in the sum); each rank packs its buckets behind one another (the ranges are in rank order, so
// one prefix sum gives every destination's segment), grouped send / recv, then the received entries are appended behind
// the rank's own, in rank order.  The chunk overflow flags are combined by max.
void exchange_buckets(L& h, const KnnPanelPlan& pp, const KnnPanelSymDev& sd, int rb_per) {
  const int G = h.world, me = h.rank;
  const int32_t nb_all = pp.npad / 32, nbp = rb_per * 4, stride = nbp * G, cap = pp.bucket_cap;
  auto b0 = [&](int q) { return std::min(nb_all, q * nbp); };
  DevBuf<int32_t> all_cnt, clamped, off, sums, src_off_d;
  DevBuf<int64_t> seg_off_d;
  all_cnt.alloc((size_t)G * stride);
  HIP_CHECK(hipMemsetAsync(all_cnt.p, 0, (size_t)G * stride * 4, h.stream));
  HIP_CHECK(hipMemcpyAsync(all_cnt.p + (size_t)me * stride, sd.bucket_cnt, (size_t)nb_all * 4, hipMemcpyDeviceToDevice, h.stream));
  h.comm->allgather(all_cnt.p, (size_t)stride * 4, h.stream);
  std::vector<int32_t> cnt((size_t)G * stride);
  HIP_CHECK(hipMemcpyAsync(cnt.data(), all_cnt.p, cnt.size() * 4, hipMemcpyDeviceToHost, h.stream));
  // my buckets, packed: off[b] = entries before bucket b
  clamped.alloc((size_t)nb_all);
  off.alloc((size_t)nb_all);
  sums.alloc(scan_blocks(nb_all) + 1);
  launch_bucket_clamp(sd.bucket_cnt, nb_all, cap, clamped.p, h.stream);
  exclusive_scan_i32(clamped.p, off.p, nb_all, sums.p, h.stream);
  sync(h);
  auto held = [&](int p, int b) { return (int64_t)std::min(cnt[(size_t)p * stride + b], cap); };
  std::vector<int64_t> seg_start((size_t)G + 1, 0);  // my packed buffer: where each destination's segment starts
  for (int q = 0; q < G; ++q) {
    int64_t n = 0;
    for (int b = b0(q); b < b0(q + 1); ++b) n += held(me, b);
    seg_start[(size_t)q + 1] = seg_start[(size_t)q] + n;
  }
  const int32_t nb_mine = b0(me + 1) - b0(me);
  std::vector<int64_t> seg_off((size_t)G + 1, 0);                    // the receive buffer: one segment per source rank
  std::vector<int32_t> src_off((size_t)G * std::max(1, nb_mine), 0);  // (source, my bucket) -> offset inside that segment
  for (int p = 0; p < G; ++p) {
    int64_t n = 0;
    for (int w = 0; w < nb_mine; ++w) {
      src_off[(size_t)p * nb_mine + w] = (int32_t)n;
      if (p != me) n += held(p, b0(me) + w);
    }
    if (n >= ((int64_t)1 << 31)) throw Unsupported("sharded half sweep: more than 2^31 hits for one rank's rows from one peer");
    seg_off[(size_t)p + 1] = seg_off[(size_t)p] + n;
  }
  DevBuf<unsigned long long> send, recv;
  send.alloc((size_t)std::max<int64_t>(1, seg_start[(size_t)G]));
  recv.alloc((size_t)std::max<int64_t>(1, seg_off[(size_t)G]));
  launch_bucket_pack(sd.bucket_ent, sd.bucket_cnt, off.p, nb_all, cap, send.p, h.stream);
  std::vector<CommXfer> sends, recvs;
  for (int q = 0; q < G; ++q) {
    if (q == me) continue;
    const int64_t ns = seg_start[(size_t)q + 1] - seg_start[(size_t)q], nr = seg_off[(size_t)q + 1] - seg_off[(size_t)q];
    if (ns > 0) sends.push_back(CommXfer{send.p + seg_start[(size_t)q], (size_t)ns * 8, q});
    if (nr > 0) recvs.push_back(CommXfer{recv.p + seg_off[(size_t)q], (size_t)nr * 8, q});
  }
  h.comm->exchange(sends, recvs, h.stream);
  if (nb_mine > 0) {
    src_off_d.alloc(src_off.size());
    seg_off_d.alloc(seg_off.size());
    HIP_CHECK(hipMemcpyAsync(src_off_d.p, src_off.data(), src_off.size() * 4, hipMemcpyHostToDevice, h.stream));
    HIP_CHECK(hipMemcpyAsync(seg_off_d.p, seg_off.data(), seg_off.size() * 8, hipMemcpyHostToDevice, h.stream));
    launch_bucket_merge(sd.bucket_ent, sd.bucket_cnt, all_cnt.p, src_off_d.p, seg_off_d.p, recv.p, b0(me), nb_mine, stride, cap, me, G,
                        h.stream);
  }
  h.comm->allreduce(sd.flags, (size_t)pp.S, COMM_I32, COMM_MAX, h.stream);
  sync(h);  // the host vectors and the temporaries above are in use until here
}

void build_graph(L& h) {
  const double t0 = now_ms();
  drop_order(h);  // the build works on the API's row order
  const int32_t N = (int32_t)h.N;
  h.k_eff = std::min<int32_t>(h.k_eff, std::max<int32_t>(1, N - 1));  // lattice.py:60
  h.have_ustar = false;
  if (N <= 1) {  // graph.py:30-32
    alloc_ell(h, 1);
    const float one_em6 = 1e-6f;  // sqrt(max(0, 1e-12))
    std::vector<float> sd((size_t)h.N, one_em6);
    HIP_CHECK(hipMemcpyAsync(h.sqrt_deg.p, sd.data(), sd.size() * 4, hipMemcpyHostToDevice, h.stream));
    sync(h);
    h.knn_k = 0;
    h.have_graph = true;
    h.nnz = 0;
    h.max_deg = 0;
    h.build_ms = now_ms() - t0;
    return;
  }
  const int32_t k = h.k_eff;
  // k <= 128: register-resident streaming lists (exact / prefilter / small-dense routes below).  Larger k (the
  // reference takes any k <= N - 1, lattice.py:60): dense similarity rows in chunks + a radix select per row.
  const bool any_k = k > 128;
  const int32_t ldn = ((h.D + 31) / 32) * 32;
  DevBuf<float> Yn;
  Yn.alloc((size_t)h.N * ldn);
  launch_normalize_rows(h.Y.p, h.ld, Yn.p, ldn, h.N, h.D, h.stream);
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, h.device));
  const int slots = prop.multiProcessorCount * (k <= 64 ? 2 : 1);
  // multi-GPU: row-block-sharded build -- this rank computes the top-k lists of its 128-row blocks against all
  // columns, then one all-gather of the (idx, sim) lists; mutual test / cap / Laplacian weights run on every rank.
  const int all_rb = (N + 127) / 128;
  // OSC_KNN_FAKE_SHARDS=G (test hook): run the G per-rank passes of a sharded build one after another on this GPU
  const int fake = h.knn_fake_shards;
  const bool sharded = h.comm != nullptr && h.world > 1;
  const int parts = sharded ? h.world : (fake > 1 ? fake : 1);
  const int rb_per = (all_rb + parts - 1) / parts;
  const size_t list_rows = parts > 1 ? (size_t)rb_per * 128 * parts : (size_t)h.N;
  h.knn_val.alloc(list_rows * k);
  h.knn_idx.alloc(list_rows * k);
  h.knn_k = k;
  HIP_CHECK(hipMemsetAsync(h.knn_val.p, 0, list_rows * k * 4, h.stream));
  HIP_CHECK(hipMemsetAsync(h.knn_idx.p, 0xFF, list_rows * k * 4, h.stream));
  // Two ways to the per-row top-k lists (identical results):
  //  exact     : fp32 MFMA similarity tiles + running top-k.
  //  prefilter : fp16 MFMA tiles keep the best KC >= k+16 candidates per row, exact fp32 re-scoring picks the k;
  //              a row is accepted only if the worst-case fp16 error bound proves no left-out column can belong
  //              to its top-k, otherwise the row is redone by the exact kernel.
  // kept candidates per row: k plus a margin; rows whose margin turns out too thin are redone exactly
  const int keep_f = std::min(96, k + std::max(12, k / 2));
  constexpr bool dense_small = true;
  constexpr int dense_max = 8192;
  // small lattices go through the dense similarity matrix (below); beyond that the fp16 prefilter pays
  bool prefilter = (keep_f >= k + 8) && N >= 4096 && !(dense_small && parts == 1 && N <= dense_max);
  // OSC_KNN_MODE = exact | prefilter | panel: force one route (tests, A/B)
  if (h.knn_mode == 1) prefilter = false;
  if (h.knn_mode == 2) prefilter = (keep_f >= k + 8);
  if (any_k) prefilter = false;
  // The prefilter's GEMM has two shapes: "panel" (knn_gemm.hip: query panel in registers, thresholds from a column
  // sample, hits appended -- D <= 768 and enough row blocks for the sample) and the older 128 x 128 tile with
  // register-resident sorted lists (k_knn_pref), which serves everything else.
  constexpr int panel_min = 8193;  // (up to 8192 rows: the dense route)
  // (a hit entry packs the column index into 25 bits, next to its two side flags)
  // (D > 768: the same route on the tile core, k_tile_thr -- half sweep only, so single-process builds only)
  // (round 5: the half sweep also under sharding -- the ranks split the work ITEMS of the one sweep and exchange the hits of
  // each other's rows, below -- so a sharded build issues the single-GPU build's MFMA work, not twice it, and D > 768 keeps
  // the threshold route instead of falling back to the list-maintaining tile prefilter)
  bool sym_ok = h.knn_sym;
  if (sym_ok && prefilter && N >= panel_min && N < (1 << 25)) {
    // The half sweep delivers every hit to a bucket per 32 receiving rows: (npad / 32) x bucket_cap entries of 8 bytes --
    // 2.9 GB of temporaries at N = 1M (config 4), growing with N x the threshold sample's hit bound (the full sweep's
    // lists: 0.2-0.5 GB).  Beyond a budget, or where the device cannot spare it, the build takes the full sweep (D <= 768)
    // or the tile prefilter (D > 768) instead of failing in the allocator.
    const KnnPanelPlan sp = knn_panel_plan(N, h.D, keep_f, prop.multiProcessorCount, false, true, h.knn_tune);
    const double bucket_bytes = (double)(sp.npad / 32) * (double)sp.bucket_cap * 8.0;
    size_t mem_free = 0, mem_total = 0;
    HIP_CHECK(hipMemGetInfo(&mem_free, &mem_total));
    constexpr double kSymBucketBudget = 12.0 * 1024 * 1024 * 1024;
    if (sp.ok && (bucket_bytes > kSymBucketBudget || bucket_bytes > 0.5 * (double)mem_free)) sym_ok = false;
  }
  const bool depth_ok = knn_panel_nkt(h.D) != 0 || (sym_ok && knn_tile_nkt(h.D) != 0);
  bool panel = prefilter && depth_ok && N >= panel_min && N < (1 << 25) &&
               knn_panel_plan(N, h.D, keep_f, prop.multiProcessorCount, false, sym_ok, h.knn_tune).ok;
  if (h.knn_mode == 3) panel = prefilter = (keep_f >= k + 8) && !any_k && depth_ok && N >= 6144 && N < (1 << 25);
  if (h.knn_mode == 2) panel = false;
  h.knn_panel = panel;
  DevBuf<float> cand_val, cval;
  DevBuf<int32_t> cand_idx, cidx, fail_rows, fail_count;
  DevBuf<float> Yh;  // fp16 image, viewed as float slots
  const int32_t ldh = ((h.D + 63) / 64) * 64;
  h.knn_fallback_rows = 0;
  h.knn_prefilter = prefilter;
  KnnPanelPlan pp{};
  DevBuf<float> p_img, p_smp, p_tmax, p_tau;
  DevBuf<unsigned long long> p_hits;
  DevBuf<int32_t> p_hcnt;
  DevBuf<unsigned> p_queue;
  KnnPanelSymDev sym_dev{};
  if (panel) {
    // (image rows scattered over the lattice rows in single-process builds: knn_gemm.hpp, KnnPanelPlan::scatter)
    // single-process builds sweep only the column tiles J >= I of every row block (knn_gemm.hip: symmetric half sweep);
    // a sharded build's ranks own row blocks and would have to exchange the column-side hits, so they keep the full sweep
    // (OSC_KNN_PANEL_SCATTER=0 / OSC_KNN_PANEL_SYM=0: A/B and tests)
    pp = knn_panel_plan(N, h.D, keep_f, prop.multiProcessorCount, h.knn_scatter && parts == 1, sym_ok, h.knn_tune);
    p_img.alloc((size_t)(pp.npad + 128) * pp.ldh / 2);  // (+ one zero tile: k_tile_thr2 sweeps row blocks and column tiles in pairs)
    HIP_CHECK(hipMemsetAsync(p_img.p + (size_t)pp.npad * pp.ldh / 2, 0, (size_t)128 * pp.ldh * 2, h.stream));
    p_smp.alloc((size_t)pp.sample_tiles * 128 * pp.ldh / 2);
    p_tmax.alloc((size_t)pp.npad * pp.sample_groups);
    p_tau.alloc(std::max((size_t)pp.npad, (size_t)rb_per * 128 * parts));  // (whole equal chunks for the all-gather of a sharded half sweep)
    p_queue.alloc(1);
    launch_panel_image(Yn.p, ldn, p_img.p, pp, N, h.D, h.stream);
    launch_panel_sample(p_img.p, p_smp.p, pp, N, h.stream);
  }
  if (prefilter) {
    if (!panel) {
      Yh.alloc((size_t)h.N * ldh / 2);
      launch_to_f16(Yn.p, ldn, Yh.p, ldh, h.N, h.D, h.stream);
    }
    cval.alloc((size_t)h.N * keep_f);
    cidx.alloc((size_t)h.N * keep_f);
    fail_rows.alloc((size_t)h.N);
    fail_count.alloc(1);
    HIP_CHECK(hipMemsetAsync(fail_count.p, 0, 4, h.stream));
  }
  // worst-case |fp16-path score - exact score| for unit rows: (2u + u^2) with u = 2^-11, plus fp32 accumulation
  const float delta = 9.8e-4f + 1.2e-7f * (float)h.D;
  const bool sym_sharded = panel && pp.sym && parts > 1;
  if (sym_sharded) {
    // Half sweep of a sharded build (graph.py:35-65 cut over the ranks): thresholds of a rank's own row blocks, all-gathered;
    // then ONE sweep of the tiles J >= I whose work items the ranks take in turn (item = rank, rank + parts, ...: items of a
    // chunk stay neighbours), every rank delivering into buckets of ALL rows; then the entries of each rank's own rows travel
    // to it (exchange_buckets).  OSC_KNN_FAKE_SHARDS runs the ranks' passes one after another into the same buckets.
    ProfScope ps(h, 3);
    for (int part = 0; part < parts; ++part) {
      if (sharded && part != h.rank) continue;
      const int rb_begin = std::min(all_rb, part * rb_per), rb_count = std::max(0, std::min(rb_per, all_rb - rb_begin));
      const int nsets = (rb_count + pp.nrg - 1) / pp.nrg;
      launch_panel_tilemax(p_img.p, p_smp.p, pp, N, rb_begin, rb_count, p_tmax.p, p_queue.p,
                           std::max(1, std::min(prop.multiProcessorCount, nsets * pp.SA)), h.stream);
    }
    launch_panel_tau(p_tmax.p, pp, N, p_tau.p, h.stream);  // (rows of other ranks' blocks: overwritten by the all-gather)
    if (sharded) h.comm->allgather(p_tau.p, (size_t)rb_per * 128 * 4, h.stream);
    const size_t nb = (size_t)pp.npad / 32;
    p_hits.alloc(nb * pp.bucket_cap);
    p_hcnt.alloc(nb + (size_t)pp.S);
    HIP_CHECK(hipMemsetAsync(p_hcnt.p, 0, (nb + (size_t)pp.S) * 4, h.stream));
    sym_dev.bucket_ent = p_hits.p;
    sym_dev.bucket_cnt = p_hcnt.p;
    sym_dev.flags = p_hcnt.p + nb;
    const int sgrid = std::max(1, std::min(prop.multiProcessorCount, (pp.nitems + parts - 1) / parts));
    for (int part = 0; part < parts; ++part) {
      if (sharded && part != h.rank) continue;
      launch_panel_filter(p_img.p, pp, N, 0, pp.nrb, p_tau.p, p_hits.p, p_hcnt.p, p_queue.p, sgrid, h.stream, &sym_dev, part, parts);
    }
    if (sharded) exchange_buckets(h, pp, sym_dev, rb_per);
  }
  for (int part = 0; part < parts; ++part) {
    if (sharded && part != h.rank) continue;
    const int rb_begin = std::min(all_rb, part * rb_per);
    const int rb_count = std::max(0, std::min(rb_per, all_rb - rb_begin));
    if (any_k) {
      // chunks of up to ~1 GiB of similarity rows (multiple of 128 rows)
      const int32_t ldS = ((N + 31) / 32) * 32;
      const int64_t cap_rows = std::max<int64_t>(128, (((int64_t)1 << 28) / ldS) / 128 * 128);
      const int32_t row_lo = rb_begin * 128, row_hi = std::min(N, (rb_begin + rb_count) * 128);
      const int32_t chunk = (int32_t)std::min<int64_t>(cap_rows, ((row_hi - row_lo + 127) / 128) * 128);
      if (row_hi > row_lo) {
        DevBuf<float> Sm;
        Sm.alloc((size_t)chunk * ldS);
        ProfScope ps(h, 3);
        for (int32_t r = row_lo; r < row_hi; r += chunk)
          launch_knn_rows_any(Yn.p, ldn, N, k, r, std::min(chunk, row_hi - r), Sm.p, ldS, h.knn_val.p, h.knn_idx.p, h.stream);
        sync(h);  // Sm goes back to the pool at scope exit
      }
    } else if (panel) {
      const KnnPlan plan = knn_plan(N, keep_f, slots, rb_begin, rb_count, true, h.knn_splits);  // row range + keep for the re-scoring
      const int nsets = (rb_count + pp.nrg - 1) / pp.nrg;  // work items per column split (knn_gemm.hip)
      const int grid = std::max(1, std::min(prop.multiProcessorCount, nsets * pp.S));
      if (!sym_sharded) {  // (a sharded half sweep has its thresholds and buckets already: above)
        ProfScope ps(h, 3);
        launch_panel_tilemax(p_img.p, p_smp.p, pp, N, rb_begin, rb_count, p_tmax.p, p_queue.p,
                             std::max(1, std::min(prop.multiProcessorCount, nsets * pp.SA)), h.stream);
        launch_panel_tau(p_tmax.p, pp, N, p_tau.p, h.stream);
        if (pp.sym) {
          const size_t nb = (size_t)pp.npad / 32;
          p_hits.alloc(nb * pp.bucket_cap);  // one bucket per group of 32 receiving rows
          p_hcnt.alloc(nb + (size_t)pp.S);   // [bucket counts | chunk flags]
          HIP_CHECK(hipMemsetAsync(p_hcnt.p, 0, (nb + (size_t)pp.S) * 4, h.stream));
          sym_dev.bucket_ent = p_hits.p;
          sym_dev.bucket_cnt = p_hcnt.p;
          sym_dev.flags = p_hcnt.p + nb;
          const int sgrid = std::max(1, std::min(prop.multiProcessorCount, pp.nitems));
          launch_panel_filter(p_img.p, pp, N, rb_begin, rb_count, p_tau.p, p_hits.p, p_hcnt.p, p_queue.p, sgrid, h.stream, &sym_dev);
        } else {
          p_hits.alloc((size_t)rb_count * pp.S * 4 * pp.hit_cap);  // one list per (work item, wave)
          p_hcnt.alloc((size_t)rb_count * pp.S * 4);
          launch_panel_filter(p_img.p, pp, N, rb_begin, rb_count, p_tau.p, p_hits.p, p_hcnt.p, p_queue.p, grid, h.stream);
        }
      }
      launch_panel_select(pp, rb_begin, rb_count, N, p_hits.p, p_hcnt.p, cval.p, cidx.p, fail_rows.p, fail_count.p,
                          h.stream, pp.sym ? &sym_dev : nullptr);
      launch_knn_rescore(plan, Yn.p, ldn, h.D, N, cidx.p, cval.p, k, delta, h.knn_val.p, h.knn_idx.p, fail_rows.p,
                         fail_count.p, h.stream);
    } else if (prefilter) {
      const KnnPlan plan = knn_plan(N, keep_f, slots, rb_begin, rb_count, true, h.knn_splits);
      const size_t ncand = (size_t)h.N * plan.S * plan.KC;
      cand_val.alloc(ncand);
      cand_idx.alloc(ncand);
      {
        ProfScope ps(h, 3);
        launch_knn_topk(plan, Yh.p, ldh / 2, N, cand_val.p, cand_idx.p, h.stream);
      }
      HIP_CHECK(hipMemsetAsync(cidx.p, 0xFF, (size_t)h.N * keep_f * 4, h.stream));
      launch_knn_merge(plan, cand_val.p, cand_idx.p, N, keep_f, cval.p, cidx.p, 0, h.stream);
      launch_knn_rescore(plan, Yn.p, ldn, h.D, N, cidx.p, cval.p, k, delta, h.knn_val.p, h.knn_idx.p, fail_rows.p,
                         fail_count.p, h.stream);
    } else if (parts == 1 && N <= dense_max && dense_small) {
      // small lattices: dense S + per-row argmax selection (the streaming kernel's first-tile inserts dominate here)
      const int32_t ldS = ((N + 31) / 32) * 32;
      DevBuf<float> Sm;
      Sm.alloc((size_t)N * ldS);
      ProfScope ps(h, 3);
      launch_knn_dense(Yn.p, ldn, N, k, Sm.p, ldS, h.knn_val.p, h.knn_idx.p, h.stream);
      sync(h);  // Sm goes back to the pool at scope exit
    } else {
      const KnnPlan plan = knn_plan(N, k, slots, rb_begin, rb_count, false, h.knn_splits);
      const size_t ncand = (size_t)h.N * plan.S * plan.KC;
      cand_val.alloc(ncand);
      cand_idx.alloc(ncand);
      {
        ProfScope ps(h, 3);
        launch_knn_topk(plan, Yn.p, ldn, N, cand_val.p, cand_idx.p, h.stream);
      }
      launch_knn_merge(plan, cand_val.p, cand_idx.p, N, k, h.knn_val.p, h.knn_idx.p, 1, h.stream);
    }
  }
  if (prefilter) {
    int32_t nfail = 0;
    HIP_CHECK(hipMemcpyAsync(&nfail, fail_count.p, 4, hipMemcpyDeviceToHost, h.stream));
    sync(h);
    DevBuf<int32_t> fail_rows2, fail_count2;
    int32_t* fail_list = fail_rows.p;
    if (panel && pp.sym && nfail > 0) {  // second-stage proof from the rows' whole buckets (knn_gemm.hip: k_bucket_rescore)
      fail_rows2.alloc((size_t)nfail);
      fail_count2.alloc(1);
      HIP_CHECK(hipMemsetAsync(fail_count2.p, 0, 4, h.stream));
      launch_bucket_rescore(pp, sym_dev, Yn.p, ldn, N, fail_rows.p, nfail, p_tau.p, k, delta, h.knn_val.p, h.knn_idx.p,
                            fail_rows2.p, fail_count2.p, h.stream);
      HIP_CHECK(hipMemcpyAsync(&nfail, fail_count2.p, 4, hipMemcpyDeviceToHost, h.stream));
      sync(h);
      fail_list = fail_rows2.p;
    }
    h.knn_fallback_rows = nfail;
    bool few_done = false;
    if (nfail > 0 && nfail <= 32) {  // a handful of rows: stream the columns once, select per row (0.15 vs 3.9 ms at N = 100k)
      const int32_t ldS = ((N + 31) / 32) * 32;
      DevBuf<float> Sm;
      Sm.alloc((size_t)nfail * ldS);
      few_done = launch_knn_few_rows(Yn.p, ldn, N, k, fail_list, nfail, Sm.p, ldS, h.knn_val.p, h.knn_idx.p, h.stream);
      if (few_done) sync(h);  // Sm goes back to the pool at scope exit
    }
    if (nfail > 0 && !few_done) {  // redo the unproven rows with the exact kernel (ties / dense clusters of near-equal scores)
      KnnPlan plan = knn_plan(N, k, slots, 0, (nfail + 127) / 128, false, h.knn_splits);
      plan.qrows = fail_list;
      plan.nq = nfail;
      const size_t ncand = (size_t)h.N * plan.S * plan.KC;
      cand_val.alloc(ncand);
      cand_idx.alloc(ncand);
      launch_knn_topk(plan, Yn.p, ldn, N, cand_val.p, cand_idx.p, h.stream);
      launch_knn_merge(plan, cand_val.p, cand_idx.p, N, k, h.knn_val.p, h.knn_idx.p, 1, h.stream);
    }
  }
  if (sharded) {
    const size_t cnt = (size_t)rb_per * 128 * k;  // equal chunk per rank, in place
    h.comm->allgather(h.knn_val.p, cnt * 4, h.stream);
    h.comm->allgather(h.knn_idx.p, cnt * 4, h.stream);
  }
  alloc_ell(h, k);
  launch_mutual_ell(h.knn_val.p, h.knn_idx.p, N, k, h.width, h.ell_col.p, h.ell_a.p, h.deg.p, h.stream);
  DevBuf<float> scale;
  scale.alloc((size_t)h.N);
  launch_cap_and_normalize(h.ell_a.p, h.ell_w.p, h.ell_col.p, h.deg.p, h.width, N, h.row_cap, 1, scale.p,
                           h.sqrt_deg.p, h.stream);
  graph_counts(h);  // synchronises
  h.have_graph = true;
  maybe_reorder(h);
  h.build_ms = now_ms() - t0;
}

// ---- operators ------------------------------------------------------------------------------------
bool path_active(const L& h) { return h.chain_present && h.lamP > 0.0f; }

OpParams settle_op(const L& h, float dt, int precond) {
  OpParams o{};
  const float lp_op = path_active(h) ? h.lamP : 0.0f;
  o.cs_const = 1.0f + dt * (h.lamG + h.lamC + lp_op);  // X + dt (lamG X + lamC (X - W X) + lamP (X - Wp X))
  o.cs_B = dt * h.lamQ;
  o.cW = dt * h.lamC;
  o.cP = dt * lp_op;
  o.md_const = 1.0f + dt * (h.lamG + (h.chain_present ? h.lamP : 0.0f));  // lattice.py:187-192
  o.md_B = dt * h.lamQ;
  o.precond = precond;
  o.rbU = 1.0f;
  o.rbY = dt * h.lamG;
  o.rbB = dt * h.lamQ;
  return o;
}
OpParams ustar_op(const L& h) {
  OpParams o{};
  const float lp_op = path_active(h) ? h.lamP : 0.0f;
  o.cs_const = h.lamG + h.lamC + lp_op;
  o.cs_B = h.lamQ;
  o.cW = h.lamC;
  o.cP = lp_op;
  o.md_const = h.lamG + (h.chain_present ? h.lamP : 0.0f);  // lattice.py:257-259
  o.md_B = h.lamQ;
  o.precond = 1;
  o.rbU = 0.0f;
  o.rbY = h.lamG;
  o.rbB = h.lamQ;
  return o;
}

// Operator apply, optionally split into column slabs so the gathered operand slab (N x slab x 4 B) stays resident
// in the 256 MB Infinity Cache while its rows are re-read ~deg times (MI355X_MICROARCH.md, Infinity Cache rule).
int32_t auto_slab(const L& h, int32_t ncols) {
  constexpr int32_t kMaxWindow = 2048;  // widest column window one launch covers (8 x 256 floats per row)
  if (h.spmm_slab > 0) return std::min(h.spmm_slab, kMaxWindow);
  if (h.spmm_slab < 0) return std::min(ncols, kMaxWindow);  // OSC_SPMM_SLAB=-1: split only when it must
  // a lattice stored in a local row order gathers from its XCD's L2 whatever the slab: 256 columns (one 1 KB row piece per
  // wave, eight of them in flight: k_spmm's UDEEP variant) ran fastest on 1000 clusters x 100 rows at N = 100k, D = 768
  // (0.69 ms per apply at 64 columns, 0.45 at 128, 0.41 at 256 and 512, 0.44 at 768)
  if (h.reordered && h.spmm_deep && ncols > 256) return 256;
  // keep the gathered slab (N x slab x 4 B) around 50 MB so it and the streams beside it stay inside 256 MB
  const double budget = 56.0 * 1024 * 1024;
  if ((double)h.N * ncols * 4.0 <= 2.0 * budget) return std::min(ncols, kMaxWindow);
  int32_t slab = 64;
  for (int32_t w : {128, 256, 384, 512, 768, 1024, 2048})
    if ((double)h.N * w * 4.0 <= budget) slab = w;
  return slab;
}

// XCD-affine 32-column slabs (SpmmArgs::xs): one launch covers the window; returns the workgroups per XCD to use, 0 = no.
// Pays when the gathered operand is far larger than an XCD's L2 and the graph has no row locality to exploit: each
// XCD then keeps 4 MB / (N x 128 B) of ITS slab in L2 (31 % at N = 100k) instead of 4 MB / (N x 512 B) of a slab all
// eight share.  With fewer than 8 slabs (or a count that is not a multiple of 8) the XCDs pair up: gcd(8, slabs) slab
// groups, the XCDs of a group split the rows.  Needs 128-byte-aligned rows and the slabs in flight (groups x N x 128 B)
// inside the Infinity Cache: measured 1.11 vs 1.26 ms per apply at N = 100k, D = 768; no gain at N = 200k, D = 1536 with
// 8 slabs (205 MB) in flight, 4 % with 4 (xs_groups_for); 36 % slower at N = 1M, D = 384.
// (the counts themselves: host_logic.hpp)
int xs_groups(int32_t ncols, int cap = 8) { return host::xs_groups(ncols, cap); }
int xs_groups_for(const L& h, int32_t ncols) { return host::xs_groups_for(h.N, ncols, h.xs_groups_cap, h.xs_groups_min); }
int blocked_plan(const L& h, bool with_path);
int xs_plan(const L& h, int32_t ncols, int grid) {
  if (grid < 8 || (grid & 7) != 0) return 0;
  const int nb = std::max(1, std::min(grid / 8, h.xs_nb > 0 ? h.xs_nb : 96));
  if (h.spmm_xs == 0) return 0;
  if (h.spmm_xs == 1) return nb;
  if (h.spmm_slab != 0 || (h.ld & 31) != 0 || (h.c0 & 31) != 0) return 0;
  // A lattice stored in BFS order gathers from its XCD's L2 on the general path already, so the slab
  // mode is off for it -- except large narrow ones, where the source-blocked matvec on top of the local order wins
  // (round 4, scripts/exp/r04_bfs_blocked_sweep.py, clustered anchors, per settle: 300k x 128 k 16 2.17 -> 1.95 ms, 300k x
  // 256 k 32 6.15 -> 5.0-5.3, 400k x 256 6.35 -> 5.35, 600k x 128 4.79 -> 3.92, 1M x 128 8.15 -> 6.72; at 384 columns a tie
  // (400k 8.06 / 7.98, 1M 20.4 / 20.7), at 200k rows a loss (128 columns: 1.25 -> 1.32)).
  if (h.reordered) return (h.N >= 300000 && ncols <= 256 && blocked_plan(h, false) > 0) ? nb : 0;
  // from N = 32768 on, and from 6144 (16384 until round 3) for windows of >= 256 columns (N = 20000, D = 256: apply 43.5 -> 31.4 us)
  // narrower windows: 32768 rows, but 12288 where the window is whole groups of four slabs (every XCD pair a slab of its
  // own) and 24576 for other windows of >= 128 columns (scripts/exp/xs_narrow_sweep.py, k = 16, per settle: 20000 x 128
  // 246 -> 223 us, 32000 x 128 347 -> 301, 12000 x 128 192 -> 186, 32000 x 192 484 -> 438, 24000 x 192 393 -> 378, 20000 x
  // 192 345 -> 360; 64 and 32 columns: a tie or a loss up to 32000 rows)
  const int narrow_rows = h.xs_min_rows_narrow > 0 ? h.xs_min_rows_narrow : ncols < 128 ? 32768 : (ncols % 128) == 0 ? 12288 : 24576;
  if (h.N < h.xs_min_rows || (h.N < narrow_rows && ncols < 256) || ncols < h.xs_min_cols) return 0;
  // below 16384 rows (round 3: the floor was 16384) a 32-column slab is at most 2 MB -- it sits in its XCD's L2 whole,
  // where the general path spreads N x window over all eight L2s -- which pays once a row has enough gathers: per settle
  // 6500 x 768 k 32 0.520 -> 0.437 ms, 9000 x 1024 k 32 0.925 -> 0.697, 8192 x 1536 k 32 1.32 -> 0.91, 14000 x 256 k 32 0.406
  // -> 0.329, 9000 x 256 k 16 0.236 -> 0.219, 7000 x 512 k 16 0.299 -> 0.280; at k = 8 it loses (14000 x 320: 0.307 -> 0.329)
  if (h.N < 16384 && (double)h.nnz < 10.0 * (double)h.N) return 0;
  const int xg = xs_groups_for(h, ncols);
  if (xg == 0) return 0;
  // Two slab groups (262k < N <= 524k: four XCDs share a slab) pay only under the blocked matvec -- measured in round 3
  // against the general path: 300k x 768 k 32 25.96 -> 22.33 ms per settle, 400k x 512 k 32 22.70 -> 19.10, 300k x 768 k 64
  // 43.1 -> 36.3, 500k x 384 k 16 a tie; the plain slab apply at two groups loses (config 5's shape: 57.1 vs 56.4 ms) and one
  // group loses either way (700k x 384: 22.8 -> 24.5, config 4: 32.3 -> 34.4)
  if (xg < 4 && xg != xs_groups(ncols, h.xs_groups_cap) && blocked_plan(h, false) == 0) return 0;
  return nb;
}

// workgroups per XCD a shape of the blocked apply gets resident
int blocked_resident(const L& h, int shape) {
  if (h.blk_resident[shape] < 0) {
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, h.device));
    h.blk_resident[shape] = blocked_resident_per_cu(shape) * std::max(1, prop.multiProcessorCount / 8);
  }
  return h.blk_resident[shape];
}

// Kernel shape of the blocked matvec for a window cut into xg slab groups (cg_kernels.hip: kBlkShapes).  The wide shapes
// (one workgroup per CU, four gather rounds in flight, no tests in the rounds) carry their group count as a template
// constant -- the smallest that holds the lattice's groups is used -- and are taken from 96 000 rows on, where they win
// at every width measured except one slab per XCD below 150k rows; below 96k rows they are within +-2 % of shape 0 with
// single wins and losses of 5-7 % either way, so shape 0 stays there.
int blocked_shape_for(const L& h, int xg, int grid) {
  if (h.blk_variant >= 0) return h.blk_variant;
  const int wide_last = blocked_variants() - 1;
  const host::BlockedGeom g = host::blocked_geometry(h.N, xg, grid, blocked_resident(h, wide_last), blocked_groups_max(wide_last),
                                                     blocked_gather_waves(wide_last));
  // Measured against shape 0 (profiles/r05_blk_shape_sweep.txt, per AP launch, exact-fit group counts): 20k x 768 -7.5 %, 20k x
  // 128 k 16 +5.8 %, 30k-80k x 768 -0.8 ... +2.7 %, 100k x 768 -4.6 %, 100k x 384 k 16 -4.8 %, 100k x 1024 k 48 -4.0 %, 100k x
  // 96 (rank 0 of 8's window of config 3) -10.9 %, 100k x 192 -2.7 %, 160k x 768 -10.9 %, 200k x 768 -9.6 %, 200k x 64 -13.0 %,
  // 260k x 512 -15.0 %, 400k x 384 k 16 -7.1 %; one slab per XCD: 100k x 64 k 16 +5.0 %, 100k x 128 k 16 +1.4 %, 130k x 256
  // +1.4 ... +3.6 % -- there the wide shapes wait for N = 150k.
  const int64_t min_rows = h.blk_wide_min_rows > 0 ? h.blk_wide_min_rows : 96000;
  const int slabs_per_group = ((h.c1 - h.c0 + 31) / 32 + xg - 1) / std::max(1, xg);
  if (h.N < min_rows || (h.blk_wide_min_rows <= 0 && slabs_per_group < 2 && h.N < 150000)) return 0;
  for (int v = 1; v <= wide_last; ++v)
    if (g.groups <= blocked_groups_max(v)) return v;
  return 0;
}

// Source blocks of the blocked CG matvec (k_apply_blocked): 0 = use the plain apply.
int blocked_plan(const L& h, bool with_path) {
  if (h.spmm_blocked == 0 || (with_path && (h.prows < 1 || h.prows > OSC_CHAIN_FIX_MAX_ROWS)) || (int64_t)h.N * h.width >= ((int64_t)1 << 28) || h.N >= ((int64_t)1 << 24) ||
      (int64_t)h.N * h.ld * 4 >= ((int64_t)1 << 32))
    return 0;
  if (h.spmm_blocked > 0) return std::min(h.spmm_blocked, OSC_MAX_SRC_BLOCKS);
  // block count from the mean degree and the lattice size (host_logic.hpp: blocked_edges_per_block)
  // overrides the edges a row should have per block
  const double mean_deg = h.N > 0 ? (double)h.nnz / (double)h.N : 0.0;
  // (a lattice in BFS order: 2.2 edges per block -- x4 of x2 / x3 / x4 / x6 / x8 at mean degree 8.3, x8 of x6 / x8 / x12 at 20.2)
  const int ncols = h.c1 - h.c0, xg0 = xs_groups_for(h, ncols);
  const bool wide = blocked_shape_for(h, xg0 > 0 ? xg0 : xs_groups(ncols, h.xs_groups_cap), cg_grid(h)) > 0;
  const double e = h.blk_edges > 0.0 ? h.blk_edges
                   : h.reordered     ? 2.2
                   : wide            ? host::blocked_edges_per_block_wide(h.N)
                                     : host::blocked_edges_per_block(h.N);
  const int nb = host::blocked_block_count(mean_deg, e, OSC_MAX_SRC_BLOCKS);
  if (h.spmm_blocked == -2) return nb;  // "whenever possible" (experiments)
  // ... and wherever the XCD-affine slab mode itself runs from a 2 MiB slab (N = 16384) on.  Measured against the plain
  // apply (k = 32 unless noted): N = 20k x 768 -7 %, 35k x 768 -26 %, 40k x 256 (k 8) -25 %, 50k x 512 -30 %, 65k x 256
  // (k 16) -30 %, 60k x 1024 (k 24) -29 %, 80k x 768 -39 %, 100k x 768 -39 % (k 16, D 384: -33 %; k 48: -47 %; k 64:
  // -45 %), 100k x 128 (k 16) -35 %, 110k x 768 -40 %, 130k x 256 -43 %; round 3: 160k x 768 -31 %, 200k x 768 -37 %
  // (k 64: -46 %), 260k x 768 -22 % (k 64: -37 %).
  const double slab = (double)h.N * 128.0;
  if (slab < h.blk_mb * 1024.0 * 1024.0) return 0;
  // narrow windows of small lattices: the plain slab apply is ahead (round 4 shape sweep: 16384 x 128 k 16 0.205 vs 0.221 ms per
  // settle; from 20000 rows on a tie or a win)
  if (h.N < 20000 && h.c1 - h.c0 <= 128) return 0;
  return nb;
}

BlockedView blocked_view(L& h, int nb) {
  if (h.blk_nb != nb) {
    DevBuf<unsigned> cnt;
    cnt.alloc(1);
    HIP_CHECK(hipMemsetAsync(cnt.p, 0, 4, h.stream));
    launch_blocked_count(h.ell_col.p, h.deg.p, h.width, (int32_t)h.N, nb, cnt.p, h.stream);
    unsigned over = 0;
    HIP_CHECK(hipMemcpyAsync(&over, cnt.p, 4, hipMemcpyDeviceToHost, h.stream));
    sync(h);
    // the apply's list wave copies whole row groups: up to 8 x gather-waves slot rows past the lattice's end
    // (host_logic.hpp: blocked_list_extent <= N - 1 + 8 x gather waves, swept in tests/host_logic)
    constexpr size_t kPadRows = 8192;
    for (int v = 0; v < blocked_variants(); ++v)
      if ((size_t)blocked_gather_waves(v) * 8 > kPadRows) throw std::runtime_error("blocked graph copy: padding too small");
    const size_t nslots = (size_t)nb * h.N * OSC_BLK_SLOTS, npad = kPadRows * OSC_BLK_SLOTS;
    h.blk_slots.alloc(nslots + npad);
    HIP_CHECK(hipMemsetAsync(h.blk_slots.p + nslots, 0, npad * sizeof(int2), h.stream));  // {row 0, 0.0f}
    h.blk_over.alloc((size_t)over + 1);
    h.blk_rest.alloc((size_t)h.N);
    HIP_CHECK(hipMemsetAsync(cnt.p, 0, 4, h.stream));
    launch_blocked_fill(h.ell_col.p, h.ell_w.p, h.deg.p, h.width, (int32_t)h.N, nb, h.blk_slots.p, h.blk_rest.p, h.blk_over.p, cnt.p,
                        h.stream);
    sync(h);  // cnt goes out of scope
    h.blk_nb = nb;
  }
  BlockedView v{};
  v.slots = h.blk_slots.p;
  v.rest = h.blk_rest.p;
  v.over = h.blk_over.p;
  v.nb = nb;
  return v;
}

void spmm_slabbed(L& h, int mode, SpmmArgs sa, int grid, int iter = 0) {
  const int32_t c0 = sa.c0, c1 = sa.c1;
  sa.deep = (h.reordered && h.spmm_deep) ? 1 : 0;
  ProfScope ps(h, mode == SPMM_INIT ? 4 : 0, iter);  // slot 0: AP applies (the CG matvec); slot 4: the INIT apply
  if (const int nb = xs_plan(h, c1 - c0, grid)) {
    // workgroups per XCD: 3 per CU when the operand is row-major (2: 1.37, 4: 1.15 ms vs 1.11), 4 per CU when it is
    // slab-major (3: 1.09, 4: 1.05 ms)
    sa.xs = (h.xs_nb <= 0 && sa.xblk != 0) ? std::min(grid / 8, 128) : nb;
    const int xg = xs_groups_for(h, c1 - c0);
    sa.xs_groups = xg > 0 ? xg : xs_groups(c1 - c0, h.xs_groups_cap);  // forced mode: natural count
    launch_spmm(mode, sa, grid, h.stream);
    return;
  }
  const int32_t slab = auto_slab(h, c1 - c0);
  for (int32_t s0 = c0; s0 < c1; s0 += slab) {
    sa.c0 = s0;
    sa.c1 = std::min(c1, s0 + slab);
    launch_spmm(mode, sa, grid, h.stream);
  }
}

// elementwise CG kernels cover at most 2048 columns per launch: wider states run as several column windows
template <typename F>
void for_windows(UpdateArgs ua, F&& launch) {
  const int32_t c0 = ua.c0, c1 = ua.c1;
  for (int32_t s0 = c0; s0 < c1; s0 += 2048) {
    ua.c0 = s0;
    ua.c1 = std::min(c1, s0 + 2048);
    launch(ua);
  }
}

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

// cg_solve (solver.py:6-37) on the device; returns once the last residual is out (what may still be queued then touches
// scratch arrays only, and later calls are ordered behind it by the stream).
// The host enqueues iteration it+1 before it reads iteration it's residual.  On one GPU every kernel of a speculative
// iteration carries a gate (residual of the previous iteration, tol) and is a no-op once the CG has converged; under
// a communicator it carries none and writes scratch arrays only (the x update of an iteration is applied by its
// successor's p update or by the host's order, never speculatively).  Either way the reference's "stop before the
// beta/p update" semantics hold exactly while the stream never drains between iterations.
// Small lattices: the whole solve in ONE launch with the state in LDS (small_kernels.hip).  Returns false when the
// lattice does not fit that path (or its barrier timed out) and the general path must run.
bool run_cg_small(L& h, const OpParams& op, const CgBuffers& b, bool with_path, int max_iters, float tol,
                  CgResult& out) {
  if (!h.small_path || h.comm != nullptr || b.c0 != 0 || b.c1 != h.dcols || b.ld != h.dcols || max_iters > 4096) return false;
  const int C = small_pick_cols((int32_t)h.N, b.ld);
  if (C <= 0) return false;
  const size_t nslots = (size_t)max_iters + 2;
  const size_t nctl = 2 * nslots + 2;  // [residual slots | arrival counters | status | finish counter]
  ensure_ctrl(h, nctl);
  uint32_t* ctl = ctrl_segment(h, nctl);
  // the kernel's last workgroup publishes residuals + a "done" word into host-mapped memory and the host polls that
  // word
  const bool polled = h.mapped_residual;
  constexpr uint32_t kPending = 0xFFFFFFFFu;
  volatile uint32_t* host_words = reinterpret_cast<volatile uint32_t*>(h.res_host);
  if (polled) host_words[nslots] = kPending;
  SmallArgs a{};
  if (!h.ell_t_ready) {
    h.ell_col_t.alloc((size_t)h.N * h.width);
    h.ell_w_t.alloc((size_t)h.N * h.width);
    launch_transpose_ell(h.ell_col.p, h.ell_w.p, (int32_t)h.N, h.width, h.ell_col_t.p, h.ell_w_t.p, h.stream);
    h.ell_t_ready = true;
  }
  a.g = graph_view(h, with_path);
  a.col_t = h.ell_col_t.p;
  a.w_t = h.ell_w_t.p;
  a.op = op;
  float* xout = b.X;
  if (b.X == b.x0 || b.X == b.rhsU || b.X == b.rhsY) {  // never hand the one-launch kernel an aliased output
    if (!b.Xalt) return false;
    xout = b.Xalt;
  }
  a.x0 = b.x0;
  a.X = xout;
  a.U = b.rhsU;
  a.Y = b.rhsY;
  a.B = b.B;
  a.psi = b.psi;
  a.res_bits = ctl;
  a.arrive = ctl + nslots;
  a.status = ctl + 2 * nslots;
  a.finish = ctl + 2 * nslots + 1;
  a.host_words = polled ? reinterpret_cast<uint32_t*>(h.res_host_dev) : nullptr;
  a.N = (int32_t)h.N;
  a.ld = b.ld;
  a.max_iters = max_iters;
  a.tol = tol;
  launch_settle_small(a, C, h.stream);
  if (polled) {
    const double t_start = now_ms();
    for (uint64_t spin = 1; host_words[nslots] == kPending; ++spin) {
      if ((spin & 0x3FFF) == 0) {  // every ~16k polls: has the stream died or drained without the word?
        const hipError_t q = hipStreamQuery(h.stream);
        if (q != hipSuccess && q != hipErrorNotReady) hip_check(q, "hipStreamQuery (one-launch solve)", __FILE__, __LINE__);
        if (q == hipSuccess && host_words[nslots] == kPending) throw HipError("one-launch solve finished without its done word");
        if (now_ms() - t_start > 120000.0) throw HipError("timeout waiting for the one-launch solve");
      }
      __builtin_ia32_pause();
    }
    if (host_words[nslots] != 0u) {
      sync(h);       // (the kernel's other workgroups are on their way out)
      return false;  // barrier timeout (GPU shared with other persistent work): take the general path
    }
  } else {
    HIP_CHECK(hipMemcpyAsync(h.res_host, ctl, nctl * 4, hipMemcpyDeviceToHost, h.stream));
    sync(h);
    uint32_t st;
    std::memcpy(&st, h.res_host + 2 * nslots, 4);
    if (st != 0) return false;  // barrier timeout (GPU shared with other persistent work): take the general path
  }
  h.history.clear();
  out = CgResult{max_iters, 0.f, xout};
  for (int it = 1; it <= max_iters; ++it) {
    const float res = h.res_host[it];
    h.history.push_back(res);
    out.res = res;
    if ((double)res <= (double)tol) {
      out.iters = it;
      break;
    }
  }
  h.small_solves += 1;
  return true;
}

CgResult run_cg_rows(L& h, const OpParams& op, const CgBuffers& b, bool with_path, int max_iters, float tol);
bool row_mode(const L& h);

CgResult run_cg(L& h, const OpParams& op, const CgBuffers& b, bool with_path, int max_iters, float tol) {
  if (row_mode(h) && b.ld == h.ld) return run_cg_rows(h, op, b, with_path, max_iters, tol);
  {
    CgResult one{};
    if (run_cg_small(h, op, b, with_path, max_iters, tol, one)) return one;
  }
  const int grid = cg_grid(h);
  const size_t nslots = (size_t)max_iters + 2;
  ensure_ctrl(h, nslots);
  uint32_t* const res_slots = ctrl_segment(h, 2 * nslots);  // zeroed: [residual per iteration | arrival counter per iteration]
  uint32_t* done_ctr = res_slots + nslots;
  // Single GPU: the last workgroup of each iteration's beta reduction writes the residual into host-mapped memory and
  // the host polls that word (no 4-byte copy, event record and event wait per iteration).  Under a communicator the
  // residual first goes through the all-reduce (below; OSC_COMM_OVERLAP=0: in the solve's stream,
  // read back by copy + event).
  const bool mapped = h.comm == nullptr && h.mapped_residual;
  // Sharded (column windows): the stop test needs max over the ranks of the residual -- a 4-byte all-reduce per iteration,
  // tens of microseconds of latency on xGMI next to ~180 us of kernels per iteration in an 8-rank window of config 3.
  // With the x update deferred (below) a speculative iteration writes scratch arrays only (r, p, Ap, alpha, beta), so it
  // needs no gate and the solve's stream never waits for the all-reduce: that goes to a second stream behind an event
  // per iteration, followed by a one-thread kernel that publishes the reduced word into the host-mapped slot the host
  // polls, as on one GPU.  The host alone decides when to stop; a wrong guess of the last iteration costs one iteration
  // of device time instead of five gated-off launches.
  const bool xdefer = h.x_defer;
  // What it costs (one-rank RCCL communicator, all-reduce latency ~0: DESIGN.md section 6): ~17 us once per solve for the
  // second stream's hand-over at the last iteration, and the expected last iteration's own form (an ungated speculative
  // iteration must not touch x): 47 us at 768 columns, 6 at 96.  What it saves: every all-reduce latency but the last.
  // Hence by default from four ranks on (narrow windows, 15-30 us per all-reduce); OSC_COMM_OVERLAP=1 / 0 force it.
  const bool want_overlap = h.comm_overlap == 1 || (h.comm_overlap < 0 && h.world >= 4);
  const bool overlap = h.comm != nullptr && want_overlap && h.mapped_residual && xdefer;
  if (overlap) {
    if (!h.comm_stream) h.comm_stream = acquire_stream(h.device);
    while (h.step_events.size() < nslots) {
      hipEvent_t e;
      HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      h.step_events.push_back(e);
    }
    h.comm_stream_busy = true;  // from here on (also if the solve is abandoned half way): drained before the slots are reused
  }
  const bool polled = mapped || overlap;
  constexpr uint32_t kPending = 0xFFFFFFFFu;  // never a residual (those are sqrt(...) >= 0 or a canonical NaN)
  if (polled)
    for (size_t i = 0; i < nslots; ++i) reinterpret_cast<volatile uint32_t*>(h.res_host)[i] = kPending;
  const float* res_dev = reinterpret_cast<const float*>(res_slots);
  SpmmArgs sa{};
  sa.g = graph_view(h, with_path);
  sa.op = op;
  sa.B = b.B;
  sa.psi = b.psi;
  sa.N = h.N;
  sa.ld = b.ld;
  sa.c0 = b.c0;
  sa.c1 = b.c1;
  sa.part = h.part0.p;
  // r = b - A x0 ; z ; p ; rz
  sa.X = b.x0;
  sa.OUT = b.X;
  sa.R = b.R;
  sa.P = b.P;
  sa.U = b.rhsU;
  sa.Y = b.rhsY;
  sa.gate = nullptr;
  // slab-major search direction: only where the XCD-affine slab apply runs (its gathers then read contiguous slabs)
  // and P is private to this solve (N x ld floats either way; needs whole 32-column slabs inside the pitch)
  const bool pblk = h.p_blocked && xs_plan(h, b.c1 - b.c0, grid) > 0 && (b.ld & 31) == 0 && (b.c0 & 31) == 0 &&
                    b.ld == h.ld;
  sa.pblk = pblk ? h.N : 0;
  // source-blocked CG matvec (k_apply_blocked) where the slab an XCD gathers from is far larger than its L2
  BlkArgs ba{};
  ChainFixArgs cf{};
  int blk_shape = 0;
  if (pblk && b.c0 == h.c0 && b.c1 == h.c1) {
    if (const int nb = blocked_plan(h, with_path)) {
      const BlockedView bv = blocked_view(h, nb);
      ba.X = b.P;
      ba.OUT = b.AP;
      ba.B = b.B;
      ba.part = h.part0.p;
      ba.slots = bv.slots;
      ba.rest = bv.rest;
      ba.over = bv.over;
      ba.cs_const = op.cs_const;
      ba.cs_B = op.cs_B;
      ba.cW = op.cW;
      ba.N = (int32_t)h.N;
      ba.ld = b.ld;
      ba.c0 = b.c0;
      ba.c1 = b.c1;
      ba.nb = nb;
      // workgroups per XCD (what is resident at once), slab groups, row groups per wave, destination slices
      const int xg0 = xs_groups_for(h, b.c1 - b.c0), xg = xg0 > 0 ? xg0 : xs_groups(b.c1 - b.c0, h.xs_groups_cap);
      blk_shape = blocked_shape_for(h, xg, grid);
      const host::BlockedGeom geom = host::blocked_geometry(h.N, xg, grid, blocked_resident(h, blk_shape), blocked_groups_max(blk_shape),
                                                           blocked_gather_waves(blk_shape));
      ba.xs = geom.xs;
      ba.xs_groups = geom.xs_groups;
      ba.slices = geom.slices;
      ba.groups = geom.groups;
      if (with_path && op.cP != 0.f) {  // the chain prior's few rows: a small launch behind every blocked apply
        cf.X = b.P;
        cf.OUT = b.AP;
        cf.part = h.part0.p;
        cf.prow = h.prow.p;
        cf.pcol = h.pcol.p;
        cf.pw = h.pw.p;
        cf.pdeg = h.pdeg.p;
        cf.cP = op.cP;
        cf.prows = h.prows;
        cf.pwidth = h.pwidth;
        cf.N = (int32_t)h.N;
        cf.ld = b.ld;
        cf.c0 = b.c0;
        cf.c1 = b.c1;
        cf.part_row0 = grid;
        cf.chunks = chain_fix_chunks(h.prows);
      }
    }
  }

  // (an inertia start hands over x0 IN the AP array, which the blocked matvec would overwrite with A x0 before
  // init_finish has read x0: such a solve keeps the gathering INIT kernel, which reads x0 completely first)
  int init_part_rows = grid;  // rows of r . z partials the INIT pass leaves (+ the chain fix-up's behind the fused pass)
  float* Pbuf = b.P;   // search direction / operator output: the fused INIT pass below leaves p in the AP array and
  float* APbuf = b.AP;  // swaps the two for the rest of the solve
  // (the rhs rows the fused pass can take besides x0 itself: one -- the state term must be x0 or absent, and if y is a
  // third array the solution array must be x0)
  const bool fuse_u = b.rhsU == b.x0 || op.rbU == 0.f;
  const bool fuse_y = b.rhsY == b.x0 || b.X == b.x0;
  if (ba.nb > 0 && h.blk_init && h.blk_init_fused && fuse_u && fuse_y && b.x0 != b.AP) {
    // r = b - A x0 INSIDE the blocked matvec (the in-place warm-started settle: x0 is also the rhs state term and the
    // solution array; the U* solve: x0 is Y, no state term): x0 -> slab-major (into P), then one launch gathers A x0 and
    // leaves r, z (slab-major, in the AP array), x0 in the solution array and the r . z column sums
    ProfScope ps(h, 4, 0);
    launch_rows_to_slab(b.x0, b.P, h.N, b.ld, b.c0, b.c1, grid, h.stream);
    BlkInit bi{};
    bi.Y = b.rhsY == b.x0 ? nullptr : b.rhsY;
    bi.Xcopy = b.X == b.x0 ? nullptr : b.X;
    bi.R = b.R;
    bi.Z = b.AP;
    bi.psi = b.psi;
    bi.rbU = b.rhsU == b.x0 ? op.rbU : 0.f;
    bi.rbY = op.rbY;
    bi.rbB = op.rbB;
    bi.md_B = op.precond ? op.md_B : 0.f;
    bi.md_const = op.precond ? op.md_const : 1.f;
    ba.gate = nullptr;
    ba.OUT = nullptr;
    launch_apply_blocked(ba, grid, h.stream, &bi, blk_shape);
    if (cf.chunks > 0) {  // the chain prior's rows: their r, z and r . z still lack the chain term
      ChainFixArgs ci = cf;
      ci.gate = nullptr;
      ci.initR = b.R;
      ci.initZ = b.AP;
      ci.B = b.B;
      ci.md_B = bi.md_B;
      ci.md_const = bi.md_const;
      launch_chain_fix(ci, h.stream);
      init_part_rows = grid + cf.chunks;
    }
    std::swap(Pbuf, APbuf);
    ba.X = Pbuf;
    ba.OUT = APbuf;
    cf.X = Pbuf;
    cf.OUT = APbuf;
  } else if (ba.nb > 0 && h.blk_init && b.x0 != b.AP) {
    // r = b - A x0 around the blocked matvec: x0 -> slab-major (into P), A x0 -> AP, then r, z, p = z, r . z
    ProfScope ps(h, 4, 0);
    launch_rows_to_slab(b.x0, b.P, h.N, b.ld, b.c0, b.c1, grid, h.stream);
    ba.gate = nullptr;
    launch_apply_blocked(ba, grid, h.stream, nullptr, blk_shape);
    if (cf.chunks > 0) {
      cf.gate = nullptr;
      launch_chain_fix(cf, h.stream);
    }
    InitFinishArgs fa{};
    fa.AP = b.AP;
    fa.X0 = b.x0;
    fa.X = b.X;
    fa.R = b.R;
    fa.P = b.P;
    fa.U = b.rhsU;
    fa.Y = b.rhsY;
    fa.B = b.B;
    fa.psi = b.psi;
    fa.part = h.part0.p;
    fa.op = op;
    fa.N = h.N;
    fa.pblk = h.N;
    fa.ld = b.ld;
    for (int32_t s0 = b.c0; s0 < b.c1; s0 += 2048) {
      fa.c0 = s0;
      fa.c1 = std::min(b.c1, s0 + 2048);
      launch_init_finish(fa, grid, h.stream);
    }
  } else {
    spmm_slabbed(h, SPMM_INIT, sa, grid);
  }
  launch_reduce_init(h.part0.p, init_part_rows, b.ld, b.c0, b.c1, h.rz.p, h.stream);
  UpdateArgs ua{};
  ua.pblk = pblk ? h.N : 0;
  ua.temporal = 5.0 * (double)h.N * (double)(b.c1 - b.c0) * 4.0 <= h.temporal_mb * 1048576.0;
  ua.X = b.X;
  ua.R = b.R;
  ua.P = Pbuf;
  ua.AP = APbuf;
  ua.B = b.B;
  ua.alpha = h.alpha.p;
  ua.beta = h.beta.p;
  ua.part_rr = h.part0.p;
  ua.part_rz = h.part1.p;
  ua.op = op;
  ua.N = h.N;
  ua.ld = b.ld;
  ua.c0 = b.c0;
  ua.c1 = b.c1;
  sa.X = Pbuf;
  sa.OUT = APbuf;
  sa.xblk = pblk ? h.N : 0;
  sa.pblk = 0;
  h.blk_last = ba.nb;
  h.blk_shape_last = ba.nb > 0 ? blk_shape : 0;
  // Deferred x update: iteration it's x += alpha p is applied by iteration it + 1's p update, which reads p anyway (x, r,
  // p in / x, p out there, r, Ap in / r out in the x-r kernel: 8 array passes per iteration instead of 9), or by
  // finish_x behind an iteration that has no successor enqueued.  The iteration expected to be the last (the count of
  // the handle's previous solve of this kind, or max_iters) takes k_update_xr's "last" form instead: x finished next
  // to the r update, the new r not stored (five passes instead of three there and three in finish_x).  Which launch
  // carries which update is decided by host::CgXSchedule (host_logic.hpp; swept against a model of the device's
  // gating on the CPU box, tests/host_logic/sweep_host_logic.cpp).
  const int stop_guess = h.predicted_iters[b.kind];
  host::CgXSchedule xs;
  xs.xdefer = xdefer;
  xs.last_form = h.x_last_form;
  xs.ungated = overlap;
  xs.stop_guess = stop_guess;
  xs.max_iters = max_iters;
  auto finish_x = [&](int it) {
    ua.gate = nullptr;
    ua.xmode = OSC_XMODE_XR_SKIPS_X | OSC_XMODE_P_APPLIES_X;
    for_windows(ua, [&](const UpdateArgs& w) { launch_update_x(w, grid, h.stream); });
    xs.finished(it);
  };
  auto enqueue_iter = [&](int it, bool speculative) {  // everything of iteration `it` up to its residual, gated on iteration it-1
    // (overlap: no gates -- an iteration writes scratch arrays only until the host has seen its predecessor unconverged)
    const Gate g{it > 1 && !overlap ? res_dev + (it - 1) : nullptr, tol};
    sa.gate = g.p;
    sa.gate_tol = tol;
    ua.gate = g.p;
    ua.gate_tol = tol;
    const host::CgXSchedule::IterForm form = xs.enqueue(it, speculative);
    if (it > 1) {
      ProfScope ps(h, 2, it);
      // p = z + beta p (solver.py:32-36), and iteration it - 1's x += alpha p (solver.py:27) with the p it replaces
      ua.xmode = form.p_applies_x ? OSC_XMODE_P_APPLIES_X : 0;
      for_windows(ua, [&](const UpdateArgs& w) { launch_update_p(w, grid, h.stream); });
    }
    if (ba.nb > 0) {  // Ap and column sums of p.Ap
      ProfScope ps(h, 0, it);
      ba.gate = g.p;
      ba.gate_tol = tol;
      unsigned long long* stamps = nullptr;
      if (h.blk_stamp && h.prof_on) {  // diagnostic: per-wave cycle counters of where the launch's time goes
        const size_t words = (size_t)grid * (size_t)(blocked_gather_waves(blk_shape) + 1) * 4;
        if (h.blk_stamps.n != words || h.blk_stamp_grid != grid) {
          h.blk_stamps.alloc(words);
          HIP_CHECK(hipMemsetAsync(h.blk_stamps.p, 0, words * 8, h.stream));
          h.blk_stamp_launches = 0;
          h.blk_stamp_grid = grid;
        }
        stamps = h.blk_stamps.p;
        h.blk_stamp_launches += 1;
      }
      launch_apply_blocked(ba, grid, h.stream, nullptr, blk_shape, stamps);
      if (cf.chunks > 0) {
        cf.gate = g.p;
        cf.gate_tol = tol;
        launch_chain_fix(cf, h.stream);
      }
      h.blk_applies += 1;
    } else {
      spmm_slabbed(h, SPMM_AP, sa, grid, it);
    }
    launch_reduce_alpha(h.part0.p, grid + (ba.nb > 0 ? cf.chunks : 0), b.ld, b.c0, b.c1, h.rz.p, h.alpha.p, g, h.stream);
    {
      ProfScope ps(h, 1, it);
      ua.xmode = form.xr == host::CgXSchedule::XR_LAST ? OSC_XMODE_XR_LAST : form.xr == host::CgXSchedule::XR_SKIPS_X ? OSC_XMODE_XR_SKIPS_X : 0;
      for_windows(ua, [&](const UpdateArgs& w) { launch_update_xr(w, grid, h.stream); });
    }
    if (mapped) {
      launch_reduce_beta(h.part0.p, h.part1.p, grid, b.ld, b.c0, b.c1, h.rz.p, h.beta.p, res_slots + it, g, h.stream,
                         done_ctr + it, h.res_host_dev + it);
      return;
    }
    launch_reduce_beta(h.part0.p, h.part1.p, grid, b.ld, b.c0, b.c1, h.rz.p, h.beta.p, res_slots + it, g, h.stream);
    if (overlap) {  // max over the shards (solver.py:29) and its way to the host, beside the next iteration's first kernels
      HIP_CHECK(hipEventRecord(h.step_events[(size_t)it], h.stream));
      HIP_CHECK(hipStreamWaitEvent(h.comm_stream, h.step_events[(size_t)it], 0));
      h.comm->allreduce(res_slots + it, 1, COMM_F32, COMM_MAX, h.comm_stream);
      launch_publish_word(res_slots + it, reinterpret_cast<uint32_t*>(h.res_host_dev + it), h.comm_stream);
      return;
    }
    // column-sharded: the stop test is the max over all shards (solver.py:29)
    if (h.comm) h.comm->allreduce(res_slots + it, 1, COMM_F32, COMM_MAX, h.stream);
    HIP_CHECK(hipMemcpyAsync(h.res_host + it, res_slots + it, 4, hipMemcpyDeviceToHost, h.stream));
    HIP_CHECK(hipEventRecord(h.iter_events[(size_t)it], h.stream));
  };
  auto wait_residual = [&](int it) -> float {
    if (!polled) {
      HIP_CHECK(hipEventSynchronize(h.iter_events[(size_t)it]));
      return h.res_host[it];
    }
    volatile uint32_t* slot = reinterpret_cast<volatile uint32_t*>(h.res_host) + it;
    const double t_start = now_ms();
    for (uint64_t spin = 1;; ++spin) {
      const uint32_t bits = *slot;
      if (bits != kPending) {
        float v;
        std::memcpy(&v, &bits, 4);
        return v;
      }
      if ((spin & 0x3FFF) == 0) {  // every ~16k polls: has the stream died or drained without publishing?
        hipError_t q = hipStreamQuery(h.stream);
        if (q == hipSuccess && overlap) q = hipStreamQuery(h.comm_stream);  // the word comes out of the second stream
        if (q != hipSuccess && q != hipErrorNotReady) hip_check(q, "hipStreamQuery (CG residual wait)", __FILE__, __LINE__);
        if (q == hipSuccess && *slot == kPending) throw HipError("CG iteration finished without publishing its residual");
        if (now_ms() - t_start > 120000.0) throw HipError("timeout waiting for a CG residual");
      }
      __builtin_ia32_pause();
    }
  };

  h.history.clear();
  CgResult out{max_iters, 0.f, b.X};
  const size_t prof_mark = h.prof_pending.size();
  // Iteration it + 1 is enqueued before iteration it's residual is read -- except behind the iteration the previous
  // solve of this handle converged in: repeated settles of one lattice take the same count, and the five gated-off
  // launches of a needless speculative iteration cost ~22 us (8 % of a settle at N = 20000, D = 128; ungated under a
  // communicator: a whole iteration).  A wrong guess the other way costs one host round trip: the iteration is then
  // enqueued after its predecessor's residual has been read.
  // (Every rank of a sharded solve sees the same residuals, hence takes the same decisions.)
  int enqueued = 1;
  enqueue_iter(1, false);
  for (int it = 1; it <= max_iters; ++it) {
    if (it < max_iters && it != stop_guess && enqueued == it) {
      enqueue_iter(++enqueued, true);  // speculative: no-ops if `it` converged (overlap: ungated, scratch arrays only)
    } else if (xs.finish_before_wait(it)) {
      // nothing is enqueued behind this iteration for now (the expected last one): its x update goes out at once.  The
      // host has seen iteration it - 1 unconverged, so iteration `it` is a real one whatever its residual will say.
      finish_x(it);
    }
    const float res = wait_residual(it);
    h.history.push_back(res);
    out.res = res;
    if ((double)res <= (double)tol) {
      out.iters = it;
      break;
    }
    if (it < max_iters && enqueued == it) {  // the guess was wrong: go on
      if (xs.restore_r(it)) {  // ... from the r this iteration computed but did not keep
        ua.gate = nullptr;
        ua.xmode = OSC_XMODE_XR_SKIPS_X;
        for_windows(ua, [&](const UpdateArgs& w) { launch_update_xr(w, grid, h.stream); });
      }
      enqueue_iter(++enqueued, false);
    }
  }
  h.predicted_iters[b.kind] = out.iters;
  // the last iteration's x update rode in a gated p update that did not run (the solve converged under a speculative
  // iteration): alpha and p are still that iteration's
  if (xs.finish_at_end(out.iters)) finish_x(out.iters);
  // The solution is complete once the last residual is out; what may still be queued are the gated-off launches of
  // the speculative iteration (they return at once and write nothing).  With the mapped read-back the stream is left
  // to drain on its own -- later calls are ordered behind it anyway; the copy + event path keeps its full wait.
  // (overlap: the same; the second stream is drained by whoever next touches the residual slots it writes -- drain_comm_stream)
  if (!polled || h.prof_on) sync(h);
  for (size_t i = prof_mark; i < h.prof_pending.size(); ++i)  // speculative (gated-off) launches are not samples
    if (h.prof_pending[i].iter > out.iters) h.prof_pending[i].which = -1;
  return out;
}

// Column-sharded runs: every rank owns columns [c0, c1) of an N x ld array.  Make the whole array valid on every
// rank: one ncclBroadcast of each rank's packed slab (slab widths may differ by 4 columns, so not an all-gather).
// Collective: every rank must call it.
void gather_columns(L& h, float* arr) {
  if (!h.comm || h.world <= 1) return;
  const int32_t q = h.dcols / 4;
  int32_t wmax = 0;
  for (int r = 0; r < h.world; ++r) wmax = std::max(wmax, (int32_t)(((int64_t)q * (r + 1) / h.world - (int64_t)q * r / h.world) * 4));
  h.comm_buf.alloc((size_t)h.N * wmax);
  for (int r = 0; r < h.world; ++r) {
    const int32_t lo = (int32_t)((int64_t)q * r / h.world) * 4, hi = (int32_t)((int64_t)q * (r + 1) / h.world) * 4;
    const int32_t w = hi - lo;
    if (w <= 0) continue;
    if (r == h.rank)
      HIP_CHECK(hipMemcpy2DAsync(h.comm_buf.p, (size_t)w * 4, arr + lo, (size_t)h.ld * 4, (size_t)w * 4, (size_t)h.N,
                                 hipMemcpyDeviceToDevice, h.stream));
    h.comm->broadcast_group({CommXfer{h.comm_buf.p, (size_t)h.N * w * 4, r}}, h.stream);
    if (r != h.rank)
      HIP_CHECK(hipMemcpy2DAsync(arr + lo, (size_t)h.ld * 4, h.comm_buf.p, (size_t)w * 4, (size_t)w * 4, (size_t)h.N,
                                 hipMemcpyDeviceToDevice, h.stream));
  }
}

// ---- row-sharded CG (BASELINE north_star wording) --------------------------------------------------------------
// Rank r owns rows [N r/G, N (r+1)/G) of every N x D array and of the lattice graph.  Per iteration: the local rows of
// the search direction p are exchanged so every rank holds all of p for the neighbour gathers ("halo": on i.i.d.
// anchors ~all rows are somebody's neighbour, so the halo is the whole array), and the column sums (p.Ap, then
// [r.r, r.z]) are completed with all-reduces of fp64 D-vectors before alpha / beta / the residual are formed.
struct RowShard {
  int64_t r0, r1;
};

std::vector<RowShard> row_shards(const L& h) {
  std::vector<RowShard> v;
  if (h.comm) {
    v.push_back({h.N * h.rank / h.world, h.N * (h.rank + 1) / h.world});
  } else {
    const int V = std::max(1, h.fake_row_shards);
    for (int s = 0; s < V; ++s) v.push_back({h.N * s / V, h.N * (s + 1) / V});
  }
  return v;
}

// make every rank's copy of `arr` complete: each rank broadcasts its own row block (grouped, in place)
void exchange_rows(L& h, float* arr, int32_t ld) {
  if (!h.comm) return;  // (a 1-rank communicator still runs the calls: that is how one GPU exercises this path)
  std::vector<CommXfer> pieces;
  for (int r = 0; r < h.world; ++r) {
    const int64_t a = h.N * r / h.world, b = h.N * (r + 1) / h.world;
    pieces.push_back(CommXfer{arr + (size_t)a * ld, (size_t)(b - a) * ld * 4, r});
  }
  h.comm->broadcast_group(pieces, h.stream);
}

void allreduce_sums(L& h, double* buf, size_t n) {
  if (!h.comm) return;
  h.comm->allreduce(buf, n, COMM_F64, COMM_SUM, h.stream);
}


// ---- halo lists ---------------------------------------------------------------------------------------------------
// Which rows of the search direction a rank needs from its peers: the off-partition column ids its ELL rows (and its
// rows of the chain's path graph) reference.  The adjacency is symmetric (by construction of the build, enforced on
// injection), so "peer q needs my row i" == "my row i has a neighbour in q's row block": both lists of a pair of
// ranks follow from each rank's OWN rows, sorted by row id on both sides, and no index lists are exchanged -- only the
// counts, once, as a consistency check and to take the same full-exchange decision everywhere.
void build_halo_plan(L& h) {
  L::HaloPlan& hp = h.halo;
  const int G = h.world, me = h.rank;
  auto lo = [&](int r) { return host::row_lo(h.N, G, r); };
  const int64_t r0 = lo(me), r1 = lo(me + 1), nloc = r1 - r0;
  std::vector<int32_t> col((size_t)nloc * h.width), deg((size_t)nloc);
  if (nloc > 0) {
    HIP_CHECK(hipMemcpyAsync(col.data(), h.ell_col.p + (size_t)r0 * h.width, col.size() * 4, hipMemcpyDeviceToHost, h.stream));
    HIP_CHECK(hipMemcpyAsync(deg.data(), h.deg.p + r0, (size_t)nloc * 4, hipMemcpyDeviceToHost, h.stream));
  }
  sync(h);
  std::vector<std::pair<int64_t, int64_t>> chain_edges;  // path graph: consecutive chain nodes (graph.py:96-111), device row ids
  if (h.chain_present && h.lamP > 0.0f) {
    auto id = [&](int32_t v) { return permuted(h) ? h.inv_h[(size_t)v] : v; };
    for (size_t t = 0; t + 1 < h.chain_nodes.size(); ++t) chain_edges.emplace_back(id(h.chain_nodes[t]), id(h.chain_nodes[t + 1]));
  }
  host::HaloLists hl = host::build_halo_lists(h.N, G, me, h.width, col.data(), deg.data(), chain_edges);
  hp.give_off = hl.give_off;
  hp.need_off = hl.need_off;
  const std::vector<int32_t>&gi = hl.give_idx, &ni = hl.need_idx;
  hp.give_rows = (int64_t)gi.size();
  hp.need_rows = (int64_t)ni.size();
  // counts of every (rank, peer) pair, all-gathered: row r = [need from 0..G-1 | give to 0..G-1] of rank r
  DevBuf<int32_t> cnt_d;
  cnt_d.alloc((size_t)G * 2 * G);
  std::vector<int32_t> mine((size_t)2 * G), all((size_t)G * 2 * G);
  for (int q = 0; q < G; ++q) {
    mine[(size_t)q] = (int32_t)(hp.need_off[(size_t)q + 1] - hp.need_off[(size_t)q]);
    mine[(size_t)G + q] = (int32_t)(hp.give_off[(size_t)q + 1] - hp.give_off[(size_t)q]);
  }
  HIP_CHECK(hipMemcpyAsync(cnt_d.p + (size_t)me * 2 * G, mine.data(), (size_t)2 * G * 4, hipMemcpyHostToDevice, h.stream));
  h.comm->allgather(cnt_d.p, (size_t)2 * G * 4, h.stream);
  HIP_CHECK(hipMemcpyAsync(all.data(), cnt_d.p, all.size() * 4, hipMemcpyDeviceToHost, h.stream));
  sync(h);
  const host::HaloDecision dec = host::halo_decide(h.N, G, all);
  if (!dec.consistent) throw CommError("halo plan: need / give counts of a rank pair differ (asymmetric lattice graph?)");
  hp.need_rows_max = dec.need_rows_max;
  bool full = dec.full;
  const int force = h.halo_force;  // OSC_HALO = full | lists: force one exchange form (tests, A/B); same on every rank
  if (force == 1) full = true;
  if (force == 2) full = false;
  hp.full = full;
  hp.give_idx.alloc(std::max<size_t>(1, gi.size()));
  hp.need_idx.alloc(std::max<size_t>(1, ni.size()));
  if (!gi.empty()) HIP_CHECK(hipMemcpyAsync(hp.give_idx.p, gi.data(), gi.size() * 4, hipMemcpyHostToDevice, h.stream));
  if (!ni.empty()) HIP_CHECK(hipMemcpyAsync(hp.need_idx.p, ni.data(), ni.size() * 4, hipMemcpyHostToDevice, h.stream));
  if (!full) {
    hp.send.alloc(std::max<size_t>(1, gi.size() * (size_t)h.ld));
    hp.recv.alloc(std::max<size_t>(1, ni.size() * (size_t)h.ld));
  }
  sync(h);
  hp.epoch = h.graph_epoch;
}

// the per-iteration halo exchange of `arr` (N x ld, every rank's own row block current): afterwards the rows this
// rank's operator gathers from are current too
void halo_exchange(L& h, float* arr, int32_t ld) {
  if (!h.comm) return;
  if (h.halo.epoch != h.graph_epoch) build_halo_plan(h);
  L::HaloPlan& hp = h.halo;
  if (hp.full || ld != h.ld) {
    exchange_rows(h, arr, ld);
    return;
  }
  if (hp.give_rows > 0) launch_move_rows(hp.send.p, arr, hp.give_idx.p, hp.give_rows, ld, false, h.stream);  // pack
  std::vector<CommXfer> sends, recvs;
  for (int q = 0; q < h.world; ++q) {
    const int64_t g0 = hp.give_off[(size_t)q], g1 = hp.give_off[(size_t)q + 1];
    const int64_t n0 = hp.need_off[(size_t)q], n1 = hp.need_off[(size_t)q + 1];
    if (g1 > g0) sends.push_back(CommXfer{hp.send.p + (size_t)g0 * ld, (size_t)(g1 - g0) * ld * 4, q});
    if (n1 > n0) recvs.push_back(CommXfer{hp.recv.p + (size_t)n0 * ld, (size_t)(n1 - n0) * ld * 4, q});
  }
  h.comm->exchange(sends, recvs, h.stream);
  if (hp.need_rows > 0) launch_move_rows(arr, hp.recv.p, hp.need_idx.p, hp.need_rows, ld, true, h.stream);  // unpack
}

CgResult run_cg_rows(L& h, const OpParams& op, const CgBuffers& b, bool with_path, int max_iters, float tol) {
  const std::vector<RowShard> shards = row_shards(h);
  const int V = (int)shards.size();
  const int grid = cg_grid(h);
  const size_t pn = (size_t)V * grid * b.ld;  // one block of partial rows per local shard
  if (h.part0.n < pn) h.part0.alloc(pn);
  if (h.part1.n < pn) h.part1.alloc(pn);
  h.sums.alloc((size_t)2 * b.ld);
  double* s0 = h.sums.p;
  double* s1 = h.sums.p + b.ld;
  HIP_CHECK(hipMemsetAsync(h.res_bits.p, 0, ((size_t)max_iters + 2) * 4, h.stream));
  if (h.res_host_n < (size_t)max_iters + 2) {
    if (h.res_host) (void)hipHostFree(h.res_host);
    h.res_host = nullptr;
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h.res_host), ((size_t)max_iters + 2) * 4, hipHostMallocDefault));
    h.res_host_n = (size_t)max_iters + 2;
  }
  while (h.iter_events.size() < (size_t)max_iters + 2) {
    hipEvent_t e;
    HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    h.iter_events.push_back(e);
  }
  const float* res_dev = reinterpret_cast<const float*>(h.res_bits.p);
  SpmmArgs sa{};
  sa.g = graph_view(h, with_path);
  sa.op = op;
  sa.B = b.B;
  sa.psi = b.psi;
  sa.ld = b.ld;
  sa.c0 = b.c0;
  sa.c1 = b.c1;
  UpdateArgs ua{};
  ua.X = b.X;
  ua.R = b.R;
  ua.P = b.P;
  ua.AP = b.AP;
  ua.B = b.B;
  ua.alpha = h.alpha.p;
  ua.beta = h.beta.p;
  ua.op = op;
  ua.ld = b.ld;
  ua.c0 = b.c0;
  ua.c1 = b.c1;
  auto for_shards_spmm = [&](int mode, int iter) {
    for (int s = 0; s < V; ++s) {
      sa.row0 = shards[(size_t)s].r0;
      sa.N = shards[(size_t)s].r1;
      sa.part = h.part0.p + (size_t)s * grid * b.ld;
      spmm_slabbed(h, mode, sa, grid, iter);
    }
  };
  // r = b - A x0 ; z ; p ; rz
  sa.X = b.x0;
  sa.OUT = b.X;
  sa.R = b.R;
  sa.P = b.P;
  sa.U = b.rhsU;
  sa.Y = b.rhsY;
  sa.gate = nullptr;
  for_shards_spmm(SPMM_INIT, 0);
  launch_reduce_sum(h.part0.p, V * grid, b.ld, b.c0, b.c1, s0, h.stream);
  allreduce_sums(h, s0 + b.c0, (size_t)(b.c1 - b.c0));
  launch_finish_init(s0, b.c0, b.c1, h.rz.p, h.stream);
  halo_exchange(h, b.P, b.ld);
  sa.X = b.P;
  sa.OUT = b.AP;

  auto enqueue_iter = [&](int it) {
    const Gate g{it > 1 ? res_dev + (it - 1) : nullptr, tol};
    sa.gate = g.p;
    sa.gate_tol = tol;
    ua.gate = g.p;
    ua.gate_tol = tol;
    if (it > 1) {
      for (int s = 0; s < V; ++s) {
        ua.row0 = shards[(size_t)s].r0;
        ua.N = shards[(size_t)s].r1;
        ProfScope ps(h, 2, it);
        for_windows(ua, [&](const UpdateArgs& w) { launch_update_p(w, grid, h.stream); });
      }
      halo_exchange(h, b.P, b.ld);  // the halo exchange of this iteration
    }
    for_shards_spmm(SPMM_AP, it);
    launch_reduce_sum_gated(h.part0.p, V * grid, b.ld, b.c0, b.c1, s0, g, h.stream);
    allreduce_sums(h, s0 + b.c0, (size_t)(b.c1 - b.c0));
    launch_finish_alpha(s0, b.c0, b.c1, h.rz.p, h.alpha.p, g, h.stream);
    for (int s = 0; s < V; ++s) {
      ua.row0 = shards[(size_t)s].r0;
      ua.N = shards[(size_t)s].r1;
      ua.part_rr = h.part0.p + (size_t)s * grid * b.ld;
      ua.part_rz = h.part1.p + (size_t)s * grid * b.ld;
      ProfScope ps(h, 1, it);
      for_windows(ua, [&](const UpdateArgs& w) { launch_update_xr(w, grid, h.stream); });
    }
    launch_reduce_sum_gated(h.part0.p, V * grid, b.ld, b.c0, b.c1, s0, g, h.stream);
    launch_reduce_sum_gated(h.part1.p, V * grid, b.ld, b.c0, b.c1, s1, g, h.stream);
    allreduce_sums(h, s0, (size_t)2 * b.ld);  // [r.r | r.z] in one message
    launch_finish_beta(s0, s1, b.c0, b.c1, h.rz.p, h.beta.p, h.res_bits.p + it, g, h.stream);
    HIP_CHECK(hipMemcpyAsync(h.res_host + it, h.res_bits.p + it, 4, hipMemcpyDeviceToHost, h.stream));
    HIP_CHECK(hipEventRecord(h.iter_events[(size_t)it], h.stream));
  };

  h.history.clear();
  CgResult out{max_iters, 0.f, b.X};
  const size_t prof_mark = h.prof_pending.size();
  enqueue_iter(1);
  for (int it = 1; it <= max_iters; ++it) {
    if (it < max_iters) enqueue_iter(it + 1);
    HIP_CHECK(hipEventSynchronize(h.iter_events[(size_t)it]));
    const float res = h.res_host[it];
    h.history.push_back(res);
    out.res = res;
    if ((double)res <= (double)tol) {
      out.iters = it;
      break;
    }
  }
  exchange_rows(h, b.X, b.ld);  // every rank leaves with the whole solution
  sync(h);
  for (size_t i = prof_mark; i < h.prof_pending.size(); ++i)
    if (h.prof_pending[i].iter > out.iters) h.prof_pending[i].which = -1;
  return out;
}

bool row_mode(const L& h) { return h.shard_mode == 1 && (h.comm != nullptr || h.fake_row_shards > 1); }

// The ONE place the per-handle OSC_* switches are read (osc_create, osc_rebuild_graph).  Process-wide ones are read where
// the process-wide object is made: OSC_POOL_MB (device memory pool), OSC_PINNED_DL / OSC_COPY_THREADS (read-back staging),
// OSC_LOOPBACK_TIMEOUT_S / OSC_RCCL_PROXY (communicator backends, comm.hip), OSC_LD (osc_create, before the arrays are sized).
bool env_num(const char* name, int& out) {
  const char* e = getenv(name);
  if (e) out = atoi(e);
  return e != nullptr;
}

// The switches a handle reads ONCE, at creation: how its solves run and how it is sharded.  They stay what they were when
// the graph is rebuilt (osc_rebuild_graph): the column window, the halo plan and the communicator were laid out for them
// (a handle whose OSC_SHARD changed under it would solve over a partial window).
void read_env_solver(L& h) {
  auto num = env_num;
  int v = 0;
  if (num("OSC_SPMM_SLAB", v)) h.spmm_slab = v < 0 ? -1 : (v / 4) * 4;
  if (num("OSC_SPMM_XS", v)) h.spmm_xs = v != 0 ? 1 : 0;
  if (num("OSC_XS_NB", v)) h.xs_nb = std::max(1, v);
  if (num("OSC_XS_GROUPS", v)) h.xs_groups_cap = v >= 8 ? 8 : v >= 4 ? 4 : v >= 2 ? 2 : 1;
  if (num("OSC_P_BLOCKED", v)) h.p_blocked = v != 0;
  if (num("OSC_SPMM_BLOCKED", v)) h.spmm_blocked = v;
  if (num("OSC_BLK_STAMP", v)) h.blk_stamp = v != 0;
  if (num("OSC_BLK_VARIANT", v)) h.blk_variant = (v >= 0 && v < blocked_variants()) ? v : -1;
  if (num("OSC_BLK_WIDE_MIN_ROWS", v)) h.blk_wide_min_rows = std::max(0, v);
  if (num("OSC_BLK_INIT", v)) {
    h.blk_init = v != 0;
    h.blk_init_fused = v == 1;
  }
  if (num("OSC_SPMM_DEEP", v)) h.spmm_deep = v != 0;
  if (num("OSC_COMM_OVERLAP", v)) h.comm_overlap = v != 0 ? 1 : 0;
  if (num("OSC_X_DEFER", v)) {
    h.x_defer = v != 0;
    h.x_last_form = v == 1;
  }
  if (num("OSC_SMALL_PATH", v)) h.small_path = v != 0;
  if (const char* e = getenv("OSC_SHARD")) h.shard_mode = !strcmp(e, "row") ? 1 : 0;
  if (num("OSC_ROW_FAKE_SHARDS", v)) h.fake_row_shards = std::max(0, v);
  h.fake_col_w = 0;
  if (const char* e = getenv("OSC_FAKE_COL_SHARD")) {  // "r/w"
    int r = 0, w = 1;
    if (sscanf(e, "%d/%d", &r, &w) == 2 && w >= 1 && r >= 0 && r < w) h.fake_col_r = r, h.fake_col_w = w;
  }
}

// The switches of the lattice build: read at creation and again by every rebuild.
void read_env_build(L& h) {
  auto num = env_num;
  int v = 0;
  if (num("OSC_REORDER", v)) h.reorder = v != 0 ? 1 : 0;
  else h.reorder = -1;
  h.knn_mode = 0;
  if (const char* e = getenv("OSC_KNN_MODE")) h.knn_mode = !strcmp(e, "exact") ? 1 : !strcmp(e, "prefilter") ? 2 : !strcmp(e, "panel") ? 3 : 0;
  h.knn_fake_shards = 0;
  if (num("OSC_KNN_FAKE_SHARDS", v)) h.knn_fake_shards = std::max(0, v);
  h.knn_splits = 0;
  if (num("OSC_KNN_SPLITS", v)) h.knn_splits = std::max(1, v);
  h.knn_scatter = !(num("OSC_KNN_PANEL_SCATTER", v) && v == 0);
  h.knn_sym = !(num("OSC_KNN_PANEL_SYM", v) && v == 0);
  h.knn_tune = KnnPanelTune{};
  if (num("OSC_KNN_PANEL_NRG", v)) h.knn_tune.nrg = v;
  if (const char* e = getenv("OSC_KNN_PANEL_RHO")) h.knn_tune.rho = atof(e);
  if (num("OSC_KNN_PANEL_T", v)) h.knn_tune.T = v;
  if (num("OSC_KNN_PANEL_RANK", v)) h.knn_tune.rank = v;
  if (num("OSC_KNN_TILE_WIDE", v)) h.knn_tune.tile_wide = v != 0 ? 1 : 0;
  if (num("OSC_KNN_TILE_GROUP_MB", v)) h.knn_tune.tile_group_mb = v;
  h.bfs_host = num("OSC_BFS_HOST", v) && v != 0;
  h.halo_force = 0;
  if (const char* e = getenv("OSC_HALO")) h.halo_force = !strcmp(e, "full") ? 1 : !strcmp(e, "lists") ? 2 : 0;
}

void read_env(L& h) {
  read_env_solver(h);
  read_env_build(h);
}

void require_graph(L& h) {
  if (!h.have_graph) throw StateError("no lattice graph: build it (osc_create build_graph=1) or inject one (osc_set_csr)");
}

template <typename F>
int guarded(osc_handle h, F&& f) {
  if (!h) return OSC_E_INVALID;
  try {
    use_device(*h);
    alloc_ctx() = AllocCtx{h->device, h->stream};
    f(*h);
    return OSC_OK;
  } catch (const Invalid& e) {
    h->err = e.what();
    return OSC_E_INVALID;
  } catch (const StateError& e) {
    h->err = e.what();
    return OSC_E_STATE;
  } catch (const Unsupported& e) {
    h->err = e.what();
    return OSC_E_UNSUPPORTED;
  } catch (const CommError& e) {
    h->err = e.what();
    return OSC_E_COMM;
  } catch (const HipError& e) {
    h->err = e.what();
    return OSC_E_HIP;
  } catch (const std::exception& e) {
    h->err = e.what();
    return OSC_E_HIP;
  }
}

}  // namespace

// =====================================================================================================
extern "C" {

const char* osc_version(void) { return "oscillink-hip 0.1.0 (gfx950)"; }

int osc_device_count(int32_t* n) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
  if (n) *n = c;
  return OSC_OK;
}

int osc_device_name(int32_t device, char* out, int32_t cap) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return OSC_E_NODEVICE;
  snprintf(out, (size_t)cap, "%s|%s|CUs=%d", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return OSC_OK;
}

int osc_device_synchronize(int32_t device) {
  if (hipSetDevice(device) != hipSuccess) return OSC_E_NODEVICE;
  return hipDeviceSynchronize() == hipSuccess ? OSC_OK : OSC_E_HIP;
}

const char* osc_last_error(osc_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int osc_create(const float* Y, int64_t N, int32_t D, int32_t k, float row_cap, int32_t deterministic, int64_t seed,
               int32_t device, int32_t build, osc_handle* out) {
  if (!out) return OSC_E_INVALID;
  *out = nullptr;
  if (!Y || N < 1 || D < 1 || k < 1 || N >= (int64_t)1 << 31) {
    g_create_error = "osc_create: need Y != NULL, 1 <= N < 2^31, D >= 1, k >= 1";
    return OSC_E_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev) {
    g_create_error = "osc_create: no usable HIP device (this library has no CPU fallback)";
    return OSC_E_NODEVICE;
  }
  std::unique_ptr<osc_lattice> h(new osc_lattice());
  try {
    h->device = device;
    HIP_CHECK(hipSetDevice(device));
    {
      static std::mutex arch_mu;
      static std::map<int, std::string> arch;  // hipGetDeviceProperties is slow: ask once per device
      std::lock_guard<std::mutex> lk(arch_mu);
      auto it = arch.find(device);
      if (it == arch.end()) {
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device));
        it = arch.emplace(device, std::string(prop.gcnArchName)).first;
      }
      if (it->second.find("gfx950") == std::string::npos) {
        g_create_error = std::string("osc_create: device is ") + it->second + ", this build targets gfx950 only";
        return OSC_E_NODEVICE;
      }
    }
    h->stream = acquire_stream(device);
    alloc_ctx() = AllocCtx{device, h->stream};
    h->N = N;
    h->D = D;
    h->dcols = ((D + 3) / 4) * 4;
    h->ld = h->dcols;
    // Row pitch.  Gathered rows that straddle 128-byte lines cost the operator apply (stand-alone: pitch 772 vs 768
    // floats, 1.5-2.2 vs 1.06 ms; in the library at D = 1000: 12.2 ms per settle at pitch 1000, 10.7 at 1024), and a
    // pitch that is a multiple of 4 KB piles the rows of a column slab onto few L2 channels (D = 1000: 9.7 ms at pitch
    // 1056; 768 -> 800 changes nothing), so large lattices get line-aligned rows plus one line when the pitch would
    // be a multiple of 4 KB; small ones keep the dense pitch (their state lives in LDS / L2 anyway).
    // OSC_LD overrides (multiple of 4, >= D).
    if ((int64_t)N * D >= (int64_t)1 << 22) {
      h->ld = ((D + 31) / 32) * 32;
      if ((h->ld * 4) % 4096 == 0) h->ld += 32;
    }
    if (const char* e = getenv("OSC_LD")) {  // (row pitch override: needed before the arrays are sized, hence not in read_env)
      const int v = atoi(e);
      if (v >= h->dcols && v % 4 == 0) h->ld = v;
    }
    h->c0 = 0;
    h->c1 = h->dcols;
    h->k_eff = (int32_t)std::min<int64_t>(k, std::max<int64_t>(1, N - 1));
    h->row_cap = row_cap;
    h->deterministic = deterministic;
    h->seed = seed;
    read_env(*h);
    if (h->fake_col_w > 0) {  // measurement hook: work on rank r's column slab of w, no communicator
      std::tie(h->c0, h->c1) = host::column_shard(h->dcols, h->fake_col_r, h->fake_col_w);
      if (h->c1 <= h->c0) throw Invalid("OSC_FAKE_COL_SHARD: more ranks than 4-column groups");
    }
    const size_t n = (size_t)N * h->ld;
    for (DevBuf<float>* b : {&h->Y, &h->U, &h->X, &h->R, &h->P, &h->AP, &h->Ustar}) b->alloc(n);
    HIP_CHECK(hipMemsetAsync(h->Y.p, 0, n * 4, h->stream));
    upload_rows(*h, h->Y.p, Y);
    HIP_CHECK(hipMemcpyAsync(h->U.p, h->Y.p, n * 4, hipMemcpyDeviceToDevice, h->stream));
    for (DevBuf<float>* b : {&h->X, &h->R, &h->P, &h->AP, &h->Ustar}) HIP_CHECK(hipMemsetAsync(b->p, 0, n * 4, h->stream));
    h->B.alloc((size_t)N);
    std::vector<float> ones((size_t)N, 1.0f);
    HIP_CHECK(hipMemcpyAsync(h->B.p, ones.data(), (size_t)N * 4, hipMemcpyHostToDevice, h->stream));
    h->psi.alloc((size_t)h->ld);
    HIP_CHECK(hipMemsetAsync(h->psi.p, 0, (size_t)h->ld * 4, h->stream));
    sync(*h);
    if (build) build_graph(*h);
  } catch (const Unsupported& e) {
    g_create_error = e.what();
    return OSC_E_UNSUPPORTED;
  } catch (const std::exception& e) {
    g_create_error = e.what();
    return OSC_E_HIP;
  }
  *out = h.release();
  return OSC_OK;
}

int osc_host_alloc(int64_t bytes, void** out) {
  if (!out || bytes <= 0) return OSC_E_INVALID;
  *out = host_pool_alloc((size_t)bytes);
  return *out ? OSC_OK : OSC_E_HIP;
}

int osc_host_free(void* p) {
  if (!p) return OSC_OK;
  return host_pool_free(p) ? OSC_OK : OSC_E_INVALID;
}

int osc_destroy(osc_handle h) {
  if (!h) return OSC_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  alloc_ctx() = AllocCtx{h->device, nullptr};  // drained: the blocks go back to the pool without further waits
  delete h;
  return OSC_OK;
}

int osc_rebuild_graph(osc_handle h, int32_t k, float row_cap, int32_t deterministic, int64_t seed) {
  return guarded(h, [&](L& l) {
    if (k < 1) throw Invalid("kneighbors must be >= 1");
    read_env_build(l);  // (the build's switches only: solver and sharding switches are fixed at creation)
    l.k_eff = (int32_t)std::min<int64_t>(k, std::max<int64_t>(1, l.N - 1));
    l.row_cap = row_cap;
    l.deterministic = deterministic;
    l.seed = seed;
    build_graph(l);
  });
}

int osc_graph_stats(osc_handle h, int64_t* nnz, int32_t* max_deg, double* build_ms) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (nnz) *nnz = l.nnz;
    if (max_deg) *max_deg = l.max_deg;
    if (build_ms) *build_ms = l.build_ms;
  });
}

int osc_build_info(osc_handle h, int32_t* prefilter, int32_t* fallback_rows, int64_t* small_solves) {
  return guarded(h, [&](L& l) {
    if (prefilter) *prefilter = l.knn_prefilter ? (l.knn_panel ? 2 : 1) : 0;
    if (fallback_rows) *fallback_rows = l.knn_fallback_rows;
    if (small_solves) *small_solves = l.small_solves;
  });
}

int osc_apply_info(osc_handle h, int32_t* src_blocks, int64_t* blocked_applies) {
  return guarded(h, [&](L& l) {
    if (src_blocks) *src_blocks = l.blk_last;
    if (blocked_applies) *blocked_applies = l.blk_applies;
  });
}

int osc_get_blocked_copy(osc_handle h, int32_t nb, int32_t* slot_col, float* slot_w, int32_t* over_first, int32_t* over_count,
                         int32_t* over_col, float* over_w, int32_t over_cap) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (nb < 1 || nb > OSC_MAX_SRC_BLOCKS) throw Invalid("osc_get_blocked_copy: 1 <= nb <= 32");
    if (permuted(l)) throw Unsupported("osc_get_blocked_copy: the lattice is stored in an internal row order");
    const BlockedView bv = blocked_view(l, nb);
    const size_t ns = (size_t)nb * l.N * OSC_BLK_SLOTS;
    std::vector<int2> hs(ns), hr((size_t)l.N);
    HIP_CHECK(hipMemcpyAsync(hs.data(), bv.slots, ns * sizeof(int2), hipMemcpyDeviceToHost, l.stream));
    HIP_CHECK(hipMemcpyAsync(hr.data(), bv.rest, (size_t)l.N * sizeof(int2), hipMemcpyDeviceToHost, l.stream));
    sync(l);
    int64_t total = 0;
    for (int64_t i = 0; i < l.N; ++i) total = std::max<int64_t>(total, (int64_t)hr[(size_t)i].x + hr[(size_t)i].y);
    std::vector<int2> ho((size_t)total);
    if (total > 0) {
      HIP_CHECK(hipMemcpyAsync(ho.data(), bv.over, (size_t)total * sizeof(int2), hipMemcpyDeviceToHost, l.stream));
      sync(l);
    }
    for (size_t t = 0; t < ns; ++t) {
      if (slot_col) slot_col[t] = hs[t].x;
      if (slot_w) std::memcpy(slot_w + t, &hs[t].y, 4);
    }
    for (int64_t i = 0; i < l.N; ++i) {
      if (over_first) over_first[i] = hr[(size_t)i].x;
      if (over_count) over_count[i] = hr[(size_t)i].y;
    }
    if (total > over_cap && (over_col || over_w)) throw Invalid("osc_get_blocked_copy: over_cap too small");
    for (int64_t t = 0; t < total; ++t) {
      if (over_col) over_col[t] = ho[(size_t)t].x;
      if (over_w) std::memcpy(over_w + t, &ho[(size_t)t].y, 4);
    }
  });
}

int osc_order_info(osc_handle h, int32_t* reordered, double* clustering) {
  return guarded(h, [&](L& l) {
    if (reordered) *reordered = l.reordered ? 1 : 0;
    if (clustering) *clustering = l.clustering;
  });
}

int osc_get_row_order(osc_handle h, int32_t* perm) {
  return guarded(h, [&](L& l) {
    if (!perm) throw Invalid("osc_get_row_order: perm is NULL");
    for (int64_t i = 0; i < l.N; ++i) perm[i] = l.perm_h.empty() ? (int32_t)i : l.perm_h[(size_t)i];
  });
}

int osc_spmm_plan(osc_handle h, int32_t* launches, int32_t* slab_cols, int32_t* xs_workgroups) {
  return guarded(h, [&](L& l) {
    const int32_t ncols = l.c1 - l.c0;
    const int nb = xs_plan(l, ncols, cg_grid(l));
    const int32_t slab = nb ? 32 : auto_slab(l, ncols);
    if (launches) *launches = nb ? 1 : (ncols + slab - 1) / slab;
    if (slab_cols) *slab_cols = nb ? ncols : slab;
    if (xs_workgroups) *xs_workgroups = nb;
  });
}

int osc_get_csr(osc_handle h, int64_t* rowptr, int32_t* col, float* a, float* w, float* sqrt_deg) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    const size_t n = (size_t)l.N * l.width;
    std::vector<int32_t> hc(n), hd((size_t)l.N);
    std::vector<float> ha(n), hw(n);
    HIP_CHECK(hipMemcpyAsync(hc.data(), l.ell_col.p, n * 4, hipMemcpyDeviceToHost, l.stream));
    HIP_CHECK(hipMemcpyAsync(ha.data(), l.ell_a.p, n * 4, hipMemcpyDeviceToHost, l.stream));
    HIP_CHECK(hipMemcpyAsync(hw.data(), l.ell_w.p, n * 4, hipMemcpyDeviceToHost, l.stream));
    HIP_CHECK(hipMemcpyAsync(hd.data(), l.deg.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    if (sqrt_deg) HIP_CHECK(hipMemcpyAsync(sqrt_deg, l.sqrt_deg.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    sync(l);
    std::vector<float> hsd;
    if (sqrt_deg && permuted(l)) {  // downloaded in device order above: put it back into API order
      hsd.assign(sqrt_deg, sqrt_deg + l.N);
      for (int64_t i = 0; i < l.N; ++i) sqrt_deg[l.perm_h[(size_t)i]] = hsd[(size_t)i];
    }
    int64_t pos = 0;
    std::vector<std::pair<int32_t, size_t>> ent;  // (API column id, ELL slot) of one row, sorted by column
    for (int64_t r = 0; r < l.N; ++r) {  // r = API row
      const int64_t i = permuted(l) ? l.inv_h[(size_t)r] : r;  // device row
      if (rowptr) rowptr[r] = pos;
      ent.clear();
      for (int e = 0; e < hd[(size_t)i]; ++e) {
        const size_t o = (size_t)i * l.width + e;
        ent.emplace_back(permuted(l) ? l.perm_h[(size_t)hc[o]] : hc[o], o);
      }
      if (permuted(l)) std::sort(ent.begin(), ent.end());
      for (auto& pe : ent) {
        if (col) col[pos] = pe.first;
        if (a) a[pos] = ha[pe.second];
        if (w) w[pos] = hw[pe.second];
        ++pos;
      }
    }
    if (rowptr) rowptr[l.N] = pos;
  });
}

int osc_edge_prefix(osc_handle h, int32_t cap, int64_t* pairs, int32_t* n_out) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (cap < 0 || (cap > 0 && !pairs) || !n_out) throw Invalid("osc_edge_prefix: bad arguments");
    int32_t n = 0;
    std::vector<int32_t> hc, hd;
    std::vector<int32_t> ent;
    // rows are fetched in chunks until `cap` edges are collected (a few dozen rows at k = 32), not the whole graph
    for (int64_t r0 = 0; r0 < l.N && n < cap;) {
      const int64_t chunk = std::min<int64_t>(l.N - r0, permuted(l) ? 1 : 256);
      const int64_t d0 = permuted(l) ? l.inv_h[(size_t)r0] : r0;  // device row (chunk == 1 when rows are permuted)
      hc.resize((size_t)chunk * l.width);
      hd.resize((size_t)chunk);
      HIP_CHECK(hipMemcpyAsync(hc.data(), l.ell_col.p + (size_t)d0 * l.width, hc.size() * 4, hipMemcpyDeviceToHost, l.stream));
      HIP_CHECK(hipMemcpyAsync(hd.data(), l.deg.p + d0, (size_t)chunk * 4, hipMemcpyDeviceToHost, l.stream));
      sync(l);
      for (int64_t t = 0; t < chunk && n < cap; ++t) {
        ent.clear();
        for (int e = 0; e < hd[(size_t)t]; ++e) {
          const int32_t c = hc[(size_t)t * l.width + e];
          ent.push_back(permuted(l) ? l.perm_h[(size_t)c] : c);
        }
        if (permuted(l)) std::sort(ent.begin(), ent.end());
        for (int32_t c : ent) {
          if (n >= cap) break;
          pairs[2 * (size_t)n] = r0 + t;
          pairs[2 * (size_t)n + 1] = c;
          ++n;
        }
      }
      r0 += chunk;
    }
    *n_out = n;
  });
}

int osc_set_csr(osc_handle h, const int64_t* rowptr, const int32_t* col, const float* a) {
  return guarded(h, [&](L& l) {
    // validated and packed on the host (host_logic.hpp: sorted columns, no diagonal, no duplicates, symmetric)
    host::PackedEll pk;
    try {
      pk = host::pack_csr(l.N, rowptr, col, a);
    } catch (const host::InvalidArg& e) {
      throw Invalid(e.what());
    }
    const size_t W = (size_t)pk.width, n = (size_t)l.N * W;
    const std::vector<int32_t>&hc = pk.col, &hd = pk.deg;
    const std::vector<float>& ha = pk.a;
    // everything above ran on the host: a refused graph leaves the handle untouched
    if (l.have_graph) drop_order(l);  // the injected ids are API ids
    else {
      l.perm_h.clear();
      l.inv_h.clear();
    }
    alloc_ell(l, (int32_t)W);
    HIP_CHECK(hipMemcpyAsync(l.ell_col.p, hc.data(), n * 4, hipMemcpyHostToDevice, l.stream));
    HIP_CHECK(hipMemcpyAsync(l.ell_a.p, ha.data(), n * 4, hipMemcpyHostToDevice, l.stream));
    HIP_CHECK(hipMemcpyAsync(l.deg.p, hd.data(), (size_t)l.N * 4, hipMemcpyHostToDevice, l.stream));
    launch_cap_and_normalize(l.ell_a.p, l.ell_w.p, l.ell_col.p, l.deg.p, l.width, (int32_t)l.N, 0.f, 0, nullptr,
                             l.sqrt_deg.p, l.stream);
    graph_counts(l);
    l.have_graph = true;
    l.have_ustar = false;
    l.knn_k = 0;
    maybe_reorder(l);
  });
}

int osc_get_knn_lists(osc_handle h, int32_t* idx, float* val, int32_t* k_eff) {
  return guarded(h, [&](L& l) {
    if (k_eff) *k_eff = l.knn_k;
    if (l.knn_k <= 0) return;
    const size_t n = (size_t)l.N * l.knn_k;
    if (idx) HIP_CHECK(hipMemcpyAsync(idx, l.knn_idx.p, n * 4, hipMemcpyDeviceToHost, l.stream));
    if (val) HIP_CHECK(hipMemcpyAsync(val, l.knn_val.p, n * 4, hipMemcpyDeviceToHost, l.stream));
    sync(l);
  });
}

int osc_set_query(osc_handle h, const float* psi, const float* gates) {
  return guarded(h, [&](L& l) {
    if (psi) {
      HIP_CHECK(hipMemsetAsync(l.psi.p, 0, (size_t)l.ld * 4, l.stream));
      HIP_CHECK(hipMemcpyAsync(l.psi.p, psi, (size_t)l.D * 4, hipMemcpyHostToDevice, l.stream));
    }
    std::vector<float> gp;
    if (gates && permuted(l)) {
      gp.resize((size_t)l.N);
      for (int64_t i = 0; i < l.N; ++i) gp[(size_t)i] = gates[l.perm_h[(size_t)i]];
      gates = gp.data();
    }
    if (gates) HIP_CHECK(hipMemcpyAsync(l.B.p, gates, (size_t)l.N * 4, hipMemcpyHostToDevice, l.stream));
    sync(l);
    l.have_ustar = false;
  });
}

int osc_set_chain(osc_handle h, const int32_t* chain, const float* weights, int32_t len, float lamP) {
  return guarded(h, [&](L& l) {
    if (lamP < 0) throw Invalid("lamP must be >= 0");
    if (len < 2 || !chain) throw Invalid("chain must contain at least two indices");
    for (int i = 0; i < len; ++i)
      if (chain[i] < 0 || chain[i] >= l.N) throw Invalid("chain indices out of bounds");
    l.chain_nodes.assign(chain, chain + len);
    if (weights) l.chain_w.assign(weights, weights + (len - 1));
    else l.chain_w.clear();
    l.chain_present = true;
    install_chain(l);
    l.lamP = lamP;
    l.have_ustar = false;
  });
}

int osc_clear_chain(osc_handle h) {
  return guarded(h, [&](L& l) {
    l.chain_present = false;
    l.lamP = 0.0f;
    l.have_ustar = false;
    ++l.graph_epoch;
  });
}

int osc_set_lams(osc_handle h, float lamG, float lamC, float lamQ) {
  return guarded(h, [&](L& l) {
    if (!(lamG > 0)) throw Invalid("lamG must be > 0 for SPD");
    if (lamC < 0) throw Invalid("lamC must be >= 0");
    if (lamQ < 0) throw Invalid("lamQ must be >= 0");
    if (l.lamG != lamG || l.lamC != lamC || l.lamQ != lamQ) l.have_ustar = false;  // U* belongs to the old lams
    l.lamG = lamG;
    l.lamC = lamC;
    l.lamQ = lamQ;
  });
}

int osc_get_U(osc_handle h, float* out) {
  return guarded(h, [&](L& l) {
    if (!out) throw Invalid("osc_get_U: out is NULL");
    if (l.u_sharded) {  // collective in column-sharded runs
      gather_columns(l, l.U.p);
      l.u_sharded = false;
    }
    download_api_order(l, out, l.U.p);
  });
}

int osc_get_Y(osc_handle h, float* out) {
  return guarded(h, [&](L& l) {
    if (!out) throw Invalid("osc_get_Y: out is NULL");
    download_api_order(l, out, l.Y.p);
  });
}

int osc_set_U(osc_handle h, const float* U) {
  return guarded(h, [&](L& l) {
    if (U) {
      if (permuted(l)) {  // API order -> device order through the scratch array
        upload_rows(l, l.AP.p, U);
        launch_move_rows(l.U.p, l.AP.p, l.perm_d.p, l.N, l.ld, false, l.stream);
      } else {
        upload_rows(l, l.U.p, U);
      }
      l.u_sharded = false;
    } else if (l.comm && l.world > 1 && l.shard_mode == 0) {  // column-sharded: only this rank's slab is needed
      HIP_CHECK(hipMemcpy2DAsync(l.U.p + l.c0, (size_t)l.ld * 4, l.Y.p + l.c0, (size_t)l.ld * 4,
                                 (size_t)(l.c1 - l.c0) * 4, (size_t)l.N, hipMemcpyDeviceToDevice, l.stream));
      l.u_sharded = true;
    } else {
      HIP_CHECK(hipMemcpyAsync(l.U.p, l.Y.p, (size_t)l.N * l.ld * 4, hipMemcpyDeviceToDevice, l.stream));
      l.u_sharded = false;
    }
    if (U) sync(l);  // (the caller's buffer is free on return; the device-side reset is ordered by the stream)
  });
}

int osc_settle(osc_handle h, float dt, int32_t max_iters, float tol, int32_t precond, int32_t warm_start, float inertia,
               int32_t* iters, float* res, double* ms) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (max_iters < 1) throw Invalid("max_iters must be >= 1");
    ensure_cg_scratch(l, max_iters);
    const OpParams op = settle_op(l, dt, precond == OSC_PRECOND_JACOBI ? 1 : 0);
    sync(l);
    const double t0 = now_ms();
    // x0 (lattice.py:751-758)
    const float* x0 = l.U.p;
    if (!warm_start) {
      x0 = l.Y.p;
    } else {
      const float w = std::max(0.0f, std::min(1.0f, inertia));
      if (w > 0.0f) {
        launch_axpby(l.AP.p, l.Y.p, 1.0f - w, l.U.p, w, (int64_t)l.N * l.ld, l.stream);
        x0 = l.AP.p;  // AP is free until the first operator apply overwrites it (INIT gathers x0 before that)
      }
    }
    // Warm start from U itself (the default): the CG runs in place on U -- the old state is only read by the INIT pass
    // (as x0 and as the rhs term), which then has no x0 copy to write, and there is nothing to swap afterwards.
    const bool in_place = x0 == l.U.p && !row_mode(l);
    CgBuffers b{x0, in_place ? l.U.p : l.X.p, l.R.p, l.P.p, l.AP.p, l.U.p, l.Y.p, l.B.p, l.psi.p, l.ld, l.c0, l.c1};
    if (in_place) b.Xalt = l.X.p;  // free in an in-place solve
    // when x0 aliases AP the INIT pass reads it completely before the first SPMM_AP launch writes AP: same stream
    const CgResult r = run_cg(l, op, b, path_active(l), max_iters, tol);
    if (r.sol == l.X.p) l.U.swap(l.X);  // U <- U+ (lattice.py:206); an in-place solve left it in U already
    if (l.comm && l.world > 1 && l.shard_mode == 0) {
      // the swapped-in buffer only holds this rank's columns; the others are refreshed lazily by osc_get_U.
      l.u_sharded = true;
    }
    if (ms) *ms = now_ms() - t0;
    if (iters) *iters = r.iters;
    if (res) *res = r.res;
  });
}

int osc_solve_ustar(osc_handle h, float tol, int32_t max_iters, float* Ustar_out, int32_t* iters, float* res,
                    double* ms) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (max_iters < 1) throw Invalid("max_iters must be >= 1");
    ensure_cg_scratch(l, max_iters);
    const OpParams op = ustar_op(l);
    sync(l);
    const double t0 = now_ms();
    CgBuffers b{l.Y.p, l.X.p, l.R.p, l.P.p, l.AP.p, l.U.p, l.Y.p, l.B.p, l.psi.p, l.ld, l.c0, l.c1};
    b.kind = 1;
    const CgResult r = run_cg(l, op, b, path_active(l), max_iters, tol);
    l.Ustar.swap(l.X);
    if (l.shard_mode == 0) gather_columns(l, l.Ustar.p);  // receipts read whole rows of U* (row mode: already whole)
    l.have_ustar = true;
    if (ms) *ms = now_ms() - t0;
    if (iters) *iters = r.iters;
    if (res) *res = r.res;
    if (Ustar_out) download_api_order(l, Ustar_out, l.Ustar.p);
  });
}

int osc_has_ustar(osc_handle h, int32_t* yes) {
  return guarded(h, [&](L& l) {
    if (yes) *yes = l.have_ustar ? 1 : 0;
  });
}

int osc_get_ustar(osc_handle h, float* out) {
  return guarded(h, [&](L& l) {
    if (!out) throw Invalid("osc_get_ustar: out is NULL");
    if (!l.have_ustar) throw StateError("osc_get_ustar: no resident U* (call osc_solve_ustar first)");
    download_api_order(l, out, l.Ustar.p);
  });
}

int osc_get_rows(osc_handle h, int32_t which, const int32_t* rows, int32_t n, float* out) {
  return guarded(h, [&](L& l) {
    if (n < 0 || (n > 0 && (!rows || !out))) throw Invalid("osc_get_rows: bad arguments");
    const float* src = which == 0 ? l.Y.p : which == 1 ? l.U.p : which == 2 ? l.Ustar.p : nullptr;
    if (!src) throw Invalid("osc_get_rows: which must be 0 (Y), 1 (U) or 2 (U*)");
    if (which == 2 && !l.have_ustar) throw StateError("osc_get_rows: no resident U* (call osc_solve_ustar first)");
    if (which == 1 && l.u_sharded) {  // collective in column-sharded runs, like osc_get_U
      gather_columns(l, l.U.p);
      l.u_sharded = false;
    }
    if (n == 0) return;
    std::vector<int32_t> dev_rows((size_t)n);
    for (int32_t i = 0; i < n; ++i) {
      if (rows[i] < 0 || rows[i] >= l.N) throw Invalid("osc_get_rows: row out of range");
      dev_rows[(size_t)i] = permuted(l) ? l.inv_h[(size_t)rows[i]] : rows[i];
    }
    DevBuf<int32_t> idx;
    DevBuf<float> tmp;
    idx.alloc((size_t)n);
    tmp.alloc((size_t)n * l.ld);
    HIP_CHECK(hipMemcpyAsync(idx.p, dev_rows.data(), (size_t)n * 4, hipMemcpyHostToDevice, l.stream));
    launch_move_rows(tmp.p, src, idx.p, n, l.ld, false, l.stream);  // tmp[i] = src[rows[i]]
    HIP_CHECK(hipMemcpy2DAsync(out, (size_t)l.D * 4, tmp.p, (size_t)l.ld * 4, (size_t)l.D * 4, (size_t)n,
                               hipMemcpyDeviceToHost, l.stream));
    sync(l);
  });
}

int osc_residual_history(osc_handle h, float* out, int32_t cap, int32_t* n) {
  return guarded(h, [&](L& l) {
    const int32_t m = std::min<int32_t>(cap, (int32_t)l.history.size());
    if (out)
      for (int32_t i = 0; i < m; ++i) out[i] = l.history[(size_t)i];
    if (n) *n = m;
  });
}

int osc_cg_single_rhs(osc_handle h, float gamma, const float* s, float tol, int32_t max_iters, float* h_out,
                      int32_t* iters, float* res) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (!(gamma > 0)) throw Invalid("gamma must be > 0 for SPD");
    if (max_iters < 1) throw Invalid("max_iters must be >= 1");
    if (!s || !h_out) throw Invalid("osc_cg_single_rhs: NULL buffer");
    ensure_cg_scratch(l, max_iters);
    // N x 1 problem stored with pitch 4 (columns 1..3 stay zero)
    const int32_t ld1 = 4;
    const size_t n = (size_t)l.N * ld1;
    DevBuf<float> S, X0, X, R, P, AP, psi0;
    for (DevBuf<float>* b : {&S, &X0, &X, &R, &P, &AP}) {
      b->alloc(n);
      HIP_CHECK(hipMemsetAsync(b->p, 0, n * 4, l.stream));
    }
    psi0.alloc(ld1);
    HIP_CHECK(hipMemsetAsync(psi0.p, 0, ld1 * 4, l.stream));
    std::vector<float> sp;
    if (permuted(l)) {  // API order -> device order
      sp.resize((size_t)l.N);
      for (int64_t i = 0; i < l.N; ++i) sp[(size_t)i] = s[l.perm_h[(size_t)i]];
      s = sp.data();
    }
    // host <-> device contiguous through a scratch vector, re-pitched on the device: a strided copy of N 4-byte rows
    // from/to pageable memory costs more than the whole solve
    DevBuf<float> flat;
    flat.alloc((size_t)l.N);
    HIP_CHECK(hipMemcpyAsync(flat.p, s, (size_t)l.N * 4, hipMemcpyHostToDevice, l.stream));
    HIP_CHECK(hipMemcpy2DAsync(S.p, ld1 * 4, flat.p, 4, 4, (size_t)l.N, hipMemcpyDeviceToDevice, l.stream));
    OpParams op{};
    op.cs_const = 1.0f + gamma;  // (L_sym + gamma I) x = (1 + gamma) x - W x
    op.cs_B = 0.f;
    op.cW = 1.0f;
    op.cP = 0.f;
    op.md_const = 1.0f + gamma;  // diag(L_sym) + gamma (diffusion.py:138-139), diag(L_sym) = 1
    op.md_B = 0.f;
    op.precond = 1;
    op.rbU = 0.f;
    op.rbY = 1.0f;
    op.rbB = 0.f;
    CgBuffers b{X0.p, X.p, R.p, P.p, AP.p, S.p, S.p, l.B.p, psi0.p, ld1, 0, ld1};
    b.kind = 2;
    // scratch sized for ld >= 4 already
    std::unique_ptr<Comm> saved = std::move(l.comm);  // the diffusion solve is replicated, not sharded
    CgResult r;
    try {
      r = run_cg(l, op, b, false, max_iters, tol);
    } catch (...) {
      l.comm = std::move(saved);
      throw;
    }
    l.comm = std::move(saved);
    HIP_CHECK(hipMemcpy2DAsync(flat.p, 4, X.p, ld1 * 4, 4, (size_t)l.N, hipMemcpyDeviceToDevice, l.stream));
    HIP_CHECK(hipMemcpyAsync(h_out, flat.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    sync(l);
    to_api_order(l, h_out);
    if (iters) *iters = r.iters;
    if (res) *res = r.res;
  });
}

// cosine of every row of a resident N x ld array to one host query (rows and query normalised with the +1e-12 of the
// reference); result in API row order
static void rows_cosine_to(L& l, const float* rows, const float* psi, float* out) {
  l.vec_q.alloc((size_t)l.D);
  l.vec_n.alloc((size_t)l.N);
  std::vector<float> qn((size_t)l.D);
  float ss = 0.f;
  for (int c = 0; c < l.D; ++c) ss += psi[c] * psi[c];
  const float inv = 1.0f / (std::sqrt(ss) + 1e-12f);
  for (int c = 0; c < l.D; ++c) qn[(size_t)c] = psi[c] * inv;
  HIP_CHECK(hipMemcpyAsync(l.vec_q.p, qn.data(), (size_t)l.D * 4, hipMemcpyHostToDevice, l.stream));
  launch_rows_cosine(rows, l.ld, l.vec_q.p, l.vec_n.p, l.N, l.D, l.stream);
  HIP_CHECK(hipMemcpyAsync(out, l.vec_n.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
  sync(l);
  to_api_order(l, out);
}

int osc_cosine_to(osc_handle h, const float* psi, float* out) {
  return guarded(h, [&](L& l) {
    if (!psi || !out) throw Invalid("osc_cosine_to: NULL buffer");
    rows_cosine_to(l, l.Y.p, psi, out);
  });
}

int osc_cosine_to_row(osc_handle h, int64_t row, float* out) {
  return guarded(h, [&](L& l) {
    if (!out) throw Invalid("osc_cosine_to_row: out is NULL");
    if (row < 0 || row >= l.N) throw Invalid("osc_cosine_to_row: row out of range");
    const int64_t dev_row = permuted(l) ? l.inv_h[(size_t)row] : row;
    std::vector<float> q((size_t)l.D);
    HIP_CHECK(hipMemcpyAsync(q.data(), l.Y.p + (size_t)dev_row * l.ld, (size_t)l.D * 4, hipMemcpyDeviceToHost, l.stream));
    sync(l);
    rows_cosine_to(l, l.Y.p, q.data(), out);
  });
}

int osc_mmr(osc_handle h, const float* scores, int32_t k, float lambda_div, int32_t* out_idx, int32_t* out_count) {
  return guarded(h, [&](L& l) {
    if (!scores || !out_idx || !out_count) throw Invalid("osc_mmr: NULL buffer");
    *out_count = 0;
    const int32_t want = (int32_t)std::min<int64_t>(std::max(k, 0), l.N);
    if (want <= 0) return;
    const int32_t N = (int32_t)l.N;
    // (1 - lambda) * score in fp64 and in device row order, like the NumPy float64 arithmetic this replaces
    std::vector<double> base((size_t)N);
    for (int32_t i = 0; i < N; ++i)
      base[(size_t)i] = (1.0 - (double)lambda_div) * (double)scores[permuted(l) ? l.perm_h[(size_t)i] : i];
    DevBuf<double> d_base, d_maxsim, d_pval;
    DevBuf<unsigned char> d_alive;
    DevBuf<int32_t> d_pid, d_prow, d_chosen;
    const int nblocks = (int)std::max<int64_t>(1, std::min<int64_t>((N + 255) / 256, 256));
    d_base.alloc((size_t)N);
    d_maxsim.alloc((size_t)N);
    d_alive.alloc((size_t)N);
    d_pval.alloc((size_t)nblocks);
    d_pid.alloc((size_t)nblocks);
    d_prow.alloc((size_t)nblocks);
    d_chosen.alloc((size_t)want);
    l.vec_q.alloc((size_t)l.D);
    HIP_CHECK(hipMemcpyAsync(d_base.p, base.data(), (size_t)N * 8, hipMemcpyHostToDevice, l.stream));
    HIP_CHECK(hipMemsetAsync(d_alive.p, 1, (size_t)N, l.stream));
    MmrArgs a{};
    a.Y = l.Y.p;
    a.base = d_base.p;
    a.maxsim = d_maxsim.p;
    a.alive = d_alive.p;
    a.api_id = permuted(l) ? l.perm_d.p : nullptr;
    a.q = l.vec_q.p;
    a.pval = d_pval.p;
    a.pid = d_pid.p;
    a.prow = d_prow.p;
    a.chosen_api = d_chosen.p;
    a.N = N;
    a.D = l.D;
    a.ld = l.ld;
    a.nblocks = nblocks;
    a.lambda = (double)lambda_div;
    for (int step = 0; step < want; ++step) launch_mmr_step(a, step, l.stream);
    std::vector<int32_t> chosen((size_t)want);
    HIP_CHECK(hipMemcpyAsync(chosen.data(), d_chosen.p, (size_t)want * 4, hipMemcpyDeviceToHost, l.stream));
    sync(l);
    int32_t n = 0;
    for (; n < want && chosen[(size_t)n] >= 0; ++n) out_idx[n] = chosen[(size_t)n];
    *out_count = n;
  });
}

int osc_ustar_cosine_to(osc_handle h, const float* psi, float* out) {
  return guarded(h, [&](L& l) {
    if (!psi || !out) throw Invalid("osc_ustar_cosine_to: NULL buffer");
    if (!l.have_ustar) throw StateError("osc_ustar_cosine_to: no resident U* (call osc_solve_ustar first)");
    rows_cosine_to(l, l.Ustar.p, psi, out);
  });
}

// sum (A - B) . M (A - B) with the stationary operator M = lamG I + lamC L + lamQ B (+ lamP L_path)  (receipts.py:21-25)
// over this rank's share (column window / row block), completed over the communicator
static double quad_form_of_difference(L& l, const float* A, const float* B) {
  ensure_cg_scratch(l, 1);
  launch_axpby(l.P.p, A, 1.0f, B, -1.0f, (int64_t)l.N * l.ld, l.stream);
  const int grid = cg_grid(l);
  SpmmArgs sa{};
  sa.g = graph_view(l, path_active(l));
  sa.op = ustar_op(l);
  sa.X = l.P.p;
  sa.B = l.B.p;
  sa.psi = l.psi.p;
  sa.ld = l.ld;
  sa.c0 = l.c0;
  sa.c1 = l.c1;
  sa.gate = nullptr;
  int nb = grid;
  if (row_mode(l)) {  // each rank (or fake shard) sums its own rows; the scalar is all-reduced below
    const std::vector<RowShard> shards = row_shards(l);
    nb = (int)shards.size() * grid;
    if (l.part0.n < (size_t)nb * l.ld) l.part0.alloc((size_t)nb * l.ld);
    for (size_t si = 0; si < shards.size(); ++si) {
      sa.row0 = shards[si].r0;
      sa.N = shards[si].r1;
      sa.part = l.part0.p + si * grid * l.ld;
      spmm_slabbed(l, SPMM_DOT, sa, grid);
    }
  } else {
    sa.part = l.part0.p;
    sa.N = l.N;
    spmm_slabbed(l, SPMM_DOT, sa, grid);
  }
  launch_reduce_sum(l.part0.p, nb, l.ld, l.c0, l.c1, l.colsum.p, l.stream);
  std::vector<double> cs((size_t)l.ld, 0.0);
  HIP_CHECK(hipMemcpyAsync(cs.data() + l.c0, l.colsum.p + l.c0, (size_t)(l.c1 - l.c0) * 8, hipMemcpyDeviceToHost,
                           l.stream));
  sync(l);
  double tot = 0.0;
  for (int c = l.c0; c < l.c1; ++c) tot += cs[(size_t)c];
  if (l.comm) {
    DevBuf<double> t;
    t.alloc(1);
    HIP_CHECK(hipMemcpyAsync(t.p, &tot, 8, hipMemcpyHostToDevice, l.stream));
    l.comm->allreduce(t.p, 1, COMM_F64, COMM_SUM, l.stream);
    HIP_CHECK(hipMemcpyAsync(&tot, t.p, 8, hipMemcpyDeviceToHost, l.stream));
    sync(l);
  }
  return tot;
}

int osc_deltaH(osc_handle h, double* dH) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (!l.have_ustar) throw StateError("osc_deltaH: no resident U* (call osc_solve_ustar first)");
    if (!dH) throw Invalid("osc_deltaH: dH is NULL");
    *dH = quad_form_of_difference(l, l.U.p, l.Ustar.p);  // diff = U - U*
  });
}

// N x D host array in API row order -> N x ld device array in device row order (AP is scratch between solves)
static void upload_api_order(L& l, float* dst, const float* src) {
  if (permuted(l)) {
    upload_rows(l, l.AP.p, src);
    launch_move_rows(dst, l.AP.p, l.perm_d.p, l.N, l.ld, false, l.stream);
  } else {
    upload_rows(l, dst, src);
  }
}

int osc_dynamics_snapshot(osc_handle h) {
  return guarded(h, [&](L& l) {
    if (l.u_sharded) {  // collective in column-sharded runs, like osc_get_U
      gather_columns(l, l.U.p);
      l.u_sharded = false;
    }
    l.Uprev.alloc((size_t)l.N * l.ld);
    HIP_CHECK(hipMemcpyAsync(l.Uprev.p, l.U.p, (size_t)l.N * l.ld * 4, hipMemcpyDeviceToDevice, l.stream));
    l.have_uprev = true;
  });
}

int osc_dynamics(osc_handle h, const float* U_prev, const float* U_next, double* move2_mean, float* move2_max,
                 double* step_deltaH, double* flow_total, int32_t top_cap, int32_t* top_i, int32_t* top_j,
                 double* top_flow, int32_t* top_n, int32_t* radius) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (top_cap < 0 || top_cap > 32 || (top_cap > 0 && (!top_i || !top_j || !top_flow || !top_n)))
      throw Invalid("osc_dynamics: top_cap must be 0..32 with output arrays");
    const size_t n = (size_t)l.N * l.ld;
    if (U_prev) {
      l.Uprev.alloc(n);
      HIP_CHECK(hipMemsetAsync(l.Uprev.p, 0, n * 4, l.stream));
      upload_api_order(l, l.Uprev.p, U_prev);
      l.have_uprev = true;
    }
    if (!l.have_uprev) throw StateError("osc_dynamics: no previous state (call osc_dynamics_snapshot before the settle)");
    if (l.u_sharded) {
      gather_columns(l, l.U.p);
      l.u_sharded = false;
    }
    const float* next = l.U.p;
    DevBuf<float> next_buf;
    if (U_next) {
      next_buf.alloc(n);
      HIP_CHECK(hipMemsetAsync(next_buf.p, 0, n * 4, l.stream));
      upload_api_order(l, next_buf.p, U_next);
      next = next_buf.p;
    }
    // step energy (lattice.py:843-857): the quadratic form of the stationary operator on U_prev - U_next
    const double dH = quad_form_of_difference(l, l.Uprev.p, next);
    if (step_deltaH) *step_deltaH = dH;
    // movement per node and structural energy drop per edge: the receipt-rows pass with (Y, U*) := (U_prev, U_next)
    DevBuf<float> move2, flows;
    move2.alloc((size_t)l.N);
    const size_t ne = (size_t)l.N * l.width;
    flows.alloc(ne);
    HIP_CHECK(hipMemsetAsync(flows.p, 0, ne * 4, l.stream));
    ReceiptArgs a{};
    a.Y = l.Uprev.p;
    a.Ustar = next;
    a.psi = l.psi.p;
    a.B = l.B.p;
    a.sqrt_deg = l.sqrt_deg.p;
    a.col = l.ell_col.p;
    a.adj = l.ell_a.p;
    a.deg = l.deg.p;
    a.width = l.width;
    a.lamG = 1.0f;  // anchor term of the pass = ||U_next - U_prev||^2 per node
    a.lamC = l.lamC;
    a.lamQ = 0.0f;
    a.z_th = 0.0f;
    a.anchor = move2.p;
    a.N = (int32_t)l.N;
    a.D = l.D;
    a.ld = l.ld;
    a.api_id = nullptr;
    a.edge_flow = flows.p;
    launch_receipt_rows(a, l.stream);
    // reductions: sum / max of the movement, sum of the flows
    const int nb1 = (int)std::max<int64_t>(1, std::min<int64_t>((l.N + 255) / 256, 256));
    const int nb2 = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)ne + 255) / 256, 256));
    DevBuf<double> ps;
    DevBuf<float> pm;
    ps.alloc((size_t)nb1 + nb2);
    pm.alloc((size_t)nb1 + nb2);
    launch_sum_max(move2.p, l.N, nb1, ps.p, pm.p, l.stream);
    launch_sum_max(flows.p, (int64_t)ne, nb2, ps.p + nb1, pm.p + nb1, l.stream);
    // top flows: per-block candidates (twice the cap, so both directions of a tied pair survive a block's cut)
    const int K = top_cap > 0 ? 32 : 0;
    const int nb3 = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)ne + 4095) / 4096, 256));
    DevBuf<float> tv;
    DevBuf<int64_t> ti;
    DevBuf<int32_t> tc;
    std::vector<float> htv;
    std::vector<int64_t> hti;
    std::vector<int32_t> htc;
    if (K > 0) {
      tv.alloc((size_t)nb3 * K);
      ti.alloc((size_t)nb3 * K);
      tc.alloc((size_t)nb3 * K);
      launch_top_select(flows.p, l.ell_col.p, permuted(l) ? l.perm_d.p : nullptr, l.width, (int64_t)ne, nb3, K, tv.p, ti.p, tc.p,
                        l.stream);
      htv.resize((size_t)nb3 * K);
      hti.resize((size_t)nb3 * K);
      htc.resize((size_t)nb3 * K);
      HIP_CHECK(hipMemcpyAsync(htv.data(), tv.p, htv.size() * 4, hipMemcpyDeviceToHost, l.stream));
      HIP_CHECK(hipMemcpyAsync(hti.data(), ti.p, hti.size() * 8, hipMemcpyDeviceToHost, l.stream));
      HIP_CHECK(hipMemcpyAsync(htc.data(), tc.p, htc.size() * 4, hipMemcpyDeviceToHost, l.stream));
    }
    std::vector<double> hps((size_t)nb1 + nb2);
    std::vector<float> hpm((size_t)nb1 + nb2);
    HIP_CHECK(hipMemcpyAsync(hps.data(), ps.p, hps.size() * 8, hipMemcpyDeviceToHost, l.stream));
    HIP_CHECK(hipMemcpyAsync(hpm.data(), pm.p, hpm.size() * 4, hipMemcpyDeviceToHost, l.stream));
    sync(l);
    double msum = 0.0, fsum = 0.0;
    float mmax = 0.f;
    for (int b = 0; b < nb1; ++b) {
      msum += hps[(size_t)b];
      mmax = (hpm[(size_t)b] != hpm[(size_t)b] || mmax != mmax) ? NAN : std::max(mmax, hpm[(size_t)b]);
    }
    for (int b = 0; b < nb2; ++b) fsum += hps[(size_t)nb1 + b];
    if (move2_mean) *move2_mean = msum / (double)l.N;
    if (move2_max) *move2_max = mmax;
    if (flow_total) *flow_total = fsum;
    if (K > 0) {
      struct Cand {
        double f;
        int32_t i, j;
      };
      std::vector<Cand> cands;
      for (size_t t = 0; t < hti.size(); ++t) {
        if (hti[t] < 0) continue;
        const int32_t row = (int32_t)(hti[t] / l.width);
        cands.push_back({(double)htv[t], permuted(l) ? l.perm_h[(size_t)row] : row,
                         permuted(l) ? l.perm_h[(size_t)htc[t]] : htc[t]});
      }
      // the reference keeps the flows in argwhere order and sorts them stably by flow, descending (lattice.py:879-881)
      std::sort(cands.begin(), cands.end(), [](const Cand& x, const Cand& y) {
        if (x.f != y.f) return x.f > y.f;
        if (x.i != y.i) return x.i < y.i;
        return x.j < y.j;
      });
      const int32_t m = (int32_t)std::min<size_t>((size_t)top_cap, cands.size());
      for (int32_t t = 0; t < m; ++t) {
        top_i[t] = cands[(size_t)t].i;
        top_j[t] = cands[(size_t)t].j;
        top_flow[t] = cands[(size_t)t].f;
      }
      *top_n = m;
    } else if (top_n) {
      *top_n = 0;
    }
    // coherence radius (lattice.py:885-892, 905-927): BFS from the nodes that moved at least 10 % of the largest move
    if (radius) {
      *radius = 0;
      const float maxinf = std::sqrt(mmax + 1e-12f);
      if (l.N > 0 && maxinf > 1e-9f) {
        const float thr = (float)(0.1 * (double)maxinf);
        DevBuf<int32_t> dist, changed;
        dist.alloc((size_t)l.N);
        changed.alloc(1);
        launch_bfs_seeds(move2.p, l.N, thr, dist.p, l.stream);
        int32_t level = 0;
        for (; level < l.N; ++level) {
          int32_t ch = 0;
          HIP_CHECK(hipMemsetAsync(changed.p, 0, 4, l.stream));
          launch_bfs_level(l.ell_col.p, l.deg.p, l.width, l.N, dist.p, level, changed.p, l.stream);
          HIP_CHECK(hipMemcpyAsync(&ch, changed.p, 4, hipMemcpyDeviceToHost, l.stream));
          sync(l);
          if (!ch) break;
        }
        *radius = level;
      }
    }
  });
}

// null-point candidates per device row -> compact list in API row order (the reference walks rows 0..N-1)
static int32_t compact_nulls(const L& l, const std::vector<int32_t>& hj, const std::vector<float>& hz,
                             const std::vector<float>& hr, int32_t* i_out, int32_t* j_out, float* z_out, float* r_out) {
  int32_t n = 0;
  for (int64_t r = 0; r < l.N; ++r) {  // r = API row
    const size_t i = (size_t)(l.perm_h.empty() ? r : l.inv_h[(size_t)r]);
    if (hj[i] < 0) continue;
    if (i_out) i_out[n] = (int32_t)r;
    if (j_out) j_out[n] = l.perm_h.empty() ? hj[i] : l.perm_h[(size_t)hj[i]];
    if (z_out) z_out[n] = hz[i];
    if (r_out) r_out[n] = hr[i];
    ++n;
  }
  return n;
}

static void receipt_rows(L& l, float z_th, DevBuf<float>& coh, DevBuf<float>& an, DevBuf<float>& qu,
                         DevBuf<int32_t>& nj, DevBuf<float>& nz, DevBuf<float>& nr, bool want_comp, bool want_null) {
  ReceiptArgs a{};
  a.Y = l.Y.p;
  a.Ustar = l.Ustar.p;
  a.psi = l.psi.p;
  a.B = l.B.p;
  a.sqrt_deg = l.sqrt_deg.p;
  a.col = l.ell_col.p;
  a.adj = l.ell_a.p;
  a.deg = l.deg.p;
  a.width = l.width;
  a.lamG = l.lamG;
  a.lamC = l.lamC;
  a.lamQ = l.lamQ;
  a.z_th = z_th;
  a.N = (int32_t)l.N;
  a.D = l.D;
  a.ld = l.ld;
  a.api_id = permuted(l) ? l.perm_d.p : nullptr;
  if (want_comp) {
    coh.alloc((size_t)l.N);
    an.alloc((size_t)l.N);
    qu.alloc((size_t)l.N);
    a.coh = coh.p;
    a.anchor = an.p;
    a.query = qu.p;
  }
  if (want_null) {
    nj.alloc((size_t)l.N);
    nz.alloc((size_t)l.N);
    nr.alloc((size_t)l.N);
    a.null_j = nj.p;
    a.null_z = nz.p;
    a.null_r = nr.p;
  }
  launch_receipt_rows(a, l.stream);
}

int osc_receipt_components(osc_handle h, float* coh, float* anchor, float* query) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (!l.have_ustar) throw StateError("osc_receipt_components: no resident U*");
    DevBuf<float> c, a, q, nz, nr;
    DevBuf<int32_t> nj;
    receipt_rows(l, 3.0f, c, a, q, nj, nz, nr, true, false);
    if (coh) HIP_CHECK(hipMemcpyAsync(coh, c.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    if (anchor) HIP_CHECK(hipMemcpyAsync(anchor, a.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    if (query) HIP_CHECK(hipMemcpyAsync(query, q.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    sync(l);
    to_api_order(l, coh);
    to_api_order(l, anchor);
    to_api_order(l, query);
  });
}

int osc_null_points(osc_handle h, float z_th, int32_t* i_out, int32_t* j_out, float* z_out, float* r_out,
                    int32_t* count) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (!l.have_ustar) throw StateError("osc_null_points: no resident U*");
    if (!count) throw Invalid("osc_null_points: count is NULL");
    DevBuf<float> c, a, q, nz, nr;
    DevBuf<int32_t> nj;
    receipt_rows(l, z_th, c, a, q, nj, nz, nr, false, true);
    std::vector<int32_t> hj((size_t)l.N);
    std::vector<float> hz((size_t)l.N), hr((size_t)l.N);
    HIP_CHECK(hipMemcpyAsync(hj.data(), nj.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    HIP_CHECK(hipMemcpyAsync(hz.data(), nz.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    HIP_CHECK(hipMemcpyAsync(hr.data(), nr.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    sync(l);
    *count = compact_nulls(l, hj, hz, hr, i_out, j_out, z_out, r_out);
  });
}

int osc_receipt_rows(osc_handle h, float z_th, float* coh, float* anchor, float* query, int32_t* i_out, int32_t* j_out,
                     float* z_out, float* r_out, int32_t* count) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (!l.have_ustar) throw StateError("osc_receipt_rows: no resident U*");
    if (!count) throw Invalid("osc_receipt_rows: count is NULL");
    DevBuf<float> c, a, q, nz, nr;
    DevBuf<int32_t> nj;
    receipt_rows(l, z_th, c, a, q, nj, nz, nr, true, true);
    std::vector<int32_t> hj((size_t)l.N);
    std::vector<float> hz((size_t)l.N), hr((size_t)l.N);
    if (coh) HIP_CHECK(hipMemcpyAsync(coh, c.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    if (anchor) HIP_CHECK(hipMemcpyAsync(anchor, a.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    if (query) HIP_CHECK(hipMemcpyAsync(query, q.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    HIP_CHECK(hipMemcpyAsync(hj.data(), nj.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    HIP_CHECK(hipMemcpyAsync(hz.data(), nz.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    HIP_CHECK(hipMemcpyAsync(hr.data(), nr.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    sync(l);
    to_api_order(l, coh);
    to_api_order(l, anchor);
    to_api_order(l, query);
    *count = compact_nulls(l, hj, hz, hr, i_out, j_out, z_out, r_out);
  });
}

int osc_profile_enable(osc_handle h, int32_t on) {
  return guarded(h, [&](L& l) {
    if (!on) prof_drain(l);
    l.prof_on = on != 0;
  });
}
int osc_profile_reset(osc_handle h) {
  return guarded(h, [&](L& l) {
    prof_drain(l);
    for (int i = 0; i < 5; ++i) {
      l.prof_count[i] = 0;
      l.prof_ms[i] = 0.0;
    }
    if (l.blk_stamps.n) {
      HIP_CHECK(hipMemsetAsync(l.blk_stamps.p, 0, l.blk_stamps.n * 8, l.stream));
      l.blk_stamp_launches = 0;
    }
  });
}
int osc_profile_get(osc_handle h, int32_t which, int64_t* launches, double* total_ms) {
  return guarded(h, [&](L& l) {
    if (which == 14) {  // the kernel shape of the last blocked matvec
      if (launches) *launches = l.blk_shape_last;
      if (total_ms) *total_ms = 0.0;
      return;
    }
    if (which >= 8 && which <= 13) {
      // cycle stamps of the blocked matvec (OSC_BLK_STAMP=1): mean over the waves of a role of the shader cycles summed
      // over the stamped launches.  8-11: gathering waves' lifetime / gather rounds / barrier / epilogue; 12-13: the list
      // waves' fetch / barrier.  *launches = stamped launches (speculative, gated-off ones included: they add ~nothing).
      prof_drain(l);
      sync(l);
      const int wpg = blocked_gather_waves(l.blk_shape_last) + 1;
      std::vector<unsigned long long> w(l.blk_stamps.n);
      if (!w.empty()) HIP_CHECK(hipMemcpy(w.data(), l.blk_stamps.p, w.size() * 8, hipMemcpyDeviceToHost));
      double sum = 0.0;
      int64_t cnt = 0;
      for (size_t i = 0; i + 3 < w.size(); i += 4) {
        const bool list_wave = (int)((i / 4) % (size_t)wpg) == wpg - 1;
        if (w[i] == 0 || list_wave != (which >= 12)) continue;  // (workgroups that took no part have no stamps)
        sum += (double)w[i + (which >= 12 ? which - 11 : which - 8)];
        cnt += 1;
      }
      if (launches) *launches = l.blk_stamp_launches;
      if (total_ms) *total_ms = cnt ? sum / (double)cnt : 0.0;
      return;
    }
    if (which < 0 || which > 4) throw Invalid("osc_profile_get: which must be 0..4 (or 8..14: blocked matvec diagnostics)");
    prof_drain(l);
    if (launches) *launches = l.prof_count[which];
    if (total_ms) *total_ms = l.prof_ms[which];
  });
}

int osc_comm_unique_id(char id_out[128]) {
  try {
    comm_rccl_id(id_out);
  } catch (const std::exception&) {
    return OSC_E_COMM;
  }
  return OSC_OK;
}

int osc_comm_backend_version(int32_t* version) {
  if (!version) return OSC_E_INVALID;
  *version = comm_rccl_version();
  return OSC_OK;
}

int osc_comm_loopback_id(char id_out[128]) {
  comm_loopback_id(id_out);
  return OSC_OK;
}

int osc_comm_init(osc_handle h, const char id[128], int32_t rank, int32_t world) {
  return guarded(h, [&](L& l) {
    if (world < 1 || rank < 0 || rank >= world) throw Invalid("osc_comm_init: bad rank/world");
    drain_comm_stream(l);
    l.comm.reset();
    l.rank = rank;
    l.world = world;
    l.fake_window = false;
    std::tie(l.c0, l.c1) = host::column_shard(l.dcols, rank, world);  // slabs in units of 4 floats, as even as possible
    if (l.shard_mode == 1) {  // row-sharded CG: every rank works on all columns of its row block
      l.c0 = 0;
      l.c1 = l.dcols;
    }
    if (world == 1 && l.shard_mode == 0 && l.fake_col_w > 0) {  // measurement hook (OSC_FAKE_COL_SHARD, read at osc_create): one
      std::tie(l.c0, l.c1) = host::column_shard(l.dcols, l.fake_col_r, l.fake_col_w);  // rank's window of a wider solve, now WITH
      l.fake_window = l.fake_col_w > 1;  // the communicator machinery around it; osc_comm_info says so
    }
    if (l.c1 <= l.c0) throw Invalid("osc_comm_init: more ranks than 4-column groups");
    l.comm = comm_create(id, rank, world, l.device);
    ++l.graph_epoch;
    l.have_ustar = false;
  });
}

int osc_comm_allreduce_f64(osc_handle h, double* vals, int32_t n, int32_t op) {
  return guarded(h, [&](L& l) {
    if (n < 0 || (n > 0 && !vals) || (op != 0 && op != 1)) throw Invalid("osc_comm_allreduce_f64: bad arguments");
    sync(l);  // also the barrier use: everything this rank enqueued so far has finished
    if (!l.comm || n == 0) {
      if (l.comm) {  // n == 0: pure barrier
        DevBuf<double> t;
        t.alloc(1);
        HIP_CHECK(hipMemsetAsync(t.p, 0, 8, l.stream));
        l.comm->allreduce(t.p, 1, COMM_F64, COMM_SUM, l.stream);
        sync(l);
      }
      return;
    }
    DevBuf<double> t;
    t.alloc((size_t)n);
    HIP_CHECK(hipMemcpyAsync(t.p, vals, (size_t)n * 8, hipMemcpyHostToDevice, l.stream));
    l.comm->allreduce(t.p, (size_t)n, COMM_F64, op == 0 ? COMM_SUM : COMM_MAX, l.stream);
    HIP_CHECK(hipMemcpyAsync(vals, t.p, (size_t)n * 8, hipMemcpyDeviceToHost, l.stream));
    sync(l);
  });
}

int osc_comm_info(osc_handle h, int32_t* rank, int32_t* world, int32_t* shard_mode, char* kind_out, int32_t cap) {
  return guarded(h, [&](L& l) {
    if (rank) *rank = l.comm ? l.rank : 0;
    if (world) *world = l.comm ? l.world : 1;
    if (shard_mode) *shard_mode = l.shard_mode;
    if (kind_out && cap > 0)
      snprintf(kind_out, (size_t)cap, "%s%s", l.comm ? l.comm->kind() : "none", l.comm && l.fake_window ? "+fake-window" : "");
  });
}

int osc_halo_info(osc_handle h, int64_t* need_rows, int64_t* need_rows_max, int64_t* remote_rows,
                  int64_t* bytes_per_iteration, int32_t* full_exchange) {
  return guarded(h, [&](L& l) {
    if (!l.comm || l.shard_mode != 1) throw StateError("osc_halo_info: no row-sharded communicator on this handle");
    require_graph(l);
    if (l.halo.epoch != l.graph_epoch) build_halo_plan(l);
    const int64_t own = l.N * (l.rank + 1) / l.world - l.N * l.rank / l.world;
    if (need_rows) *need_rows = l.halo.need_rows;
    if (need_rows_max) *need_rows_max = l.halo.need_rows_max;
    if (remote_rows) *remote_rows = l.N - own;
    if (bytes_per_iteration) *bytes_per_iteration = (l.halo.full ? (l.N - own) : l.halo.need_rows) * (int64_t)l.ld * 4;
    if (full_exchange) *full_exchange = l.halo.full ? 1 : 0;
  });
}

int osc_comm_shard(osc_handle h, int32_t* c0, int32_t* c1) {
  return guarded(h, [&](L& l) {
    if (c0) *c0 = l.c0;
    if (c1) *c1 = std::min(l.c1, l.D);
  });
}

}  // extern "C"
