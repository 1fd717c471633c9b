// Process-wide pools, transfers, profiling events and per-handle scratch of liboscillink_hip.so (see osc_internal.hpp).
#include "osc_internal.hpp"

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
std::map<int, std::vector<CtrlBlock>> g_ctrl_pool;

// ---- caching device allocator (common.hpp) ----
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
void prof_drain(L& h, bool nothrow) {
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

