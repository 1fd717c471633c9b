// Communicator backends (comm.hpp): RCCL, and the in-process loopback used to run the multi-rank paths on one GPU.
#include "comm.hpp"

#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

namespace osc {
namespace {

// ---- RCCL ---------------------------------------------------------------------------------------------------------
void nccl_ok(ncclResult_t r, const char* what) {
  if (r != ncclSuccess) throw CommError(std::string(what) + " failed: " + ncclGetErrorString(r));
}

// Measurement stand-in (OSC_RCCL_PROXY=1, one-rank communicators only): at world 1 ncclAllReduce launches nothing, so the
// one GPU a round gets cannot show what the stop test's all-reduce KERNEL costs beside a resident operator apply.  This
// kernel has the launch shape of RCCL 2.27's ncclDevKernel_Generic on gfx950 as its code object declares it (one
// workgroup of 256 threads, 248 VGPRs, 37 664 bytes of LDS, 696 bytes of scratch per lane: `llvm-readelf --notes` of the
// gfx950 bundle of librccl.so.1.0.70200) and touches one word; the communicator then reports kind "rccl+proxy".
__global__ void __launch_bounds__(256) k_rccl_shape_proxy(uint32_t* __restrict__ word) {
  __shared__ uint32_t lds[37664 / 4];
  asm volatile("v_mov_b32 v247, 0" ::: "v247");  // raises the kernel's VGPR count to RCCL's 248
  lds[threadIdx.x] = threadIdx.x == 0 ? word[0] : 0u;
  __syncthreads();
  if (threadIdx.x == 0) word[0] = lds[0];
}

class RcclComm final : public Comm {
 public:
  RcclComm(const char id[128], int rank, int world) {
    rank_ = rank;
    world_ = world;
    if (const char* e = getenv("OSC_RCCL_PROXY")) proxy_ = world == 1 && atoi(e) != 0;
    ncclUniqueId uid;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(&uid, id, 128);
    nccl_ok(ncclCommInitRank(&comm_, world, uid, rank), "ncclCommInitRank");
  }
  ~RcclComm() override {
    if (comm_) (void)ncclCommDestroy(comm_);
  }
  const char* kind() const override { return proxy_ ? "rccl+proxy" : "rccl"; }
  void allreduce(void* buf, size_t n, CommDType t, CommOp op, hipStream_t s) override {
    if (proxy_ && n > 0) {
      hipLaunchKernelGGL(k_rccl_shape_proxy, dim3(1), dim3(256), 0, s, static_cast<uint32_t*>(buf));
      HIP_CHECK(hipGetLastError());
    }
    const ncclDataType_t dt = t == COMM_F32 ? ncclFloat : t == COMM_F64 ? ncclDouble : ncclInt32;
    nccl_ok(ncclAllReduce(buf, buf, n, dt, op == COMM_SUM ? ncclSum : ncclMax, comm_, s), "ncclAllReduce");
  }
  void allgather(void* buf, size_t chunk_bytes, hipStream_t s) override {
    nccl_ok(ncclAllGather(static_cast<char*>(buf) + (size_t)rank_ * chunk_bytes, buf, chunk_bytes, ncclChar, comm_, s),
            "ncclAllGather");
  }
  void broadcast_group(const std::vector<CommXfer>& pieces, hipStream_t s) override {
    nccl_ok(ncclGroupStart(), "ncclGroupStart");
    for (const CommXfer& p : pieces)
      if (p.bytes) nccl_ok(ncclBroadcast(p.ptr, p.ptr, p.bytes, ncclChar, p.peer, comm_, s), "ncclBroadcast");
    nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
  }
  void exchange(const std::vector<CommXfer>& sends, const std::vector<CommXfer>& recvs, hipStream_t s) override {
    nccl_ok(ncclGroupStart(), "ncclGroupStart");
    for (const CommXfer& p : sends)
      if (p.bytes) nccl_ok(ncclSend(p.ptr, p.bytes, ncclChar, p.peer, comm_, s), "ncclSend");
    for (const CommXfer& p : recvs)
      if (p.bytes) nccl_ok(ncclRecv(p.ptr, p.bytes, ncclChar, p.peer, comm_, s), "ncclRecv");
    nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
  }

 private:
  ncclComm_t comm_ = nullptr;
  bool proxy_ = false;
};

// ---- loopback -----------------------------------------------------------------------------------------------------
constexpr char kLoopMagic[8] = {'O', 'S', 'C', 'L', 'O', 'O', 'P', '1'};

template <typename T, int OP>
__global__ void k_loop_reduce(T* __restrict__ acc, const T* __restrict__ src, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const T a = acc[i], b = src[i];
    if (OP == COMM_SUM) acc[i] = a + b;
    else acc[i] = (a != a || b != b) ? (a != a ? a : b) : (a > b ? a : b);  // NaN wins, like ncclMax on floats here
  }
}

std::mutex g_loop_mu;
std::map<uint64_t, std::weak_ptr<LoopGroup>> g_loop_groups;
std::atomic<uint64_t> g_loop_next{1};

class LoopbackComm final : public Comm {
 public:
  LoopbackComm(const char id[128], int rank, int world) {
    rank_ = rank;
    world_ = world;
    uint64_t key = 0;
    std::memcpy(&key, id + 8, 8);
    std::lock_guard<std::mutex> lk(g_loop_mu);
    auto it = g_loop_groups.find(key);
    if (it != g_loop_groups.end()) g_ = it->second.lock();
    if (!g_) {
      g_ = std::make_shared<LoopGroup>();
      g_->world = world;
      g_->slots.resize((size_t)world);
      if (const char* e = getenv("OSC_LOOPBACK_TIMEOUT_S")) g_->timeout_s = std::max(1.0, atof(e));
      g_loop_groups[key] = g_;
    }
    // (a constructor that throws never runs the destructor: check before counting this rank in)
    if (g_->world != world) throw CommError("loopback: ranks of one group disagree on the world size");
    if (g_->joined >= world) throw CommError("loopback: more ranks joined than the world size");
    ++g_->joined;
  }
  ~LoopbackComm() override {
    std::lock_guard<std::mutex> lk(g_loop_mu);
    --g_->joined;
    for (auto it = g_loop_groups.begin(); it != g_loop_groups.end();)  // groups whose last rank has gone
      it = (it->second.expired() || (it->second.lock() == g_ && g_->joined == 0)) ? g_loop_groups.erase(it) : std::next(it);
  }
  const char* kind() const override { return "loopback"; }

  // Protocol of every collective: publish my pointers, drain my stream (my data is final), barrier, read the peers'
  // memory on my stream, drain, barrier (now every peer may overwrite what I read).
  void allreduce(void* buf, size_t n, CommDType t, CommOp op, hipStream_t s) override {
    const size_t esz = t == COMM_F64 ? 8 : 4;
    enter(buf, nullptr, s);
    tmp_.alloc((n * esz + 3) / 4);
    try {
      HIP_CHECK(hipMemcpyAsync(tmp_.p, g_->slots[0].ptr, n * esz, hipMemcpyDeviceToDevice, s));
      for (int r = 1; r < world_; ++r) reduce_into(tmp_.p, g_->slots[(size_t)r].ptr, n, t, op, s);  // fixed order: same bits everywhere
      HIP_CHECK(hipStreamSynchronize(s));
    } catch (...) {
      g_->fail();
      throw;
    }
    g_->barrier();
    HIP_CHECK(hipMemcpyAsync(buf, tmp_.p, n * esz, hipMemcpyDeviceToDevice, s));
    // tmp_ is shared by every stream this rank runs collectives on (the solve's and the stop test's second stream): the
    // copy out of it must have finished before another collective may refill it
    HIP_CHECK(hipStreamSynchronize(s));
  }
  void allgather(void* buf, size_t chunk_bytes, hipStream_t s) override {
    enter(buf, nullptr, s);
    try {
      for (int r = 0; r < world_; ++r) {
        if (r == rank_ || !chunk_bytes) continue;
        const size_t off = (size_t)r * chunk_bytes;
        HIP_CHECK(hipMemcpyAsync(static_cast<char*>(buf) + off, static_cast<const char*>(g_->slots[(size_t)r].ptr) + off,
                                 chunk_bytes, hipMemcpyDeviceToDevice, s));
      }
      HIP_CHECK(hipStreamSynchronize(s));
    } catch (...) {
      g_->fail();
      throw;
    }
    g_->barrier();
  }
  void broadcast_group(const std::vector<CommXfer>& pieces, hipStream_t s) override {
    enter(nullptr, &pieces, s);
    try {
      for (size_t i = 0; i < pieces.size(); ++i) {
        const CommXfer& p = pieces[i];
        if (p.peer == rank_ || !p.bytes) continue;
        if (p.peer < 0 || p.peer >= world_) throw CommError("loopback broadcast: bad root");
        const std::vector<CommXfer>& theirs = *g_->slots[(size_t)p.peer].list;
        if (theirs.size() != pieces.size() || theirs[i].bytes != p.bytes || theirs[i].peer != p.peer)
          throw CommError("loopback broadcast: the ranks' piece lists differ");
        HIP_CHECK(hipMemcpyAsync(p.ptr, theirs[i].ptr, p.bytes, hipMemcpyDeviceToDevice, s));
      }
      HIP_CHECK(hipStreamSynchronize(s));
    } catch (...) {
      g_->fail();
      throw;
    }
    g_->barrier();
  }
  void exchange(const std::vector<CommXfer>& sends, const std::vector<CommXfer>& recvs, hipStream_t s) override {
    enter(nullptr, &sends, s);
    try {
      std::vector<size_t> cursor((size_t)world_, 0);  // next unmatched send of each peer
      for (const CommXfer& rv : recvs) {
        if (rv.peer < 0 || rv.peer >= world_) throw CommError("loopback recv: bad peer");
        const std::vector<CommXfer>& theirs = *g_->slots[(size_t)rv.peer].list;
        size_t& c = cursor[(size_t)rv.peer];
        while (c < theirs.size() && theirs[c].peer != rank_) ++c;
        if (c >= theirs.size() || theirs[c].bytes != rv.bytes) throw CommError("loopback exchange: unmatched recv");
        if (rv.bytes) HIP_CHECK(hipMemcpyAsync(rv.ptr, theirs[c].ptr, rv.bytes, hipMemcpyDeviceToDevice, s));
        ++c;
      }
      HIP_CHECK(hipStreamSynchronize(s));
    } catch (...) {
      g_->fail();
      throw;
    }
    g_->barrier();
  }

 private:
  void enter(void* ptr, const std::vector<CommXfer>* list, hipStream_t s) {
    g_->slots[(size_t)rank_].ptr = ptr;
    g_->slots[(size_t)rank_].list = list;
    const hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
      g_->fail();
      hip_check(e, "hipStreamSynchronize (loopback collective)", __FILE__, __LINE__);
    }
    g_->barrier();
  }
  static void reduce_into(void* acc, const void* src, size_t n, CommDType t, CommOp op, hipStream_t s) {
    if (!n) return;
    const int grid = (int)std::min<size_t>((n + 255) / 256, 1024);
#define OSC_LOOP_CASE(T, TT)                                                                                   \
  if (t == T) {                                                                                                \
    if (op == COMM_SUM) hipLaunchKernelGGL((k_loop_reduce<TT, COMM_SUM>), dim3(grid), dim3(256), 0, s,         \
                                           static_cast<TT*>(acc), static_cast<const TT*>(src), n);            \
    else hipLaunchKernelGGL((k_loop_reduce<TT, COMM_MAX>), dim3(grid), dim3(256), 0, s, static_cast<TT*>(acc), \
                            static_cast<const TT*>(src), n);                                                   \
  }
    OSC_LOOP_CASE(COMM_F32, float)
    OSC_LOOP_CASE(COMM_F64, double)
    OSC_LOOP_CASE(COMM_I32, int32_t)
#undef OSC_LOOP_CASE
    HIP_CHECK(hipGetLastError());
  }

  std::shared_ptr<LoopGroup> g_;
  DevBuf<uint32_t> tmp_;
};

}  // namespace

void comm_rccl_id(char id_out[128]) {
  ncclUniqueId id;
  nccl_ok(ncclGetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(id_out, &id, 128);
}

int comm_rccl_version() {
  int v = 0;
  return ncclGetVersion(&v) == ncclSuccess ? v : 0;
}

void comm_loopback_id(char id_out[128]) {
  std::memset(id_out, 0, 128);
  std::memcpy(id_out, kLoopMagic, 8);
  const uint64_t key = g_loop_next.fetch_add(1);
  std::memcpy(id_out + 8, &key, 8);
}

std::unique_ptr<Comm> comm_create(const char id[128], int rank, int world, int device) {
  (void)device;
  if (std::memcmp(id, kLoopMagic, 8) == 0) return std::unique_ptr<Comm>(new LoopbackComm(id, rank, world));
  return std::unique_ptr<Comm>(new RcclComm(id, rank, world));
}

}  // namespace osc
