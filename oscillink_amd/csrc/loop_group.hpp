// The host barrier of the in-process loopback communicator (comm.hip), free of HIP types so that it also builds with a
// plain host compiler: tests/host_logic/tsan_loop_group.cpp hammers it under -fsanitize=thread on the CPU box.
#pragma once
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <stdexcept>
#include <vector>

namespace osc {

struct CommError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// one piece of a grouped transfer: `bytes` at `ptr`; peer = destination (send), source (recv) or root (broadcast)
struct CommXfer {
  void* ptr;
  size_t bytes;
  int peer;
};

struct LoopGroup {
  int world = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  bool broken = false;
  int joined = 0;
  double timeout_s = 300.0;  // a rank may build a large lattice (N ~ 1M: seconds) before its first collective
  struct Slot {
    void* ptr = nullptr;
    const std::vector<CommXfer>* list = nullptr;
  };
  std::vector<Slot> slots;

  // every rank of the group passes, or every rank throws
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (broken) throw CommError("loopback communicator is broken (a rank failed or timed out earlier)");
    const uint64_t gen = generation;
    if (++arrived == world) {
      arrived = 0;
      ++generation;
      cv.notify_all();
      return;
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s);
    while (generation == gen && !broken) {
      if (cv.wait_until(lk, deadline) == std::cv_status::timeout && generation == gen) {
        broken = true;  // a rank never reached this collective: mismatched call sequences
        cv.notify_all();
      }
    }
    if (generation == gen) throw CommError("loopback barrier timed out: the ranks' collective sequences differ");
  }
  void fail() {
    std::lock_guard<std::mutex> lk(mu);
    broken = true;
    cv.notify_all();
  }
};

}  // namespace osc
