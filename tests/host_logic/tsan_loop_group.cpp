// The loopback communicator's host barrier (oscillink_amd/csrc/loop_group.hpp, the struct comm.hip uses) under
// -fsanitize=thread: `world` threads pass thousands of generations exchanging values through the group's slots the way
// the collectives do (publish, barrier, read the peers', barrier); then one rank stays away and every other rank must
// get a CommError instead of hanging, and the group must stay broken.
#include <atomic>
#include <cstdio>
#include <thread>

#include "../../oscillink_amd/csrc/loop_group.hpp"

int main() {
  int fails = 0;
  for (int world : {2, 3, 8}) {
    osc::LoopGroup g;
    g.world = world;
    g.slots.resize((size_t)world);
    g.timeout_s = 20.0;
    std::vector<long> mine((size_t)world, 0);
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r)
      th.emplace_back([&, r] {
        for (int it = 0; it < 3000; ++it) {
          mine[(size_t)r] = 1000L * it + r;
          g.slots[(size_t)r].ptr = &mine[(size_t)r];
          g.barrier();
          long sum = 0;
          for (int q = 0; q < world; ++q) sum += *static_cast<long*>(g.slots[(size_t)q].ptr);
          if (sum != 1000L * it * world + (long)world * (world - 1) / 2) ++bad;
          g.barrier();
        }
      });
    for (auto& t : th) t.join();
    if (bad.load()) {
      std::fprintf(stderr, "world %d: %d wrong sums\n", world, bad.load());
      ++fails;
    }
    // a rank that never arrives: the others time out together, the group stays broken
    g.timeout_s = 0.3;
    std::atomic<int> thrown{0};
    th.clear();
    for (int r = 0; r + 1 < world; ++r)
      th.emplace_back([&] {
        try {
          g.barrier();
        } catch (const osc::CommError&) {
          ++thrown;
        }
      });
    for (auto& t : th) t.join();
    if (thrown.load() != world - 1) {
      std::fprintf(stderr, "world %d: %d of %d waiting ranks got the timeout\n", world, thrown.load(), world - 1);
      ++fails;
    }
    bool again = false;
    try {
      g.barrier();
    } catch (const osc::CommError&) {
      again = true;
    }
    if (!again) {
      std::fprintf(stderr, "world %d: a broken group let a rank through\n", world);
      ++fails;
    }
  }
  if (fails) return 1;
  std::printf("loop group ok\n");
  return 0;
}
