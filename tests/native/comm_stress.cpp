// Stress test of the C++ host runtime, built WITHOUT Python or HIP so that AddressSanitizer + UBSan and
// ThreadSanitizer can instrument all of it (tests/native/Makefile; run by tests/test_native_sanitizers.py):
//
//   1. TcpStore: 8 client threads -- counters, parked gets answered by later sets, 200 barriers, oversize keys,
//      clients that die mid-request, and the shutdown race (the master's store is destroyed while the last
//      barrier's replies are still unread by the other ranks).
//   2. BucketReducerCore + HostBackend at world 4 (one thread per rank): random ready orders, chunked buckets,
//      unused parameters, results against the exact mean.
//   3. Watchdog: one rank stops calling all-reduce; the others must fail within the timeout, not hang.
//   4. plan_buckets on random parameter lists (every parameter exactly once, aligned offsets, caps respected).
#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "comm/host_backend.h"
#include "comm/reducer_core.h"
#include "comm/store_core.h"

using namespace dpa;

#define REQUIRE(cond)                                                              \
  do {                                                                             \
    if (!(cond)) {                                                                 \
      fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);            \
      std::abort();                                                                \
    }                                                                              \
  } while (0)

static int raw_connect(int port) {
  int fd = ::socket(AF_INET, SOCK_STREAM, 0);
  sockaddr_in a{};
  a.sin_family = AF_INET;
  a.sin_port = htons((uint16_t)port);
  a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
  REQUIRE(::connect(fd, reinterpret_cast<sockaddr*>(&a), sizeof(a)) == 0);
  return fd;
}

static void store_stress(int rounds) {
  const int world = 8;
  for (int round = 0; round < rounds; ++round) {
    auto master = std::make_unique<store::TcpStore>("127.0.0.1", 0, true, 20.0, 5.0);
    const int port = master->port();
    std::atomic<int> errors{0};
    std::vector<std::thread> ts;
    // misbehaving peers: a half-sent request then death, an oversized key, a parked get whose owner dies
    {
      int fd = raw_connect(port);
      const char half[3] = {2, 9, 0};
      (void)!::write(fd, half, sizeof(half));
      ::close(fd);
      fd = raw_connect(port);
      std::string big;
      big.push_back(1);
      uint32_t klen = 1u << 20;
      big.append(reinterpret_cast<const char*>(&klen), 4);
      big.append(64, 'k');
      (void)!::write(fd, big.data(), big.size());
      char c;
      REQUIRE(::read(fd, &c, 1) <= 0);          // the server hung up on us
      ::close(fd);
      fd = raw_connect(port);
      std::string get;
      get.push_back(2);
      klen = 5;
      uint64_t vlen = 0;
      get.append(reinterpret_cast<const char*>(&klen), 4);
      get.append("never");
      get.append(reinterpret_cast<const char*>(&vlen), 8);
      (void)!::write(fd, get.data(), get.size());
      ::close(fd);                               // dies with its get parked
    }
    for (int r = 1; r < world; ++r) {
      ts.emplace_back([&, r] {
        try {
          store::Client c("127.0.0.1", port, 20.0);
          for (int i = 0; i < 50; ++i) {
            c.add("counter", 1);
            if (i % 10 == 0) {
              // ring of parked gets: rank r waits for what rank r+1 publishes a little later
              const std::string mine = "ring/" + std::to_string(i) + "/" + std::to_string(r);
              const std::string next = "ring/" + std::to_string(i) + "/" + std::to_string(r % (world - 1) + 1);
              if (r % 2) std::this_thread::sleep_for(std::chrono::milliseconds(2));
              c.set(mine, std::string(1000 + r, (char)('a' + r)));
              REQUIRE(c.get(next).size() == (size_t)(1000 + r % (world - 1) + 1));
            }
            c.barrier("b", world);
          }
          REQUIRE(c.add("counter", 0) == 50 * world);
          c.barrier("final", world);
        } catch (const std::exception& e) {
          fprintf(stderr, "client %d: %s\n", r, e.what());
          errors++;
        }
      });
    }
    // rank 0: same script on the master's own client, then it destroys the store IMMEDIATELY after the final
    // barrier -- the other ranks' release replies may still be unread (the round-1 shutdown race)
    try {
      store::Client& c = master->client();
      for (int i = 0; i < 50; ++i) {
        c.add("counter", 1);
        c.barrier("b", world);
      }
      REQUIRE(c.check("counter"));
      REQUIRE(!c.check("never"));
      c.barrier("final", world);
    } catch (const std::exception& e) {
      fprintf(stderr, "master: %s\n", e.what());
      errors++;
    }
    master.reset();
    for (auto& t : ts) t.join();
    REQUIRE(errors.load() == 0);
  }
  // a client whose master is gone fails fast instead of burning its whole timeout twice
  {
    int port;
    {
      store::TcpStore m("127.0.0.1", 0, true, 5.0, 0.2);
      port = m.port();
    }
    const auto t0 = std::chrono::steady_clock::now();
    bool threw = false;
    try {
      store::Client c("127.0.0.1", port, 1.0);
    } catch (const std::exception&) {
      threw = true;
    }
    REQUIRE(threw);
    REQUIRE(std::chrono::steady_clock::now() - t0 < std::chrono::seconds(4));
  }
  printf("store stress ok (%d rounds)\n", rounds);
}

struct RankState {
  std::vector<std::vector<float>> buckets;
};

static void reducer_stress(bool with_dead_rank) {
  const int world = 4, nparams = 23;
  store::TcpStore master("127.0.0.1", 0, true, 30.0, 5.0);
  const int port = master.port();
  // parameters -> 4 buckets, one of them big enough to be chunked into several slices
  std::vector<int> param_bucket(nparams);
  std::vector<size_t> counts = {257, 70001, 33, 4096};
  for (int p = 0; p < nparams; ++p) param_bucket[p] = p % 4;
  std::atomic<int> errors{0}, failed_ranks{0};
  std::vector<std::thread> ts;
  const int iters = with_dead_rank ? 3 : 12;
  for (int r = 0; r < world; ++r) {
    ts.emplace_back([&, r] {
      try {
        std::vector<std::vector<float>> bufs;
        std::vector<uintptr_t> ptrs;
        std::vector<int> dtypes;
        for (size_t c : counts) {
          bufs.emplace_back(c);
          dtypes.push_back(0);
        }
        for (auto& b : bufs) ptrs.push_back(reinterpret_cast<uintptr_t>(b.data()));
        auto be = std::make_shared<comm::HostBackend>("127.0.0.1", port, r, world, 30.0,
                                                      with_dead_rank ? "dead" : "live");
        comm::BucketReducerCore red(be, ptrs, counts, dtypes, param_bucket, /*chunk_bytes=*/16384,
                                    /*timeout_s=*/with_dead_rank ? 1.0 : 20.0, comm::OnTimeout::kFlagOnly, 1.0);
        std::mt19937 rng(1234 + r);          // every rank marks in its OWN order: only bucket order must agree
        for (int it = 0; it < iters; ++it) {
          for (size_t b = 0; b < bufs.size(); ++b)
            for (size_t i = 0; i < bufs[b].size(); ++i) bufs[b][i] = (float)((r + 1) * (it + 1)) + (float)(i % 7);
          if (with_dead_rank && it == iters - 1 && r == world - 1) {
            std::this_thread::sleep_for(std::chrono::seconds(4));     // this rank never calls all-reduce again
            return;
          }
          std::vector<int> order(nparams);
          for (int p = 0; p < nparams; ++p) order[p] = p;
          std::shuffle(order.begin(), order.end(), rng);
          const int skip = it % 3 == 2 ? order.back() : -1;            // an unused parameter every third step
          for (int p : order)
            if (p != skip) {
              red.mark_ready(p, 0);
              if (p % 5 == 0) red.mark_ready(p, 0);                      // shared weight reported twice
            }
          red.finalize(0);
          red.prepare();
          const float want_base = (float)(it + 1) * (1 + 2 + 3 + 4) / 4.0f;
          for (size_t b = 0; b < bufs.size(); ++b)
            for (size_t i = 0; i < bufs[b].size(); ++i)
              REQUIRE(std::fabs(bufs[b][i] - (want_base + (float)(i % 7))) < 1e-4f);
          REQUIRE((int)red.observed_order().size() == nparams - (skip >= 0 ? 1 : 0));
        }
        REQUIRE(red.collectives() > (long)iters * 4);                   // the big bucket travelled in slices
      } catch (const std::exception& e) {
        if (with_dead_rank && std::string(e.what()).find("reducer failed") != std::string::npos) failed_ranks++;
        else {
          fprintf(stderr, "rank %d: %s\n", r, e.what());
          errors++;
        }
      }
    });
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (auto& t : ts) t.join();
  REQUIRE(errors.load() == 0);
  if (with_dead_rank) {
    REQUIRE(failed_ranks.load() == world - 1);
    REQUIRE(std::chrono::steady_clock::now() - t0 < std::chrono::seconds(20));
    printf("watchdog ok (3 survivors failed within the timeout)\n");
  } else {
    printf("reducer stress ok (world %d, %d iterations)\n", world, iters);
  }
}

static void planner_fuzz() {
  std::mt19937 rng(7);
  for (int trial = 0; trial < 300; ++trial) {
    const int n = 1 + (int)(rng() % 40);
    std::vector<long> numels(n), esz(n), dk(n), order(n);
    const long sizes[] = {1, 7, 64, 1000, 4096, 250000, 3000000};
    for (int i = 0; i < n; ++i) {
      numels[i] = sizes[rng() % 7];
      dk[i] = rng() % 4 == 0;
      esz[i] = dk[i] ? 2 : 4;
      order[i] = i;
    }
    std::shuffle(order.begin(), order.end(), rng);
    const long caps[] = {1L << 12, 1L << 20, 1L << 24, 1L << 40};
    const long cap = caps[rng() % 4], first = caps[rng() % 4], last = rng() % 2 ? 0 : caps[rng() % 4];
    const long align = rng() % 2 ? 16 : 256;
    auto plan = comm::plan_buckets(numels, esz, dk, order, cap, first, align, last);
    std::set<long> seen;
    for (auto& b : plan) {
      const auto& pids = std::get<1>(b);
      const auto& offs = std::get<2>(b);
      REQUIRE(pids.size() == offs.size() && !pids.empty());
      long end = 0;
      for (size_t i = 0; i < pids.size(); ++i) {
        REQUIRE(seen.insert(pids[i]).second);
        REQUIRE(dk[pids[i]] == std::get<0>(b));
        REQUIRE(offs[i] * esz[pids[i]] % align == 0 && offs[i] >= end);
        end = offs[i] + numels[pids[i]];
      }
      REQUIRE(std::get<3>(b) >= end);
    }
    REQUIRE((int)seen.size() == n);
  }
  bool threw = false;
  try {
    comm::plan_buckets({1}, {4}, {0}, {3}, 1, 1, 256, 0);
  } catch (const std::invalid_argument&) {
    threw = true;
  }
  REQUIRE(threw);
  printf("planner fuzz ok\n");
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 20;
  store_stress(rounds);
  reducer_stress(false);
  reducer_stress(true);
  planner_fuzz();
  printf("ALL OK\n");
  return 0;
}
