// Host-memory reduce backend: the gradient reducer's state machine (reducer_core.h) over the TcpStore.
//
// A worker thread plays the communication stream: launch() enqueues a slice, the worker publishes this rank's
// bytes under a per-collective key, fetches every peer's, sums in RANK ORDER in double precision (bit-identical
// on every rank) and writes the mean back in place.  O(world^2) bytes through one server -- this is the CPU
// plumbing configuration (BASELINE config #1) and the way the C++ reducer is exercised at world sizes the
// one-GPU test box cannot offer; gradients of a real job travel over RCCL (comm.cpp).
#pragma once
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "reducer_core.h"
#include "store_core.h"

namespace dpa {
namespace comm {

class HostBackend : public ReduceBackend {
 public:
  // opens a store connection of its own (the training thread keeps using the process group's)
  HostBackend(const std::string& host, int port, int rank, int world, double timeout_s, const std::string& tag);
  ~HostBackend() override;
  int world() const override { return world_; }
  void record_ready(int bucket, uintptr_t) override;   // host memory: program order is the ordering (time stamp only)
  void launch(int bucket, uintptr_t base, size_t offset_elems, size_t count, int dtype, int wire) override;
  void record_done(int bucket) override;
  void wait_done(int bucket, uintptr_t) override;
  bool query_done(int bucket) override;
  std::string async_error() override;
  void abort() override;
  void resize(int num_buckets) override;
  bool set_timing(bool on) override;
  void mark_tail(uintptr_t) override;
  std::vector<double> bucket_timings(int num_buckets) override;
  long collectives() const { return seq_; }

 private:
  struct Task {
    int bucket;
    bool marker;            // done marker of `bucket`
    void* ptr;
    size_t count;
    int dtype;
    int wire;               // -1: travels as stored; 1: fp32 storage rounded to bf16 on the wire
    long seq;
  };
  void run();
  void reduce(const Task& t);

  store::Client client_;
  int rank_, world_;
  std::string tag_;
  long seq_ = 0;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Task> queue_;
  std::vector<long> issued_, done_;     // done markers per bucket: issued vs retired
  bool stop_ = false, aborted_ = false, timing_ = false;
  using Stamp = std::chrono::steady_clock::time_point;
  std::vector<Stamp> t_ready_, t_done_;  // host clock at the ready point / when the worker retired the done marker
  Stamp t_tail_;
  std::string error_;
  std::thread thread_;
};

}  // namespace comm
}  // namespace dpa
