// Bucketed gradient reducer: the state machine of DistributedDataParallel, independent of where the bytes
// travel (no Python, no HIP in this header: tests/native/ builds it with the sanitizers).
//
//  * Gradients live as VIEWS inside a few flat buckets.  When backward has produced every gradient of a bucket
//    the compute queue records a ready point, the communication queue waits for it and all-reduces (mean) the
//    bucket in place while backward keeps running.  Buckets are issued in index order so every rank enqueues
//    the same collective sequence; finalize() issues the stragglers and joins the communication queue back.
//  * A bucket larger than `chunk_bytes` (VGG-16 fc6.weight: 392 MiB) travels as back-to-back slices, so no
//    single collective monopolises the links and the watchdog sees progress at slice granularity.
//  * The order in which gradients became ready during an iteration is recorded; the wrapper re-plans the
//    buckets from it after the first iteration (what c10d's Reducer does for imagenet_ddp.py:127).
//  * A watchdog thread polls the done points of the in-flight buckets and the backend's asynchronous error
//    state.  A bucket that does not complete within `timeout_s` (a peer died, or stopped calling all-reduce)
//    aborts the backend (ncclCommAbort on RCCL) and fails the rank: the next reducer call throws, and if the
//    training thread is stuck for another `grace_s` the process exits with kWatchdogExitCode so that the
//    launcher tears the job down instead of hanging.
//
// Backends: HIP events + RCCL on the communicator's stream (comm.cpp), and a host-memory backend over the
// TcpStore with a worker thread as its "stream" (host_backend.cpp) that runs the SAME state machine at any
// world size without a GPU.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

namespace dpa {
namespace comm {

constexpr int kWatchdogExitCode = 86;

size_t dtype_bytes(int code);   // dpa::DType codes + 3 = int64, 4 = uint8

class ReduceBackend {
 public:
  virtual ~ReduceBackend() = default;
  virtual int world() const = 0;
  // ordering point: everything enqueued on `compute_stream` so far precedes this bucket's collectives
  virtual void record_ready(int bucket, uintptr_t compute_stream) = 0;
  // one in-place mean all-reduce of `count` elements at base + offset, behind the bucket's ready point.
  // `wire` is the dtype code the values TRAVEL in (-1 or == dtype: as stored).  A 16-bit wire halves the bytes on
  // the links (Apex O2 all-reduces fp16 gradients, imagenet_ddp_apex.py:169-183): the backend casts the slice on
  // its own queue, reduces, and writes the result back in the storage dtype -- every rank ends with the same bits.
  virtual void launch(int bucket, uintptr_t base, size_t offset_elems, size_t count, int dtype, int wire) = 0;
  virtual void record_done(int bucket) = 0;
  // the compute queue (or the calling thread, for a host backend) waits for the bucket's done point
  virtual void wait_done(int bucket, uintptr_t compute_stream) = 0;
  virtual bool query_done(int bucket) = 0;    // non-blocking, watchdog thread
  virtual std::string async_error() = 0;      // "" while healthy, watchdog thread
  virtual void abort() = 0;                   // any thread: unblocks everything in flight
  virtual void resize(int num_buckets) = 0;
  // ---- self-diagnosis (optional; default: not supported).  With timing on, the ready / done points carry time
  // stamps and finalize() marks the end of backward on the compute queue, so a step can report, per bucket, how long
  // its collectives took behind the ready point and how much of that stuck out past the last backward kernel.
  virtual bool set_timing(bool /*on*/) { return false; }
  virtual void mark_tail(uintptr_t /*compute_stream*/) {}
  // blocks until the last issued bucket is done; 2 values per bucket in milliseconds: (ready -> done,
  // end of backward -> done; negative = finished that long BEFORE backward ended, i.e. fully hidden)
  virtual std::vector<double> bucket_timings(int /*num_buckets*/) { return {}; }
};

enum class OnTimeout : int { kExitProcess = 0, kFlagOnly = 1 };

class BucketReducerCore {
 public:
  BucketReducerCore(std::shared_ptr<ReduceBackend> backend, std::vector<uintptr_t> ptrs, std::vector<size_t> counts,
                    std::vector<int> dtypes, std::vector<int> param_bucket, size_t chunk_bytes, double timeout_s,
                    OnTimeout on_timeout = OnTimeout::kExitProcess, double grace_s = 10.0, int wire_dtype = -1);
  ~BucketReducerCore();

  void prepare();
  void mark_ready(int idx, uintptr_t compute_stream);   // gradient of parameter idx is enqueued on the stream
  void flush(uintptr_t compute_stream, bool force);
  void finalize(uintptr_t compute_stream);

  int num_buckets() const { return (int)buckets_.size(); }
  int launched() const { return launched_; }
  long collectives() const { return collectives_; }       // slices issued since construction
  long wire_bytes() const { return wire_bytes_; }         // bytes handed to the links since construction
  int wire_dtype() const { return wire_; }
  std::vector<int> observed_order() const { return observed_; }   // last complete iteration's ready order
  bool failed() const { return failed_.load(); }
  // the watchdog must not query events while a hipGraph capture is recording them (an illegal call there can
  // invalidate the capture); paused, it only sleeps
  void set_paused(bool on) { paused_.store(on); }
  bool set_timing(bool on) { return backend_->set_timing(on); }
  std::vector<double> bucket_timings() { return backend_->bucket_timings((int)buckets_.size()); }
  std::string failure() const;

 private:
  struct Bucket {
    uintptr_t ptr = 0;
    size_t count = 0;
    int dtype = 0;
    int nparams = 0, pending = 0;
    bool launched = false;
  };
  void launch_bucket(size_t b, uintptr_t compute_stream);
  void check_failed();
  void fail(const std::string& why);
  void watchdog();

  std::shared_ptr<ReduceBackend> backend_;
  std::vector<Bucket> buckets_;
  std::vector<int> param_bucket_;
  std::vector<char> seen_;            // per parameter: marked ready this iteration
  std::vector<int> order_, observed_;
  size_t chunk_bytes_;
  size_t next_ = 0;
  int launched_ = 0;
  long collectives_ = 0;
  long wire_bytes_ = 0;
  int wire_ = -1;

  // watchdog state
  double timeout_s_, grace_s_;
  OnTimeout on_timeout_;
  mutable std::mutex mu_;
  std::vector<char> inflight_;
  std::vector<std::chrono::steady_clock::time_point> since_;
  std::atomic<bool> failed_{false}, acked_{false}, stop_{false}, paused_{false};
  std::string failure_;
  std::thread thread_;
};

// ---- bucket planner ----------------------------------------------------------------------------------------
// Packs parameters, given in expected gradient-ready order, into dtype-homogeneous flat buckets.  The first
// bucket of each dtype is small (its all-reduce starts early), the parameters that become ready LAST form a
// small bucket of their own (its all-reduce is the only one that cannot overlap with backward), everything
// in between fills `cap` buckets sized for xGMI.  A bucket smaller than `min_bytes` is never closed: it takes the
// next tensor even past the cap (no latency-bound collective for a lone bias).  Every view offset is aligned to
// `align` bytes.
// -> (dtype_key, param ids, element offsets, padded element total) per bucket
std::vector<std::tuple<long, std::vector<long>, std::vector<long>, long>> plan_buckets(
    const std::vector<long>& numels, const std::vector<long>& esz, const std::vector<long>& dkeys,
    const std::vector<long>& order, long cap, long first_cap, long align, long last_cap, long min_bytes = 1L << 20);

}  // namespace comm
}  // namespace dpa
