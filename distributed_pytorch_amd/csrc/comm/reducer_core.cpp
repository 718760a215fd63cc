// Reducer state machine, watchdog and bucket planner -- see reducer_core.h.
#include "reducer_core.h"

#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <stdexcept>
#include <utility>

namespace dpa {
namespace comm {

using Clock = std::chrono::steady_clock;

size_t dtype_bytes(int code) {
  switch (code) {
    case 0: return 4;
    case 1: case 2: return 2;
    case 3: return 8;
    default: return 1;
  }
}

BucketReducerCore::BucketReducerCore(std::shared_ptr<ReduceBackend> backend, std::vector<uintptr_t> ptrs,
                                     std::vector<size_t> counts, std::vector<int> dtypes,
                                     std::vector<int> param_bucket, size_t chunk_bytes, double timeout_s,
                                     OnTimeout on_timeout, double grace_s, int wire_dtype)
    : backend_(std::move(backend)), param_bucket_(std::move(param_bucket)), chunk_bytes_(chunk_bytes),
      wire_(wire_dtype), timeout_s_(timeout_s), grace_s_(grace_s), on_timeout_(on_timeout) {
  if (!backend_) throw std::invalid_argument("reducer: no backend");
  if (ptrs.size() != counts.size() || ptrs.size() != dtypes.size()) throw std::invalid_argument("bucket arrays");
  buckets_.resize(ptrs.size());
  for (size_t b = 0; b < ptrs.size(); ++b) {
    buckets_[b].ptr = ptrs[b];
    buckets_[b].count = counts[b];
    buckets_[b].dtype = dtypes[b];
  }
  for (int b : param_bucket_) {
    if (b < 0 || b >= (int)buckets_.size()) throw std::invalid_argument("param -> bucket index out of range");
    buckets_[b].nparams++;
  }
  seen_.assign(param_bucket_.size(), 0);
  inflight_.assign(buckets_.size(), 0);
  since_.assign(buckets_.size(), Clock::now());
  backend_->resize((int)buckets_.size());
  prepare();
  if (timeout_s_ > 0) thread_ = std::thread([this] { watchdog(); });
}

BucketReducerCore::~BucketReducerCore() {
  stop_.store(true);
  if (thread_.joinable()) thread_.join();
}

void BucketReducerCore::prepare() {
  for (Bucket& k : buckets_) {
    k.pending = k.nparams;
    k.launched = false;
  }
  if (!order_.empty()) observed_.swap(order_);
  order_.clear();
  std::fill(seen_.begin(), seen_.end(), 0);
  next_ = 0;
  launched_ = 0;
}

std::string BucketReducerCore::failure() const {
  std::lock_guard<std::mutex> lock(mu_);
  return failure_;
}

void BucketReducerCore::check_failed() {
  if (!failed_.load()) return;
  acked_.store(true);
  throw std::runtime_error("gradient reducer failed: " + failure());
}

void BucketReducerCore::mark_ready(int idx, uintptr_t compute_stream) {
  check_failed();
  if (idx < 0 || idx >= (int)param_bucket_.size()) throw std::out_of_range("reducer: bad parameter index");
  if (seen_[idx]) return;          // a parameter used twice is complete at its first (= last-use) report
  seen_[idx] = 1;
  order_.push_back(idx);
  Bucket& k = buckets_[param_bucket_[idx]];
  if (k.pending <= 0) return;
  if (--k.pending == 0) {
    backend_->record_ready(param_bucket_[idx], compute_stream);
    flush(compute_stream, false);
  }
}

void BucketReducerCore::launch_bucket(size_t b, uintptr_t /*compute_stream*/) {
  Bucket& k = buckets_[b];
  const size_t esz = dtype_bytes(k.dtype);
  size_t step = chunk_bytes_ > 0 ? std::max<size_t>(1, chunk_bytes_ / esz) : k.count;
  if (step >= k.count) step = k.count;
  // only fp32 storage is narrowed on the wire; 16-bit buckets (O3) already travel as they are
  const int wire = (wire_ >= 0 && k.dtype == 0 && wire_ != k.dtype) ? wire_ : -1;
  for (size_t off = 0; off < k.count; off += step) {
    const size_t n = std::min(step, k.count - off);
    backend_->launch((int)b, k.ptr, off, n, k.dtype, wire);
    ++collectives_;
    wire_bytes_ += (long)(n * (wire >= 0 ? dtype_bytes(wire) : esz));
  }
  backend_->record_done((int)b);
  k.launched = true;
  ++launched_;
  {
    std::lock_guard<std::mutex> lock(mu_);
    inflight_[b] = 1;
    since_[b] = Clock::now();
  }
}

// issue, in index order, every bucket whose gradients are complete (all of them when `force`)
void BucketReducerCore::flush(uintptr_t compute_stream, bool force) {
  check_failed();
  while (next_ < buckets_.size()) {
    Bucket& k = buckets_[next_];
    if (k.pending > 0) {
      if (!force) break;
      k.pending = 0;   // unused parameters: their (zeroed) gradient slots are reduced as they are
      backend_->record_ready((int)next_, compute_stream);
    }
    launch_bucket(next_, compute_stream);
    ++next_;
  }
}

// end of backward: issue the stragglers and make the compute queue wait for every bucket
void BucketReducerCore::finalize(uintptr_t compute_stream) {
  flush(compute_stream, true);
  backend_->mark_tail(compute_stream);      // (no-op unless the backend is timing this step)
  // buckets are issued in index order on ONE in-order queue: the last bucket's done point implies every earlier
  // one (a cross-stream wait costs ~15 us of host time on HIP, and a dependency edge on the GPU, per call)
  if (!buckets_.empty()) backend_->wait_done((int)buckets_.size() - 1, compute_stream);
  check_failed();
}

void BucketReducerCore::fail(const std::string& why) {
  {
    std::lock_guard<std::mutex> lock(mu_);
    if (failed_.load()) return;
    failure_ = why;
    failed_.store(true);
  }
  fprintf(stderr, "[dpa watchdog] %s -- aborting the communicator\n", why.c_str());
  fflush(stderr);
  try {
    backend_->abort();
  } catch (...) {
  }
}

void BucketReducerCore::watchdog() {
  const auto period = std::chrono::milliseconds(
      (long)std::min(500.0, std::max(10.0, timeout_s_ * 1000.0 / 8.0)));
  // plain sleeps, no timed condition wait: libstdc++'s wait_for is pthread_cond_clockwait, which gcc 11's
  // ThreadSanitizer does not model (it then reports the mutex as double-locked)
  while (!stop_.load()) {
    for (auto slept = std::chrono::milliseconds(0); slept < period && !stop_.load();
         slept += std::chrono::milliseconds(10))
      std::this_thread::sleep_for(std::chrono::milliseconds(10));
    if (stop_.load()) break;
    if (paused_.load()) continue;
    std::unique_lock<std::mutex> lock(mu_);
    std::string why;
    const auto now = Clock::now();
    for (size_t b = 0; b < inflight_.size() && why.empty(); ++b) {
      if (!inflight_[b]) continue;
      if (backend_->query_done((int)b)) {
        inflight_[b] = 0;
        continue;
      }
      const double waited = std::chrono::duration<double>(now - since_[b]).count();
      if (waited > timeout_s_)
        why = "all-reduce of gradient bucket " + std::to_string(b) + " did not complete within " +
              std::to_string(timeout_s_) + " s (a peer died or stopped calling all-reduce)";
    }
    if (why.empty()) {
      std::string err = backend_->async_error();
      if (!err.empty()) why = "communicator reported an asynchronous error: " + err;
    }
    if (why.empty()) continue;
    lock.unlock();
    fail(why);
    if (on_timeout_ == OnTimeout::kExitProcess) {
      // the training thread gets `grace_s` to notice (its next reducer call throws) and then `grace_s` more to
      // unwind; a thread parked inside the runtime forever must not keep the job alive
      const auto t0 = Clock::now();
      bool acked_once = false;
      auto limit = t0 + std::chrono::milliseconds((long)(grace_s_ * 1000));
      while (!stop_.load() && Clock::now() < limit) {
        if (acked_.load() && !acked_once) {
          acked_once = true;
          limit = Clock::now() + std::chrono::milliseconds((long)(grace_s_ * 1000));
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
      }
      if (!stop_.load()) {
        fprintf(stderr, "[dpa watchdog] rank did not unwind after the abort; exiting with code %d\n",
                kWatchdogExitCode);
        fflush(stderr);
        _exit(kWatchdogExitCode);
      }
    }
    return;
  }
}

// ---- bucket planner ----------------------------------------------------------------------------------------
namespace {

struct PlannedBucket {
  long dtype_key = 0;
  std::vector<long> pids, offsets;
  long total = 0;     // elements, padded
};

void plan_range(const std::vector<long>& numels, const std::vector<long>& esz, const std::vector<long>& dkeys,
                const std::vector<long>& order, size_t begin, size_t end, long cap, long first_cap, long align,
                long min_bytes, std::vector<PlannedBucket>& out) {
  std::vector<std::pair<long, int>> open;       // dtype -> index into out (few dtypes: linear search)
  std::vector<long> first_done;
  auto find_open = [&](long dk) -> int {
    for (size_t i = 0; i < open.size(); ++i)
      if (open[i].first == dk) return (int)i;
    return -1;
  };
  auto done = [&](long dk) { return std::find(first_done.begin(), first_done.end(), dk) != first_done.end(); };
  for (size_t i = begin; i < end; ++i) {
    const long pid = order[i];
    const long dk = dkeys[pid], es = esz[pid], n = numels[pid];
    int slot = find_open(dk);
    if (slot < 0) {
      open.emplace_back(dk, -1);
      slot = (int)open.size() - 1;
    }
    int& cur = open[slot].second;
    const long limit = done(dk) ? cap : first_cap;
    // A bucket is closed when the next tensor would push it past the cap -- unless it is still tiny: a 16 KB bias
    // in front of a 392 MiB weight would otherwise travel as a latency-bound collective of its own, queued in
    // front of the big one (in-order issue).  Such a bucket takes the tensor anyway and is closed by the next.
    if (cur >= 0 && !out[cur].pids.empty() && out[cur].total * es + n * es > limit) {
      if (!done(dk)) first_done.push_back(dk);
      if (out[cur].total * es >= min_bytes) cur = -1;
    }
    if (cur < 0) {
      out.emplace_back();
      out.back().dtype_key = dk;
      cur = (int)out.size() - 1;
    }
    PlannedBucket& b = out[cur];
    b.pids.push_back(pid);
    b.offsets.push_back(b.total);
    const long align_el = std::max<long>(1, align / es);
    b.total += (n + align_el - 1) / align_el * align_el;
  }
}

}  // namespace

std::vector<std::tuple<long, std::vector<long>, std::vector<long>, long>> plan_buckets(
    const std::vector<long>& numels, const std::vector<long>& esz, const std::vector<long>& dkeys,
    const std::vector<long>& order, long cap, long first_cap, long align, long last_cap, long min_bytes) {
  if (numels.size() != esz.size() || numels.size() != dkeys.size())
    throw std::invalid_argument("plan_buckets: per-parameter lists differ in length");
  for (long pid : order)
    if (pid < 0 || (size_t)pid >= numels.size()) throw std::invalid_argument("plan_buckets: bad parameter index");
  for (long e : esz)
    if (e <= 0) throw std::invalid_argument("plan_buckets: bad element size");
  std::vector<PlannedBucket> out;
  size_t split = order.size();
  if (last_cap > 0 && order.size() > 1) {
    long acc = 0;
    while (split > 0) {
      const long pid = order[split - 1];
      const long sz = numels[pid] * esz[pid];
      if (split < order.size() && acc + sz > last_cap) break;
      acc += sz;
      --split;
    }
    if (split == 0) split = order.size();       // everything fits the tail: plan normally
  }
  plan_range(numels, esz, dkeys, order, 0, split, cap, first_cap, align, min_bytes, out);
  const long inf = 1L << 62;
  plan_range(numels, esz, dkeys, order, split, order.size(), inf, inf, align, min_bytes, out);
  std::vector<std::tuple<long, std::vector<long>, std::vector<long>, long>> res;
  res.reserve(out.size());
  for (auto& b : out) res.emplace_back(b.dtype_key, std::move(b.pids), std::move(b.offsets), b.total);
  return res;
}

}  // namespace comm
}  // namespace dpa
