// Host-memory reduce backend over the TcpStore -- see host_backend.h.
#include "host_backend.h"

#include <cstring>
#include <stdexcept>

namespace dpa {
namespace comm {
namespace {

inline float bf16_to_float(uint16_t v) {
  uint32_t u = (uint32_t)v << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
inline uint16_t float_to_bf16(float f) {   // round to nearest even, NaN preserved
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

}  // namespace

HostBackend::HostBackend(const std::string& host, int port, int rank, int world, double timeout_s,
                         const std::string& tag)
    : client_(host, port, timeout_s), rank_(rank), world_(world), tag_(tag) {
  if (world < 1 || rank < 0 || rank >= world) throw std::invalid_argument("HostBackend: bad rank / world size");
  thread_ = std::thread([this] { run(); });
}

HostBackend::~HostBackend() {
  {
    std::lock_guard<std::mutex> lock(mu_);
    stop_ = true;
  }
  client_.interrupt();
  cv_.notify_all();
  if (thread_.joinable()) thread_.join();
}

void HostBackend::resize(int num_buckets) {
  std::lock_guard<std::mutex> lock(mu_);
  issued_.assign((size_t)num_buckets, 0);
  done_.assign((size_t)num_buckets, 0);
  t_ready_.assign((size_t)num_buckets, Stamp());
  t_done_.assign((size_t)num_buckets, Stamp());
}

void HostBackend::record_ready(int bucket, uintptr_t) {
  std::lock_guard<std::mutex> lock(mu_);
  if (timing_) t_ready_.at((size_t)bucket) = std::chrono::steady_clock::now();
}

bool HostBackend::set_timing(bool on) {
  std::lock_guard<std::mutex> lock(mu_);
  timing_ = on;
  return true;
}

void HostBackend::mark_tail(uintptr_t) {
  std::lock_guard<std::mutex> lock(mu_);
  if (timing_) t_tail_ = std::chrono::steady_clock::now();
}

std::vector<double> HostBackend::bucket_timings(int n) {
  std::vector<double> out;
  if (n <= 0) return out;
  wait_done(n - 1, 0);
  std::lock_guard<std::mutex> lock(mu_);
  if (!timing_ || (size_t)n > t_done_.size()) return out;
  auto ms = [](Stamp a, Stamp b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  for (int b = 0; b < n; ++b) {
    out.push_back(ms(t_ready_[(size_t)b], t_done_[(size_t)b]));
    out.push_back(ms(t_tail_, t_done_[(size_t)b]));
  }
  return out;
}

void HostBackend::launch(int bucket, uintptr_t base, size_t offset_elems, size_t count, int dtype, int wire) {
  const size_t esz = dtype_bytes(dtype);
  if (dtype != 0 && dtype != 1) throw std::invalid_argument("HostBackend: float32 / bfloat16 gradients only");
  if (wire >= 0 && wire != dtype && !(dtype == 0 && wire == 1))
    throw std::invalid_argument("HostBackend: the only narrowed wire format is bfloat16 for float32 gradients");
  if (wire == dtype) wire = -1;
  std::lock_guard<std::mutex> lock(mu_);
  queue_.push_back(Task{bucket, false, reinterpret_cast<char*>(base) + offset_elems * esz, count, dtype, wire,
                        seq_++});
  cv_.notify_all();
}

void HostBackend::record_done(int bucket) {
  std::lock_guard<std::mutex> lock(mu_);
  issued_.at((size_t)bucket)++;
  queue_.push_back(Task{bucket, true, nullptr, 0, 0, -1, 0});
  cv_.notify_all();
}

bool HostBackend::query_done(int bucket) {
  std::lock_guard<std::mutex> lock(mu_);
  return done_.at((size_t)bucket) >= issued_.at((size_t)bucket);
}

void HostBackend::wait_done(int bucket, uintptr_t) {
  std::unique_lock<std::mutex> lock(mu_);
  cv_.wait(lock, [&] { return done_.at((size_t)bucket) >= issued_.at((size_t)bucket) || aborted_ || stop_; });
  if (!error_.empty() && !aborted_) throw std::runtime_error("host all-reduce failed: " + error_);
}

std::string HostBackend::async_error() {
  std::lock_guard<std::mutex> lock(mu_);
  return error_;
}

void HostBackend::abort() {
  {
    std::lock_guard<std::mutex> lock(mu_);
    aborted_ = true;
  }
  client_.interrupt();     // the worker's blocked get fails within one poll period
  cv_.notify_all();
}

void HostBackend::reduce(const Task& t) {
  if (world_ == 1) return;
  const bool narrow = t.wire == 1 && t.dtype == 0;      // fp32 storage, bf16 on the wire
  const int wdt = narrow ? 1 : t.dtype;
  const size_t esz = dtype_bytes(wdt);
  const std::string base = "hr/" + tag_ + "/" + std::to_string(t.seq);
  std::string mine;
  if (narrow) {
    mine.resize(t.count * 2);
    const float* src = static_cast<const float*>(t.ptr);
    uint16_t* dst = reinterpret_cast<uint16_t*>(&mine[0]);
    for (size_t i = 0; i < t.count; ++i) dst[i] = float_to_bf16(src[i]);
  } else {
    mine.assign(static_cast<const char*>(t.ptr), t.count * esz);
  }
  client_.set(base + "/" + std::to_string(rank_), mine);
  std::vector<double> acc(t.count, 0.0);
  for (int r = 0; r < world_; ++r) {           // rank order on every rank: bit-identical sums
    const std::string part = r == rank_ ? mine : client_.get(base + "/" + std::to_string(r));
    if (part.size() != t.count * esz) throw std::runtime_error("HostBackend: peer sent a different slice size");
    if (wdt == 0) {
      const float* p = reinterpret_cast<const float*>(part.data());
      for (size_t i = 0; i < t.count; ++i) acc[i] += (double)p[i];
    } else {
      const uint16_t* p = reinterpret_cast<const uint16_t*>(part.data());
      for (size_t i = 0; i < t.count; ++i) acc[i] += (double)bf16_to_float(p[i]);
    }
  }
  const double inv = 1.0 / world_;
  if (narrow) {        // the mean is rounded to the wire format once, like the GPU backends' gather phase delivers it
    float* o = static_cast<float*>(t.ptr);
    for (size_t i = 0; i < t.count; ++i) o[i] = bf16_to_float(float_to_bf16((float)(acc[i] * inv)));
  } else if (t.dtype == 0) {
    float* o = static_cast<float*>(t.ptr);
    for (size_t i = 0; i < t.count; ++i) o[i] = (float)(acc[i] * inv);
  } else {
    uint16_t* o = static_cast<uint16_t*>(t.ptr);
    for (size_t i = 0; i < t.count; ++i) o[i] = float_to_bf16((float)(acc[i] * inv));
  }
  if (client_.add(base + "/read", 1) == world_) {   // the last reader cleans up
    for (int r = 0; r < world_; ++r) client_.remove(base + "/" + std::to_string(r));
    client_.remove(base + "/read");
  }
}

void HostBackend::run() {
  for (;;) {
    Task t;
    {
      std::unique_lock<std::mutex> lock(mu_);
      cv_.wait(lock, [&] { return stop_ || !queue_.empty(); });
      if (stop_) return;
      t = queue_.front();
      queue_.pop_front();
      if (t.marker) {
        done_.at((size_t)t.bucket)++;
        if (timing_) t_done_.at((size_t)t.bucket) = std::chrono::steady_clock::now();
        cv_.notify_all();
        continue;
      }
      if (aborted_ || !error_.empty()) continue;    // drain without touching the network
    }
    try {
      reduce(t);
    } catch (const std::exception& e) {
      std::lock_guard<std::mutex> lock(mu_);
      if (error_.empty()) error_ = e.what();
      cv_.notify_all();
    }
  }
}

}  // namespace comm
}  // namespace dpa
