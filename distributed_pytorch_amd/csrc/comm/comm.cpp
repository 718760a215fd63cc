// Native data-parallel communication runtime for one-process-per-GPU training on MI355X.
//
//  * Communicator: owns an RCCL communicator (ncclCommInitRank from a unique id that Python
//    exchanges through the rendezvous store) and a dedicated HIP stream, so
//    collectives run beside compute instead of in its queue.
//  * BucketReducer: the DDP gradient reducer.  Gradients live as VIEWS inside a few large flat
//    buckets (no flatten / unflatten copies).  When backward has produced every gradient of a
//    bucket, the compute stream records an event, the comm stream waits on it and issues ONE
//    in-place ncclAllReduce(avg) for the whole bucket while backward keeps running; buckets are
//    always issued in index order so every rank enqueues the same collective sequence.
//    finalize() makes the compute stream wait for the buckets' completion events -- no host
//    synchronisation anywhere, so the whole step stays asynchronous.
//
// xGMI sizing: 8 GPUs are fully connected with 7 links per GPU; RCCL's collectives are per-link
// bandwidth bound and have a fixed launch cost, so buckets default to 64 MiB (vs the 25 MiB
// PCIe/IB-era default of c10d) -- ResNet-50's 102 MB of fp32 gradients travel as two collectives.
//
// RCCL is bound at run time with dlopen("librccl.so.1") so the extension shares the copy torch
// already loaded (one RCCL / one HIP runtime per process) and CPU-only hosts can import it.
//
// Replaces c10d ProcessGroupNCCL + Reducer and Apex DistributedDataParallel (reference
// nd_imagenet.py:130-154, imagenet_ddp.py:104-127, imagenet_ddp_apex.py:124-183)
// -- SURVEY.md N-C1..N-C5.
#include "comm.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <pybind11/stl.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../kernels/kernels.h"
#include "host_backend.h"
#include "reducer_core.h"

namespace py = pybind11;
using dpa::comm::BucketReducerCore;
using dpa::comm::HostBackend;
using dpa::comm::OnTimeout;
using dpa::comm::ReduceBackend;
using dpa::comm::StreamCollective;
using dpa::comm::dtype_bytes;

namespace {

#define CHECK_HIP(expr)                                                                  \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclReduceScatter) ReduceScatter = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;

  template <typename F> void sym(F& f, const char* name) {
    f = reinterpret_cast<F>(dlsym(handle, name));
    if (!f) throw std::runtime_error(std::string("RCCL symbol missing: ") + name);
  }
  void load() {
    if (handle) return;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (handle) break;
    }
    if (!handle) throw std::runtime_error(std::string("cannot load RCCL: ") + dlerror());
    sym(GetUniqueId, "ncclGetUniqueId");
    sym(CommInitRank, "ncclCommInitRank");
    sym(CommDestroy, "ncclCommDestroy");
    sym(CommAbort, "ncclCommAbort");
    sym(AllReduce, "ncclAllReduce");
    sym(Broadcast, "ncclBroadcast");
    sym(AllGather, "ncclAllGather");
    sym(ReduceScatter, "ncclReduceScatter");
    sym(GroupStart, "ncclGroupStart");
    sym(GroupEnd, "ncclGroupEnd");
    sym(GetErrorString, "ncclGetErrorString");
    sym(GetVersion, "ncclGetVersion");
    sym(CommGetAsyncError, "ncclCommGetAsyncError");
  }
};
Rccl& rccl() {
  static Rccl r;
  r.load();
  return r;
}

void check_nccl(ncclResult_t r, const char* what) {
  if (r != ncclSuccess && r != ncclInProgress)
    throw std::runtime_error(std::string(what) + ": " + rccl().GetErrorString(r));
}

ncclDataType_t nccl_dtype(int code) {  // dpa::DType codes
  switch (code) {
    case 0: return ncclFloat32;
    case 1: return ncclBfloat16;
    case 2: return ncclFloat16;
    case 3: return ncclInt64;
    case 4: return ncclUint8;
    default: throw std::invalid_argument("unsupported comm dtype");
  }
}
class Communicator : public StreamCollective {
 public:
  Communicator(py::bytes id, int rank, int world, int device) : rank_(rank), world_(world), device_(device) {
    std::string s = id;
    if (s.size() != sizeof(ncclUniqueId)) throw std::invalid_argument("bad unique id size");
    ncclUniqueId uid;
    memcpy(&uid, s.data(), sizeof(uid));
    CHECK_HIP(hipSetDevice(device));
    {
      py::gil_scoped_release nogil;
      ncclComm_t c = nullptr;
      check_nccl(rccl().CommInitRank(&c, world, uid, rank), "ncclCommInitRank");
      comm_.store(c);
    }
    // Normal priority by default.  Cause of the round-1 deadlock, established in round 2
    // (profiles/r2/priority_stream_hw_queues.txt): a high-priority stream needs a hardware queue of its own
    // priority class; with GPU_MAX_HW_QUEUES=4 (the pool's setting, HIP's default) and this job's default, copy
    // and RCCL streams already on the normal queues it ends up sharing a hardware queue with a stream it
    // event-waits on, and the cyclic wait compute -> comm -> compute never drains.  The identical run with
    // GPU_MAX_HW_QUEUES=8 completes (12.2k img/s, forced-DDP ResNet-50).  DPA_COMM_PRIORITY=1 therefore takes
    // effect only when the process was started with at least 8 hardware queues; otherwise it is refused loudly.
    const char* prio = getenv("DPA_COMM_PRIORITY");
    bool want_prio = prio && std::string(prio) == "1";
    if (want_prio) {
      const char* q = getenv("GPU_MAX_HW_QUEUES");
      const int queues = q ? atoi(q) : 4;
      if (queues < 8) {
        fprintf(stderr, "[dpa comm] DPA_COMM_PRIORITY=1 ignored: GPU_MAX_HW_QUEUES=%d < 8 would alias the priority "
                        "stream onto a queue it waits on (deadlock); using a normal-priority stream\n", queues);
        want_prio = false;
      }
    }
    if (want_prio) {
      int lo = 0, hi = 0;
      CHECK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
      CHECK_HIP(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, hi));
    } else {
      CHECK_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    }
  }
  ~Communicator() override {
    ncclComm_t c = comm_.exchange(nullptr);
    if (c) rccl().CommDestroy(c);
    if (stream_) (void)hipStreamDestroy(stream_);
    if (staging_) (void)hipFree(staging_);
  }

  // ---- StreamCollective: what the reducer's backend calls ------------------------------------------------------
  // A narrowed wire format (fp32 gradients travelling as bf16 / fp16, what Apex O2 does with its fp16 gradients,
  // imagenet_ddp_apex.py:169-183) is three enqueues on the communicator's stream: cast into the staging buffer,
  // ncclAllReduce(avg) of the 16-bit values, cast back into the fp32 bucket.  The casts stream at HBM rate and sit
  // behind backward like the collective itself; the links carry half the bytes.
  void reserve_wire(size_t max_count, int dtype, int wire) override {
    if (wire < 0 || wire == dtype) return;
    const size_t need = max_count * dtype_bytes(wire);
    if (need <= staging_bytes_) return;
    CHECK_HIP(hipSetDevice(device_));
    CHECK_HIP(hipStreamSynchronize(stream_));
    if (staging_) CHECK_HIP(hipFree(staging_));
    staging_ = nullptr;
    CHECK_HIP(hipMalloc(&staging_, need));
    staging_bytes_ = need;
  }
  void reduce_mean(uintptr_t p, size_t count, int dtype, int wire) override {
    if (wire < 0 || wire == dtype) {
      all_reduce(p, count, dtype, true, 0);
      return;
    }
    if (count * dtype_bytes(wire) > staging_bytes_)
      throw std::runtime_error("RCCL wire staging smaller than the slice (reserve_wire was not called)");
    dpa::cast_launch(reinterpret_cast<const void*>(p), dtype, staging_, wire, (long)count, stream_);
    all_reduce(reinterpret_cast<uintptr_t>(staging_), count, wire, true, 0);
    dpa::cast_launch(staging_, wire, reinterpret_cast<void*>(p), dtype, (long)count, stream_);
  }
  static py::bytes unique_id() {
    ncclUniqueId uid;
    check_nccl(rccl().GetUniqueId(&uid), "ncclGetUniqueId");
    return py::bytes(reinterpret_cast<const char*>(&uid), sizeof(uid));
  }
  static int version() {
    int v = 0;
    check_nccl(rccl().GetVersion(&v), "ncclGetVersion");
    return v;
  }
  hipStream_t pick(uintptr_t s) const { return s ? reinterpret_cast<hipStream_t>(s) : stream_; }
  ncclComm_t live() const {
    ncclComm_t c = comm_.load();
    if (!c) throw std::runtime_error("RCCL communicator was aborted");
    return c;
  }

  // Every enqueue holds `mu_` from the moment it loads the handle until RCCL has returned, and abort() takes the
  // same lock before it invalidates the handle: the watchdog thread cannot destroy a communicator the training
  // thread is about to pass to ncclAllReduce (c10d's ProcessGroupNCCL guards its communicators the same way).
  // The lock is timed on the abort side only -- an enqueue wedged inside RCCL must not keep the abort out.
  void all_reduce(uintptr_t p, size_t count, int dtype, bool avg, uintptr_t s) {
    std::lock_guard<std::timed_mutex> guard(mu_);
    check_nccl(rccl().AllReduce(reinterpret_cast<void*>(p), reinterpret_cast<void*>(p), count, nccl_dtype(dtype),
                                avg ? ncclAvg : ncclSum, live(), pick(s)),
               "ncclAllReduce");
  }
  void all_reduce_max(uintptr_t p, size_t count, int dtype, uintptr_t s) {
    std::lock_guard<std::timed_mutex> guard(mu_);
    check_nccl(rccl().AllReduce(reinterpret_cast<void*>(p), reinterpret_cast<void*>(p), count, nccl_dtype(dtype),
                                ncclMax, live(), pick(s)),
               "ncclAllReduce");
  }
  void broadcast(uintptr_t p, size_t nbytes, int root, uintptr_t s) {
    std::lock_guard<std::timed_mutex> guard(mu_);
    check_nccl(rccl().Broadcast(reinterpret_cast<void*>(p), reinterpret_cast<void*>(p), nbytes, ncclUint8, root, live(),
                                pick(s)),
               "ncclBroadcast");
  }
  void all_gather(uintptr_t send, uintptr_t recv, size_t count, int dtype, uintptr_t s) {
    std::lock_guard<std::timed_mutex> guard(mu_);
    check_nccl(rccl().AllGather(reinterpret_cast<void*>(send), reinterpret_cast<void*>(recv), count, nccl_dtype(dtype),
                                live(), pick(s)),
               "ncclAllGather");
  }
  // ncclCommAbort: kernels of this communicator that are spinning on a dead peer exit, the streams drain.
  // Callable from any thread (the reducer's watchdog); the communicator is unusable afterwards.
  void abort() override {
    std::unique_lock<std::timed_mutex> guard(mu_, std::chrono::seconds(2));   // proceeds unlocked after 2 s
    ncclComm_t c = comm_.exchange(nullptr);
    if (c) rccl().CommAbort(c);
  }
  bool aborted() const override { return comm_.load() == nullptr; }
  std::string async_error() override {
    ncclComm_t c = comm_.load();
    if (!c) return "communicator aborted";
    ncclResult_t r = ncclSuccess;
    rccl().CommGetAsyncError(c, &r);
    return (r == ncclSuccess || r == ncclInProgress) ? std::string() : std::string(rccl().GetErrorString(r));
  }
  // order `consumer` after everything enqueued so far on the comm stream (and vice versa)
  void comm_waits_for(uintptr_t producer) {
    hipEvent_t e = scratch_event();
    CHECK_HIP(hipEventRecord(e, reinterpret_cast<hipStream_t>(producer)));
    CHECK_HIP(hipStreamWaitEvent(stream_, e, 0));
  }
  void wait_comm(uintptr_t consumer) {
    hipEvent_t e = scratch_event();
    CHECK_HIP(hipEventRecord(e, stream_));
    CHECK_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(consumer), e, 0));
  }
  void synchronize() {
    py::gil_scoped_release nogil;
    CHECK_HIP(hipStreamSynchronize(stream_));
  }
  int rank() const { return rank_; }
  int world() const override { return world_; }
  int device() const override { return device_; }
  uintptr_t stream_handle() const { return reinterpret_cast<uintptr_t>(stream_); }
  hipStream_t stream() const override { return stream_; }

 private:
  hipEvent_t scratch_event() {
    if (events_.size() < 8) {
      hipEvent_t e;
      CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      events_.push_back(e);
      return e;
    }
    hipEvent_t e = events_[next_event_++ % events_.size()];
    return e;
  }
  std::atomic<ncclComm_t> comm_{nullptr};
  std::timed_mutex mu_;
  hipStream_t stream_ = nullptr;
  int rank_, world_, device_;
  std::vector<hipEvent_t> events_;
  size_t next_event_ = 0;
  void* staging_ = nullptr;         // 16-bit wire copy of the slice in flight (one at a time: one stream)
  size_t staging_bytes_ = 0;
};

// The reducer's communication queue on a GPU: the communicator's HIP stream, HIP events as ready / done points,
// one in-place mean all-reduce per slice (ncclAllReduce on RCCL, the two-shot window kernels on the peer
// communicator).  No host synchronisation anywhere -- the step stays asynchronous.
class StreamBackend : public ReduceBackend {
 public:
  explicit StreamBackend(std::shared_ptr<StreamCollective> comm) : comm_(std::move(comm)) {}
  ~StreamBackend() override { drop_events(); }
  int world() const override { return comm_->world(); }
  void resize(int n) override {
    const unsigned flags = timing_ ? hipEventDefault : hipEventDisableTiming;
    while ((int)ready_.size() < n) {
      hipEvent_t r, d;
      CHECK_HIP(hipEventCreateWithFlags(&r, flags));
      CHECK_HIP(hipEventCreateWithFlags(&d, flags));
      ready_.push_back(r);
      done_.push_back(d);
    }
  }
  // Timing swaps every event for one that carries a time stamp (and back).  Called between steps only: nothing
  // recorded on the old events is waited for any more once the compute stream has joined the last bucket.
  bool set_timing(bool on) override {
    if (on == timing_) return true;
    CHECK_HIP(hipSetDevice(comm_->device()));
    CHECK_HIP(hipStreamSynchronize(comm_->stream()));
    const int n = (int)ready_.size();
    drop_events();
    timing_ = on;
    resize(n);
    if (on) CHECK_HIP(hipEventCreateWithFlags(&tail_, hipEventDefault));
    return true;
  }
  void mark_tail(uintptr_t compute_stream) override {
    if (timing_ && tail_) CHECK_HIP(hipEventRecord(tail_, reinterpret_cast<hipStream_t>(compute_stream)));
  }
  std::vector<double> bucket_timings(int n) override {
    std::vector<double> out;
    if (!timing_ || !tail_ || n <= 0 || n > (int)done_.size()) return out;
    CHECK_HIP(hipEventSynchronize(done_.at(n - 1)));
    CHECK_HIP(hipEventSynchronize(tail_));
    for (int b = 0; b < n; ++b) {
      float busy = 0.f, exposed = 0.f;
      if (hipEventElapsedTime(&busy, ready_.at(b), done_.at(b)) != hipSuccess) busy = -1.f;
      // HIP reports stop-before-start as an error or as a negative span depending on the runtime: try both orders
      if (hipEventElapsedTime(&exposed, tail_, done_.at(b)) != hipSuccess) {
        float back = 0.f;
        exposed = hipEventElapsedTime(&back, done_.at(b), tail_) == hipSuccess ? -back : 0.f;
      }
      (void)hipGetLastError();
      out.push_back(busy);
      out.push_back(exposed);
    }
    return out;
  }
  void record_ready(int b, uintptr_t compute_stream) override {
    CHECK_HIP(hipEventRecord(ready_.at(b), reinterpret_cast<hipStream_t>(compute_stream)));
    CHECK_HIP(hipStreamWaitEvent(comm_->stream(), ready_.at(b), 0));
  }
  void launch(int, uintptr_t base, size_t offset_elems, size_t count, int dtype, int wire) override {
    comm_->reduce_mean(base + offset_elems * dtype_bytes(dtype), count, dtype, wire);
  }
  void reserve_wire(size_t max_count, int dtype, int wire) { comm_->reserve_wire(max_count, dtype, wire); }
  void record_done(int b) override { CHECK_HIP(hipEventRecord(done_.at(b), comm_->stream())); }
  void wait_done(int b, uintptr_t compute_stream) override {
    CHECK_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(compute_stream), done_.at(b), 0));
  }
  bool query_done(int b) override {
    thread_local int bound_device = -1;
    if (bound_device != comm_->device()) {
      (void)hipSetDevice(comm_->device());
      bound_device = comm_->device();
    }
    return hipEventQuery(done_.at(b)) != hipErrorNotReady;
  }
  std::string async_error() override { return comm_->aborted() ? std::string() : comm_->async_error(); }
  void abort() override { comm_->abort(); }

 private:
  void drop_events() {
    for (hipEvent_t e : ready_) (void)hipEventDestroy(e);
    for (hipEvent_t e : done_) (void)hipEventDestroy(e);
    if (tail_) (void)hipEventDestroy(tail_);
    ready_.clear();
    done_.clear();
    tail_ = nullptr;
  }
  std::shared_ptr<StreamCollective> comm_;
  std::vector<hipEvent_t> ready_, done_;
  hipEvent_t tail_ = nullptr;
  bool timing_ = false;
};

// Python handle of the reducer (either backend).  `stream` arguments are hipStream_t integers (0 on the host).
class PyReducer {
 public:
  PyReducer(std::shared_ptr<ReduceBackend> backend, std::vector<uintptr_t> ptrs, std::vector<size_t> counts,
            std::vector<int> dtypes, std::vector<int> param_bucket, size_t chunk_bytes, double timeout_s,
            bool exit_on_timeout, double grace_s, int wire_dtype)
      : core_(reserve(backend, counts, dtypes, chunk_bytes, wire_dtype), std::move(ptrs), std::move(counts),
              std::move(dtypes), std::move(param_bucket), chunk_bytes, timeout_s,
              exit_on_timeout ? OnTimeout::kExitProcess : OnTimeout::kFlagOnly, grace_s, wire_dtype) {}
  // the staging of a narrowed wire format is sized for the largest slice before the first launch
  static std::shared_ptr<ReduceBackend> reserve(std::shared_ptr<ReduceBackend> backend,
                                                const std::vector<size_t>& counts, const std::vector<int>& dtypes,
                                                size_t chunk_bytes, int wire) {
    auto* sb = dynamic_cast<StreamBackend*>(backend.get());
    if (sb && wire >= 0) {
      for (size_t b = 0; b < counts.size() && b < dtypes.size(); ++b) {
        if (dtypes[b] != 0 || wire == dtypes[b]) continue;
        size_t n = counts[b];
        if (chunk_bytes > 0) n = std::min(n, std::max<size_t>(1, chunk_bytes / dtype_bytes(dtypes[b])));
        sb->reserve_wire(n, dtypes[b], wire);
      }
    }
    return backend;
  }
  void prepare() { core_.prepare(); }
  void mark_ready(int idx, uintptr_t s) { core_.mark_ready(idx, s); }
  void flush(uintptr_t s, bool force) { core_.flush(s, force); }
  void finalize(uintptr_t s) {
    py::gil_scoped_release nogil;     // a host backend blocks here until its worker thread has retired the buckets
    core_.finalize(s);
  }
  int num_buckets() const { return core_.num_buckets(); }
  int launched() const { return core_.launched(); }
  long collectives() const { return core_.collectives(); }
  long wire_bytes() const { return core_.wire_bytes(); }
  std::vector<int> observed_order() const { return core_.observed_order(); }
  bool failed() const { return core_.failed(); }
  std::string failure() const { return core_.failure(); }
  void set_paused(bool on) { core_.set_paused(on); }
  bool set_timing(bool on) { return core_.set_timing(on); }
  std::vector<double> bucket_timings() {
    py::gil_scoped_release nogil;
    return core_.bucket_timings();
  }

 private:
  BucketReducerCore core_;
};

}  // namespace

void bind_comm(py::module_& m) {
  m.def("plan_buckets", &dpa::comm::plan_buckets, py::arg("numels"), py::arg("elem_sizes"), py::arg("dtype_keys"),
        py::arg("order"), py::arg("cap_bytes"), py::arg("first_cap_bytes"), py::arg("align_bytes") = 256,
        py::arg("last_cap_bytes") = 0, py::arg("min_bytes") = 1L << 20);
  m.attr("WATCHDOG_EXIT_CODE") = (int)dpa::comm::kWatchdogExitCode;
  m.def("rccl_version", &Communicator::version);
  m.def("rccl_unique_id", &Communicator::unique_id);
  py::class_<StreamCollective, std::shared_ptr<StreamCollective>>(m, "StreamCollective");
  py::class_<Communicator, StreamCollective, std::shared_ptr<Communicator>>(m, "Communicator")
      .def(py::init<py::bytes, int, int, int>(), py::arg("unique_id"), py::arg("rank"), py::arg("world_size"),
           py::arg("device"))
      .def("all_reduce", &Communicator::all_reduce, py::arg("ptr"), py::arg("count"), py::arg("dtype"),
           py::arg("average") = false, py::arg("stream") = 0)
      .def("all_reduce_max", &Communicator::all_reduce_max, py::arg("ptr"), py::arg("count"), py::arg("dtype"),
           py::arg("stream") = 0)
      .def("broadcast", &Communicator::broadcast, py::arg("ptr"), py::arg("nbytes"), py::arg("root") = 0,
           py::arg("stream") = 0)
      .def("all_gather", &Communicator::all_gather, py::arg("send"), py::arg("recv"), py::arg("count"),
           py::arg("dtype"), py::arg("stream") = 0)
      .def("comm_waits_for", &Communicator::comm_waits_for)
      .def("wait_comm", &Communicator::wait_comm)
      .def("synchronize", &Communicator::synchronize)
      .def("abort", &Communicator::abort, py::call_guard<py::gil_scoped_release>())
      .def("async_error", &Communicator::async_error)
      .def_property_readonly("aborted", &Communicator::aborted)
      .def_property_readonly("rank", &Communicator::rank)
      .def_property_readonly("world_size", &Communicator::world)
      .def_property_readonly("stream", &Communicator::stream_handle);
  py::class_<ReduceBackend, std::shared_ptr<ReduceBackend>>(m, "ReduceBackend");
  py::class_<StreamBackend, ReduceBackend, std::shared_ptr<StreamBackend>>(m, "StreamBackend")
      .def(py::init<std::shared_ptr<StreamCollective>>(), py::arg("comm"));
  m.attr("RcclBackend") = m.attr("StreamBackend");
  py::class_<HostBackend, ReduceBackend, std::shared_ptr<HostBackend>>(m, "HostBackend")
      .def(py::init([](const std::string& host, int port, int rank, int world, double timeout_s,
                       const std::string& tag) {
             py::gil_scoped_release nogil;
             return std::make_shared<HostBackend>(host, port, rank, world, timeout_s, tag);
           }),
           py::arg("host"), py::arg("port"), py::arg("rank"), py::arg("world_size"), py::arg("timeout_s") = 300.0,
           py::arg("tag") = "0")
      .def_property_readonly("collectives", &HostBackend::collectives);
  py::class_<PyReducer>(m, "BucketReducer")
      .def(py::init<std::shared_ptr<ReduceBackend>, std::vector<uintptr_t>, std::vector<size_t>, std::vector<int>,
                    std::vector<int>, size_t, double, bool, double, int>(),
           py::arg("backend"), py::arg("bucket_ptrs"), py::arg("bucket_counts"), py::arg("bucket_dtypes"),
           py::arg("param_bucket"), py::arg("chunk_bytes") = (size_t)128 << 20, py::arg("timeout_s") = 0.0,
           py::arg("exit_on_timeout") = true, py::arg("grace_s") = 10.0, py::arg("wire_dtype") = -1)
      .def("prepare", &PyReducer::prepare)
      .def("mark_ready", &PyReducer::mark_ready)
      .def("flush", &PyReducer::flush)
      .def("finalize", &PyReducer::finalize)
      .def("observed_order", &PyReducer::observed_order)
      .def("set_paused", &PyReducer::set_paused)
      .def("set_timing", &PyReducer::set_timing)
      .def("bucket_timings", &PyReducer::bucket_timings)
      .def_property_readonly("num_buckets", &PyReducer::num_buckets)
      .def_property_readonly("launched", &PyReducer::launched)
      .def_property_readonly("collectives", &PyReducer::collectives)
      .def_property_readonly("wire_bytes", &PyReducer::wire_bytes)
      .def_property_readonly("failed", &PyReducer::failed)
      .def_property_readonly("failure", &PyReducer::failure);
}
