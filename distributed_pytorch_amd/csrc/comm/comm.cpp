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
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

namespace py = pybind11;

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
size_t dtype_bytes(int code) {
  switch (code) {
    case 0: return 4;
    case 1: case 2: return 2;
    case 3: return 8;
    default: return 1;
  }
}

class Communicator {
 public:
  Communicator(py::bytes id, int rank, int world, int device) : rank_(rank), world_(world), device_(device) {
    std::string s = id;
    if (s.size() != sizeof(ncclUniqueId)) throw std::invalid_argument("bad unique id size");
    ncclUniqueId uid;
    memcpy(&uid, s.data(), sizeof(uid));
    CHECK_HIP(hipSetDevice(device));
    {
      py::gil_scoped_release nogil;
      check_nccl(rccl().CommInitRank(&comm_, world, uid, rank), "ncclCommInitRank");
    }
    // Normal priority on purpose.  A high-priority stream that event-waits on work of torch's default
    // (null) stream DEADLOCKED on MI355X / ROCm 7.0 runtime with GPU_MAX_HW_QUEUES=4 (both streams stay
    // busy forever, measured 2026-11: profiles/r1_notes.md); DPA_COMM_PRIORITY=1 opts back in.
    const char* prio = getenv("DPA_COMM_PRIORITY");
    if (prio && std::string(prio) == "1") {
      int lo = 0, hi = 0;
      CHECK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
      CHECK_HIP(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, hi));
    } else {
      CHECK_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    }
  }
  ~Communicator() {
    if (comm_) {
      if (aborted_) rccl().CommAbort(comm_);
      else rccl().CommDestroy(comm_);
    }
    if (stream_) (void)hipStreamDestroy(stream_);
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

  void all_reduce(uintptr_t p, size_t count, int dtype, bool avg, uintptr_t s) {
    check_nccl(rccl().AllReduce(reinterpret_cast<void*>(p), reinterpret_cast<void*>(p), count, nccl_dtype(dtype),
                                avg ? ncclAvg : ncclSum, comm_, pick(s)),
               "ncclAllReduce");
  }
  void broadcast(uintptr_t p, size_t nbytes, int root, uintptr_t s) {
    check_nccl(rccl().Broadcast(reinterpret_cast<void*>(p), reinterpret_cast<void*>(p), nbytes, ncclUint8, root, comm_,
                                pick(s)),
               "ncclBroadcast");
  }
  void all_gather(uintptr_t send, uintptr_t recv, size_t count, int dtype, uintptr_t s) {
    check_nccl(rccl().AllGather(reinterpret_cast<void*>(send), reinterpret_cast<void*>(recv), count, nccl_dtype(dtype),
                                comm_, pick(s)),
               "ncclAllGather");
  }
  void abort() { aborted_ = true; }
  std::string async_error() {
    ncclResult_t r = ncclSuccess;
    rccl().CommGetAsyncError(comm_, &r);
    return r == ncclSuccess ? std::string() : std::string(rccl().GetErrorString(r));
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
  int world() const { return world_; }
  uintptr_t stream_handle() const { return reinterpret_cast<uintptr_t>(stream_); }
  ncclComm_t raw() const { return comm_; }
  hipStream_t stream() const { return stream_; }

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
  ncclComm_t comm_ = nullptr;
  hipStream_t stream_ = nullptr;
  int rank_, world_, device_;
  bool aborted_ = false;
  std::vector<hipEvent_t> events_;
  size_t next_event_ = 0;
};

class BucketReducer {
 public:
  struct Bucket {
    uintptr_t ptr;
    size_t count;
    int dtype;
    int nparams;
    int pending;
    bool launched;
    hipEvent_t ready, done;
  };
  BucketReducer(std::shared_ptr<Communicator> comm, std::vector<uintptr_t> ptrs, std::vector<size_t> counts,
                std::vector<int> dtypes, std::vector<int> param_bucket)
      : comm_(std::move(comm)), param_bucket_(std::move(param_bucket)) {
    if (ptrs.size() != counts.size() || ptrs.size() != dtypes.size()) throw std::invalid_argument("bucket arrays");
    buckets_.resize(ptrs.size());
    for (size_t b = 0; b < ptrs.size(); ++b) {
      Bucket& k = buckets_[b];
      k.ptr = ptrs[b]; k.count = counts[b]; k.dtype = dtypes[b]; k.nparams = 0; k.pending = 0; k.launched = false;
      CHECK_HIP(hipEventCreateWithFlags(&k.ready, hipEventDisableTiming));
      CHECK_HIP(hipEventCreateWithFlags(&k.done, hipEventDisableTiming));
    }
    for (int b : param_bucket_) {
      if (b < 0 || b >= (int)buckets_.size()) throw std::invalid_argument("param -> bucket index out of range");
      buckets_[b].nparams++;
    }
    prepare();
  }
  ~BucketReducer() {
    for (Bucket& k : buckets_) {
      (void)hipEventDestroy(k.ready);
      (void)hipEventDestroy(k.done);
    }
  }
  void prepare() {
    for (Bucket& k : buckets_) {
      k.pending = k.nparams;
      k.launched = false;
    }
    next_ = 0;
    launched_ = 0;
  }
  // gradient of parameter `idx` has been enqueued on `compute_stream`
  void mark_ready(int idx, uintptr_t compute_stream) {
    Bucket& k = buckets_.at(param_bucket_.at(idx));
    if (k.pending <= 0) return;  // parameter used twice: already counted
    if (--k.pending == 0) {
      CHECK_HIP(hipEventRecord(k.ready, reinterpret_cast<hipStream_t>(compute_stream)));
      flush(compute_stream, false);
    }
  }
  // issue, in index order, every bucket whose gradients are complete (all of them when `force`)
  void flush(uintptr_t compute_stream, bool force) {
    while (next_ < buckets_.size()) {
      Bucket& k = buckets_[next_];
      if (k.pending > 0) {
        if (!force) break;
        k.pending = 0;  // unused parameters: their (zeroed) gradient slots are reduced as they are
        CHECK_HIP(hipEventRecord(k.ready, reinterpret_cast<hipStream_t>(compute_stream)));
      }
      CHECK_HIP(hipStreamWaitEvent(comm_->stream(), k.ready, 0));
      comm_->all_reduce(k.ptr, k.count, k.dtype, true, 0);
      CHECK_HIP(hipEventRecord(k.done, comm_->stream()));
      k.launched = true;
      ++launched_;
      ++next_;
    }
  }
  // end of backward: issue the stragglers and make the compute stream wait for every bucket
  void finalize(uintptr_t compute_stream) {
    flush(compute_stream, true);
    for (Bucket& k : buckets_) CHECK_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(compute_stream), k.done, 0));
  }
  int num_buckets() const { return (int)buckets_.size(); }
  int launched() const { return launched_; }

 private:
  std::shared_ptr<Communicator> comm_;
  std::vector<Bucket> buckets_;
  std::vector<int> param_bucket_;
  size_t next_ = 0;
  int launched_ = 0;
};


// ---- bucket planner (host logic of the reducer) ----------------------------------------------------
// Packs parameters, given in expected gradient-ready order, into dtype-homogeneous flat buckets.  The first
// bucket of each dtype is small (its all-reduce starts early), the parameters that become ready LAST form a
// small bucket of their own (its all-reduce is the only one that cannot overlap with backward), everything
// in between fills `cap_bytes` buckets sized for xGMI.  Every view offset is aligned to `align_bytes`.
struct PlannedBucket {
  long dtype_key = 0;
  std::vector<long> pids, offsets;
  long total = 0;     // elements, padded
};

void plan_range(const std::vector<long>& numels, const std::vector<long>& esz, const std::vector<long>& dkeys,
                const std::vector<long>& order, size_t begin, size_t end, long cap, long first_cap, long align,
                std::vector<PlannedBucket>& out) {
  std::vector<std::pair<long, int>> open;       // dtype -> index into out (few dtypes: linear search)
  std::vector<long> first_done;
  auto find_open = [&](long dk) -> int* {
    for (auto& kv : open)
      if (kv.first == dk) return &kv.second;
    return nullptr;
  };
  auto done = [&](long dk) {
    for (long d : first_done)
      if (d == dk) return true;
    return false;
  };
  for (size_t i = begin; i < end; ++i) {
    const long pid = order[i];
    const long dk = dkeys[pid], es = esz[pid], n = numels[pid];
    int* cur = find_open(dk);
    const long limit = done(dk) ? cap : first_cap;
    if (cur != nullptr && *cur >= 0 && !out[*cur].pids.empty() && out[*cur].total * es + n * es > limit) {
      if (!done(dk)) first_done.push_back(dk);
      *cur = -1;
    }
    if (cur == nullptr) {
      open.emplace_back(dk, -1);
      cur = &open.back().second;
    }
    if (*cur < 0) {
      out.emplace_back();
      out.back().dtype_key = dk;
      *cur = (int)out.size() - 1;
    }
    PlannedBucket& b = out[*cur];
    b.pids.push_back(pid);
    b.offsets.push_back(b.total);
    const long align_el = std::max<long>(1, align / es);
    b.total += (n + align_el - 1) / align_el * align_el;
  }
}

std::vector<std::tuple<long, std::vector<long>, std::vector<long>, long>> plan_buckets(
    const std::vector<long>& numels, const std::vector<long>& esz, const std::vector<long>& dkeys,
    const std::vector<long>& order, long cap, long first_cap, long align, long last_cap) {
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
  plan_range(numels, esz, dkeys, order, 0, split, cap, first_cap, align, out);
  const long inf = 1L << 62;
  plan_range(numels, esz, dkeys, order, split, order.size(), inf, inf, align, out);
  std::vector<std::tuple<long, std::vector<long>, std::vector<long>, long>> res;
  res.reserve(out.size());
  for (auto& b : out) res.emplace_back(b.dtype_key, std::move(b.pids), std::move(b.offsets), b.total);
  return res;
}

}  // namespace

void bind_comm(py::module_& m) {
  m.def("plan_buckets", &plan_buckets, py::arg("numels"), py::arg("elem_sizes"), py::arg("dtype_keys"),
        py::arg("order"), py::arg("cap_bytes"), py::arg("first_cap_bytes"), py::arg("align_bytes") = 256,
        py::arg("last_cap_bytes") = 0);
  m.def("rccl_version", &Communicator::version);
  m.def("rccl_unique_id", &Communicator::unique_id);
  py::class_<Communicator, std::shared_ptr<Communicator>>(m, "Communicator")
      .def(py::init<py::bytes, int, int, int>(), py::arg("unique_id"), py::arg("rank"), py::arg("world_size"),
           py::arg("device"))
      .def("all_reduce", &Communicator::all_reduce, py::arg("ptr"), py::arg("count"), py::arg("dtype"),
           py::arg("average") = false, py::arg("stream") = 0)
      .def("broadcast", &Communicator::broadcast, py::arg("ptr"), py::arg("nbytes"), py::arg("root") = 0,
           py::arg("stream") = 0)
      .def("all_gather", &Communicator::all_gather, py::arg("send"), py::arg("recv"), py::arg("count"),
           py::arg("dtype"), py::arg("stream") = 0)
      .def("comm_waits_for", &Communicator::comm_waits_for)
      .def("wait_comm", &Communicator::wait_comm)
      .def("synchronize", &Communicator::synchronize)
      .def("abort", &Communicator::abort)
      .def("async_error", &Communicator::async_error)
      .def_property_readonly("rank", &Communicator::rank)
      .def_property_readonly("world_size", &Communicator::world)
      .def_property_readonly("stream", &Communicator::stream_handle);
  py::class_<BucketReducer>(m, "BucketReducer")
      .def(py::init<std::shared_ptr<Communicator>, std::vector<uintptr_t>, std::vector<size_t>, std::vector<int>,
                    std::vector<int>>(),
           py::arg("comm"), py::arg("bucket_ptrs"), py::arg("bucket_counts"), py::arg("bucket_dtypes"),
           py::arg("param_bucket"))
      .def("prepare", &BucketReducer::prepare)
      .def("mark_ready", &BucketReducer::mark_ready)
      .def("flush", &BucketReducer::flush)
      .def("finalize", &BucketReducer::finalize)
      .def_property_readonly("num_buckets", &BucketReducer::num_buckets)
      .def_property_readonly("launched", &BucketReducer::launched);
}
