// One-shot peer-memory all-reduce for small messages (SyncBN statistics, packed metrics).
//
// The reference sends these through NCCL one tiny collective at a time: 53 x 2 all-reduces of (2C+1) floats per
// SyncBN step (Apex syncbn, imagenet_ddp_apex.py:146-148) and 3 scalar all-reduces per logging step
// (reduce_tensor, imagenet_ddp_apex.py:562-566).  At <= 16 KB a collective is pure latency: RCCL's launch,
// its proxy hand-shake and the stream hop to the communication stream cost more than moving the bytes.
//
// On an MI355X node every GPU reaches every other GPU's HBM directly over xGMI, so for small payloads each rank
//   1. stores its vector into its OWN fine-grained slot and publishes a sequence number (system-scope release),
//   2. polls the 7 peers' sequence numbers (system-scope acquire, bounded by a wall-clock deadline),
//   3. reads the 7 peer slots over xGMI and sums all 8 vectors in rank order (bit-identical on every rank,
//      deterministic run to run).
// One kernel, one workgroup, on the CALLER's stream: no communication stream, no events, no host involvement.
// Slots are double buffered by sequence parity: a rank can only pass step 2 of call s+1 after every peer has
// entered call s+1, i.e. finished reading call s, so slot (s & 1) is free again when call s+2 overwrites it.
//
// Peer buffers are exchanged as hipIpc handles through the rendezvous store and mapped with hipIpcOpenMemHandle.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstring>
#include <vector>

#include "../common.h"
#include "comm.h"

namespace py = pybind11;

namespace {

constexpr int kMaxRanks = 16;
constexpr int kThreads = 512;
constexpr int kHeaderBytes = 256;       // flags live in their own 256-byte line in front of the slots

struct PeerTable {
  const uint32_t* base[kMaxRanks];      // mapped buffer of every rank (own entry = local allocation)
};

__device__ __forceinline__ float load_sys(const float* p) {
  // system-scope load: never served from a stale line of this GPU's caches
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_SYSTEM));
}

__global__ __launch_bounds__(kThreads) void oneshot_allreduce_kernel(float* __restrict__ data, int n, PeerTable peers,
                                                                     uint32_t* __restrict__ own, int rank, int world,
                                                                     uint32_t seq, long cap_floats,
                                                                     long long deadline_ticks,
                                                                     uint32_t* __restrict__ status) {
  __shared__ int timed_out;
  const int tid = threadIdx.x;
  const int par = (int)(seq & 1u);
  float* own_slot = reinterpret_cast<float*>(reinterpret_cast<char*>(own) + kHeaderBytes) + (long)par * cap_floats;
  if (tid == 0) timed_out = 0;
  for (int i = tid; i < n; i += kThreads) own_slot[i] = data[i];
  __threadfence_system();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(own + par, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);

  if (tid < world && tid != rank) {
    const uint32_t* flag = peers.base[tid] + par;
    const long long t0 = wall_clock64();
    // sequence numbers only grow; the signed difference keeps the comparison valid across wrap-around
    while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
      __builtin_amdgcn_s_sleep(16);
      if (wall_clock64() - t0 > deadline_ticks) {       // a peer died or never made the call: give up, never hang
        timed_out = 1;
        break;
      }
    }
  }
  __syncthreads();
  if (timed_out) {
    // The sum never happened.  Poison the output so that whatever consumes it next (the loss, the scaler's finite
    // check, a checkpointed running statistic) shows it, and raise the host-visible status word: the host checks
    // it at every later all_reduce(), optimizer step, barrier and checkpoint (NativeComm.check_health).  This
    // object must not be used again -- the ranks' sequence numbers no longer agree.
    for (int i = tid; i < n; i += kThreads) data[i] = __uint_as_float(0x7fc00000u);
    if (tid == 0) __hip_atomic_store(status, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  __threadfence_system();
  for (int i = tid; i < n; i += kThreads) {
    float acc = 0.f;
    for (int r = 0; r < world; ++r) {
      const float* slot = reinterpret_cast<const float*>(reinterpret_cast<const char*>(peers.base[r]) + kHeaderBytes) +
                          (long)par * cap_floats;
      acc += (r == rank) ? own_slot[i] : load_sys(slot + i);
    }
    data[i] = acc;
  }
}

class PeerAllReduce {
 public:
  PeerAllReduce(int rank, int world, int device, long cap_floats, double timeout_s)
      : rank_(rank), world_(world), device_(device), cap_(cap_floats) {
    if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world)
      throw std::invalid_argument("PeerAllReduce: world size must be 1..16");
    if (cap_floats < 1 || cap_floats > (1 << 22)) throw std::invalid_argument("PeerAllReduce: bad capacity");
    DPA_CHECK_HIP(hipSetDevice(device_));
    bytes_ = kHeaderBytes + 2 * (size_t)cap_ * sizeof(float);
    // fine-grained: stores become visible to peers while the kernel runs (coarse-grained memory is only
    // coherent at kernel boundaries)
    DPA_CHECK_HIP(hipExtMallocWithFlags(&local_, bytes_, hipDeviceMallocFinegrained));
    DPA_CHECK_HIP(hipMemset(local_, 0, bytes_));
    DPA_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&status_), 64, hipHostMallocMapped));
    *status_ = 0;
    DPA_CHECK_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&status_dev_), status_, 0));
    DPA_CHECK_HIP(hipDeviceSynchronize());
    for (int r = 0; r < kMaxRanks; ++r) table_.base[r] = nullptr;
    table_.base[rank_] = static_cast<const uint32_t*>(local_);
    deadline_ticks_ = (long long)(timeout_s * 1e8);       // wall_clock64 ticks at 100 MHz
    opened_ = world_ == 1;
  }

  ~PeerAllReduce() {
    (void)hipSetDevice(device_);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < world_; ++r)
      if (r != rank_ && mapped_[r] != nullptr) (void)hipIpcCloseMemHandle(mapped_[r]);
    if (local_) (void)hipFree(local_);
    if (status_) (void)hipHostFree(status_);
  }

  py::bytes handle() const {
    hipIpcMemHandle_t h;
    DPA_CHECK_HIP(hipIpcGetMemHandle(&h, local_));
    return py::bytes(reinterpret_cast<const char*>(&h), sizeof(h));
  }

  // handles[r] = handle() of rank r (the own entry is ignored)
  void open_peers(const std::vector<std::string>& handles) {
    if ((int)handles.size() != world_) throw std::invalid_argument("PeerAllReduce: need one handle per rank");
    DPA_CHECK_HIP(hipSetDevice(device_));
    for (int r = 0; r < world_; ++r) {
      if (r == rank_) continue;
      if (handles[r].size() != sizeof(hipIpcMemHandle_t)) throw std::invalid_argument("PeerAllReduce: bad handle");
      hipIpcMemHandle_t h;
      std::memcpy(&h, handles[r].data(), sizeof(h));
      DPA_CHECK_HIP(hipIpcOpenMemHandle(&mapped_[r], h, hipIpcMemLazyEnablePeerAccess));
      table_.base[r] = static_cast<const uint32_t*>(mapped_[r]);
    }
    opened_ = true;
  }

  void all_reduce(uintptr_t data, long n, uintptr_t stream) {
    if (!opened_) throw std::runtime_error("PeerAllReduce: peers not opened");
    if (n < 0 || n > cap_) throw std::invalid_argument("PeerAllReduce: message larger than the slot");
    if (*status_ != 0) throw std::runtime_error("PeerAllReduce: a peer did not arrive before the deadline");
    if (n == 0 || world_ == 1) return;
    ++seq_;
    hipLaunchKernelGGL(oneshot_allreduce_kernel, dim3(1), dim3(kThreads), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<float*>(data), (int)n, table_, static_cast<uint32_t*>(local_), rank_, world_,
                       seq_, cap_, deadline_ticks_, status_dev_);
    DPA_CHECK_HIP(hipGetLastError());
  }

  uint32_t status() const { return *status_; }
  long capacity() const { return cap_; }

 private:
  int rank_, world_, device_;
  long cap_;
  size_t bytes_ = 0;
  void* local_ = nullptr;
  void* mapped_[kMaxRanks] = {};
  PeerTable table_;
  uint32_t seq_ = 0;
  uint32_t* status_ = nullptr;
  uint32_t* status_dev_ = nullptr;
  long long deadline_ticks_ = 0;
  bool opened_ = false;
};

}  // namespace

void bind_p2p(py::module_& m) {
  py::class_<PeerAllReduce>(m, "PeerAllReduce")
      .def(py::init<int, int, int, long, double>(), py::arg("rank"), py::arg("world_size"), py::arg("device"),
           py::arg("capacity_floats") = 16384, py::arg("timeout_s") = 30.0)
      .def("handle", &PeerAllReduce::handle)
      .def("open_peers", &PeerAllReduce::open_peers)
      .def("all_reduce", &PeerAllReduce::all_reduce, py::arg("ptr"), py::arg("count"), py::arg("stream"))
      .def("status", &PeerAllReduce::status)
      .def_property_readonly("capacity", &PeerAllReduce::capacity);
}
