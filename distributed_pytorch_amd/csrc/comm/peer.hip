// Peer-memory communicator: collectives over hipIpc-mapped windows, no RCCL.
//
// Why it exists.  On an MI355X node every GPU reaches every other GPU's HBM directly over its own xGMI link
// (7 links x ~153 GB/s per GPU, point to point, no switch).  The bandwidth-optimal all-reduce on that topology is
// not a ring (one link per direction busy, 2(W-1)/W * S bytes through it) but the direct TWO-SHOT form: every
// rank reduces its own 1/W slice reading the W-1 peers' copies over W-1 different links at once, then every rank
// gathers the W-1 reduced slices the same way -- 2 * S/W bytes per link, 7x less link time at W = 8.
// It is also the only way to run MORE THAN ONE RANK through the GPU data-parallel path on a box with a single
// device: RCCL refuses two ranks on one GPU, hipIpc does not care whether the peer's window lives on another GPU
// or belongs to another process on the same one.
//
// Mechanism.  Each rank owns
//   * a WINDOW (IPC-exported, fine-grained so remote reads are never served from a stale cache line) with two
//     regions: A = this rank's input in wire format, B = the slice this rank reduced;
//   * a CONTROL page (IPC-exported, fine-grained) holding the sequence number of the last barrier it entered;
//   * private state in plain device memory: its barrier counter and a "failed" latch.
// One all-reduce of n elements (n <= window capacity; longer messages loop) is five launches on one stream:
//     pack      D (storage dtype) -> A_own (wire dtype: fp32 gradients may travel as bf16 / fp16)
//     barrier   every rank has packed
//     reduce    slice r: sum_p A_p[slice r] in RANK ORDER in fp32, x scale -> B_own (wire) and D[slice r]
//     barrier   every rank has reduced
//     gather    slice s != r: B_s -> D[slice s]
// Every rank ends with bit-identical values (each element is computed once, by its slice's owner, and everybody
// else copies the rounded wire value).  No trailing barrier is needed between all-reduces: A is only rewritten
// after barrier 2 (all readers of A are done), B only after the NEXT barrier 1, which a peer enters only after
// its gather.  Broadcast / all-gather read region A of the source rank(s) and end with a barrier.
//
// The barrier is a one-workgroup kernel: publish own sequence number (system-scope release), poll the peers'
// (system-scope acquire) with a wall-clock deadline and a host-raised abort word -- it never hangs.  The
// sequence number is counted ON THE DEVICE, so a step captured into a hipGraph keeps counting when replayed.
// Data moves only in ordinary kernels between barriers: no kernel spins while it holds more than one workgroup.
//
// Replaces nothing in the reference (it has no communication code of its own); it is the xGMI-native alternative to
// the NCCL all-reduce behind imagenet_ddp.py:127 / imagenet_ddp_apex.py:183, selected with DPA_COMM=peer.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../common.h"
#include "comm.h"

namespace py = pybind11;
using dpa::bf16_t;
using dpa::f16_t;
using dpa::from_float;
using dpa::to_float;

namespace {

constexpr int kMaxRanks = 16;
constexpr size_t kCtrlBytes = 4096;
constexpr int kThreads = 256;
constexpr int kMaxBlocks = 2048;

struct PeerPtrs {
  char* win[kMaxRanks];                 // every rank's window (own entry = local allocation)
  const uint32_t* flag[kMaxRanks];      // every rank's control page
};

struct LocalState {
  uint32_t seq;        // barriers entered so far
  uint32_t failed;     // a barrier gave up: every later data kernel poisons what it writes
};

enum Op : int { kSum = 0, kMax = 1 };

// ---- barrier ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void peer_barrier_kernel(LocalState* st, uint32_t* own_flag, PeerPtrs peers, int rank,
                                                          int world, long long deadline_ticks, uint32_t* status,
                                                          const uint32_t* abort_word) {
  __shared__ uint32_t s_seq;
  __shared__ int s_fail;
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_seq = st->seq + 1;
    st->seq = s_seq;
    s_fail = (int)st->failed;
    // everything earlier launches of this stream wrote is in memory (kernel boundary); the release orders this
    // kernel's own view before the flag
    __hip_atomic_store(own_flag, s_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  const uint32_t seq = s_seq;
  if (tid < world && tid != rank && !s_fail) {
    const uint32_t* flag = peers.flag[tid];
    const long long t0 = wall_clock64();
    // sequence numbers only grow; the signed difference keeps the comparison valid across wrap-around
    while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
      __builtin_amdgcn_s_sleep(32);
      if (wall_clock64() - t0 > deadline_ticks ||
          __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) {
        s_fail = 1;          // a peer died, never made the call, or the host aborted: give up, never hang
        break;
      }
    }
  }
  __syncthreads();
  if (tid == 0 && s_fail && !st->failed) {
    st->failed = 1;
    __hip_atomic_store(status, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---- data kernels ----------------------------------------------------------------------------------------------
// A "unit" is 16 bytes of wire format: 4 fp32 or 8 16-bit elements.
template <typename T> struct Unit {
  static constexpr int N = 16 / sizeof(T);
};

template <typename T> __device__ __forceinline__ void load_unit(const T* p, float (&v)[Unit<T>::N]) {
  dpa::Vec16<T> r;
  r.load(p);
#pragma unroll
  for (int j = 0; j < Unit<T>::N; ++j) v[j] = r.get(j);
}
template <typename T, int N> __device__ __forceinline__ void store_vals(T* p, const float (&v)[N]) {
  // N values of T at p (16-byte aligned, N * sizeof(T) a multiple of 16)
  constexpr int per = 16 / sizeof(T);
#pragma unroll
  for (int c = 0; c < N / per; ++c) {
    dpa::Vec16<T> r;
#pragma unroll
    for (int j = 0; j < per; ++j) r.set(j, v[c * per + j]);
    r.store(p + c * per);
  }
}
template <typename T, int N> __device__ __forceinline__ void load_vals(const T* p, float (&v)[N]) {
  constexpr int per = 16 / sizeof(T);
#pragma unroll
  for (int c = 0; c < N / per; ++c) {
    dpa::Vec16<T> r;
    r.load(p + c * per);
#pragma unroll
    for (int j = 0; j < per; ++j) v[c * per + j] = r.get(j);
  }
}

__device__ __forceinline__ float poison() { return __uint_as_float(0x7fc00000u); }

// D (storage) -> A_own (wire)
template <typename SrcT, typename WireT>
__global__ __launch_bounds__(kThreads) void peer_pack_kernel(const SrcT* __restrict__ src, WireT* __restrict__ dst,
                                                             long n, int vec_ok) {
  constexpr int N = Unit<WireT>::N;
  const long stride = (long)gridDim.x * kThreads;
  const long units = vec_ok ? n / N : 0;
  for (long u = (long)blockIdx.x * kThreads + threadIdx.x; u < units; u += stride) {
    float v[N];
    load_vals<SrcT, N>(src + u * N, v);
    store_vals<WireT, N>(dst + u * N, v);
  }
  for (long i = units * N + (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride)
    dst[i] = from_float<WireT>(to_float(src[i]));
}

// slice [lo, hi) of the message: sum of every rank's A in rank order -> B_own (wire) and D (storage)
template <typename WireT, typename OutT, int OP>
__global__ __launch_bounds__(kThreads) void peer_reduce_kernel(PeerPtrs peers, int rank, int world, long lo, long hi,
                                                               long b_off_bytes, OutT* __restrict__ out, float scale,
                                                               int vec_ok, const LocalState* __restrict__ st) {
  constexpr int N = Unit<WireT>::N;
  const bool bad = st->failed != 0;
  WireT* own_b = reinterpret_cast<WireT*>(peers.win[rank] + b_off_bytes);
  const long n = hi - lo;
  const long stride = (long)gridDim.x * kThreads;
  const long units = vec_ok ? n / N : 0;
  for (long u = (long)blockIdx.x * kThreads + threadIdx.x; u < units; u += stride) {
    float acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = OP == kMax ? -INFINITY : 0.f;
    for (int r = 0; r < world; ++r) {
      float v[N];
      load_unit<WireT>(reinterpret_cast<const WireT*>(peers.win[r]) + lo + u * N, v);
#pragma unroll
      for (int j = 0; j < N; ++j) acc[j] = OP == kMax ? fmaxf(acc[j], v[j]) : acc[j] + v[j];
    }
    float w[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const float a = bad ? poison() : acc[j] * scale;
      w[j] = to_float(from_float<WireT>(a));            // what the peers will receive: everybody keeps THESE bits
    }
    store_vals<WireT, N>(own_b + u * N, w);
    store_vals<OutT, N>(out + lo + u * N, w);
  }
  for (long i = units * N + (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    float acc = OP == kMax ? -INFINITY : 0.f;
    for (int r = 0; r < world; ++r) {
      const float v = to_float(reinterpret_cast<const WireT*>(peers.win[r])[lo + i]);
      acc = OP == kMax ? fmaxf(acc, v) : acc + v;
    }
    const WireT w = from_float<WireT>(bad ? poison() : acc * scale);
    own_b[i] = w;
    out[lo + i] = from_float<OutT>(to_float(w));
  }
}

// slices of the other ranks: B_s -> D[slice s]   (grid.y = source rank)
template <typename WireT, typename OutT>
__global__ __launch_bounds__(kThreads) void peer_gather_kernel(PeerPtrs peers, int rank, long count, long per,
                                                               long b_off_bytes, OutT* __restrict__ out, int vec_ok,
                                                               const LocalState* __restrict__ st) {
  constexpr int N = Unit<WireT>::N;
  const int s = blockIdx.y;
  if (s == rank) return;
  const long lo_raw = (long)s * per, hi_raw = lo_raw + per;
  const long lo = lo_raw < count ? lo_raw : count, hi = hi_raw < count ? hi_raw : count;
  const long n = hi - lo;
  if (n <= 0) return;
  const bool bad = st->failed != 0;
  const WireT* src = reinterpret_cast<const WireT*>(peers.win[s] + b_off_bytes);
  const long stride = (long)gridDim.x * kThreads;
  const long units = vec_ok ? n / N : 0;
  for (long u = (long)blockIdx.x * kThreads + threadIdx.x; u < units; u += stride) {
    float v[N];
    load_unit<WireT>(src + u * N, v);
    if (bad) {
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = poison();
    }
    store_vals<OutT, N>(out + lo + u * N, v);
  }
  for (long i = units * N + (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride)
    out[lo + i] = from_float<OutT>(bad ? poison() : to_float(src[i]));
}

// raw bytes (broadcast, all-gather): region A of rank `src` -> dst
__global__ __launch_bounds__(kThreads) void peer_copy_kernel(const char* __restrict__ src, char* __restrict__ dst,
                                                             long nbytes, int vec_ok) {
  const long stride = (long)gridDim.x * kThreads;
  const long units = vec_ok ? nbytes / 16 : 0;
  for (long u = (long)blockIdx.x * kThreads + threadIdx.x; u < units; u += stride)
    reinterpret_cast<uint4*>(dst)[u] = reinterpret_cast<const uint4*>(src)[u];
  for (long i = units * 16 + (long)blockIdx.x * kThreads + threadIdx.x; i < nbytes; i += stride) dst[i] = src[i];
}
// all-gather: region A of every rank -> recv[rank * nbytes ...]   (grid.y = source rank)
__global__ __launch_bounds__(kThreads) void peer_gather_bytes_kernel(PeerPtrs peers, char* __restrict__ recv,
                                                                     long nbytes, int vec_ok) {
  const int s = blockIdx.y;
  const char* src = peers.win[s];
  char* dst = recv + (long)s * nbytes;
  const long stride = (long)gridDim.x * kThreads;
  const long units = vec_ok ? nbytes / 16 : 0;
  for (long u = (long)blockIdx.x * kThreads + threadIdx.x; u < units; u += stride)
    reinterpret_cast<uint4*>(dst)[u] = reinterpret_cast<const uint4*>(src)[u];
  for (long i = units * 16 + (long)blockIdx.x * kThreads + threadIdx.x; i < nbytes; i += stride) dst[i] = src[i];
}

inline int blocks_for(long work_items) {
  long b = (work_items + kThreads - 1) / kThreads;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)b;
}
inline size_t wire_size(int code) {
  switch (code) {
    case 0: return 4;
    case 1: case 2: return 2;
    default: throw std::invalid_argument("peer communicator: float32 / bfloat16 / float16 only");
  }
}
inline bool aligned16(uintptr_t p) { return (p & 15u) == 0; }

class PeerComm : public dpa::comm::StreamCollective {
 public:
  PeerComm(int rank, int world, int device, size_t window_bytes, double timeout_s, bool coarse_window)
      : rank_(rank), world_(world), device_(device) {
    if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world)
      throw std::invalid_argument("PeerComm: world size must be 1..16");
    if (window_bytes < (1u << 16)) throw std::invalid_argument("PeerComm: window smaller than 64 KiB");
    DPA_CHECK_HIP(hipSetDevice(device_));
    a_bytes_ = (window_bytes + 4095) / 4096 * 4096;
    b_bytes_ = ((a_bytes_ + world - 1) / world + 4095) / 4096 * 4096 + 4096;
    // fine-grained: a peer's read is never served from a stale line of ITS caches (remote fine-grained memory is
    // uncached on the reader's side); DPA_PEER_WINDOW_COARSE=1 takes plain hipMalloc memory, which relies on the
    // kernel-boundary acquire alone
    if (coarse_window) {
      DPA_CHECK_HIP(hipMalloc(&window_, a_bytes_ + b_bytes_));
    } else {
      DPA_CHECK_HIP(hipExtMallocWithFlags(&window_, a_bytes_ + b_bytes_, hipDeviceMallocFinegrained));
    }
    DPA_CHECK_HIP(hipExtMallocWithFlags(&ctrl_, kCtrlBytes, hipDeviceMallocFinegrained));
    DPA_CHECK_HIP(hipMemset(ctrl_, 0, kCtrlBytes));
    DPA_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&state_), 256));
    DPA_CHECK_HIP(hipMemset(state_, 0, 256));
    DPA_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&host_words_), 128, hipHostMallocMapped));
    hw(0) = 0;     // status: sequence number of the barrier that gave up
    hw(16) = 0;    // abort word (own cache line)
    DPA_CHECK_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&host_words_dev_), host_words_, 0));
    DPA_CHECK_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    DPA_CHECK_HIP(hipDeviceSynchronize());
    for (int r = 0; r < kMaxRanks; ++r) {
      table_.win[r] = nullptr;
      table_.flag[r] = nullptr;
    }
    table_.win[rank_] = static_cast<char*>(window_);
    table_.flag[rank_] = static_cast<const uint32_t*>(ctrl_);
    deadline_ticks_ = (long long)(timeout_s * 1e8);       // wall_clock64 ticks at 100 MHz
    opened_ = world_ == 1;
  }

  ~PeerComm() override {
    (void)hipSetDevice(device_);
    hw(16) = 1;                                           // anything still spinning leaves
    (void)hipDeviceSynchronize();
    for (int r = 0; r < world_; ++r) {
      if (r == rank_) continue;
      if (mapped_win_[r]) (void)hipIpcCloseMemHandle(mapped_win_[r]);
      if (mapped_ctrl_[r]) (void)hipIpcCloseMemHandle(mapped_ctrl_[r]);
    }
    if (stream_) (void)hipStreamDestroy(stream_);
    if (window_) (void)hipFree(window_);
    if (ctrl_) (void)hipFree(ctrl_);
    if (state_) (void)hipFree(state_);
    if (host_words_) (void)hipHostFree(host_words_);
    for (hipEvent_t e : events_) (void)hipEventDestroy(e);
  }

  // 2 x hipIpcMemHandle_t: window, control page
  py::bytes handle() const {
    hipIpcMemHandle_t h[2];
    DPA_CHECK_HIP(hipIpcGetMemHandle(&h[0], window_));
    DPA_CHECK_HIP(hipIpcGetMemHandle(&h[1], ctrl_));
    return py::bytes(reinterpret_cast<const char*>(h), sizeof(h));
  }
  static size_t handle_bytes() { return 2 * sizeof(hipIpcMemHandle_t); }

  void open_peers(const std::vector<std::string>& handles) {
    if ((int)handles.size() != world_) throw std::invalid_argument("PeerComm: need one handle per rank");
    DPA_CHECK_HIP(hipSetDevice(device_));
    for (int r = 0; r < world_; ++r) {
      if (r == rank_) continue;
      if (handles[r].size() != handle_bytes()) throw std::invalid_argument("PeerComm: bad handle");
      hipIpcMemHandle_t h[2];
      std::memcpy(h, handles[r].data(), sizeof(h));
      DPA_CHECK_HIP(hipIpcOpenMemHandle(&mapped_win_[r], h[0], hipIpcMemLazyEnablePeerAccess));
      DPA_CHECK_HIP(hipIpcOpenMemHandle(&mapped_ctrl_[r], h[1], hipIpcMemLazyEnablePeerAccess));
      table_.win[r] = static_cast<char*>(mapped_win_[r]);
      table_.flag[r] = static_cast<const uint32_t*>(mapped_ctrl_[r]);
    }
    opened_ = true;
  }

  // ---- collectives (stream 0 = the communicator's own stream) ----------------------------------------------------
  void all_reduce(uintptr_t p, size_t count, int dtype, int op, bool average, int wire, uintptr_t s) {
    check_usable();
    if (count == 0) return;
    if (wire < 0) wire = dtype;
    const size_t ws = wire_size(wire);
    (void)wire_size(dtype);
    if (dtype != 0 && wire != dtype) throw std::invalid_argument("PeerComm: only float32 storage takes a narrower wire");
    hipStream_t st = pick(s);
    if (world_ == 1) return;       // the mean / sum / max of one rank is what is already there
    const float scale = average ? 1.0f / (float)world_ : 1.0f;
    const size_t cap = a_bytes_ / ws / 64 * 64;          // elements per pass
    const size_t dsz = wire_size(dtype);
    for (size_t off = 0; off < count; off += cap) {
      const size_t n = std::min(cap, count - off);
      one_pass(p + off * dsz, (long)n, dtype, wire, op, scale, st);
    }
  }

  void broadcast(uintptr_t p, size_t nbytes, int root, uintptr_t s) {
    check_usable();
    if (root < 0 || root >= world_) throw std::invalid_argument("PeerComm: bad root");
    if (nbytes == 0 || world_ == 1) return;
    hipStream_t st = pick(s);
    for (size_t off = 0; off < nbytes; off += a_bytes_) {
      const long n = (long)std::min(a_bytes_, nbytes - off);
      char* ptr = reinterpret_cast<char*>(p + off);
      const int vec = aligned16(reinterpret_cast<uintptr_t>(ptr)) ? 1 : 0;
      if (rank_ == root)
        hipLaunchKernelGGL(peer_copy_kernel, dim3(blocks_for(n / 16 + 1)), dim3(kThreads), 0, st, ptr,
                           table_.win[rank_], n, vec);
      barrier_on(st);
      if (rank_ != root)
        hipLaunchKernelGGL(peer_copy_kernel, dim3(blocks_for(n / 16 + 1)), dim3(kThreads), 0, st,
                           (const char*)table_.win[root], ptr, n, vec);
      barrier_on(st);             // the root's window is free again
    }
    DPA_CHECK_HIP(hipGetLastError());
  }

  void all_gather(uintptr_t send, uintptr_t recv, size_t nbytes, uintptr_t s) {
    check_usable();
    if (nbytes == 0) return;
    if (nbytes > a_bytes_) throw std::invalid_argument("PeerComm: all_gather message larger than the window");
    hipStream_t st = pick(s);
    const int vec = (aligned16(send) && aligned16(recv) && nbytes % 16 == 0) ? 1 : 0;
    if (world_ == 1) {
      hipLaunchKernelGGL(peer_copy_kernel, dim3(blocks_for((long)nbytes / 16 + 1)), dim3(kThreads), 0, st,
                         reinterpret_cast<const char*>(send), reinterpret_cast<char*>(recv), (long)nbytes, vec);
      DPA_CHECK_HIP(hipGetLastError());
      return;
    }
    hipLaunchKernelGGL(peer_copy_kernel, dim3(blocks_for((long)nbytes / 16 + 1)), dim3(kThreads), 0, st,
                       reinterpret_cast<const char*>(send), table_.win[rank_], (long)nbytes,
                       aligned16(send) ? 1 : 0);
    barrier_on(st);
    hipLaunchKernelGGL(peer_gather_bytes_kernel, dim3(blocks_for((long)nbytes / 16 + 1), world_), dim3(kThreads), 0, st,
                       table_, reinterpret_cast<char*>(recv), (long)nbytes, vec);
    barrier_on(st);
    DPA_CHECK_HIP(hipGetLastError());
  }

  void barrier(uintptr_t s) {
    check_usable();
    if (world_ == 1) return;
    barrier_on(pick(s));
    DPA_CHECK_HIP(hipGetLastError());
  }

  // ---- StreamCollective ------------------------------------------------------------------------------------------
  void reduce_mean(uintptr_t p, size_t count, int dtype, int wire) override {
    all_reduce(p, count, dtype, kSum, true, wire, 0);
  }
  void reserve_wire(size_t, int, int) override {}      // the window is the staging
  void abort() override {
    aborted_.store(true);
    hw(16) = 1;
  }
  bool aborted() const override { return aborted_.load(); }
  std::string async_error() override {
    if (hw(0) != 0)
      return "peer barrier " + std::to_string(hw(0)) + " timed out: a rank did not arrive (dead or hung)";
    return std::string();
  }
  hipStream_t stream() const override { return stream_; }
  int device() const override { return device_; }
  int world() const override { return world_; }

  // ---- stream plumbing (same surface as the RCCL communicator) -----------------------------------------------
  void comm_waits_for(uintptr_t producer) {
    hipEvent_t e = scratch_event();
    DPA_CHECK_HIP(hipEventRecord(e, reinterpret_cast<hipStream_t>(producer)));
    DPA_CHECK_HIP(hipStreamWaitEvent(stream_, e, 0));
  }
  void wait_comm(uintptr_t consumer) {
    hipEvent_t e = scratch_event();
    DPA_CHECK_HIP(hipEventRecord(e, stream_));
    DPA_CHECK_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(consumer), e, 0));
  }
  void synchronize() {
    py::gil_scoped_release nogil;
    DPA_CHECK_HIP(hipStreamSynchronize(stream_));
  }
  int rank() const { return rank_; }
  uint32_t status() const { return hw(0); }
  uintptr_t stream_handle() const { return reinterpret_cast<uintptr_t>(stream_); }
  size_t window_bytes() const { return a_bytes_; }
  long passes() const { return passes_; }

 private:
  volatile uint32_t& hw(int i) const { return *reinterpret_cast<volatile uint32_t*>(host_words_ + i); }
  hipStream_t pick(uintptr_t s) const { return s ? reinterpret_cast<hipStream_t>(s) : stream_; }
  void check_usable() const {
    if (!opened_) throw std::runtime_error("PeerComm: peers not opened");
    if (aborted_.load()) throw std::runtime_error("PeerComm: communicator was aborted");
    if (hw(0) != 0)
      throw std::runtime_error("PeerComm: a peer did not arrive at barrier " + std::to_string(hw(0)) +
                               " before the deadline");
  }
  void barrier_on(hipStream_t st) {
    hipLaunchKernelGGL(peer_barrier_kernel, dim3(1), dim3(64), 0, st, state_, static_cast<uint32_t*>(ctrl_), table_,
                       rank_, world_, deadline_ticks_, host_words_dev_, host_words_dev_ + 16);
  }
  hipEvent_t scratch_event() {
    if (events_.size() < 8) {
      hipEvent_t e;
      DPA_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      events_.push_back(e);
      return e;
    }
    return events_[next_event_++ % events_.size()];
  }

  template <typename WireT, typename OutT>
  void launch_pass(OutT* data, long n, int op, float scale, hipStream_t st) {
    constexpr int N = Unit<WireT>::N;
    // slices: multiples of 64 elements so every slice start is 16-byte aligned in both formats
    const long per = ((n + world_ - 1) / world_ + 63) / 64 * 64;
    const long lo = std::min(n, (long)rank_ * per), hi = std::min(n, (long)(rank_ + 1) * per);
    const int vec = aligned16(reinterpret_cast<uintptr_t>(data)) ? 1 : 0;
    WireT* a_own = reinterpret_cast<WireT*>(table_.win[rank_]);
    hipLaunchKernelGGL((peer_pack_kernel<OutT, WireT>), dim3(blocks_for(n / N + 1)), dim3(kThreads), 0, st, data, a_own,
                       n, vec);
    barrier_on(st);
    if (hi > lo) {
      if (op == kMax)
        hipLaunchKernelGGL((peer_reduce_kernel<WireT, OutT, kMax>), dim3(blocks_for((hi - lo) / N + 1)), dim3(kThreads),
                           0, st, table_, rank_, world_, lo, hi, (long)a_bytes_, data, scale, vec, state_);
      else
        hipLaunchKernelGGL((peer_reduce_kernel<WireT, OutT, kSum>), dim3(blocks_for((hi - lo) / N + 1)), dim3(kThreads),
                           0, st, table_, rank_, world_, lo, hi, (long)a_bytes_, data, scale, vec, state_);
    }
    barrier_on(st);
    hipLaunchKernelGGL((peer_gather_kernel<WireT, OutT>), dim3(blocks_for(per / N + 1), world_), dim3(kThreads), 0, st,
                       table_, rank_, n, per, (long)a_bytes_, data, vec, state_);
    DPA_CHECK_HIP(hipGetLastError());
    ++passes_;
  }

  void one_pass(uintptr_t p, long n, int dtype, int wire, int op, float scale, hipStream_t st) {
    if (dtype == 0 && wire == 0) launch_pass<float, float>(reinterpret_cast<float*>(p), n, op, scale, st);
    else if (dtype == 0 && wire == 1) launch_pass<bf16_t, float>(reinterpret_cast<float*>(p), n, op, scale, st);
    else if (dtype == 0 && wire == 2) launch_pass<f16_t, float>(reinterpret_cast<float*>(p), n, op, scale, st);
    else if (dtype == 1) launch_pass<bf16_t, bf16_t>(reinterpret_cast<bf16_t*>(p), n, op, scale, st);
    else launch_pass<f16_t, f16_t>(reinterpret_cast<f16_t*>(p), n, op, scale, st);
  }

  int rank_, world_, device_;
  size_t a_bytes_ = 0, b_bytes_ = 0;
  void* window_ = nullptr;
  void* ctrl_ = nullptr;
  LocalState* state_ = nullptr;
  uint32_t* host_words_ = nullptr;        // host-mapped: [0] status, [16] abort word; read through hw()
  uint32_t* host_words_dev_ = nullptr;
  void* mapped_win_[kMaxRanks] = {};
  void* mapped_ctrl_[kMaxRanks] = {};
  PeerPtrs table_;
  hipStream_t stream_ = nullptr;
  long long deadline_ticks_ = 0;
  bool opened_ = false;
  std::atomic<bool> aborted_{false};
  std::vector<hipEvent_t> events_;
  size_t next_event_ = 0;
  long passes_ = 0;
};

}  // namespace

void bind_peer(py::module_& m) {
  py::class_<PeerComm, dpa::comm::StreamCollective, std::shared_ptr<PeerComm>>(m, "PeerCommunicator")
      .def(py::init<int, int, int, size_t, double, bool>(), py::arg("rank"), py::arg("world_size"), py::arg("device"),
           py::arg("window_bytes") = (size_t)128 << 20, py::arg("timeout_s") = 120.0, py::arg("coarse_window") = false)
      .def("handle", &PeerComm::handle)
      .def("open_peers", &PeerComm::open_peers)
      .def("all_reduce", &PeerComm::all_reduce, py::arg("ptr"), py::arg("count"), py::arg("dtype"), py::arg("op") = 0,
           py::arg("average") = false, py::arg("wire") = -1, py::arg("stream") = 0)
      .def("broadcast", &PeerComm::broadcast, py::arg("ptr"), py::arg("nbytes"), py::arg("root") = 0,
           py::arg("stream") = 0)
      .def("all_gather", &PeerComm::all_gather, py::arg("send"), py::arg("recv"), py::arg("nbytes"),
           py::arg("stream") = 0)
      .def("barrier", &PeerComm::barrier, py::arg("stream") = 0)
      .def("comm_waits_for", &PeerComm::comm_waits_for)
      .def("wait_comm", &PeerComm::wait_comm)
      .def("synchronize", &PeerComm::synchronize)
      .def("abort", &PeerComm::abort)
      .def("async_error", &PeerComm::async_error)
      .def("status", &PeerComm::status)
      .def_property_readonly("aborted", &PeerComm::aborted)
      .def_property_readonly("rank", &PeerComm::rank)
      .def_property_readonly("world_size", &PeerComm::world)
      .def_property_readonly("window_bytes", &PeerComm::window_bytes)
      .def_property_readonly("passes", &PeerComm::passes)
      .def_property_readonly("stream", &PeerComm::stream_handle);
}
