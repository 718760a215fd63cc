// Communication runtime: RCCL communicator, peer-memory communicator, bucketed gradient reducer.
#pragma once
#include <hip/hip_runtime.h>
#include <pybind11/pybind11.h>

#include <cstdint>
#include <string>

namespace dpa {
namespace comm {

// What the gradient reducer's GPU backend needs from a communicator: a HIP stream of its own to queue collectives
// on, one in-place mean all-reduce, and the failure hooks of the watchdog.  Two implementations:
//   * Communicator (comm.cpp)  -- RCCL over xGMI, the default;
//   * PeerComm     (peer.hip)  -- two-shot reduce-scatter + all-gather through hipIpc-mapped windows, which also
//                                 works between processes that SHARE one device (how the multi-rank GPU path is
//                                 exercised on a one-GPU box).
class StreamCollective {
 public:
  virtual ~StreamCollective() = default;
  virtual hipStream_t stream() const = 0;
  virtual int device() const = 0;
  virtual int world() const = 0;
  // in-place mean over ranks of `count` elements of `dtype` at p, travelling as `wire` (-1: as stored), enqueued
  // on stream()
  virtual void reduce_mean(uintptr_t p, size_t count, int dtype, int wire) = 0;
  // staging the narrowed wire format may need, allocated ahead of the first launch (never inside one)
  virtual void reserve_wire(size_t max_count, int dtype, int wire) = 0;
  virtual void abort() = 0;
  virtual bool aborted() const = 0;
  virtual std::string async_error() = 0;
};

}  // namespace comm
}  // namespace dpa

void bind_comm(pybind11::module_& m);
void bind_store(pybind11::module_& m);   // TcpStore rendezvous (store.cpp)
void bind_p2p(pybind11::module_& m);     // one-shot peer-memory all-reduce for small messages (p2p.hip)
void bind_peer(pybind11::module_& m);    // peer-memory communicator: two-shot all-reduce through IPC windows (peer.hip)
