// TcpStore core: the rendezvous key-value store of the native process group (no Python, no HIP:
// this file and store_core.cpp also build stand-alone with -fsanitize=address,undefined / thread,
// see tests/native/).
//
// Rank 0 (the "master") runs a poll()-driven server thread on MASTER_ADDR:MASTER_PORT; every rank
// (rank 0 included) talks to it through one blocking client socket.  It provides exactly what a
// one-process-per-GPU job needs to bootstrap RCCL and supervise itself:
//
//   set(key, bytes)            publish (e.g. the ncclUniqueId)
//   get(key, timeout)          block until the key exists, return its value
//   add(key, delta) -> int64   atomic counter (rank census, barriers, generation numbers)
//   check(key) / remove(key)
//   barrier(name, world)       add + wait on a generation key -- usable before any GPU communicator exists
//
// Wire format (little endian): request = u8 op | u32 klen | key | u64 vlen | value,
// reply = u8 status | u64 vlen | value.  A blocked get is parked server side and answered by the
// set/add that creates the key, so there is no polling traffic while ranks wait for each other.
//
// Shutdown contract: the master's store may be destroyed while other ranks still have requests in
// flight (the last arriver of the final barrier wakes rank 0 before it has read its own reply).
// ~Server therefore stops accepting, keeps SERVING until every client has closed its socket (that
// close is the goodbye) or `linger_s` elapsed, flushes every pending reply, and only then closes.
//
// Replaces c10d TCPStore / env:// rendezvous (reference nd_imagenet.py:130-131, imagenet_ddp.py:104-105,
// imagenet_ddp_apex.py:124-125) -- SURVEY.md N-C1.
#pragma once
#include <atomic>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace dpa {
namespace store {

constexpr uint32_t kMaxKeyBytes = 4096;
constexpr uint64_t kMaxValueBytes = 1ull << 30;

class Server {
 public:
  // `host`: interface to bind (numeric or resolvable name; loopback names bind loopback only).  Falls back to
  // every interface when the name does not belong to this machine.  port 0 = ephemeral.
  Server(const std::string& host, int port, double linger_s = 10.0);
  ~Server();
  int port() const { return port_; }
  const std::string& bound_host() const { return bound_host_; }   // address a local client should dial

 private:
  struct Conn {
    int fd = -1;
    std::string in;    // unparsed request bytes
    std::string out;   // unsent reply bytes
  };
  void reply(Conn& c, uint8_t status, const std::string& v);
  void wake_waiters(const std::string& key);
  bool handle_one(Conn& c);
  void drop(size_t i);
  void flush_all();
  void run();

  int listen_fd_ = -1, port_ = 0;
  int wake_[2] = {-1, -1};
  double linger_s_;
  std::string bound_host_;
  std::atomic<bool> stop_{false};
  std::thread thread_;
  std::vector<Conn> conns_;
  std::map<std::string, std::string> data_;
  std::map<std::string, std::vector<int>> waiters_;
};

class Client {
 public:
  Client(const std::string& host, int port, double timeout_s);
  ~Client();
  Client(const Client&) = delete;
  Client& operator=(const Client&) = delete;

  void set(const std::string& key, const std::string& value);
  std::string get(const std::string& key, double timeout_s = -1.0);
  int64_t add(const std::string& key, int64_t delta);
  bool check(const std::string& key);
  bool remove(const std::string& key);
  int64_t num_keys();
  // Every rank calls barrier(name, world) the same number of times: arrival k belongs to generation
  // (k-1)/world, whose last arriver publishes the release key the others block on.
  void barrier(const std::string& name, int world, double timeout_s = -1.0);
  // Makes every blocked or future request of this client fail at once (any thread): the watchdog's way of
  // unblocking a collective that waits for a dead peer.
  void interrupt();
  double timeout_s() const { return timeout_s_; }
  const std::string& host() const { return host_; }
  int port() const { return port_; }

 private:
  void connect_to(double budget_s);
  std::string request(uint8_t op, const std::string& key, const std::string& val, double timeout_s,
                      uint8_t* status = nullptr);

  std::string host_;
  int port_ = 0;
  int fd_ = -1;
  double timeout_s_;
  std::atomic<bool> interrupted_{false};
  std::mutex mu_;   // one request at a time per connection
};

// Master = embedded Server + a Client to it; every other rank = Client only.
class TcpStore {
 public:
  TcpStore(const std::string& host, int port, bool is_master, double timeout_s, double linger_s = 10.0);
  ~TcpStore();
  Client& client() { return *client_; }
  int port() const { return port_; }
  const std::string& host() const { return host_; }

 private:
  std::unique_ptr<Server> server_;
  std::unique_ptr<Client> client_;
  std::string host_;
  int port_ = 0;
};

}  // namespace store
}  // namespace dpa
