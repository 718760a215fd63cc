// TcpStore: the rendezvous key-value store of the native process group.
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
// Replaces c10d TCPStore / env:// rendezvous (reference nd_imagenet.py:130-131, imagenet_ddp.py:104-105,
// imagenet_ddp_apex.py:124-125) -- SURVEY.md N-C1.
#include "comm.h"

#include <arpa/inet.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <pybind11/stl.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace py = pybind11;

namespace {

enum Op : uint8_t { kSet = 1, kGet = 2, kAdd = 3, kCheck = 4, kRemove = 5, kCount = 6 };

using Clock = std::chrono::steady_clock;

[[noreturn]] void fail(const std::string& what) { throw std::runtime_error("TcpStore: " + what + ": " + strerror(errno)); }

void set_nodelay(int fd) {
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
}

// ---- blocking helpers with a deadline (client side) -----------------------------------------
void send_all(int fd, const void* buf, size_t n) {
  const char* p = static_cast<const char*>(buf);
  while (n > 0) {
    ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
    if (k < 0) {
      if (errno == EINTR) continue;
      fail("send");
    }
    p += k;
    n -= (size_t)k;
  }
}

void recv_all(int fd, void* buf, size_t n, Clock::time_point deadline) {
  char* p = static_cast<char*>(buf);
  while (n > 0) {
    auto now = Clock::now();
    if (now >= deadline) throw std::runtime_error("TcpStore: timed out waiting for the store");
    int ms = (int)std::chrono::duration_cast<std::chrono::milliseconds>(deadline - now).count();
    pollfd pf{fd, POLLIN, 0};
    int r = ::poll(&pf, 1, ms > 1000 ? 1000 : ms + 1);
    if (r < 0) {
      if (errno == EINTR) continue;
      fail("poll");
    }
    if (r == 0) continue;
    ssize_t k = ::recv(fd, p, n, 0);
    if (k == 0) throw std::runtime_error("TcpStore: connection closed by the master (rank 0 died?)");
    if (k < 0) {
      if (errno == EINTR || errno == EAGAIN) continue;
      fail("recv");
    }
    p += k;
    n -= (size_t)k;
  }
}

// ---- server ------------------------------------------------------------------------------------
struct Conn {
  int fd = -1;
  std::string in;                 // unparsed request bytes
  std::string out;                // unsent reply bytes
};

class Server {
 public:
  Server(const std::string& host, int port) {
    listen_fd_ = ::socket(AF_INET, SOCK_STREAM, 0);
    if (listen_fd_ < 0) fail("socket");
    int one = 1;
    setsockopt(listen_fd_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in addr{};
    addr.sin_family = AF_INET;
    addr.sin_port = htons((uint16_t)port);
    addr.sin_addr.s_addr = htonl(INADDR_ANY);       // other nodes reach the master by its address
    (void)host;
    if (::bind(listen_fd_, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) < 0) {
      int e = errno;
      ::close(listen_fd_);
      errno = e;
      fail("bind port " + std::to_string(port));
    }
    socklen_t len = sizeof(addr);
    getsockname(listen_fd_, reinterpret_cast<sockaddr*>(&addr), &len);
    port_ = ntohs(addr.sin_port);
    if (::listen(listen_fd_, 1024) < 0) fail("listen");
    if (::pipe(wake_) < 0) fail("pipe");
    fcntl(listen_fd_, F_SETFL, fcntl(listen_fd_, F_GETFL) | O_NONBLOCK);
    thread_ = std::thread([this] { run(); });
  }
  ~Server() {
    stop_ = true;
    char c = 1;
    (void)!::write(wake_[1], &c, 1);
    if (thread_.joinable()) thread_.join();
    for (auto& c2 : conns_) ::close(c2.fd);
    ::close(listen_fd_);
    ::close(wake_[0]);
    ::close(wake_[1]);
  }
  int port() const { return port_; }

 private:
  void reply(Conn& c, uint8_t status, const std::string& v) {
    uint64_t n = v.size();
    c.out.push_back((char)status);
    c.out.append(reinterpret_cast<const char*>(&n), 8);
    c.out.append(v);
  }
  void wake_waiters(const std::string& key) {
    auto it = waiters_.find(key);
    if (it == waiters_.end()) return;
    const std::string& v = data_[key];
    for (int fd : it->second)
      for (auto& c : conns_)
        if (c.fd == fd) reply(c, 0, v);
    waiters_.erase(it);
  }
  // returns false when more bytes are needed
  bool handle_one(Conn& c) {
    const std::string& b = c.in;
    if (b.size() < 5) return false;
    const uint8_t op = (uint8_t)b[0];
    uint32_t klen;
    memcpy(&klen, b.data() + 1, 4);
    if (b.size() < 5 + (size_t)klen + 8) return false;
    uint64_t vlen;
    memcpy(&vlen, b.data() + 5 + klen, 8);
    if (vlen > (1ull << 30)) throw std::runtime_error("TcpStore: oversized value");
    const size_t total = 5 + (size_t)klen + 8 + (size_t)vlen;
    if (b.size() < total) return false;
    const std::string key = b.substr(5, klen);
    const std::string val = b.substr(5 + klen + 8, vlen);
    c.in.erase(0, total);
    switch (op) {
      case kSet:
        data_[key] = val;
        reply(c, 0, "");
        wake_waiters(key);
        break;
      case kGet: {
        auto it = data_.find(key);
        if (it != data_.end()) reply(c, 0, it->second);
        else waiters_[key].push_back(c.fd);       // answered by the set/add that creates the key
        break;
      }
      case kAdd: {
        int64_t delta = 0, cur = 0;
        if (val.size() == 8) memcpy(&delta, val.data(), 8);
        auto it = data_.find(key);
        if (it != data_.end() && it->second.size() == 8) memcpy(&cur, it->second.data(), 8);
        cur += delta;
        std::string nv(reinterpret_cast<const char*>(&cur), 8);
        data_[key] = nv;
        reply(c, 0, nv);
        wake_waiters(key);
        break;
      }
      case kCheck:
        reply(c, data_.count(key) ? 0 : 1, "");
        break;
      case kRemove:
        reply(c, data_.erase(key) ? 0 : 1, "");
        break;
      case kCount: {
        int64_t n = (int64_t)data_.size();
        reply(c, 0, std::string(reinterpret_cast<const char*>(&n), 8));
        break;
      }
      default:
        throw std::runtime_error("TcpStore: bad opcode");
    }
    return true;
  }
  void drop(size_t i) {
    const int fd = conns_[i].fd;
    for (auto& w : waiters_) {
      auto& v = w.second;
      for (size_t k = 0; k < v.size();)
        if (v[k] == fd) v.erase(v.begin() + k);
        else ++k;
    }
    ::close(fd);
    conns_.erase(conns_.begin() + i);
  }
  void run() {
    std::vector<pollfd> pfs;
    while (!stop_) {
      pfs.clear();
      pfs.push_back({listen_fd_, POLLIN, 0});
      pfs.push_back({wake_[0], POLLIN, 0});
      for (auto& c : conns_) pfs.push_back({c.fd, (short)(POLLIN | (c.out.empty() ? 0 : POLLOUT)), 0});
      int r = ::poll(pfs.data(), pfs.size(), 500);
      if (r < 0 && errno != EINTR) break;
      if (r <= 0) continue;
      if (pfs[1].revents & POLLIN) continue;           // stop_ is re-checked by the loop condition
      const size_t nconn = conns_.size();
      for (size_t i = nconn; i-- > 0;) {               // backwards: drop() erases
        const short ev = pfs[2 + i].revents;
        Conn& c = conns_[i];
        bool dead = false;
        if (ev & POLLIN) {
          char buf[65536];
          ssize_t k = ::recv(c.fd, buf, sizeof(buf), MSG_DONTWAIT);
          if (k == 0 || (k < 0 && errno != EAGAIN && errno != EINTR)) dead = true;
          else if (k > 0) {
            c.in.append(buf, (size_t)k);
            try {
              while (handle_one(c)) {}
            } catch (...) {
              dead = true;
            }
          }
        } else if (ev & (POLLHUP | POLLERR)) {
          dead = true;
        }
        if (dead) {
          drop(i);
          continue;
        }
      }
      // flush replies (also those produced for OTHER connections by wake_waiters)
      for (size_t i = conns_.size(); i-- > 0;) {
        Conn& c = conns_[i];
        while (!c.out.empty()) {
          ssize_t k = ::send(c.fd, c.out.data(), c.out.size(), MSG_NOSIGNAL | MSG_DONTWAIT);
          if (k > 0) c.out.erase(0, (size_t)k);
          else if (k < 0 && (errno == EAGAIN || errno == EINTR)) break;
          else {
            drop(i);
            break;
          }
        }
      }
      if (pfs[0].revents & POLLIN) {
        for (;;) {
          int fd = ::accept(listen_fd_, nullptr, nullptr);
          if (fd < 0) break;
          set_nodelay(fd);
          Conn c;
          c.fd = fd;
          conns_.push_back(std::move(c));
        }
      }
    }
  }

  int listen_fd_ = -1, port_ = 0;
  int wake_[2] = {-1, -1};
  std::atomic<bool> stop_{false};
  std::thread thread_;
  std::vector<Conn> conns_;
  std::map<std::string, std::string> data_;
  std::map<std::string, std::vector<int>> waiters_;
};

// ---- client (+ optional embedded server) --------------------------------------------------------
class TcpStore {
 public:
  TcpStore(const std::string& host, int port, bool is_master, double timeout_s) : timeout_s_(timeout_s) {
    if (is_master) {
      server_.reset(new Server(host, port));
      port = server_->port();
    }
    port_ = port;
    host_ = is_master ? "127.0.0.1" : host;
    connect_to(host_, port);
  }
  ~TcpStore() {
    if (fd_ >= 0) ::close(fd_);
    server_.reset();
  }
  int port() const { return port_; }

  void set(const std::string& key, const std::string& value) { request(kSet, key, value, timeout_s_); }
  py::bytes get(const std::string& key, double timeout_s) {
    return py::bytes(request(kGet, key, "", timeout_s > 0 ? timeout_s : timeout_s_));
  }
  std::string get_str(const std::string& key, double timeout_s) {
    return request(kGet, key, "", timeout_s > 0 ? timeout_s : timeout_s_);
  }
  int64_t add(const std::string& key, int64_t delta) {
    std::string v(reinterpret_cast<const char*>(&delta), 8);
    std::string r = request(kAdd, key, v, timeout_s_);
    int64_t out = 0;
    if (r.size() == 8) memcpy(&out, r.data(), 8);
    return out;
  }
  bool check(const std::string& key) {
    uint8_t st = 0;
    request(kCheck, key, "", timeout_s_, &st);
    return st == 0;
  }
  bool remove(const std::string& key) {
    uint8_t st = 0;
    request(kRemove, key, "", timeout_s_, &st);
    return st == 0;
  }
  int64_t num_keys() {
    std::string r = request(kCount, "", "", timeout_s_);
    int64_t out = 0;
    if (r.size() == 8) memcpy(&out, r.data(), 8);
    return out;
  }
  // Every rank calls barrier(name, world) the same number of times: arrival k belongs to generation
  // (k-1)/world, whose last arriver publishes the release key the others block on.
  void barrier(const std::string& name, int world, double timeout_s) {
    const int64_t k = add("__barrier__/" + name + "/count", 1);
    const int64_t gen = (k - 1) / world;
    const std::string rel = "__barrier__/" + name + "/release/" + std::to_string(gen);
    if (k == (gen + 1) * (int64_t)world) set(rel, "1");
    request(kGet, rel, "", timeout_s > 0 ? timeout_s : timeout_s_);
  }

 private:
  void connect_to(const std::string& host, int port) {
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    const auto deadline = Clock::now() + std::chrono::milliseconds((long)(timeout_s_ * 1000));
    std::string last = "no address";
    while (Clock::now() < deadline) {
      if (getaddrinfo(host.c_str(), std::to_string(port).c_str(), &hints, &res) == 0) {
        for (addrinfo* a = res; a; a = a->ai_next) {
          int fd = ::socket(a->ai_family, a->ai_socktype, a->ai_protocol);
          if (fd < 0) continue;
          if (::connect(fd, a->ai_addr, a->ai_addrlen) == 0) {
            set_nodelay(fd);
            fd_ = fd;
            freeaddrinfo(res);
            return;
          }
          last = strerror(errno);
          ::close(fd);
        }
        freeaddrinfo(res);
        res = nullptr;
      } else {
        last = "cannot resolve " + host;
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(100));   // the master may not be up yet
    }
    throw std::runtime_error("TcpStore: cannot connect to " + host + ":" + std::to_string(port) + " (" + last + ")");
  }
  std::string request(Op op, const std::string& key, const std::string& val, double timeout_s, uint8_t* status = nullptr) {
    py::gil_scoped_release nogil;                     // a blocked get must not freeze other Python threads
    std::string msg;
    msg.push_back((char)op);
    uint32_t klen = (uint32_t)key.size();
    uint64_t vlen = val.size();
    msg.append(reinterpret_cast<const char*>(&klen), 4);
    msg.append(key);
    msg.append(reinterpret_cast<const char*>(&vlen), 8);
    msg.append(val);
    if (fd_ < 0) throw std::runtime_error("TcpStore: not connected");
    send_all(fd_, msg.data(), msg.size());
    const auto deadline = Clock::now() + std::chrono::milliseconds((long)(timeout_s * 1000));
    uint8_t st;
    uint64_t n;
    try {
      recv_all(fd_, &st, 1, deadline);
      recv_all(fd_, &n, 8, deadline);
    } catch (const std::runtime_error& e) {
      // the request may still be parked on the server: a fresh connection drops it, so a late reply can
      // never be mistaken for the answer to the next request
      ::close(fd_);
      fd_ = -1;
      std::string why = std::string(e.what()) + " (key '" + key + "')";
      try {
        connect_to(host_, port_);
      } catch (...) {
      }
      throw std::runtime_error(why);
    }
    std::string out(n, '\0');
    if (n) recv_all(fd_, &out[0], n, deadline);
    if (status) *status = st;
    return out;
  }

  std::unique_ptr<Server> server_;
  std::string host_;
  int fd_ = -1, port_ = 0;
  double timeout_s_;
};

}  // namespace

void bind_store(py::module_& m) {
  py::class_<TcpStore>(m, "TcpStore")
      .def(py::init<const std::string&, int, bool, double>(), py::arg("host"), py::arg("port"),
           py::arg("is_master"), py::arg("timeout_s") = 300.0)
      .def_property_readonly("port", &TcpStore::port)
      .def("set", [](TcpStore& s, const std::string& k, const py::bytes& v) { s.set(k, std::string(v)); })
      .def("get", &TcpStore::get, py::arg("key"), py::arg("timeout_s") = -1.0)
      .def("add", &TcpStore::add)
      .def("check", &TcpStore::check)
      .def("remove", &TcpStore::remove)
      .def("num_keys", &TcpStore::num_keys)
      .def("barrier", &TcpStore::barrier, py::arg("name"), py::arg("world"), py::arg("timeout_s") = -1.0);
}
