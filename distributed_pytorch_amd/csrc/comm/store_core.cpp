// TcpStore core implementation -- see store_core.h for the protocol and the shutdown contract.
#include "store_core.h"

#include <arpa/inet.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <stdexcept>

namespace dpa {
namespace store {
namespace {

enum Op : uint8_t { kSet = 1, kGet = 2, kAdd = 3, kCheck = 4, kRemove = 5, kCount = 6 };

using Clock = std::chrono::steady_clock;

[[noreturn]] void fail(const std::string& what) {
  throw std::runtime_error("TcpStore: " + what + ": " + strerror(errno));
}

void set_nodelay(int fd) {
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
}

Clock::time_point after(double seconds) {
  return Clock::now() + std::chrono::milliseconds((long)(std::max(0.0, seconds) * 1000));
}

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

void recv_all(int fd, void* buf, size_t n, Clock::time_point deadline, const std::atomic<bool>& interrupted) {
  char* p = static_cast<char*>(buf);
  while (n > 0) {
    if (interrupted.load(std::memory_order_relaxed)) throw std::runtime_error("TcpStore: interrupted");
    auto now = Clock::now();
    if (now >= deadline) throw std::runtime_error("TcpStore: timed out waiting for the store");
    long ms = (long)std::chrono::duration_cast<std::chrono::milliseconds>(deadline - now).count();
    pollfd pf{fd, POLLIN, 0};
    int r = ::poll(&pf, 1, (int)std::min<long>(ms + 1, 200));
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

bool wildcard(const std::string& host) { return host.empty() || host == "0.0.0.0" || host == "*"; }

}  // namespace

// ---- server ------------------------------------------------------------------------------------
Server::Server(const std::string& host, int port, double linger_s) : linger_s_(linger_s) {
  // candidates: the addresses `host` resolves to (only the master's own interface listens: a store reachable
  // from every network would let anyone who can reach the port write keys), then every interface as a fallback
  // for names that do not belong to this machine (NAT'd / externally mapped master addresses)
  std::vector<sockaddr_in> cands;
  if (!wildcard(host)) {
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    if (getaddrinfo(host.c_str(), nullptr, &hints, &res) == 0) {
      for (addrinfo* a = res; a; a = a->ai_next)
        if (a->ai_addrlen == sizeof(sockaddr_in)) cands.push_back(*reinterpret_cast<sockaddr_in*>(a->ai_addr));
      freeaddrinfo(res);
    }
  }
  sockaddr_in any{};
  any.sin_family = AF_INET;
  any.sin_addr.s_addr = htonl(INADDR_ANY);
  cands.push_back(any);
  int last_errno = 0;
  for (sockaddr_in addr : cands) {
    int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) fail("socket");
    int one = 1;
    setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    addr.sin_port = htons((uint16_t)port);
    if (::bind(fd, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) == 0) {
      listen_fd_ = fd;
      char buf[INET_ADDRSTRLEN] = {0};
      bound_host_ = addr.sin_addr.s_addr == htonl(INADDR_ANY)
                        ? std::string("127.0.0.1")
                        : std::string(inet_ntop(AF_INET, &addr.sin_addr, buf, sizeof(buf)));
      break;
    }
    last_errno = errno;
    ::close(fd);
    if (last_errno != EADDRNOTAVAIL) break;   // EADDRINUSE etc.: the port itself is the problem
  }
  if (listen_fd_ < 0) {
    errno = last_errno;
    fail("bind port " + std::to_string(port));
  }
  sockaddr_in addr{};
  socklen_t len = sizeof(addr);
  getsockname(listen_fd_, reinterpret_cast<sockaddr*>(&addr), &len);
  port_ = ntohs(addr.sin_port);
  if (::listen(listen_fd_, 1024) < 0) fail("listen");
  if (::pipe(wake_) < 0) fail("pipe");
  fcntl(listen_fd_, F_SETFL, fcntl(listen_fd_, F_GETFL) | O_NONBLOCK);
  thread_ = std::thread([this] { run(); });
}

Server::~Server() {
  stop_.store(true);
  char c = 1;
  (void)!::write(wake_[1], &c, 1);
  if (thread_.joinable()) thread_.join();     // the thread lingers until the clients said goodbye
  for (auto& c2 : conns_) ::close(c2.fd);
  if (listen_fd_ >= 0) ::close(listen_fd_);
  ::close(wake_[0]);
  ::close(wake_[1]);
}

void Server::reply(Conn& c, uint8_t status, const std::string& v) {
  uint64_t n = v.size();
  c.out.push_back((char)status);
  c.out.append(reinterpret_cast<const char*>(&n), 8);
  c.out.append(v);
}

void Server::wake_waiters(const std::string& key) {
  auto it = waiters_.find(key);
  if (it == waiters_.end()) return;
  const std::string& v = data_[key];
  for (int fd : it->second)
    for (auto& c : conns_)
      if (c.fd == fd) reply(c, 0, v);
  waiters_.erase(it);
}

// returns false when more bytes are needed; throws to drop a misbehaving connection
bool Server::handle_one(Conn& c) {
  const std::string& b = c.in;
  if (b.size() < 5) return false;
  const uint8_t op = (uint8_t)b[0];
  uint32_t klen;
  memcpy(&klen, b.data() + 1, 4);
  if (klen > kMaxKeyBytes) throw std::runtime_error("TcpStore: oversized key");
  if (b.size() < 5 + (size_t)klen + 8) return false;
  uint64_t vlen;
  memcpy(&vlen, b.data() + 5 + klen, 8);
  if (vlen > kMaxValueBytes) throw std::runtime_error("TcpStore: oversized value");
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

void Server::drop(size_t i) {
  const int fd = conns_[i].fd;
  for (auto it = waiters_.begin(); it != waiters_.end();) {
    auto& v = it->second;
    v.erase(std::remove(v.begin(), v.end(), fd), v.end());
    if (v.empty()) it = waiters_.erase(it);
    else ++it;
  }
  ::close(fd);
  conns_.erase(conns_.begin() + (long)i);
}

// replies produced this round, also those created for OTHER connections by wake_waiters
void Server::flush_all() {
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
}

void Server::run() {
  std::vector<pollfd> pfs;
  bool draining = false;
  Clock::time_point linger_deadline{};
  for (;;) {
    if (stop_.load() && !draining) {
      // goodbye phase: no new clients; the ones we have are served until they hang up
      draining = true;
      linger_deadline = after(linger_s_);
      ::close(listen_fd_);
      listen_fd_ = -1;
    }
    if (draining && (conns_.empty() || Clock::now() >= linger_deadline)) break;
    pfs.clear();
    pfs.push_back({listen_fd_, POLLIN, 0});            // fd -1 is ignored by poll()
    pfs.push_back({wake_[0], POLLIN, 0});
    for (auto& c : conns_) pfs.push_back({c.fd, (short)(POLLIN | (c.out.empty() ? 0 : POLLOUT)), 0});
    int r = ::poll(pfs.data(), pfs.size(), draining ? 50 : 500);
    if (r < 0 && errno != EINTR) break;
    if (r <= 0) continue;
    if (pfs[1].revents & POLLIN) {
      char buf[16];
      (void)!::read(wake_[0], buf, sizeof(buf));
    }
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
      if (dead) drop(i);
    }
    flush_all();
    if (listen_fd_ >= 0 && (pfs[0].revents & POLLIN)) {
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
  flush_all();
}

// ---- client ------------------------------------------------------------------------------------
Client::Client(const std::string& host, int port, double timeout_s)
    : host_(host), port_(port), timeout_s_(timeout_s) {
  connect_to(timeout_s_);
}

Client::~Client() {
  if (fd_ >= 0) ::close(fd_);     // the close is this client's goodbye to the server
}

void Client::connect_to(double budget_s) {
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_INET;
  hints.ai_socktype = SOCK_STREAM;
  const auto deadline = after(budget_s);
  std::string last = "no address";
  do {
    if (getaddrinfo(host_.c_str(), std::to_string(port_).c_str(), &hints, &res) == 0) {
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
      last = "cannot resolve " + host_;
    }
    if (interrupted_.load()) break;
    std::this_thread::sleep_for(std::chrono::milliseconds(50));   // the master may not be up yet
  } while (Clock::now() < deadline);
  throw std::runtime_error("TcpStore: cannot connect to " + host_ + ":" + std::to_string(port_) + " (" + last + ")");
}

std::string Client::request(uint8_t op, const std::string& key, const std::string& val, double timeout_s,
                            uint8_t* status) {
  if (key.size() > kMaxKeyBytes) throw std::invalid_argument("TcpStore: key longer than 4 KiB");
  if (val.size() > kMaxValueBytes) throw std::invalid_argument("TcpStore: value larger than 1 GiB");
  std::lock_guard<std::mutex> lock(mu_);
  if (interrupted_.load()) throw std::runtime_error("TcpStore: interrupted");
  std::string msg;
  msg.reserve(13 + key.size() + val.size());
  msg.push_back((char)op);
  uint32_t klen = (uint32_t)key.size();
  uint64_t vlen = val.size();
  msg.append(reinterpret_cast<const char*>(&klen), 4);
  msg.append(key);
  msg.append(reinterpret_cast<const char*>(&vlen), 8);
  msg.append(val);
  if (fd_ < 0) throw std::runtime_error("TcpStore: not connected");
  const auto deadline = after(timeout_s);
  uint8_t st = 0;
  uint64_t n = 0;
  std::string out;
  try {
    send_all(fd_, msg.data(), msg.size());
    recv_all(fd_, &st, 1, deadline, interrupted_);
    recv_all(fd_, &n, 8, deadline, interrupted_);
    if (n > kMaxValueBytes) throw std::runtime_error("TcpStore: corrupt reply");
    out.resize(n);
    if (n) recv_all(fd_, &out[0], n, deadline, interrupted_);
  } catch (const std::exception& e) {
    // The request may still be parked on the server: a fresh connection drops it, so a late reply can never
    // be mistaken for the answer to the next request.  The reconnect gets a short budget of its own -- a dead
    // master must not cost the caller a second full timeout.
    ::close(fd_);
    fd_ = -1;
    const std::string why = std::string(e.what()) + " (key '" + key + "')";
    if (!interrupted_.load()) {
      try {
        connect_to(std::min(2.0, timeout_s_));
      } catch (...) {
      }
    }
    throw std::runtime_error(why);
  }
  if (status) *status = st;
  return out;
}

void Client::set(const std::string& key, const std::string& value) { request(kSet, key, value, timeout_s_); }

std::string Client::get(const std::string& key, double timeout_s) {
  return request(kGet, key, "", timeout_s > 0 ? timeout_s : timeout_s_);
}

int64_t Client::add(const std::string& key, int64_t delta) {
  std::string v(reinterpret_cast<const char*>(&delta), 8);
  std::string r = request(kAdd, key, v, timeout_s_);
  int64_t out = 0;
  if (r.size() == 8) memcpy(&out, r.data(), 8);
  return out;
}

bool Client::check(const std::string& key) {
  uint8_t st = 0;
  request(kCheck, key, "", timeout_s_, &st);
  return st == 0;
}

bool Client::remove(const std::string& key) {
  uint8_t st = 0;
  request(kRemove, key, "", timeout_s_, &st);
  return st == 0;
}

int64_t Client::num_keys() {
  std::string r = request(kCount, "", "", timeout_s_);
  int64_t out = 0;
  if (r.size() == 8) memcpy(&out, r.data(), 8);
  return out;
}

void Client::barrier(const std::string& name, int world, double timeout_s) {
  const int64_t k = add("__barrier__/" + name + "/count", 1);
  const int64_t gen = (k - 1) / world;
  const std::string rel = "__barrier__/" + name + "/release/" + std::to_string(gen);
  if (k == (gen + 1) * (int64_t)world) {
    set(rel, "1");
    if (gen > 0) remove("__barrier__/" + name + "/release/" + std::to_string(gen - 1));   // nobody waits there now
  }
  request(kGet, rel, "", timeout_s > 0 ? timeout_s : timeout_s_);
}

void Client::interrupt() { interrupted_.store(true); }

// ---- master / worker facade ------------------------------------------------------------------------
TcpStore::TcpStore(const std::string& host, int port, bool is_master, double timeout_s, double linger_s) {
  host_ = host;
  if (is_master) {
    server_.reset(new Server(host, port, linger_s));
    port = server_->port();
    host_ = server_->bound_host();
  }
  port_ = port;
  client_.reset(new Client(host_, port, timeout_s));
}

TcpStore::~TcpStore() {
  client_.reset();    // our own goodbye first, then the server lingers for the other ranks'
  server_.reset();
}

}  // namespace store
}  // namespace dpa
