// Native parameter service: the host-memory parameter-server topology of
// /root/reference/distributed.py (variables on /job:ps/cpu:0, workers pull values
// and push gradients over TCP every step; :60-64, :102), including the server
// side of tf.train.SyncReplicasOptimizer (:97-106):
//
//   * variables           float32 arrays, registered by name, initialised by the chief
//   * async update        theta -= lr * g applied on arrival WITHOUT locking (Hogwild;
//                         ApplyGradientDescent with use_locking=False)
//   * global_step         int64 counter with atomic increment (starts at 1, :65)
//   * accumulators        one per variable: ACC_APPLY(grad, local_step) adds the gradient if
//                         local_step >= the accumulator's step and DROPS it otherwise (stale);
//                         TAKE_APPLY(R, lr) blocks until R gradients arrived, applies their
//                         mean and clears the accumulator; ACC_SET_STEP stamps the new step
//   * token queue         FIFO of int64 (the sync barrier): ENQUEUE_MANY / blocking DEQUEUE
//   * readiness           IS_READY / SET_READY (workers poll until the chief has initialised)
//
// One thread per connection; a request is a fixed 40-byte header + payload, a response a
// fixed 24-byte header + payload (little endian).  The service has no idea what a model is.
#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "dtf_runtime.h"

namespace {

constexpr uint32_t kMagic = 0x44544650;  // "DTFP"

enum Op : uint32_t {
  OP_REGISTER = 1,      // payload = name; value = numel            -> value = var id
  OP_INIT = 2,          // var, payload = float data
  OP_PULL = 3,          // var                                       -> payload = float data
  OP_PUSH_ASYNC = 4,    // var, scalar = lr, payload = gradient      (lock-free apply)
  OP_INC_STEP = 5,      //                                           -> value = new global_step
  OP_GET_STEP = 6,      //                                           -> value = global_step
  OP_SET_STEP = 7,      // value = global_step
  OP_ACC_APPLY = 8,     // var, value = local_step, payload = grad   -> aux = 1 accepted / 0 dropped
  OP_TAKE_APPLY = 9,    // var, value = R, scalar = lr               (blocks) -> aux = gradients averaged
  OP_ACC_SET_STEP = 10, // var, value = new accumulator step
  OP_TOKEN_ENQUEUE = 11,  // aux = count, value = token
  OP_TOKEN_DEQUEUE = 12,  // (blocks)                                -> value = token
  OP_SET_READY = 13,
  OP_IS_READY = 14,     //                                           -> aux = 1 when ready
  OP_SHUTDOWN = 15,
  OP_STATS = 16,        //                                           -> payload = 4 x int64 counters
  OP_WORKER_DONE = 17,  //                                           -> value = workers finished so far
};

// limits on what a request may ask for (the service trusts no size that arrives over the wire)
constexpr uint64_t kMaxNameBytes = 4096;
constexpr int64_t kMaxVariableElems = int64_t(1) << 32;      // 16 GiB of fp32 per variable
constexpr size_t kMaxVariables = 1 << 16;
constexpr uint32_t kMaxTokensPerCall = 1 << 16;

#pragma pack(push, 1)
struct ReqHeader {
  uint32_t magic, op, var, aux;
  int64_t value;
  double scalar;
  uint64_t nbytes;
};
struct RespHeader {
  uint32_t status, aux;
  int64_t value;
  uint64_t nbytes;
};
#pragma pack(pop)
static_assert(sizeof(ReqHeader) == 40 && sizeof(RespHeader) == 24, "wire layout");

bool read_full(int fd, void* buf, size_t n) {
  char* p = static_cast<char*>(buf);
  while (n) {
    ssize_t r = ::recv(fd, p, n, 0);
    if (r <= 0) return false;
    p += r;
    n -= size_t(r);
  }
  return true;
}
bool write_full(int fd, const void* buf, size_t n) {
  const char* p = static_cast<const char*>(buf);
  while (n) {
    ssize_t r = ::send(fd, p, n, MSG_NOSIGNAL);
    if (r <= 0) return false;
    p += r;
    n -= size_t(r);
  }
  return true;
}

struct Variable {
  std::string name;
  std::vector<float> value;
  bool initialized = false;
  // accumulator (sync mode)
  std::mutex acc_mu;
  std::condition_variable acc_cv;
  std::vector<float> acc_sum;
  int64_t acc_count = 0;
  int64_t acc_step = 0;
};

struct Server {
  int listen_fd = -1;
  int port = 0;
  std::thread accept_thread;
  std::vector<std::thread> conn_threads;
  std::mutex mu;                       // registry, readiness, shutdown
  std::condition_variable cv;
  std::vector<std::unique_ptr<Variable>> vars;
  std::map<std::string, uint32_t> by_name;
  bool ready = false;
  std::atomic<bool> stopping{false};
  bool shutdown_requested = false;
  std::atomic<int64_t> global_step{1};
  std::mutex tok_mu;
  std::condition_variable tok_cv;
  std::deque<int64_t> tokens;
  std::atomic<int64_t> n_push{0}, n_pull{0}, n_dropped{0}, n_updates{0}, n_done{0};
  std::vector<int> conn_fds;
};

// Hogwild update (ApplyGradientDescent with use_locking=False): concurrent connection threads update the same
// variable without synchronisation ON PURPOSE, so the function is excluded from ThreadSanitizer instrumentation.
__attribute__((no_sanitize("thread"))) void hogwild_apply(float* p, const float* g, size_t n, float lr) {
  for (size_t i = 0; i < n; ++i) p[i] -= lr * g[i];
}

void respond(int fd, uint32_t status, uint32_t aux, int64_t value, const void* payload, uint64_t nbytes) {
  RespHeader h{status, aux, value, nbytes};
  if (!write_full(fd, &h, sizeof(h))) return;
  if (nbytes) write_full(fd, payload, nbytes);
}

void serve_connection(Server* s, int fd) {
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
  std::vector<char> buf;
  for (;;) {
    ReqHeader h;
    if (!read_full(fd, &h, sizeof(h)) || h.magic != kMagic) break;
    // Sizes come from the wire: nothing is allocated or read before they have been checked against what the
    // operation can legitimately carry (a variable's bytes, a name), and a request that fails the check ends
    // the connection - the stream cannot be resynchronised behind a payload that is not going to be read.
    Variable* v = nullptr;
    const bool needs_var = h.op == OP_INIT || h.op == OP_PULL || h.op == OP_PUSH_ASYNC || h.op == OP_ACC_APPLY ||
                           h.op == OP_TAKE_APPLY || h.op == OP_ACC_SET_STEP;
    if (needs_var) {
      std::lock_guard<std::mutex> lk(s->mu);
      if (h.var < s->vars.size()) v = s->vars[h.var].get();
    }
    uint64_t max_payload = 0;
    if (h.op == OP_REGISTER) max_payload = kMaxNameBytes;
    else if (v && (h.op == OP_INIT || h.op == OP_PUSH_ASYNC || h.op == OP_ACC_APPLY)) max_payload = v->value.size() * sizeof(float);
    if (h.nbytes > max_payload) {
      respond(fd, 3, 0, 0, nullptr, 0);
      break;
    }
    buf.resize(h.nbytes);
    if (h.nbytes && !read_full(fd, buf.data(), h.nbytes)) break;
    if (needs_var && !v) {
      respond(fd, 2, 0, 0, nullptr, 0);
      continue;
    }
    switch (h.op) {
      case OP_REGISTER: {
        std::string name(buf.data(), buf.size());
        std::lock_guard<std::mutex> lk(s->mu);
        auto it = s->by_name.find(name);
        uint32_t id;
        if (it == s->by_name.end()) {
          if (h.value <= 0 || h.value > kMaxVariableElems || name.empty() || s->vars.size() >= kMaxVariables) {
            respond(fd, 3, 0, 0, nullptr, 0);
            break;
          }
          id = uint32_t(s->vars.size());
          std::unique_ptr<Variable> nv(new Variable);
          nv->name = name;
          try {
            nv->value.assign(size_t(h.value), 0.f);
            nv->acc_sum.assign(size_t(h.value), 0.f);
          } catch (const std::exception&) {         // out of memory: an error status, not a dead ps
            respond(fd, 5, 0, 0, nullptr, 0);
            break;
          }
          s->vars.push_back(std::move(nv));
          s->by_name[name] = id;
        } else {
          id = it->second;
          if (int64_t(s->vars[id]->value.size()) != h.value) {
            respond(fd, 3, 0, 0, nullptr, 0);
            break;
          }
        }
        respond(fd, 0, 0, id, nullptr, 0);
        break;
      }
      case OP_INIT: {
        if (h.nbytes != v->value.size() * sizeof(float)) { respond(fd, 3, 0, 0, nullptr, 0); break; }
        memcpy(v->value.data(), buf.data(), h.nbytes);
        {
          std::lock_guard<std::mutex> lk(s->mu);
          v->initialized = true;
        }
        respond(fd, 0, 0, 0, nullptr, 0);
        break;
      }
      case OP_PULL: {
        s->n_pull++;
        respond(fd, 0, 0, s->global_step.load(), v->value.data(), v->value.size() * sizeof(float));
        break;
      }
      case OP_PUSH_ASYNC: {
        if (h.nbytes != v->value.size() * sizeof(float)) { respond(fd, 3, 0, 0, nullptr, 0); break; }
        const float lr = float(h.scalar);
        const float* g = reinterpret_cast<const float*>(buf.data());
        float* p = v->value.data();
        const size_t n = v->value.size();
        hogwild_apply(p, g, n, lr);
        s->n_push++;
        respond(fd, 0, 0, s->global_step.load(), nullptr, 0);
        break;
      }
      case OP_INC_STEP: {
        s->n_updates++;
        respond(fd, 0, 0, s->global_step.fetch_add(h.value ? h.value : 1) + (h.value ? h.value : 1), nullptr, 0);
        break;
      }
      case OP_GET_STEP: respond(fd, 0, 0, s->global_step.load(), nullptr, 0); break;
      case OP_SET_STEP: s->global_step.store(h.value); respond(fd, 0, 0, h.value, nullptr, 0); break;
      case OP_ACC_APPLY: {
        if (h.nbytes != v->value.size() * sizeof(float)) { respond(fd, 3, 0, 0, nullptr, 0); break; }
        uint32_t accepted = 0;
        {
          std::lock_guard<std::mutex> lk(v->acc_mu);
          if (h.value >= v->acc_step) {
            const float* g = reinterpret_cast<const float*>(buf.data());
            float* a = v->acc_sum.data();
            const size_t n = v->acc_sum.size();
            for (size_t i = 0; i < n; ++i) a[i] += g[i];
            v->acc_count++;
            accepted = 1;
          }
        }
        if (accepted) v->acc_cv.notify_all(); else s->n_dropped++;
        s->n_push++;
        respond(fd, 0, accepted, 0, nullptr, 0);
        break;
      }
      case OP_TAKE_APPLY: {
        std::unique_lock<std::mutex> lk(v->acc_mu);
        v->acc_cv.wait(lk, [&] { return v->acc_count >= h.value || s->stopping.load(); });
        if (s->stopping.load()) { lk.unlock(); respond(fd, 4, 0, 0, nullptr, 0); break; }
        const int64_t cnt = v->acc_count;
        const float scale = float(h.scalar) / float(cnt);  // mean of everything accumulated
        float* p = v->value.data();
        float* a = v->acc_sum.data();
        const size_t n = v->value.size();
        for (size_t i = 0; i < n; ++i) {
          p[i] -= scale * a[i];
          a[i] = 0.f;
        }
        v->acc_count = 0;
        lk.unlock();
        respond(fd, 0, uint32_t(cnt), 0, nullptr, 0);
        break;
      }
      case OP_ACC_SET_STEP: {
        {
          std::lock_guard<std::mutex> lk(v->acc_mu);
          if (h.value > v->acc_step) v->acc_step = h.value;
        }
        respond(fd, 0, 0, 0, nullptr, 0);
        break;
      }
      case OP_TOKEN_ENQUEUE: {
        if (h.aux > kMaxTokensPerCall) { respond(fd, 3, 0, 0, nullptr, 0); break; }
        {
          std::lock_guard<std::mutex> lk(s->tok_mu);
          for (uint32_t i = 0; i < h.aux; ++i) s->tokens.push_back(h.value);
        }
        s->tok_cv.notify_all();
        respond(fd, 0, 0, 0, nullptr, 0);
        break;
      }
      case OP_TOKEN_DEQUEUE: {
        std::unique_lock<std::mutex> lk(s->tok_mu);
        s->tok_cv.wait(lk, [&] { return !s->tokens.empty() || s->stopping.load(); });
        if (s->tokens.empty()) { lk.unlock(); respond(fd, 4, 0, 0, nullptr, 0); break; }
        const int64_t t = s->tokens.front();
        s->tokens.pop_front();
        lk.unlock();
        respond(fd, 0, 0, t, nullptr, 0);
        break;
      }
      case OP_SET_READY: {
        { std::lock_guard<std::mutex> lk(s->mu); s->ready = true; }
        respond(fd, 0, 0, 0, nullptr, 0);
        break;
      }
      case OP_IS_READY: {
        bool r;
        {
          std::lock_guard<std::mutex> lk(s->mu);
          r = s->ready;
          for (auto& var : s->vars) r = r && var->initialized;
        }
        respond(fd, 0, r ? 1 : 0, s->global_step.load(), nullptr, 0);
        break;
      }
      case OP_WORKER_DONE: respond(fd, 0, 0, ++s->n_done, nullptr, 0); break;
      case OP_STATS: {
        int64_t st[5] = {s->n_push.load(), s->n_pull.load(), s->n_dropped.load(), s->n_updates.load(), s->n_done.load()};
        respond(fd, 0, 0, s->global_step.load(), st, sizeof(st));
        break;
      }
      case OP_SHUTDOWN: {
        respond(fd, 0, 0, 0, nullptr, 0);
        {
          std::lock_guard<std::mutex> lk(s->mu);
          s->shutdown_requested = true;
        }
        s->cv.notify_all();
        break;
      }
      default: respond(fd, 1, 0, 0, nullptr, 0); break;
    }
  }
  {
    // forget the descriptor before closing it: stop() must never shut down a number the process has reused since
    std::lock_guard<std::mutex> lk(s->mu);
    for (size_t i = 0; i < s->conn_fds.size(); ++i)
      if (s->conn_fds[i] == fd) {
        s->conn_fds[i] = s->conn_fds.back();
        s->conn_fds.pop_back();
        break;
      }
  }
  ::close(fd);
}

void accept_loop(Server* s) {
  for (;;) {
    sockaddr_in addr;
    socklen_t len = sizeof(addr);
    int fd = ::accept(s->listen_fd, reinterpret_cast<sockaddr*>(&addr), &len);
    if (fd < 0) {
      if (s->stopping.load()) return;
      continue;
    }
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->stopping.load()) { ::close(fd); return; }
    s->conn_fds.push_back(fd);
    s->conn_threads.emplace_back(serve_connection, s, fd);
  }
}

struct Client {
  int fd = -1;
  std::mutex mu;
};

}  // namespace

// `bind_host`: dotted IPv4 address to listen on (the task's own entry of --ps_hosts); null or empty = the loopback
// interface only.  The service has no authentication: it is reachable exactly where the cluster description says a ps
// task lives ("0.0.0.0" must be asked for by name).
DTF_RT_EXPORT void* dtf_ps_server_start_on(const char* bind_host, int port, int* actual_port) {
  std::unique_ptr<Server> s(new Server);
  s->listen_fd = ::socket(AF_INET, SOCK_STREAM, 0);
  if (s->listen_fd < 0) return nullptr;
  int one = 1;
  setsockopt(s->listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
  sockaddr_in addr{};
  addr.sin_family = AF_INET;
  addr.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
  if (bind_host && *bind_host && inet_pton(AF_INET, bind_host, &addr.sin_addr) != 1) {
    ::close(s->listen_fd);
    return nullptr;
  }
  addr.sin_port = htons(uint16_t(port));
  if (::bind(s->listen_fd, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) < 0 || ::listen(s->listen_fd, 128) < 0) {
    ::close(s->listen_fd);
    return nullptr;
  }
  socklen_t len = sizeof(addr);
  getsockname(s->listen_fd, reinterpret_cast<sockaddr*>(&addr), &len);
  s->port = ntohs(addr.sin_port);
  if (actual_port) *actual_port = s->port;
  s->accept_thread = std::thread(accept_loop, s.get());
  return s.release();
}

DTF_RT_EXPORT void* dtf_ps_server_start(int port, int* actual_port) { return dtf_ps_server_start_on(nullptr, port, actual_port); }

// Blocks until a client sends SHUTDOWN (the counterpart of server.join(), which the reference never leaves).
// Returns 1 when shut down, 0 on timeout (timeout_s < 0 waits forever).
DTF_RT_EXPORT int dtf_ps_server_wait(void* handle, double timeout_s) {
  Server* s = static_cast<Server*>(handle);
  std::unique_lock<std::mutex> lk(s->mu);
  if (timeout_s < 0) {
    s->cv.wait(lk, [&] { return s->shutdown_requested; });
    return 1;
  }
  return s->cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return s->shutdown_requested; }) ? 1 : 0;
}

DTF_RT_EXPORT void dtf_ps_server_stop(void* handle) {
  std::unique_ptr<Server> s(static_cast<Server*>(handle));
  s->stopping.store(true);
  ::shutdown(s->listen_fd, SHUT_RDWR);
  ::close(s->listen_fd);
  {
    std::lock_guard<std::mutex> lk(s->mu);
    for (int fd : s->conn_fds) ::shutdown(fd, SHUT_RDWR);
    for (auto& v : s->vars) v->acc_cv.notify_all();
  }
  s->tok_cv.notify_all();
  if (s->accept_thread.joinable()) s->accept_thread.join();
  for (auto& t : s->conn_threads)
    if (t.joinable()) t.join();
}

DTF_RT_EXPORT void* dtf_ps_connect(const char* host, int port, double timeout_s) {
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s);
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_INET;
  hints.ai_socktype = SOCK_STREAM;
  char portstr[16];
  snprintf(portstr, sizeof(portstr), "%d", port);
  for (;;) {
    if (getaddrinfo(host, portstr, &hints, &res) == 0 && res) {
      int fd = ::socket(res->ai_family, res->ai_socktype, res->ai_protocol);
      if (fd >= 0) {
        if (::connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
          int one = 1;
          setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
          freeaddrinfo(res);
          Client* c = new Client;
          c->fd = fd;
          return c;
        }
        ::close(fd);
      }
      freeaddrinfo(res);
      res = nullptr;
    }
    if (std::chrono::steady_clock::now() >= deadline) return nullptr;
    std::this_thread::sleep_for(std::chrono::milliseconds(50));   // the ps may not be up yet
  }
}

// One request/response exchange.  Returns 0 on success, >0 = server status, <0 = transport error.
DTF_RT_EXPORT int dtf_ps_call(void* handle, uint32_t op, uint32_t var, uint32_t aux, int64_t value, double scalar,
                              const void* payload, uint64_t nbytes, void* out, uint64_t out_cap, uint64_t* out_bytes,
                              int64_t* out_value, uint32_t* out_aux) {
  Client* c = static_cast<Client*>(handle);
  std::lock_guard<std::mutex> lk(c->mu);
  ReqHeader h{kMagic, op, var, aux, value, scalar, nbytes};
  if (!write_full(c->fd, &h, sizeof(h))) return -1;
  if (nbytes && !write_full(c->fd, payload, nbytes)) return -1;
  RespHeader r;
  if (!read_full(c->fd, &r, sizeof(r))) return -2;
  if (r.nbytes > out_cap) {
    std::vector<char> sink(r.nbytes);
    read_full(c->fd, sink.data(), r.nbytes);
    return -3;
  }
  if (r.nbytes && !read_full(c->fd, out, r.nbytes)) return -2;
  if (out_bytes) *out_bytes = r.nbytes;
  if (out_value) *out_value = r.value;
  if (out_aux) *out_aux = r.aux;
  return int(r.status);
}

DTF_RT_EXPORT void dtf_ps_close(void* handle) {
  Client* c = static_cast<Client*>(handle);
  if (!c) return;
  ::close(c->fd);
  delete c;
}
