// Self-test of the host runtime, meant to be built with sanitizers (make -C csrc selftest-asan / selftest-tsan):
// checkpoint bundle write -> read, crc32c vector, idx.gz reader, and the parameter service under concurrent
// clients (registration, sync accumulators + token barrier, lock-free async pushes, shutdown).
// Exit code 0 = every check passed; sanitizer reports make the run fail through halt_on_error / exitcode.
#include <zlib.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../runtime/dtf_runtime.h"

extern "C" {
uint32_t dtf_crc32c(const void*, size_t);
uint32_t dtf_crc32c_masked(const void*, size_t);
void* dtf_bundle_writer_open(const char*);
int dtf_bundle_writer_add(void*, const char*, int, const int64_t*, int, const void*, size_t);
int dtf_bundle_writer_finish(void*);
void* dtf_bundle_reader_open(const char*, char*, size_t);
int dtf_bundle_reader_num_entries(void*);
int dtf_bundle_reader_read(void*, const char*, void*, size_t);
void dtf_bundle_reader_close(void*);
void* dtf_events_open(const char*, double);
int dtf_events_scalar(void*, double, int64_t, const char*, float);
int dtf_events_close(void*);
void* dtf_idx_open(const char*, char*, size_t);
int dtf_idx_ndim(void*);
int64_t dtf_idx_dim(void*, int);
void dtf_idx_copy_scaled_f32(void*, float*);
int64_t dtf_idx_copy_one_hot(void*, float*, int);
void dtf_idx_close(void*);
void* dtf_ps_server_start(int, int*);
int dtf_ps_server_wait(void*, double);
void dtf_ps_server_stop(void*);
void* dtf_ps_connect(const char*, int, double);
int dtf_ps_call(void*, uint32_t, uint32_t, uint32_t, int64_t, double, const void*, uint64_t, void*, uint64_t, uint64_t*, int64_t*,
                uint32_t*);
void dtf_ps_close(void*);
}

static int failures = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      ++failures;                                                          \
    }                                                                      \
  } while (0)

enum { OP_REGISTER = 1, OP_INIT, OP_PULL, OP_PUSH_ASYNC, OP_INC_STEP, OP_GET_STEP, OP_SET_STEP, OP_ACC_APPLY, OP_TAKE_APPLY,
       OP_ACC_SET_STEP, OP_TOKEN_ENQUEUE, OP_TOKEN_DEQUEUE, OP_SET_READY, OP_IS_READY, OP_SHUTDOWN, OP_STATS, OP_WORKER_DONE };

static void test_crc() {
  CHECK(dtf_crc32c("123456789", 9) == 0xE3069283u);          // the CRC-32C check value
  const uint32_t c = dtf_crc32c("123456789", 9);
  CHECK(dtf_crc32c_masked("123456789", 9) == (((c >> 15) | (c << 17)) + 0xa282ead8u));
}

static void test_events(const std::string& dir) {
  const std::string path = dir + "/events.out.tfevents.selftest";
  remove(path.c_str());
  void* w = dtf_events_open(path.c_str(), 1.5);
  CHECK(w != nullptr);
  CHECK(dtf_events_scalar(w, 2.5, 7, "global_step/sec", 3.25f) == 0);
  CHECK(dtf_events_scalar(w, 2.5, -1, "x", 0.f) != 0);          // negative step refused
  CHECK(dtf_events_close(w) == 0);
  FILE* f = fopen(path.c_str(), "rb");
  CHECK(f != nullptr);
  fseek(f, 0, SEEK_END);
  // version record: 12 + 24 + 4; scalar record: 12 + (9 + 2 + 2 + (2 + (2 + 15 + 5))) + 4
  CHECK(ftell(f) == 40 + 12 + 37 + 4);
  fclose(f);
}

static void test_bundle(const std::string& dir) {
  const std::string prefix = dir + "/model.ckpt-7";
  std::vector<float> w(784 * 5), b(5);
  for (size_t i = 0; i < w.size(); ++i) w[i] = std::sin(float(i));
  for (size_t i = 0; i < b.size(); ++i) b[i] = float(i) - 2.f;
  int32_t gs = 7;
  void* wr = dtf_bundle_writer_open(prefix.c_str());
  CHECK(wr != nullptr);
  const int64_t dw[2] = {784, 5}, db[1] = {5};
  CHECK(dtf_bundle_writer_add(wr, "hid_w", 1, dw, 2, w.data(), w.size() * 4) == 0);
  CHECK(dtf_bundle_writer_add(wr, "global_step", 3, nullptr, 0, &gs, 4) == 0);
  CHECK(dtf_bundle_writer_add(wr, "hid_b", 1, db, 1, b.data(), b.size() * 4) == 0);
  CHECK(dtf_bundle_writer_finish(wr) == 0);
  char err[256] = {0};
  void* rd = dtf_bundle_reader_open(prefix.c_str(), err, sizeof(err));
  CHECK(rd != nullptr);
  if (!rd) return;
  CHECK(dtf_bundle_reader_num_entries(rd) == 3);
  std::vector<float> w2(w.size());
  CHECK(dtf_bundle_reader_read(rd, "hid_w", w2.data(), w2.size() * 4) == 0);
  CHECK(std::memcmp(w.data(), w2.data(), w.size() * 4) == 0);
  int32_t gs2 = 0;
  CHECK(dtf_bundle_reader_read(rd, "global_step", &gs2, 4) == 0 && gs2 == 7);
  CHECK(dtf_bundle_reader_read(rd, "missing", &gs2, 4) != 0);
  CHECK(dtf_bundle_reader_read(rd, "hid_b", w2.data(), 8) != 0);      // wrong size is refused
  dtf_bundle_reader_close(rd);
  CHECK(dtf_bundle_reader_open((dir + "/nothing").c_str(), err, sizeof(err)) == nullptr);
}

static void put_be32(std::vector<unsigned char>& v, uint32_t x) {
  v.push_back(x >> 24); v.push_back((x >> 16) & 255); v.push_back((x >> 8) & 255); v.push_back(x & 255);
}

static void test_idx(const std::string& dir) {
  std::vector<unsigned char> img, lab;
  put_be32(img, 0x00000803); put_be32(img, 3); put_be32(img, 2); put_be32(img, 2);
  for (int i = 0; i < 12; ++i) img.push_back((unsigned char)(i * 20));
  put_be32(lab, 0x00000801); put_be32(lab, 3);
  lab.push_back(1); lab.push_back(0); lab.push_back(9);
  const std::string pi = dir + "/img-idx3-ubyte.gz", pl = dir + "/lab-idx1-ubyte.gz";
  gzFile f = gzopen(pi.c_str(), "wb"); gzwrite(f, img.data(), (unsigned)img.size()); gzclose(f);
  f = gzopen(pl.c_str(), "wb"); gzwrite(f, lab.data(), (unsigned)lab.size()); gzclose(f);
  char err[256] = {0};
  void* h = dtf_idx_open(pi.c_str(), err, sizeof(err));
  CHECK(h != nullptr);
  if (h) {
    CHECK(dtf_idx_ndim(h) == 3 && dtf_idx_dim(h, 0) == 3 && dtf_idx_dim(h, 2) == 2);
    float px[12];
    dtf_idx_copy_scaled_f32(h, px);
    CHECK(std::fabs(px[1] - 20.f / 255.f) < 1e-7f);
    dtf_idx_close(h);
  }
  h = dtf_idx_open(pl.c_str(), err, sizeof(err));
  CHECK(h != nullptr);
  if (h) {
    float oh[30];
    CHECK(dtf_idx_copy_one_hot(h, oh, 10) == 0);       // no out-of-range labels
    CHECK(oh[1] == 1.f && oh[10] == 1.f && oh[29] == 1.f && oh[0] == 0.f);
    dtf_idx_close(h);
  }
  CHECK(dtf_idx_open((dir + "/absent.gz").c_str(), err, sizeof(err)) == nullptr);
}

static int call(void* c, uint32_t op, uint32_t var, uint32_t aux, int64_t value, double scalar, const void* payload, uint64_t nbytes,
                void* out = nullptr, uint64_t cap = 0, int64_t* oval = nullptr, uint32_t* oaux = nullptr) {
  uint64_t ob = 0;
  return dtf_ps_call(c, op, var, aux, value, scalar, payload, nbytes, out, cap, &ob, oval, oaux);
}

static void test_ps_service() {
  constexpr int W = 4, N = 257, STEPS = 12;
  int port = 0;
  void* srv = dtf_ps_server_start(0, &port);
  CHECK(srv != nullptr && port > 0);
  if (!srv) return;
  std::atomic<int> errors{0};
  auto worker = [&](int rank) {
    void* c = dtf_ps_connect("127.0.0.1", port, 10.0);
    if (!c) { ++errors; return; }
    int64_t id_sync = -1, id_async = -1, v = 0;
    uint32_t aux = 0;
    if (call(c, OP_REGISTER, 0, 0, N, 0, "sync_w", 6, nullptr, 0, &id_sync) != 0) ++errors;
    if (call(c, OP_REGISTER, 0, 0, N, 0, "async_w", 7, nullptr, 0, &id_async) != 0) ++errors;
    std::vector<float> buf(N, 0.f), g(N, 1.f);
    if (rank == 0) {   // chief: initialise, mark ready, prime the token queue
      if (call(c, OP_INIT, (uint32_t)id_sync, 0, 0, 0, buf.data(), N * 4) != 0) ++errors;
      if (call(c, OP_INIT, (uint32_t)id_async, 0, 0, 0, buf.data(), N * 4) != 0) ++errors;
      if (call(c, OP_SET_READY, 0, 0, 0, 0, nullptr, 0) != 0) ++errors;
      if (call(c, OP_TOKEN_ENQUEUE, 0, W, 1, 0, nullptr, 0) != 0) ++errors;
    }
    for (;;) {         // everybody polls readiness
      if (call(c, OP_IS_READY, 0, 0, 0, 0, nullptr, 0, nullptr, 0, &v, &aux) != 0) { ++errors; break; }
      if (aux) break;
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
    void* chief = rank == 0 ? dtf_ps_connect("127.0.0.1", port, 10.0) : nullptr;   // the chief's aggregation loop
    std::thread agg;
    if (rank == 0) {
      agg = std::thread([&, chief] {
        for (int s = 0; s < STEPS; ++s) {
          uint32_t cnt = 0;
          int64_t gs = 0;
          if (call(chief, OP_TAKE_APPLY, (uint32_t)id_sync, 0, W, 0.5, nullptr, 0, nullptr, 0, nullptr, &cnt) != 0 || cnt != (uint32_t)W) ++errors;
          if (call(chief, OP_INC_STEP, 0, 0, 1, 0, nullptr, 0, nullptr, 0, &gs) != 0) ++errors;
          if (call(chief, OP_ACC_SET_STEP, (uint32_t)id_sync, 0, gs, 0, nullptr, 0) != 0) ++errors;
          if (call(chief, OP_TOKEN_ENQUEUE, 0, W, gs, 0, nullptr, 0) != 0) ++errors;
        }
      });
    }
    // The worker loop of the sync mode: take a token (stamped with the global step it was issued for), pull,
    // contribute one gradient stamped with that step; leave when the token says training is over.  A fast
    // worker may take two tokens of one round, exactly as with the reference's token queue.
    for (;;) {
      int64_t tok = 0;
      if (call(c, OP_TOKEN_DEQUEUE, 0, 0, 0, 0, nullptr, 0, nullptr, 0, &tok) != 0) { ++errors; break; }
      if (tok >= 1 + STEPS) break;
      if (call(c, OP_PULL, (uint32_t)id_sync, 0, 0, 0, nullptr, 0, buf.data(), N * 4) != 0) ++errors;
      // every aggregated update subtracts lr * mean(gradient) = 0.5; at least tok-1 updates happened before the token
      const float k = -buf[N - 1] / 0.5f;
      if (std::fabs(k - std::round(k)) > 1e-4f || k < float(tok - 1) - 1e-4f || buf[0] != buf[N - 1]) ++errors;
      if (call(c, OP_ACC_APPLY, (uint32_t)id_sync, 0, tok, 0, g.data(), N * 4, nullptr, 0, nullptr, &aux) != 0) ++errors;
      // a gradient stamped before the accumulator's step must be dropped
      if (tok > 1 && rank == 1) {
        if (call(c, OP_ACC_APPLY, (uint32_t)id_sync, 0, 0, 0, g.data(), N * 4, nullptr, 0, nullptr, &aux) != 0 || aux != 0) ++errors;
      }
      // Hogwild traffic on the other variable, concurrently from every worker
      if (call(c, OP_PUSH_ASYNC, (uint32_t)id_async, 0, 0, 0.25, g.data(), N * 4) != 0) ++errors;
      if (call(c, OP_PULL, (uint32_t)id_async, 0, 0, 0, nullptr, 0, buf.data(), N * 4) != 0) ++errors;
    }
    if (agg.joinable()) agg.join();
    if (chief) dtf_ps_close(chief);
    if (call(c, OP_WORKER_DONE, 0, 0, 0, 0, nullptr, 0, nullptr, 0, &v) != 0) ++errors;
    dtf_ps_close(c);
  };
  std::vector<std::thread> th;
  for (int r = 0; r < W; ++r) th.emplace_back(worker, r);
  for (auto& t : th) t.join();
  CHECK(errors.load() == 0);
  void* c = dtf_ps_connect("127.0.0.1", port, 10.0);
  CHECK(c != nullptr);
  if (c) {
    int64_t gs = 0, st[5] = {0};
    CHECK(call(c, OP_GET_STEP, 0, 0, 0, 0, nullptr, 0, nullptr, 0, &gs) == 0 && gs == 1 + STEPS);
    CHECK(call(c, OP_STATS, 0, 0, 0, 0, nullptr, 0, st, sizeof(st)) == 0);
    CHECK(st[2] >= 1);                              // stale gradients were dropped
    CHECK(st[4] == W);                              // workers done
    CHECK(call(c, 99, 0, 0, 0, 0, nullptr, 0) == 1);  // unknown op
    CHECK(call(c, OP_PULL, 77, 0, 0, 0, nullptr, 0) == 2);   // unknown variable
    CHECK(call(c, OP_SHUTDOWN, 0, 0, 0, 0, nullptr, 0) == 0);
    dtf_ps_close(c);
  }
  CHECK(dtf_ps_server_wait(srv, 5.0) == 1);
  dtf_ps_server_stop(srv);
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  test_crc();
  test_bundle(dir);
  test_events(dir);
  test_idx(dir);
  test_ps_service();
  if (failures) {
    std::fprintf(stderr, "runtime selftest: %d check(s) failed\n", failures);
    return 1;
  }
  std::puts("runtime selftest: ok");
  return 0;
}
