// TensorBoard event files: what tf.train.Supervisor leaves in its logdir besides checkpoints
// (the `global_step/sec` summary written because `global_step=` was passed,
// /root/reference/distributed.py:110-111).
//
// File:   <logdir>/events.out.tfevents.<unix seconds>.<hostname>
// Record: uint64 length (LE) | masked crc32c(length) | data | masked crc32c(data)        (TFRecord framing)
// Data:   Event protobuf:  1: wall_time (double)   2: step (int64)   3: file_version (string, first record:
//         "brain.Event:2")   5: summary { 1: value { 1: tag (string)  2: simple_value (float) } }
#include <cstdio>
#include <cstring>
#include <string>

#include "dtf_runtime.h"

DTF_RT_EXPORT uint32_t dtf_crc32c(const void* data, size_t n);

namespace {

uint32_t mask(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xa282ead8u; }

void put_varint(std::string* s, uint64_t v) {
  while (v >= 0x80) {
    s->push_back(static_cast<char>(v | 0x80));
    v >>= 7;
  }
  s->push_back(static_cast<char>(v));
}
void put_bytes_field(std::string* s, int field, const std::string& b) {
  put_varint(s, (uint64_t)field << 3 | 2);
  put_varint(s, b.size());
  s->append(b);
}

struct Writer {
  FILE* f = nullptr;
};

bool write_record(Writer* w, const std::string& data) {
  uint64_t len = data.size();
  char head[12];
  memcpy(head, &len, 8);
  const uint32_t hc = mask(dtf_crc32c(head, 8));
  memcpy(head + 8, &hc, 4);
  const uint32_t dc = mask(dtf_crc32c(data.data(), data.size()));
  return fwrite(head, 1, 12, w->f) == 12 && fwrite(data.data(), 1, data.size(), w->f) == data.size() &&
         fwrite(&dc, 1, 4, w->f) == 4 && fflush(w->f) == 0;
}

std::string event_head(double wall_time, int64_t step) {
  std::string e;
  put_varint(&e, 1 << 3 | 1);
  e.append(reinterpret_cast<const char*>(&wall_time), 8);
  if (step) {
    put_varint(&e, 2 << 3 | 0);
    put_varint(&e, static_cast<uint64_t>(step));
  }
  return e;
}

}  // namespace

// Opens (appends to) an event file and, for a new file, writes the version record.
DTF_RT_EXPORT void* dtf_events_open(const char* path, double wall_time) {
  if (!path) return nullptr;
  FILE* probe = fopen(path, "rb");
  const bool fresh = probe == nullptr;
  if (probe) fclose(probe);
  FILE* f = fopen(path, "ab");
  if (!f) return nullptr;
  Writer* w = new Writer;
  w->f = f;
  if (fresh) {
    std::string e = event_head(wall_time, 0);
    put_bytes_field(&e, 3, "brain.Event:2");
    if (!write_record(w, e)) {
      fclose(f);
      delete w;
      return nullptr;
    }
  }
  return w;
}

DTF_RT_EXPORT int dtf_events_scalar(void* handle, double wall_time, int64_t step, const char* tag, float value) {
  Writer* w = static_cast<Writer*>(handle);
  if (!w || !w->f || !tag || step < 0) return 1;
  std::string v;
  put_bytes_field(&v, 1, tag);
  put_varint(&v, 2 << 3 | 5);
  v.append(reinterpret_cast<const char*>(&value), 4);
  std::string summary;
  put_bytes_field(&summary, 1, v);
  std::string e = event_head(wall_time, step);
  put_bytes_field(&e, 5, summary);
  return write_record(w, e) ? 0 : 2;
}

DTF_RT_EXPORT int dtf_events_close(void* handle) {
  Writer* w = static_cast<Writer*>(handle);
  if (!w) return 1;
  const int rc = w->f ? fclose(w->f) : 0;
  delete w;
  return rc == 0 ? 0 : 2;
}
