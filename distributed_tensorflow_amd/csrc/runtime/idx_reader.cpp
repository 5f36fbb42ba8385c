// MNIST idx file decoding (optionally gzip-compressed) and batch assembly on the host.
//
// Counterpart of the loader behind input_data.read_data_sets / DataSet.next_batch
// (/root/reference/distributed.py:6, :38, :137): big-endian idx header
//   [0x00 0x00 <dtype> <ndim>] [dim0]...[dimN-1] (uint32 BE each) then the payload,
// dtype 0x08 = unsigned byte.  Images are scaled to float32 in [0, 1] and flattened, labels
// turned into one-hot float32 rows.  gather_rows assembles a batch from a row-index list
// (the permutation walk of next_batch) with plain memcpy, one thread per call.
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "dtf_runtime.h"

namespace {

bool read_all(const std::string& path, std::vector<unsigned char>* out) {
  // gzopen reads plain files transparently, so one code path serves .gz and raw idx files
  gzFile f = gzopen(path.c_str(), "rb");
  if (!f) return false;
  gzbuffer(f, 1 << 20);
  unsigned char buf[1 << 16];
  for (;;) {
    int n = gzread(f, buf, sizeof(buf));
    if (n < 0) {
      gzclose(f);
      return false;
    }
    if (n == 0) break;
    out->insert(out->end(), buf, buf + n);
  }
  gzclose(f);
  return true;
}

struct IdxFile {
  std::vector<unsigned char> bytes;
  std::vector<int64_t> dims;
  size_t payload = 0;
};

}  // namespace

// Opens and decodes the header.  Returns a handle or nullptr; err receives a message.
DTF_RT_EXPORT void* dtf_idx_open(const char* path, char* err, size_t errlen) {
  IdxFile* f = new IdxFile;
  auto fail = [&](const char* msg) -> void* {
    if (err && errlen) snprintf(err, errlen, "%s: %s", path, msg);
    delete f;
    return nullptr;
  };
  if (!read_all(path, &f->bytes)) return fail("cannot read file");
  if (f->bytes.size() < 4) return fail("shorter than an idx header");
  if (f->bytes[0] != 0 || f->bytes[1] != 0 || f->bytes[2] != 0x08) return fail("not an unsigned-byte idx file");
  const int ndim = f->bytes[3];
  if (ndim < 1 || ndim > 8 || f->bytes.size() < size_t(4 + 4 * ndim)) return fail("bad dimension count");
  size_t total = 1;
  for (int i = 0; i < ndim; ++i) {
    const unsigned char* p = &f->bytes[4 + 4 * i];
    const int64_t d = (int64_t(p[0]) << 24) | (int64_t(p[1]) << 16) | (int64_t(p[2]) << 8) | int64_t(p[3]);
    f->dims.push_back(d);
    total *= size_t(d);
  }
  f->payload = size_t(4 + 4 * ndim);
  if (f->bytes.size() - f->payload != total) return fail("payload size does not match the header");
  return f;
}

DTF_RT_EXPORT int dtf_idx_ndim(void* h) { return int(static_cast<IdxFile*>(h)->dims.size()); }
DTF_RT_EXPORT int64_t dtf_idx_dim(void* h, int i) { return static_cast<IdxFile*>(h)->dims[size_t(i)]; }
DTF_RT_EXPORT void dtf_idx_copy_u8(void* h, unsigned char* out) {
  IdxFile* f = static_cast<IdxFile*>(h);
  memcpy(out, f->bytes.data() + f->payload, f->bytes.size() - f->payload);
}
// images: uint8 -> float32 * (1/255), flattened [n, rows*cols]
DTF_RT_EXPORT void dtf_idx_copy_scaled_f32(void* h, float* out) {
  IdxFile* f = static_cast<IdxFile*>(h);
  const unsigned char* p = f->bytes.data() + f->payload;
  const size_t n = f->bytes.size() - f->payload;
  const float s = 1.0f / 255.0f;
  for (size_t i = 0; i < n; ++i) out[i] = float(p[i]) * s;
}
// labels: uint8 class ids -> one-hot float32 [n, classes]; returns the number of out-of-range labels
DTF_RT_EXPORT int64_t dtf_idx_copy_one_hot(void* h, float* out, int classes) {
  IdxFile* f = static_cast<IdxFile*>(h);
  const unsigned char* p = f->bytes.data() + f->payload;
  const size_t n = f->bytes.size() - f->payload;
  memset(out, 0, n * size_t(classes) * sizeof(float));
  int64_t bad = 0;
  for (size_t i = 0; i < n; ++i) {
    if (p[i] < classes) out[i * size_t(classes) + p[i]] = 1.0f;
    else ++bad;
  }
  return bad;
}
DTF_RT_EXPORT void dtf_idx_close(void* h) { delete static_cast<IdxFile*>(h); }

// out[i, :] = src[rows[i], :]   (row_bytes per row)
DTF_RT_EXPORT int dtf_gather_rows(const void* src, int64_t n_src_rows, const int32_t* rows, int64_t n_rows, size_t row_bytes,
                                  void* out) {
  const char* s = static_cast<const char*>(src);
  char* o = static_cast<char*>(out);
  for (int64_t i = 0; i < n_rows; ++i) {
    const int64_t r = rows[i];
    if (r < 0 || r >= n_src_rows) return 1;
    memcpy(o + size_t(i) * row_bytes, s + size_t(r) * row_bytes, row_bytes);
  }
  return 0;
}
