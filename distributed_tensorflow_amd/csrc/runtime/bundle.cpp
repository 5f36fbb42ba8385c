// Tensor-bundle checkpoint I/O in the on-disk layout of TensorFlow's Saver V2
// (the default Saver that tf.train.Supervisor builds for
// /root/reference/distributed.py:110-111):
//
//   <prefix>.index                 sorted string table (leveldb table format, no compression)
//        ""     -> BundleHeaderProto { num_shards = 1, endianness = LITTLE, version { producer = 1 } }
//        <name> -> BundleEntryProto  { dtype, shape, shard_id, offset, size, crc32c }
//   <prefix>.data-00000-of-00001   raw little-endian tensor bytes, concatenated in key order
//
// Table format: data blocks of prefix-compressed entries
//   (varint shared, varint non_shared, varint value_len, key suffix, value),
// a restart array (uint32 offsets, every 16 entries) + its count, and a 5-byte block trailer
// (compression type 0 + masked crc32c); then an empty metaindex block, an index block (restart
// interval 1) of BlockHandles, and a 48-byte footer ending in magic 0xdb4775248b80fb57.
// Tensor checksums and block checksums use masked CRC32C: ((c >> 15) | (c << 17)) + 0xa282ead8.
//
// Written from the format description; no TensorFlow code is involved.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "dtf_runtime.h"

namespace {

// ------------------------------------------------------------------ crc32c (Castagnoli, reflected 0x82F63B78)
struct Crc32cTable {
  uint32_t t[8][256];
  Crc32cTable() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xff];
  }
};
const Crc32cTable& crc_table() {
  static Crc32cTable tab;
  return tab;
}

uint32_t crc32c_extend(uint32_t crc, const uint8_t* p, size_t n) {
  const Crc32cTable& T = crc_table();
  uint32_t c = crc ^ 0xffffffffu;
  while (n >= 8) {  // slicing-by-8
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = T.t[7][lo & 0xff] ^ T.t[6][(lo >> 8) & 0xff] ^ T.t[5][(lo >> 16) & 0xff] ^ T.t[4][lo >> 24] ^
        T.t[3][hi & 0xff] ^ T.t[2][(hi >> 8) & 0xff] ^ T.t[1][(hi >> 16) & 0xff] ^ T.t[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = T.t[0][(c ^ *p++) & 0xff] ^ (c >> 8);
  return c ^ 0xffffffffu;
}
inline uint32_t crc_mask(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xa282ead8u; }

// ------------------------------------------------------------------ varint / protobuf helpers
void put_varint(std::string* s, uint64_t v) {
  while (v >= 0x80) {
    s->push_back(static_cast<char>(v | 0x80));
    v >>= 7;
  }
  s->push_back(static_cast<char>(v));
}
void put_fixed32(std::string* s, uint32_t v) {
  char b[4] = {char(v), char(v >> 8), char(v >> 16), char(v >> 24)};
  s->append(b, 4);
}
void put_fixed64(std::string* s, uint64_t v) {
  put_fixed32(s, uint32_t(v));
  put_fixed32(s, uint32_t(v >> 32));
}
bool get_varint(const uint8_t*& p, const uint8_t* end, uint64_t* v) {
  uint64_t r = 0;
  for (int shift = 0; shift <= 63 && p < end; shift += 7) {
    uint8_t b = *p++;
    r |= uint64_t(b & 0x7f) << shift;
    if (!(b & 0x80)) {
      *v = r;
      return true;
    }
  }
  return false;
}
uint32_t get_fixed32(const uint8_t* p) { return uint32_t(p[0]) | uint32_t(p[1]) << 8 | uint32_t(p[2]) << 16 | uint32_t(p[3]) << 24; }

void pb_varint_field(std::string* s, int field, uint64_t v) {
  put_varint(s, uint64_t(field) << 3 | 0);
  put_varint(s, v);
}
void pb_bytes_field(std::string* s, int field, const std::string& v) {
  put_varint(s, uint64_t(field) << 3 | 2);
  put_varint(s, v.size());
  s->append(v);
}

struct Entry {
  std::string name;
  int dtype = 0;
  std::vector<int64_t> dims;
  int shard_id = 0;
  uint64_t offset = 0, size = 0;
  uint32_t crc = 0;  // masked
  std::string data;  // writer only
};

std::string encode_header() {
  std::string h;
  pb_varint_field(&h, 1, 1);  // num_shards = 1; endianness LITTLE (= 0) is the proto3 default and is not emitted
  std::string version;
  pb_varint_field(&version, 1, 1);  // VersionDef.producer = 1
  pb_bytes_field(&h, 3, version);
  return h;
}

std::string encode_entry(const Entry& e) {
  std::string s;
  if (e.dtype) pb_varint_field(&s, 1, uint64_t(e.dtype));
  std::string shape;  // TensorShapeProto: repeated Dim dim = 2 { int64 size = 1 }
  for (int64_t d : e.dims) {
    std::string dim;
    if (d) pb_varint_field(&dim, 1, uint64_t(d));
    pb_bytes_field(&shape, 2, dim);
  }
  pb_bytes_field(&s, 2, shape);  // present even when empty (scalar): the field is explicitly set
  if (e.shard_id) pb_varint_field(&s, 3, uint64_t(e.shard_id));
  if (e.offset) pb_varint_field(&s, 4, e.offset);
  if (e.size) pb_varint_field(&s, 5, e.size);
  put_varint(&s, 6 << 3 | 5);  // crc32c: fixed32
  put_fixed32(&s, e.crc);
  return s;
}

bool skip_field(const uint8_t*& p, const uint8_t* end, int wire) {
  uint64_t v;
  switch (wire) {
    case 0: return get_varint(p, end, &v);
    case 1: if (end - p < 8) return false; p += 8; return true;
    case 2: if (!get_varint(p, end, &v) || uint64_t(end - p) < v) return false; p += v; return true;
    case 5: if (end - p < 4) return false; p += 4; return true;
    default: return false;
  }
}

bool decode_shape(const uint8_t* p, const uint8_t* end, std::vector<int64_t>* dims) {
  while (p < end) {
    uint64_t tag;
    if (!get_varint(p, end, &tag)) return false;
    if ((tag >> 3) == 2 && (tag & 7) == 2) {
      uint64_t len;
      if (!get_varint(p, end, &len) || uint64_t(end - p) < len) return false;
      const uint8_t* q = p;
      const uint8_t* qe = p + len;
      int64_t size = 0;
      while (q < qe) {
        uint64_t t2;
        if (!get_varint(q, qe, &t2)) return false;
        if ((t2 >> 3) == 1 && (t2 & 7) == 0) {
          uint64_t v;
          if (!get_varint(q, qe, &v)) return false;
          size = int64_t(v);
        } else if (!skip_field(q, qe, int(t2 & 7))) {
          return false;
        }
      }
      dims->push_back(size);
      p = qe;
    } else if (!skip_field(p, end, int(tag & 7))) {
      return false;
    }
  }
  return true;
}

bool decode_entry(const std::string& v, Entry* e) {
  const uint8_t* p = reinterpret_cast<const uint8_t*>(v.data());
  const uint8_t* end = p + v.size();
  while (p < end) {
    uint64_t tag, x;
    if (!get_varint(p, end, &tag)) return false;
    const int field = int(tag >> 3), wire = int(tag & 7);
    if (field == 1 && wire == 0) { if (!get_varint(p, end, &x)) return false; e->dtype = int(x); }
    else if (field == 2 && wire == 2) {
      if (!get_varint(p, end, &x) || uint64_t(end - p) < x) return false;
      if (!decode_shape(p, p + x, &e->dims)) return false;
      p += x;
    }
    else if (field == 3 && wire == 0) { if (!get_varint(p, end, &x)) return false; e->shard_id = int(x); }
    else if (field == 4 && wire == 0) { if (!get_varint(p, end, &x)) return false; e->offset = x; }
    else if (field == 5 && wire == 0) { if (!get_varint(p, end, &x)) return false; e->size = x; }
    else if (field == 6 && wire == 5) { if (end - p < 4) return false; e->crc = get_fixed32(p); p += 4; }
    else if (!skip_field(p, end, wire)) return false;
  }
  return true;
}

// ------------------------------------------------------------------ table blocks
class BlockBuilder {
 public:
  explicit BlockBuilder(int restart_interval) : interval_(restart_interval) { restarts_.push_back(0); }
  void add(const std::string& key, const std::string& value) {
    size_t shared = 0;
    if (counter_ < interval_) {
      const size_t lim = std::min(last_key_.size(), key.size());
      while (shared < lim && last_key_[shared] == key[shared]) ++shared;
    } else {
      restarts_.push_back(uint32_t(buf_.size()));
      counter_ = 0;
    }
    put_varint(&buf_, shared);
    put_varint(&buf_, key.size() - shared);
    put_varint(&buf_, value.size());
    buf_.append(key, shared, std::string::npos);
    buf_.append(value);
    last_key_ = key;
    ++counter_;
  }
  std::string finish() {
    for (uint32_t r : restarts_) put_fixed32(&buf_, r);
    put_fixed32(&buf_, uint32_t(restarts_.size()));
    return buf_;
  }
  bool empty() const { return buf_.empty(); }
  size_t size_estimate() const { return buf_.size() + restarts_.size() * 4 + 4; }

 private:
  int interval_;
  int counter_ = 0;
  std::string buf_, last_key_;
  std::vector<uint32_t> restarts_;
};

struct BlockHandle {
  uint64_t offset = 0, size = 0;
  void encode(std::string* s) const { put_varint(s, offset); put_varint(s, size); }
};

BlockHandle write_block(std::string* file, const std::string& contents) {
  BlockHandle h;
  h.offset = file->size();
  h.size = contents.size();
  file->append(contents);
  const char type = 0;  // no compression
  uint32_t crc = crc32c_extend(0, reinterpret_cast<const uint8_t*>(contents.data()), contents.size());
  crc = crc32c_extend(crc, reinterpret_cast<const uint8_t*>(&type), 1);
  file->push_back(type);
  put_fixed32(file, crc_mask(crc));
  return h;
}

void short_successor(std::string* key) {
  for (size_t i = 0; i < key->size(); ++i) {
    const uint8_t b = uint8_t((*key)[i]);
    if (b != 0xff) {
      (*key)[i] = char(b + 1);
      key->resize(i + 1);
      return;
    }
  }
}

std::string shortest_separator(const std::string& start, const std::string& limit) {
  size_t min_len = std::min(start.size(), limit.size()), diff = 0;
  while (diff < min_len && start[diff] == limit[diff]) ++diff;
  if (diff >= min_len) return start;
  const uint8_t b = uint8_t(start[diff]);
  if (b < 0xff && b + 1 < uint8_t(limit[diff])) {
    std::string r = start.substr(0, diff + 1);
    r[diff] = char(b + 1);
    return r;
  }
  return start;
}

constexpr uint64_t kTableMagic = 0xdb4775248b80fb57ull;
constexpr size_t kBlockSize = 262144;

std::string build_table(const std::vector<std::pair<std::string, std::string>>& kv) {
  std::string file;
  BlockBuilder index(1);
  std::unique_ptr<BlockBuilder> data(new BlockBuilder(16));
  std::string last_key;
  bool pending = false;
  BlockHandle pending_handle;
  for (const auto& e : kv) {
    if (pending) {
      std::string sep = shortest_separator(last_key, e.first);
      std::string enc;
      pending_handle.encode(&enc);
      index.add(sep, enc);
      pending = false;
    }
    data->add(e.first, e.second);
    last_key = e.first;
    if (data->size_estimate() >= kBlockSize) {
      pending_handle = write_block(&file, data->finish());
      data.reset(new BlockBuilder(16));
      pending = true;
    }
  }
  if (!data->empty()) {
    pending_handle = write_block(&file, data->finish());
    pending = true;
  }
  BlockBuilder meta(16);
  BlockHandle meta_handle = write_block(&file, meta.finish());
  if (pending) {
    short_successor(&last_key);
    std::string enc;
    pending_handle.encode(&enc);
    index.add(last_key, enc);
  }
  BlockHandle index_handle = write_block(&file, index.finish());
  std::string footer;
  meta_handle.encode(&footer);
  index_handle.encode(&footer);
  footer.resize(40, '\0');
  put_fixed64(&footer, kTableMagic);
  file.append(footer);
  return file;
}

bool read_file(const std::string& path, std::string* out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  f.seekg(0, std::ios::end);
  std::streamoff n = f.tellg();
  f.seekg(0);
  out->resize(size_t(n));
  f.read(&(*out)[0], n);
  return bool(f);
}

bool write_file_atomic(const std::string& path, const std::string& data) {
  const std::string tmp = path + ".tempstate";
  {
    std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
    if (!f) return false;
    f.write(data.data(), std::streamsize(data.size()));
    if (!f) return false;
  }
  return std::rename(tmp.c_str(), path.c_str()) == 0;
}

// Reads a block, verifies its trailer, returns the contents.
bool read_block(const std::string& file, const BlockHandle& h, std::string* out, std::string* err) {
  if (h.offset + h.size + 5 > file.size()) { *err = "block handle out of range"; return false; }
  const uint8_t* p = reinterpret_cast<const uint8_t*>(file.data()) + h.offset;
  if (p[h.size] != 0) { *err = "compressed table blocks are not supported"; return false; }
  const uint32_t want = get_fixed32(p + h.size + 1);
  const uint32_t got = crc_mask(crc32c_extend(0, p, h.size + 1));
  if (want != got) { *err = "block checksum mismatch"; return false; }
  out->assign(reinterpret_cast<const char*>(p), h.size);
  return true;
}

bool parse_block(const std::string& blk, std::vector<std::pair<std::string, std::string>>* kv, std::string* err) {
  if (blk.size() < 4) { *err = "block too small"; return false; }
  const uint8_t* base = reinterpret_cast<const uint8_t*>(blk.data());
  const uint32_t nrestart = get_fixed32(base + blk.size() - 4);
  if (uint64_t(nrestart) * 4 + 4 > blk.size()) { *err = "bad restart count"; return false; }
  const uint8_t* p = base;
  const uint8_t* end = base + blk.size() - 4 - size_t(nrestart) * 4;
  std::string key;
  while (p < end) {
    uint64_t shared, non_shared, vlen;
    if (!get_varint(p, end, &shared) || !get_varint(p, end, &non_shared) || !get_varint(p, end, &vlen) ||
        shared > key.size() || uint64_t(end - p) < non_shared + vlen) {
      *err = "corrupt block entry";
      return false;
    }
    key.resize(shared);
    key.append(reinterpret_cast<const char*>(p), non_shared);
    p += non_shared;
    kv->emplace_back(key, std::string(reinterpret_cast<const char*>(p), vlen));
    p += vlen;
  }
  return true;
}

struct Writer {
  std::string prefix;
  std::map<std::string, Entry> entries;
};

struct Reader {
  std::string prefix;
  std::vector<Entry> entries;
  int num_shards = 1;
  std::string error;
};

std::string data_path(const std::string& prefix, int shard, int num_shards) {
  char buf[64];
  snprintf(buf, sizeof(buf), ".data-%05d-of-%05d", shard, num_shards);
  return prefix + buf;
}

void set_err(char* err, size_t n, const std::string& msg) {
  if (err && n) {
    snprintf(err, n, "%s", msg.c_str());
  }
}

}  // namespace

DTF_RT_EXPORT uint32_t dtf_crc32c(const void* data, size_t n) {
  return crc32c_extend(0, static_cast<const uint8_t*>(data), n);
}
DTF_RT_EXPORT uint32_t dtf_crc32c_masked(const void* data, size_t n) { return crc_mask(dtf_crc32c(data, n)); }

DTF_RT_EXPORT void* dtf_bundle_writer_open(const char* prefix) {
  Writer* w = new Writer;
  w->prefix = prefix;
  return w;
}

// dtype uses TensorFlow's DataType numbering (DT_FLOAT = 1, DT_INT32 = 3, DT_INT64 = 9, ...).
DTF_RT_EXPORT int dtf_bundle_writer_add(void* handle, const char* name, int dtype, const int64_t* dims, int ndims,
                                        const void* data, size_t nbytes) {
  Writer* w = static_cast<Writer*>(handle);
  if (!w || !name || ndims < 0 || (nbytes && !data)) return 1;
  if (w->entries.count(name)) return 2;
  Entry e;
  e.name = name;
  e.dtype = dtype;
  e.dims.assign(dims, dims + ndims);
  e.size = nbytes;
  e.data.assign(static_cast<const char*>(data), nbytes);
  e.crc = crc_mask(crc32c_extend(0, static_cast<const uint8_t*>(data), nbytes));
  w->entries.emplace(e.name, std::move(e));
  return 0;
}

DTF_RT_EXPORT int dtf_bundle_writer_finish(void* handle) {
  std::unique_ptr<Writer> w(static_cast<Writer*>(handle));
  if (!w) return 1;
  std::string data;
  std::vector<std::pair<std::string, std::string>> kv;
  kv.emplace_back("", encode_header());
  for (auto& it : w->entries) {  // std::map iterates in key order = data-file order
    Entry& e = it.second;
    e.offset = data.size();
    data.append(e.data);
    kv.emplace_back(e.name, encode_entry(e));
  }
  if (!write_file_atomic(data_path(w->prefix, 0, 1), data)) return 3;
  if (!write_file_atomic(w->prefix + ".index", build_table(kv))) return 4;
  return 0;
}

DTF_RT_EXPORT void dtf_bundle_writer_abort(void* handle) { delete static_cast<Writer*>(handle); }

DTF_RT_EXPORT void* dtf_bundle_reader_open(const char* prefix, char* err, size_t errlen) {
  std::unique_ptr<Reader> r(new Reader);
  r->prefix = prefix;
  std::string file, e;
  if (!read_file(r->prefix + ".index", &file)) { set_err(err, errlen, "cannot read " + r->prefix + ".index"); return nullptr; }
  if (file.size() < 48) { set_err(err, errlen, "index file shorter than a table footer"); return nullptr; }
  const uint8_t* f = reinterpret_cast<const uint8_t*>(file.data()) + file.size() - 48;
  uint64_t magic = uint64_t(get_fixed32(f + 40)) | uint64_t(get_fixed32(f + 44)) << 32;
  if (magic != kTableMagic) { set_err(err, errlen, "bad table magic number"); return nullptr; }
  const uint8_t* p = f;
  BlockHandle meta, index;
  if (!get_varint(p, f + 40, &meta.offset) || !get_varint(p, f + 40, &meta.size) || !get_varint(p, f + 40, &index.offset) ||
      !get_varint(p, f + 40, &index.size)) { set_err(err, errlen, "corrupt footer"); return nullptr; }
  std::string blk;
  std::vector<std::pair<std::string, std::string>> idx;
  if (!read_block(file, index, &blk, &e) || !parse_block(blk, &idx, &e)) { set_err(err, errlen, e); return nullptr; }
  bool have_header = false;
  for (const auto& ie : idx) {
    BlockHandle h;
    const uint8_t* q = reinterpret_cast<const uint8_t*>(ie.second.data());
    const uint8_t* qe = q + ie.second.size();
    if (!get_varint(q, qe, &h.offset) || !get_varint(q, qe, &h.size)) { set_err(err, errlen, "corrupt index entry"); return nullptr; }
    std::vector<std::pair<std::string, std::string>> kv;
    if (!read_block(file, h, &blk, &e) || !parse_block(blk, &kv, &e)) { set_err(err, errlen, e); return nullptr; }
    for (const auto& pr : kv) {
      if (pr.first.empty()) {
        have_header = true;
        const uint8_t* hp = reinterpret_cast<const uint8_t*>(pr.second.data());
        const uint8_t* he = hp + pr.second.size();
        while (hp < he) {
          uint64_t tag, v;
          if (!get_varint(hp, he, &tag)) break;
          if ((tag >> 3) == 1 && (tag & 7) == 0) { if (!get_varint(hp, he, &v)) break; r->num_shards = int(v); }
          else if ((tag >> 3) == 2 && (tag & 7) == 0) {
            if (!get_varint(hp, he, &v)) break;
            if (v != 0) { set_err(err, errlen, "big-endian bundles are not supported"); return nullptr; }
          }
          else if (!skip_field(hp, he, int(tag & 7))) break;
        }
        continue;
      }
      Entry en;
      en.name = pr.first;
      if (!decode_entry(pr.second, &en)) { set_err(err, errlen, "corrupt entry proto for " + pr.first); return nullptr; }
      r->entries.push_back(std::move(en));
    }
  }
  if (!have_header) { set_err(err, errlen, "bundle header entry missing"); return nullptr; }
  return r.release();
}

DTF_RT_EXPORT int dtf_bundle_reader_num_entries(void* handle) { return int(static_cast<Reader*>(handle)->entries.size()); }

DTF_RT_EXPORT int dtf_bundle_reader_entry(void* handle, int i, char* name, size_t namelen, int* dtype, int64_t* dims,
                                          int max_dims, int* ndims, uint64_t* nbytes) {
  Reader* r = static_cast<Reader*>(handle);
  if (i < 0 || size_t(i) >= r->entries.size()) return 1;
  const Entry& e = r->entries[size_t(i)];
  if (e.name.size() + 1 > namelen || int(e.dims.size()) > max_dims) return 2;
  memcpy(name, e.name.c_str(), e.name.size() + 1);
  *dtype = e.dtype;
  *ndims = int(e.dims.size());
  for (size_t d = 0; d < e.dims.size(); ++d) dims[d] = e.dims[d];
  *nbytes = e.size;
  return 0;
}

// Copies the tensor bytes into out (nbytes must equal the entry size) and verifies the checksum.
DTF_RT_EXPORT int dtf_bundle_reader_read(void* handle, const char* name, void* out, size_t nbytes) {
  Reader* r = static_cast<Reader*>(handle);
  for (const Entry& e : r->entries) {
    if (e.name != name) continue;
    if (e.size != nbytes) return 2;
    std::ifstream f(data_path(r->prefix, e.shard_id, r->num_shards), std::ios::binary);
    if (!f) return 3;
    f.seekg(std::streamoff(e.offset));
    f.read(static_cast<char*>(out), std::streamsize(nbytes));
    if (!f) return 4;
    if (crc_mask(crc32c_extend(0, static_cast<const uint8_t*>(out), nbytes)) != e.crc) return 5;
    return 0;
  }
  return 1;
}

DTF_RT_EXPORT void dtf_bundle_reader_close(void* handle) { delete static_cast<Reader*>(handle); }
