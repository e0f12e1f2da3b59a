// Native safetensors reader for the host runtimes (reference pegainfer-core/src/
// weight_loader.rs:18-206: load_shard_info_fixed / mmap_shards / find_tensor / load_tensor_2d{,_row_shard,_col_shard}).
// A checkpoint is one `*.safetensors` file, or a directory with `model.safetensors` or with
// `model.safetensors.index.json` naming the shard files.  Files are mmap'ed read-only; tensors are views into the
// maps (8-byte LE header length, JSON header {"name": {"dtype","shape","data_offsets"}}, then the data).
// The JSON reader covers exactly the grammar safetensors headers and HF config.json use.
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace pst {

struct Json {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;
  const Json* get(const std::string& k) const {
    for (auto& kv : obj)
      if (kv.first == k) return &kv.second;
    return nullptr;
  }
  double number_or(const std::string& k, double d) const {
    const Json* j = get(k);
    return j && j->kind == Num ? j->num : d;
  }
};

class JsonParser {
 public:
  JsonParser(const char* p, size_t n) : p_(p), e_(p + n) {}
  bool parse(Json* out) { ws(); return value(out) && (ws(), true); }

 private:
  const char *p_, *e_;
  void ws() { while (p_ < e_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_; }
  bool lit(const char* s) {
    const size_t n = std::strlen(s);
    if ((size_t)(e_ - p_) < n || std::strncmp(p_, s, n) != 0) return false;
    p_ += n;
    return true;
  }
  bool string(std::string* out) {
    if (p_ >= e_ || *p_ != '"') return false;
    ++p_;
    out->clear();
    while (p_ < e_ && *p_ != '"') {
      if (*p_ == '\\') {
        if (++p_ >= e_) return false;
        switch (*p_) {
          case 'n': out->push_back('\n'); break;
          case 't': out->push_back('\t'); break;
          case 'r': out->push_back('\r'); break;
          case 'b': out->push_back('\b'); break;
          case 'f': out->push_back('\f'); break;
          case 'u': {  // keep BMP code points as UTF-8; names and dtypes are ASCII in practice
            if (e_ - p_ < 5) return false;
            unsigned cp = 0;
            for (int i = 1; i <= 4; ++i) {
              const char c = p_[i];
              cp = cp * 16 + (c >= '0' && c <= '9' ? c - '0' : (c | 32) >= 'a' && (c | 32) <= 'f' ? (c | 32) - 'a' + 10 : 0);
            }
            p_ += 4;
            if (cp < 0x80) out->push_back((char)cp);
            else if (cp < 0x800) { out->push_back((char)(0xC0 | (cp >> 6))); out->push_back((char)(0x80 | (cp & 0x3F))); }
            else { out->push_back((char)(0xE0 | (cp >> 12))); out->push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out->push_back((char)(0x80 | (cp & 0x3F))); }
            break;
          }
          default: out->push_back(*p_);
        }
        ++p_;
      } else {
        out->push_back(*p_++);
      }
    }
    if (p_ >= e_) return false;
    ++p_;
    return true;
  }
  bool value(Json* out) {
    if (p_ >= e_) return false;
    if (*p_ == '{') {
      out->kind = Json::Obj;
      ++p_; ws();
      if (p_ < e_ && *p_ == '}') { ++p_; return true; }
      for (;;) {
        std::string k;
        ws();
        if (!string(&k)) return false;
        ws();
        if (p_ >= e_ || *p_ != ':') return false;
        ++p_; ws();
        Json v;
        if (!value(&v)) return false;
        out->obj.emplace_back(std::move(k), std::move(v));
        ws();
        if (p_ < e_ && *p_ == ',') { ++p_; continue; }
        if (p_ < e_ && *p_ == '}') { ++p_; return true; }
        return false;
      }
    }
    if (*p_ == '[') {
      out->kind = Json::Arr;
      ++p_; ws();
      if (p_ < e_ && *p_ == ']') { ++p_; return true; }
      for (;;) {
        Json v;
        ws();
        if (!value(&v)) return false;
        out->arr.push_back(std::move(v));
        ws();
        if (p_ < e_ && *p_ == ',') { ++p_; continue; }
        if (p_ < e_ && *p_ == ']') { ++p_; return true; }
        return false;
      }
    }
    if (*p_ == '"') { out->kind = Json::Str; return string(&out->str); }
    if (lit("true")) { out->kind = Json::Bool; out->b = true; return true; }
    if (lit("false")) { out->kind = Json::Bool; out->b = false; return true; }
    if (lit("null")) { out->kind = Json::Null; return true; }
    // number: copy the token into a NUL-terminated buffer first - the input is an mmap with no terminator, and
    // strtod on a file that ends in a digit would read past the mapping
    char buf[64];
    size_t len = 0;
    while (p_ + len < e_ && len + 1 < sizeof(buf)) {
      const char c = p_[len];
      if (!((c >= '0' && c <= '9') || c == '-' || c == '+' || c == '.' || c == 'e' || c == 'E')) break;
      buf[len++] = c;
    }
    if (len == 0) return false;
    buf[len] = 0;
    char* end = nullptr;
    out->num = std::strtod(buf, &end);
    if (end != buf + len) return false;
    out->kind = Json::Num;
    p_ += len;
    return true;
  }
};

struct MappedFile {
  const uint8_t* data = nullptr;
  size_t size = 0;
  ~MappedFile() { if (data) munmap(const_cast<uint8_t*>(data), size); }
  static std::unique_ptr<MappedFile> open(const std::string& path, std::string* err) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) { *err = "cannot open " + path; return nullptr; }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 8) { ::close(fd); *err = "cannot stat " + path; return nullptr; }
    void* p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (p == MAP_FAILED) { *err = "mmap failed for " + path; return nullptr; }
    auto f = std::make_unique<MappedFile>();
    f->data = static_cast<const uint8_t*>(p);
    f->size = (size_t)st.st_size;
    return f;
  }
};

struct TensorView {
  std::string dtype;
  std::vector<int64_t> shape;
  const uint8_t* data = nullptr;
  size_t nbytes = 0;
  int64_t numel() const { int64_t n = 1; for (int64_t d : shape) n *= d; return n; }
};

class Checkpoint {
 public:
  // path: a .safetensors file or a model directory
  bool open(const std::string& path, std::string* err) {
    struct stat st;
    if (stat(path.c_str(), &st) != 0) { *err = "no such checkpoint: " + path; return false; }
    if (!S_ISDIR(st.st_mode)) return add_file(path, err);
    dir_ = path;
    const std::string index = path + "/model.safetensors.index.json";
    if (stat(index.c_str(), &st) == 0) {
      auto f = MappedFile::open(index, err);
      if (!f) return false;
      Json j;
      if (!JsonParser(reinterpret_cast<const char*>(f->data), f->size).parse(&j)) { *err = "bad json: " + index; return false; }
      const Json* wm = j.get("weight_map");
      if (!wm || wm->kind != Json::Obj) { *err = "index without weight_map"; return false; }
      std::vector<std::string> files;
      for (auto& kv : wm->obj)
        if (kv.second.kind == Json::Str && std::find(files.begin(), files.end(), kv.second.str) == files.end())
          files.push_back(kv.second.str);
      for (auto& fn : files) {
        // shard names come from an untrusted index: plain file names only, inside the model directory
        if (fn.empty() || fn.find('/') != std::string::npos || fn.find('\\') != std::string::npos || fn == "." || fn == "..") {
          *err = "index names a shard outside the model directory: " + fn;
          return false;
        }
        if (!add_file(path + "/" + fn, err)) return false;
      }
      return true;
    }
    return add_file(path + "/model.safetensors", err);
  }
  const TensorView* find(const std::string& name) const {
    auto it = tensors_.find(name);
    return it == tensors_.end() ? nullptr : &it->second;
  }
  const std::map<std::string, TensorView>& tensors() const { return tensors_; }
  const std::string& dir() const { return dir_; }

 private:
  std::vector<std::unique_ptr<MappedFile>> files_;
  std::map<std::string, TensorView> tensors_;
  std::string dir_;
  bool add_file(const std::string& path, std::string* err) {
    auto f = MappedFile::open(path, err);
    if (!f) return false;
    uint64_t n = 0;
    std::memcpy(&n, f->data, 8);
    if (n > f->size - 8) { *err = "corrupt safetensors header: " + path; return false; }
    Json j;
    if (!JsonParser(reinterpret_cast<const char*>(f->data + 8), (size_t)n).parse(&j) || j.kind != Json::Obj) {
      *err = "bad safetensors header json: " + path;
      return false;
    }
    const uint8_t* base = f->data + 8 + n;
    const size_t avail = f->size - 8 - (size_t)n;
    for (auto& kv : j.obj) {
      if (kv.first == "__metadata__") continue;
      const Json *dt = kv.second.get("dtype"), *sh = kv.second.get("shape"), *off = kv.second.get("data_offsets");
      if (!dt || !sh || !off || off->arr.size() != 2) { *err = "bad tensor entry " + kv.first; return false; }
      TensorView t;
      t.dtype = dt->str;
      for (auto& d : sh->arr) {
        if (d.kind != Json::Num || d.num < 0 || d.num > 4e12) { *err = "bad shape in tensor entry " + kv.first; return false; }
        t.shape.push_back((int64_t)d.num);
      }
      const size_t a = (size_t)off->arr[0].num, b = (size_t)off->arr[1].num;
      if (a > b || b > avail) { *err = "tensor out of file bounds: " + kv.first; return false; }
      t.data = base + a;
      t.nbytes = b - a;
      {  // the byte range must be exactly numel x element size: consumers slice and upload by shape
        static const std::pair<const char*, int> kSizes[] = {{"BF16", 2}, {"F16", 2}, {"F32", 4}, {"F64", 8}, {"I64", 8},
                                                             {"I32", 4}, {"I16", 2}, {"I8", 1}, {"U8", 1}, {"BOOL", 1},
                                                             {"F8_E4M3", 1}, {"F8_E5M2", 1}};
        for (auto& ds : kSizes)
          if (t.dtype == ds.first && (uint64_t)t.numel() * (uint64_t)ds.second != (uint64_t)t.nbytes) {
            *err = "tensor " + kv.first + ": data_offsets do not match its shape";
            return false;
          }
      }
      tensors_[kv.first] = std::move(t);
    }
    files_.push_back(std::move(f));
    return true;
  }
};

}  // namespace pst
