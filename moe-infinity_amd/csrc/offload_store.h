// offload_store.h — the disk tier: reader/writer of the reference's offload directory (host only).
//
// Format compatibility with core/aio (so existing `offload_path` directories keep working):
//   <prefix>/archer_index        binary index (archer_tensor_index.cpp:101-132 Serialize/Deserialize):
//        u32 count, then per entry: u32 key(tensor id),
//        u32 file_id, i64 offset, u64 size, i64 ndim, i64 dims[ndim]      (:51-67)
//        6 x 1 byte options: pinned_memory, requires_grad, dtype(c10::ScalarType), device_index,
//        device_type, layout                                              (:11-25)
//   <prefix>/archer_param_<file_id>   tensor payloads at 4 KiB-aligned offsets, appended in store order
//        (archer_tensor_handle.cpp:53-86, kAioAlignment = 4096, archer_prio_aio_handle.h:18)
// Differences in mechanism, not format: the reference funnels every read through one AIO thread
// with O_DIRECT 1 MiB blocks (archer_prio_aio_handle.cpp:123-169).  Here get() is a plain pread() loop in large
// blocks straight into the (4 KiB-aligned) destination, O_DIRECT when the kernel accepts it; the engine's expert
// reads go through the two-priority block reader of aio_pool.h (plan_read() tells it where the payload is).
#pragma once
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <fstream>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace moeinf {

struct TensorMeta {
  uint32_t file_id = 0;
  int64_t offset = 0;
  uint64_t size = 0;
  std::vector<int64_t> shape;
  // options bytes, exactly as the reference writes them
  uint8_t pinned_memory = 0, requires_grad = 0;
  int8_t dtype = 6 /*c10::ScalarType::Float*/, device_index = -1, device_type = 0 /*CPU*/, layout = 0 /*Strided*/;
};

class OffloadStore {
 public:
  static constexpr int64_t kAlign = 4096;

  // returns "" on success, else an error message
  std::string open(const std::string& prefix) {
    prefix_ = prefix;
    if (prefix_.empty()) return "empty offload path";
    if (prefix_.back() != '/') prefix_ += '/';
    struct stat st;
    if (stat(prefix_.c_str(), &st) == -1) {
      if (mkdir(prefix_.c_str(), 0777) != 0) return "cannot create " + prefix_ + ": " + strerror(errno);
    } else if (!S_ISDIR(st.st_mode)) {
      return prefix_ + " is not a directory";
    }
    index_.clear();
    file_offset_ = 0;
    std::ifstream ifs(index_path(), std::ios::binary);
    if (ifs.good()) {
      uint32_t n = 0;
      ifs.read(reinterpret_cast<char*>(&n), sizeof n);
      for (uint32_t i = 0; i < n && ifs.good(); ++i) {
        uint32_t key = 0;
        TensorMeta m;
        int64_t nd = 0;
        ifs.read(reinterpret_cast<char*>(&key), 4);
        ifs.read(reinterpret_cast<char*>(&m.file_id), 4);
        ifs.read(reinterpret_cast<char*>(&m.offset), 8);
        ifs.read(reinterpret_cast<char*>(&m.size), 8);
        ifs.read(reinterpret_cast<char*>(&nd), 8);
        if (nd < 0 || nd > 16) return "corrupt archer_index (ndim " + std::to_string(nd) + ")";
        m.shape.resize((size_t)nd);
        for (auto& d : m.shape) ifs.read(reinterpret_cast<char*>(&d), 8);
        ifs.read(reinterpret_cast<char*>(&m.pinned_memory), 1);
        ifs.read(reinterpret_cast<char*>(&m.requires_grad), 1);
        ifs.read(reinterpret_cast<char*>(&m.dtype), 1);
        ifs.read(reinterpret_cast<char*>(&m.device_index), 1);
        ifs.read(reinterpret_cast<char*>(&m.device_type), 1);
        ifs.read(reinterpret_cast<char*>(&m.layout), 1);
        if (!ifs.good()) return "truncated archer_index";
        index_[key] = m;
        const int64_t end = m.offset + align_up((int64_t)m.size);
        if (m.file_id == 0 && end > file_offset_) file_offset_ = end;
      }
    }
    return "";
  }

  size_t count() const { return index_.size(); }
  const TensorMeta* find(uint32_t id) const {
    auto it = index_.find(id);
    return it == index_.end() ? nullptr : &it->second;
  }
  std::vector<uint32_t> ids() const {
    std::vector<uint32_t> v;
    for (auto& kv : index_) v.push_back(kv.first);
    return v;
  }

  // prefetch_handle.offload(tensor, id) -> ArcherTensorHandle::StoreTensor (archer_tensor_handle.cpp:53-86)
  std::string put(uint32_t id, const void* data, uint64_t nbytes, const int64_t* dims, int ndim, int scalar_type) {
    std::lock_guard<std::mutex> lk(mu_);
    TensorMeta m;
    auto it = index_.find(id);
    if (it != index_.end()) {
      if (it->second.size != nbytes) return "tensor " + std::to_string(id) + " size mismatch";
      m = it->second;  // rewrite in place
    } else {
      m.file_id = 0;
      m.offset = file_offset_;
      m.size = nbytes;
      m.shape.assign(dims, dims + ndim);
      m.dtype = (int8_t)scalar_type;
      file_offset_ += align_up((int64_t)nbytes);
      index_[id] = m;
    }
    const std::string fn = param_path(m.file_id);
    int fd = ::open(fn.c_str(), O_WRONLY | O_CREAT, 0644);
    if (fd < 0) return "open " + fn + ": " + strerror(errno);
    const char* p = static_cast<const char*>(data);
    uint64_t done = 0;
    while (done < nbytes) {
      ssize_t w = pwrite(fd, p + done, nbytes - done, m.offset + (int64_t)done);
      if (w < 0) { std::string e = strerror(errno); ::close(fd); return "pwrite " + fn + ": " + e; }
      done += (uint64_t)w;
    }
    // pad the payload to the alignment so O_DIRECT readers can read whole blocks
    const int64_t padded = align_up((int64_t)nbytes);
    if (padded > (int64_t)nbytes) {
      std::vector<char> z((size_t)(padded - (int64_t)nbytes), 0);
      if (pwrite(fd, z.data(), z.size(), m.offset + (int64_t)nbytes) < 0) { /* best effort */ }
    }
    ::close(fd);
    dirty_ = true;
    return "";
  }

  // ArcherTensorIndex::Serialize (archer_tensor_index.cpp:101-109)
  std::string flush() {
    std::lock_guard<std::mutex> lk(mu_);
    std::ofstream ofs(index_path(), std::ios::binary | std::ios::trunc);
    if (!ofs.good()) return "cannot write " + index_path();
    uint32_t n = (uint32_t)index_.size();
    ofs.write(reinterpret_cast<const char*>(&n), 4);
    for (auto& kv : index_) {
      const TensorMeta& m = kv.second;
      int64_t nd = (int64_t)m.shape.size();
      ofs.write(reinterpret_cast<const char*>(&kv.first), 4);
      ofs.write(reinterpret_cast<const char*>(&m.file_id), 4);
      ofs.write(reinterpret_cast<const char*>(&m.offset), 8);
      ofs.write(reinterpret_cast<const char*>(&m.size), 8);
      ofs.write(reinterpret_cast<const char*>(&nd), 8);
      for (auto d : m.shape) ofs.write(reinterpret_cast<const char*>(&d), 8);
      ofs.write(reinterpret_cast<const char*>(&m.pinned_memory), 1);
      ofs.write(reinterpret_cast<const char*>(&m.requires_grad), 1);
      ofs.write(reinterpret_cast<const char*>(&m.dtype), 1);
      ofs.write(reinterpret_cast<const char*>(&m.device_index), 1);
      ofs.write(reinterpret_cast<const char*>(&m.device_type), 1);
      ofs.write(reinterpret_cast<const char*>(&m.layout), 1);
    }
    dirty_ = false;
    return ofs.good() ? "" : "short write to " + index_path();
  }

  // ArcherTensorHandle::ReadTensor (archer_tensor_handle.cpp:189-201): tensor payload -> dst.
  // If dst and capacity allow (4 KiB-aligned dst with room for the padded size) the read goes through
  // O_DIRECT in 8 MiB blocks, bypassing the page cache like the reference's AIO path.
  std::string get(uint32_t id, void* dst, uint64_t capacity) const {
    const TensorMeta* m = find(id);
    if (!m) return "tensor " + std::to_string(id) + " not in archer_index";
    if (capacity < m->size) return "destination too small for tensor " + std::to_string(id);
    const std::string fn = param_path(m->file_id);
    const uint64_t padded = (uint64_t)align_up((int64_t)m->size);
    const bool direct_ok = ((uintptr_t)dst % kAlign == 0) && capacity >= padded && (m->offset % kAlign == 0);
    int fd = -1;
    bool direct = false;
    if (direct_ok) {
      fd = ::open(fn.c_str(), O_RDONLY | O_DIRECT);
      direct = fd >= 0;
    }
    if (fd < 0) fd = ::open(fn.c_str(), O_RDONLY);
    if (fd < 0) return "open " + fn + ": " + strerror(errno);
    const uint64_t want = direct ? padded : m->size;
    const uint64_t blk = 8ull << 20;
    uint64_t done = 0;
    char* p = static_cast<char*>(dst);
    while (done < want) {
      const uint64_t n = std::min<uint64_t>(blk, want - done);
      ssize_t r = pread(fd, p + done, n, m->offset + (int64_t)done);
      if (r < 0 && direct && errno == EINVAL) {  // filesystem refuses O_DIRECT at this geometry: fall back
        ::close(fd);
        fd = ::open(fn.c_str(), O_RDONLY);
        if (fd < 0) return "open " + fn + ": " + strerror(errno);
        direct = false;
        continue;
      }
      if (r < 0) { std::string e = strerror(errno); ::close(fd); return "pread " + fn + ": " + e; }
      if (r == 0) {
        if (done >= m->size) break;  // file ends inside the padding
        ::close(fd);
        return fn + " is shorter than the index says (tensor " + std::to_string(id) + ")";
      }
      done += (uint64_t)r;
    }
    ::close(fd);
    return "";
  }

  // where a tensor's payload lives, for readers that bring their own I/O (the priority block reader, aio_pool.h)
  struct ReadPlan { std::string path; int64_t offset = 0; uint64_t size = 0; bool direct_ok = false; };
  std::string plan_read(uint32_t id, const void* dst, uint64_t capacity, ReadPlan* out) const {
    const TensorMeta* m = find(id);
    if (!m) return "tensor " + std::to_string(id) + " not in archer_index";
    if (capacity < m->size) return "destination too small for tensor " + std::to_string(id);
    out->path = param_path(m->file_id);
    out->offset = m->offset;
    out->size = m->size;
    out->direct_ok = ((uintptr_t)dst % kAlign == 0) && capacity >= (uint64_t)align_up((int64_t)m->size) && (m->offset % kAlign == 0);
    return "";
  }

  // bytes [off, off+n) of a tensor's payload -> dst (buffered pread; used for piecewise disk -> device transfers)
  std::string get_range(uint32_t id, uint64_t off, void* dst, uint64_t n) const {
    const TensorMeta* m = find(id);
    if (!m) return "tensor " + std::to_string(id) + " not in archer_index";
    if (off + n > m->size) return "range beyond the end of tensor " + std::to_string(id);
    const std::string fn = param_path(m->file_id);
    const int fd = ::open(fn.c_str(), O_RDONLY);
    if (fd < 0) return "open " + fn + ": " + strerror(errno);
    uint64_t done = 0;
    while (done < n) {
      const ssize_t r = pread(fd, static_cast<char*>(dst) + done, n - done, m->offset + (int64_t)(off + done));
      if (r <= 0) { const std::string e = r < 0 ? strerror(errno) : "unexpected end of file"; ::close(fd); return "pread " + fn + ": " + e; }
      done += (uint64_t)r;
    }
    ::close(fd);
    return "";
  }

  const std::string& prefix() const { return prefix_; }
  bool dirty() const { return dirty_; }
  // experts of live engines that may still re-read their blob from this directory (moeinf_register_expert_from_store):
  // the store cannot be closed while the count is non-zero
  mutable int users = 0;

 private:
  static int64_t align_up(int64_t v) { return (v + kAlign - 1) & ~(kAlign - 1); }
  std::string index_path() const { return prefix_ + "archer_index"; }
  std::string param_path(uint32_t file_id) const { return prefix_ + "archer_param_" + std::to_string(file_id); }

  std::string prefix_;
  // the reference iterates an unordered_map when serialising; order inside the file is irrelevant to
  // readers (key -> meta), so a sorted map keeps the file deterministic
  std::map<uint32_t, TensorMeta> index_;
  int64_t file_offset_ = 0;
  bool dirty_ = false;
  mutable std::mutex mu_;
};

}  // namespace moeinf
