// aio_pool.h — two-priority block reader for the disk tier (host only, no HIP).
//
// What it replaces: the reference's ArcherPrioAioHandle / ArcherPrioAioContext (core/aio/archer_prio_aio_handle.cpp):
// a read is cut into 1 MiB blocks (kBlockSize, :13); a scheduler thread serves the HIGH queue first — every block of its
// first request — and otherwise ONE block of the first LOW request per turn (:123-169), so an on-demand read waits for
// at most one speculative block; files are opened O_DIRECT once and cached (:40-52, archer_aio_utils.cpp).
// Same discipline here, different mechanism: N worker threads (the reference has one: "only one SSD device") pull
// blocks, always from the high queue while it has any; a low request can be PROMOTED when the expert it is reading for
// is demanded (the reference has no such step: its demand read would queue a second read of the same bytes).
// submit() never blocks; wait() blocks the caller like ArcherPrioAioHandle::Read does.
#pragma once
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace moeinf {

class PrioAioPool {
 public:
  static constexpr int64_t kAlign = 4096;  // kAioAlignment

  struct Request {
    std::mutex mu;
    std::condition_variable cv;
    int pending = 0;       // blocks not finished yet
    std::string error;     // first error
    bool high = false;
    uint64_t seq = 0;      // submission order (diagnostics)
  };
  using Handle = std::shared_ptr<Request>;

  explicit PrioAioPool(int threads = 4, int64_t block_bytes = 1 << 20) : block_(std::max<int64_t>(kAlign, block_bytes / kAlign * kAlign)) {
    threads = std::max(1, std::min(threads, 64));
    for (int i = 0; i < threads; ++i) workers_.emplace_back(&PrioAioPool::run, this);
  }
  ~PrioAioPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      exit_ = true;
      // blocks that never started are dropped; their requests are failed so that no waiter hangs
      for (auto* q : {&high_, &low_})
        for (auto& b : *q) finish(b.req, "aio pool shut down");
      high_.clear(); low_.clear();
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
    for (auto& f : fds_) ::close(f.second);
  }
  PrioAioPool(const PrioAioPool&) = delete;
  PrioAioPool& operator=(const PrioAioPool&) = delete;

  // Read bytes [offset, offset + nbytes) of `path` into dst.  try_direct: use an O_DIRECT descriptor when dst, offset
  // and the block geometry are 4 KiB-aligned (nbytes is then read rounded UP to 4 KiB: the caller guarantees room —
  // the offload files are padded to the alignment); any block the filesystem refuses falls back to a buffered read.
  Handle submit(const std::string& path, void* dst, int64_t nbytes, int64_t offset, bool high_prio, bool try_direct) {
    auto req = std::make_shared<Request>();
    req->high = high_prio;
    if (nbytes <= 0) return req;
    const bool direct = try_direct && ((uintptr_t)dst % kAlign == 0) && (offset % kAlign == 0);
    const int64_t want = direct ? (nbytes + kAlign - 1) / kAlign * kAlign : nbytes;
    std::vector<Block> blocks;
    for (int64_t done = 0; done < want; done += block_) {
      Block b;
      b.req = req; b.path = path; b.dst = static_cast<char*>(dst) + done; b.off = offset + done;
      b.n = std::min<int64_t>(block_, want - done); b.direct = direct;
      b.payload_left = std::max<int64_t>(0, nbytes - done);  // a file may end inside the padding of its last tensor
      blocks.push_back(b);
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      req->seq = ++seq_;
      req->pending = (int)blocks.size();
      auto& q = high_prio ? high_ : low_;
      for (auto& b : blocks) q.push_back(std::move(b));
    }
    cv_.notify_all();
    return req;
  }

  // a speculative read whose data is needed NOW: its unstarted blocks move to the back of the high queue
  void promote(const Handle& h) {
    if (!h) return;
    std::lock_guard<std::mutex> lk(mu_);
    if (h->high) return;
    h->high = true;
    std::deque<Block> keep;
    for (auto& b : low_) {
      if (b.req == h) high_.push_back(std::move(b));
      else keep.push_back(std::move(b));
    }
    low_.swap(keep);
    promoted_ += 1;
  }

  static bool done(const Handle& h) {
    if (!h) return true;
    std::lock_guard<std::mutex> lk(h->mu);
    return h->pending == 0;
  }
  // blocks until the request has finished; returns "" or the first error
  static std::string wait(const Handle& h) {
    if (!h) return "";
    std::unique_lock<std::mutex> lk(h->mu);
    h->cv.wait(lk, [&] { return h->pending == 0; });
    return h->error;
  }

  struct Stats { int64_t blocks_high = 0, blocks_low = 0, bytes = 0, promoted = 0, direct_fallbacks = 0; };
  Stats stats() const {
    std::lock_guard<std::mutex> lk(mu_);
    Stats s;
    s.blocks_high = blocks_high_; s.blocks_low = blocks_low_; s.bytes = bytes_;
    s.promoted = promoted_; s.direct_fallbacks = direct_fallbacks_;
    return s;
  }
  int64_t block_bytes() const { return block_; }

 private:
  struct Block {
    Handle req;
    std::string path;
    char* dst = nullptr;
    int64_t off = 0, n = 0, payload_left = 0;
    bool direct = false;
  };

  static void finish(const Handle& r, const std::string& err) {
    std::lock_guard<std::mutex> lk(r->mu);
    if (!err.empty() && r->error.empty()) r->error = err;
    if (r->pending > 0 && --r->pending == 0) r->cv.notify_all();
  }

  int fd_for(const std::string& path, bool direct) {  // callers hold no lock
    std::lock_guard<std::mutex> lk(fd_mu_);
    const std::string key = (direct ? "D:" : "B:") + path;
    auto it = fds_.find(key);
    if (it != fds_.end()) return it->second;
    const int fd = ::open(path.c_str(), direct ? (O_RDONLY | O_DIRECT) : O_RDONLY);
    if (fd >= 0) fds_[key] = fd;
    return fd;
  }

  std::string read_block(Block& b) {
    bool direct = b.direct;
    int fd = direct ? fd_for(b.path, true) : -1;
    if (direct && fd < 0) direct = false;  // the filesystem has no O_DIRECT (tmpfs): buffered
    if (!direct) {
      fd = fd_for(b.path, false);
      if (fd < 0) return "open " + b.path + ": " + strerror(errno);
    }
    int64_t done = 0;
    while (done < b.n) {
      const ssize_t r = pread(fd, b.dst + done, (size_t)(b.n - done), b.off + done);
      if (r < 0 && errno == EINTR) continue;
      if (r < 0 && direct && errno == EINVAL) {  // refused at this geometry: this block goes buffered
        direct = false;
        fd = fd_for(b.path, false);
        if (fd < 0) return "open " + b.path + ": " + strerror(errno);
        std::lock_guard<std::mutex> lk(mu_);
        direct_fallbacks_ += 1;
        continue;
      }
      if (r < 0) return "pread " + b.path + ": " + strerror(errno);
      if (r == 0) {
        if (done >= b.payload_left) break;  // the file ends inside the alignment padding
        return b.path + " is shorter than requested";
      }
      done += r;
    }
    return "";
  }

  void run() {
    for (;;) {
      Block b;
      bool from_high = false;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return exit_ || !high_.empty() || !low_.empty(); });
        if (exit_) return;
        if (!high_.empty()) { b = std::move(high_.front()); high_.pop_front(); from_high = true; }
        else { b = std::move(low_.front()); low_.pop_front(); }
      }
      const std::string err = read_block(b);
      {
        std::lock_guard<std::mutex> lk(mu_);
        (from_high ? blocks_high_ : blocks_low_) += 1;
        bytes_ += b.n;
      }
      finish(b.req, err);
    }
  }

  const int64_t block_;
  mutable std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Block> high_, low_;
  bool exit_ = false;
  uint64_t seq_ = 0;
  int64_t blocks_high_ = 0, blocks_low_ = 0, bytes_ = 0, promoted_ = 0, direct_fallbacks_ = 0;
  std::mutex fd_mu_;
  std::map<std::string, int> fds_;
  std::vector<std::thread> workers_;
};

}  // namespace moeinf
