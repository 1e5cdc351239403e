// cache_policy.h — replacement policy of the device expert cache (host-only, no HIP).
//
// Restates the reference's LIVE policy: evict the resident expert with the minimum
// `incache_visit_count` (core/parallel/expert_dispatcher.cpp:227-258; the task-pool evictor
// uses the same counter, core/prefetch/task_scheduler.cpp:276-310), scanning expert-major /
// layer-minor so ties go to the lowest (expert, layer).  Counters are zeroed at the
// prefill->decode boundary (ExpertDispatcher::ClearExpertCacheCounts, expert_dispatcher.cpp:175-184).
// LRU (north_star's wording) is offered as an alternative.  Policy never changes numerics.
//
// Divergence, on purpose: the reference's prefetch evictor iterates candidates in DESCENDING
// visit count (task_scheduler.cpp:285-287), i.e. evicts the hottest expert first; this engine uses
// one ascending policy for demand fetches and prefetches.
#pragma once
#include <stdint.h>

#include <limits>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace moeinf {

enum { POLICY_LFU_INCACHE = 0, POLICY_LRU = 1 };

struct PolicyEntry {
  int64_t incache = 0;      // incache_visit_count
  uint64_t last_access = 0;  // logical clock of the last dispatch
  bool resident = false;
  bool is_protected = false;  // replace_cache_candidates set
  bool pinned = false;        // in use by the layer being dispatched right now
};

// Pick a victim among entries[0..n) (index order = tie-break order).  Returns -1 if none.
// Protected entries are only taken when nothing else is evictable AND allow_protected is set
// (demand fetches must make progress; prefetches must never displace the protected set).
inline int64_t pick_victim(const PolicyEntry* entries, int64_t n, int policy, bool allow_protected = true) {
  int64_t best = -1, best_prot = -1;
  int64_t best_key = std::numeric_limits<int64_t>::max(), best_prot_key = std::numeric_limits<int64_t>::max();
  for (int64_t i = 0; i < n; ++i) {
    const PolicyEntry& e = entries[i];
    if (!e.resident || e.pinned) continue;
    const int64_t key = (policy == POLICY_LRU) ? (int64_t)e.last_access : e.incache;
    if (e.is_protected) {
      if (key < best_prot_key) { best_prot_key = key; best_prot = i; }
    } else {
      if (key < best_key) { best_key = key; best = i; }
    }
  }
  return best >= 0 ? best : (allow_protected ? best_prot : -1);
}

// Standalone fixed-capacity cache driven by the same policy (tests, hit-rate studies).
class CacheSim {
 public:
  CacheSim(int slots, int policy) : slots_(slots), policy_(policy) {}
  // returns hit; *evicted = evicted id or -1
  bool access(int64_t id, int64_t* evicted) {
    *evicted = -1;
    int64_t idx = index_of(id);
    PolicyEntry& e = entries_[idx];
    bool hit = e.resident;
    if (!hit) {
      if (used_ >= slots_) {
        int64_t v = pick_victim(entries_.data(), (int64_t)entries_.size(), policy_);
        if (v >= 0) {
          entries_[v].resident = false;
          *evicted = ids_[v];
          --used_;
        }
      }
      if (used_ < slots_) {
        e.resident = true;
        ++used_;
      }
    }
    e.incache += 1;
    e.last_access = ++clock_;
    return hit;
  }
  void protect(const int64_t* ids, int n) {
    for (auto& e : entries_) e.is_protected = false;
    for (int i = 0; i < n; ++i) entries_[index_of(ids[i])].is_protected = true;
  }
  void clear_counts() {
    for (auto& e : entries_) e.incache = 0;
  }

 private:
  // entries are kept sorted by id so that index order == id order (tie-break = lowest id)
  int64_t index_of(int64_t id) {
    auto it = pos_.find(id);
    if (it != pos_.end()) return it->second;
    // insert keeping ids_ ascending
    size_t at = 0;
    while (at < ids_.size() && ids_[at] < id) ++at;
    ids_.insert(ids_.begin() + at, id);
    entries_.insert(entries_.begin() + at, PolicyEntry());
    pos_.clear();
    for (size_t i = 0; i < ids_.size(); ++i) pos_[ids_[i]] = (int64_t)i;
    return (int64_t)at;
  }
  int slots_, policy_, used_ = 0;
  uint64_t clock_ = 0;
  std::vector<int64_t> ids_;
  std::vector<PolicyEntry> entries_;
  std::unordered_map<int64_t, int64_t> pos_;
};

}  // namespace moeinf
