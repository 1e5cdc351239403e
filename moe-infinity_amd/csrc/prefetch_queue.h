// prefetch_queue.h — the pending-transfer queue of the tier mover (host-only, no HIP).
//
// Restates the queue discipline of the reference's ArcherTaskPool (core/prefetch/task_scheduler.{h,cpp}):
//   * NUM_PRIORITY = 20 FIFO levels, level 0 = on-demand / fetch, level 1 = prefetch (task_scheduler.h:24,
//     archer_prefetch_handle.cpp:207-218); the worker serves the lowest non-empty level, FIFO inside a level, and
//     removes every other queued task of the node it took (task_scheduler.cpp:451-497);
//   * EnqueueTask (:82-118): queued tasks of the SAME node with an equal or less urgent level are dropped first
//     (dedup / upgrade); with remove_layer also every task of an EARLIER layer;
//   * StartExec (:120-215), an on-demand access of a node at layer l: every queued task (all levels) of that node or
//     of a layer < l is dropped — prefetches for layers the pass has already left are stale;
//   * FetchExec (:44-80): like StartExec but on levels >= 1 and with layer <= l, then the node is queued at level 0;
//   * ReplaceCacheCandidates / ClearQueue (task_scheduler.h:55-79): levels >= 1 are emptied.
// What differs here is WHO serves it: there are no polling worker threads (task_scheduler.h:20-22, 10 us sleeps);
// the engine pumps the queue from its API calls and keeps a bounded number of prefetch copies in flight, so an
// on-demand miss (served immediately on its own high-priority stream, never queued) is never stuck behind a long
// FIFO of speculative copies on the link.
#pragma once
#include <stdint.h>

#include <deque>
#include <vector>

namespace moeinf {

constexpr int kNumPriority = 20;  // task_scheduler.h:24

struct QueuedTask {
  int64_t node = -1;
  int32_t layer = 0;     // corr_id & 0xffffffff of the reference
  int32_t priority = 1;  // level: 0 most urgent
};

class PrefetchQueue {
 public:
  PrefetchQueue() : levels_(kNumPriority) {}

  // EnqueueTask.  Returns the number of queued tasks it displaced.
  int enqueue(int64_t node, int layer, int priority, bool remove_layer = false) {
    if (priority < 0) priority = 0;
    if (priority >= kNumPriority) priority = kNumPriority - 1;
    int dropped = 0;
    for (int lv = 1; lv < kNumPriority; ++lv) {
      dropped += erase_if(lv, [&](const QueuedTask& t) {
        const bool same_less_urgent = (t.node == node) && (t.priority >= priority);
        const bool outdated = remove_layer && t.layer < layer;
        return same_less_urgent || outdated;
      });
    }
    QueuedTask t;
    t.node = node; t.layer = layer; t.priority = priority;
    levels_[priority].push_back(t);
    return dropped;
  }

  // StartExec: an on-demand access at `layer`.  node < 0: only the stale-layer rule (a layer dispatch as a whole).
  int on_demand(int64_t node, int layer) {
    int dropped = 0;
    for (int lv = 0; lv < kNumPriority; ++lv)
      dropped += erase_if(lv, [&](const QueuedTask& t) { return (node >= 0 && t.node == node) || t.layer < layer; });
    return dropped;
  }

  // FetchExec: drop levels >= 1 tasks of the node or of layers <= layer, then queue the node at level 0.
  int fetch(int64_t node, int layer, bool already_there) {
    int dropped = 0;
    for (int lv = 1; lv < kNumPriority; ++lv)
      dropped += erase_if(lv, [&](const QueuedTask& t) { return t.node == node || t.layer <= layer; });
    if (!already_there) {
      QueuedTask t;
      t.node = node; t.layer = layer; t.priority = 0;
      levels_[0].push_back(t);
    }
    return dropped;
  }

  int remove_node(int64_t node) {
    int dropped = 0;
    for (int lv = 0; lv < kNumPriority; ++lv) dropped += erase_if(lv, [&](const QueuedTask& t) { return t.node == node; });
    return dropped;
  }

  // ReplaceCacheCandidates / ClearQueue: speculative levels only
  int clear_prefetch() {
    int dropped = 0;
    for (int lv = 1; lv < kNumPriority; ++lv) { dropped += (int)levels_[lv].size(); levels_[lv].clear(); }
    return dropped;
  }

  // GPUThreadFunc: front of the lowest non-empty level; every other task of that node leaves the queue too.
  bool pop(QueuedTask* out) {
    for (int lv = 0; lv < kNumPriority; ++lv) {
      if (levels_[lv].empty()) continue;
      *out = levels_[lv].front();
      levels_[lv].pop_front();
      const int64_t node = out->node;
      for (int l2 = 0; l2 < kNumPriority; ++l2) erase_if(l2, [&](const QueuedTask& t) { return t.node == node; });
      return true;
    }
    return false;
  }

  size_t size() const {
    size_t n = 0;
    for (auto& d : levels_) n += d.size();
    return n;
  }
  bool empty() const { return size() == 0; }
  bool contains(int64_t node) const {
    for (auto& d : levels_)
      for (auto& t : d)
        if (t.node == node) return true;
    return false;
  }
  // queue content in service order (tests)
  std::vector<QueuedTask> snapshot() const {
    std::vector<QueuedTask> v;
    for (auto& d : levels_) v.insert(v.end(), d.begin(), d.end());
    return v;
  }

 private:
  template <typename F>
  int erase_if(int lv, F pred) {
    auto& d = levels_[lv];
    int n = 0;
    for (auto it = d.begin(); it != d.end();) {
      if (pred(*it)) { it = d.erase(it); ++n; } else { ++it; }
    }
    return n;
  }
  std::vector<std::deque<QueuedTask>> levels_;
};

// score in (0, 1] -> speculative level 1..19 (1 = most urgent).  The reference enqueues every prefetch at level 1 in
// descending-score order (memory/expert_prefetcher.py:42-59); with scores the same order is kept ACROSS calls too:
// a later, higher-scored request overtakes an earlier, lower-scored one that is still waiting.
inline int priority_from_score(const float* scores, int i) {
  if (!scores) return 1;
  float s = scores[i];
  if (!(s > 0.f)) return kNumPriority - 1;
  if (s > 1.f) s = 1.f;
  const int lv = 1 + (int)((1.f - s) * (kNumPriority - 2) + 0.5f);
  return lv < 1 ? 1 : (lv > kNumPriority - 1 ? kNumPriority - 1 : lv);
}

}  // namespace moeinf
