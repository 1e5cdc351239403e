"""CPU restatement of the reference's pending-transfer queue (ArcherTaskPool).  TEST INFRASTRUCTURE ONLY.

Follows core/prefetch/task_scheduler.{h,cpp} of the reference:
  task_scheduler.h:24     NUM_PRIORITY = 20 deques ("unified_queue_"); level 0 = on-demand / fetch, prefetch = level 1
                          (core/prefetch/archer_prefetch_handle.cpp:207-218)
  task_scheduler.cpp:82-118   EnqueueTask: erase from levels 1..19 every task with (same node AND priority >= new)
                              OR (remove_layer AND task layer < new layer); then push_back at the new level
  task_scheduler.cpp:158-168  StartExec (on-demand access): erase from ALL levels every task with same node OR
                              task layer < node layer
  task_scheduler.cpp:44-80    FetchExec: erase from levels 1..19 same node OR task layer <= node layer; push at level 0
                              unless the node already sits on its target device
  task_scheduler.h:55-79      ClearQueue / ReplaceCacheCandidates: levels 1..19 emptied
  task_scheduler.cpp:451-497  worker: lowest non-empty level, front task (single GPU: the first one matches);
                              erase every task of that node from all levels
``layer`` is ``corr_id & 0xffffffff``.  Parity: restated (the C++ core needs the CUDA toolkit);
the engine's queue (csrc/prefetch_queue.h) is checked against this file operation by operation on random traces.
"""
NUM_PRIORITY = 20


class RefTaskQueue:
    def __init__(self):
        self.q = [[] for _ in range(NUM_PRIORITY)]  # entries: (node, layer, priority)

    def _erase(self, levels, pred):
        n = 0
        for lv in levels:
            keep = [t for t in self.q[lv] if not pred(t)]
            n += len(self.q[lv]) - len(keep)
            self.q[lv] = keep
        return n

    def enqueue(self, node, layer, priority, remove_layer=False):
        priority = min(max(priority, 0), NUM_PRIORITY - 1)
        n = self._erase(range(1, NUM_PRIORITY),
                        lambda t: (t[0] == node and t[2] >= priority) or (remove_layer and t[1] < layer))
        self.q[priority].append((node, layer, priority))
        return n

    def on_demand(self, node, layer):
        return self._erase(range(NUM_PRIORITY), lambda t: (node >= 0 and t[0] == node) or t[1] < layer)

    def fetch(self, node, layer, already_there):
        n = self._erase(range(1, NUM_PRIORITY), lambda t: t[0] == node or t[1] <= layer)
        if not already_there:
            self.q[0].append((node, layer, 0))
        return n

    def clear_prefetch(self):
        n = sum(len(self.q[lv]) for lv in range(1, NUM_PRIORITY))
        for lv in range(1, NUM_PRIORITY):
            self.q[lv] = []
        return n

    def pop(self):
        for lv in range(NUM_PRIORITY):
            if self.q[lv]:
                t = self.q[lv].pop(0)
                self._erase(range(NUM_PRIORITY), lambda u: u[0] == t[0])
                return t
        return None

    def snapshot(self):
        return [t for lv in self.q for t in lv]
