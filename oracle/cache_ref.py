"""CPU restatement of the reference's LIVE expert-cache policy.  TEST INFRASTRUCTURE ONLY.

Follows core/parallel/expert_dispatcher.cpp:
  :218-258  on a miss with no room: scan every (expert, layer) node, expert-major, and evict the
            RESIDENT one with the smallest ``incache_visit_count`` (strict <, so ties go to the
            first scanned = lowest id); the node being fetched is not resident so never a candidate
  :262-263  fetch, then ``incache_visit_count += 1`` on EVERY dispatch (hit or miss)
  :175-184  ClearExpertCacheCounts zeroes the counters (prefill -> decode boundary)
Protected candidates (replace_cache_candidates, core/prefetch/task_scheduler.cpp:292-297) are skipped
unless nothing else is evictable.  policy 1 = LRU on the dispatch clock (north_star's alternative).
Capacity is expressed in experts (slots) instead of the reference's byte counter, whose accounting
is known-broken."""


class RefCache:
    def __init__(self, slots, policy=0):
        self.slots, self.policy = slots, policy
        self.count, self.last, self.resident, self.protected = {}, {}, set(), set()
        self.clock = 0

    def access(self, i):
        self.count.setdefault(i, 0)
        evicted = -1
        hit = i in self.resident
        if not hit:
            if len(self.resident) >= self.slots:
                key = (lambda j: self.last[j]) if self.policy == 1 else (lambda j: self.count[j])
                cands = [j for j in sorted(self.resident) if j not in self.protected] or sorted(self.resident)
                best = None
                for j in cands:  # ascending id, strict < keeps the first minimum
                    if best is None or key(j) < key(best):
                        best = j
                self.resident.discard(best)
                evicted = best
            self.resident.add(i)
        self.count[i] += 1
        self.clock += 1
        self.last[i] = self.clock
        return hit, evicted

    def protect(self, ids):
        self.protected = set(ids)

    def clear_counts(self):
        for k in self.count:
            self.count[k] = 0
