// host_fuzz.cpp — drives the host-only components of the engine (cache policy, tracer, offload store, priority reader)
// under AddressSanitizer + UBSan ("new engine should run its host tests under
// TSan/ASan").  Built and run by tests/test_host_sanitizers.py; no HIP.
#include <stdio.h>
#include <stdlib.h>

#include <random>
#include <string>
#include <vector>

#include "aio_pool.h"
#include "cache_policy.h"
#include "offload_store.h"
#include "tracer.h"

using namespace moeinf;

#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "REQUIRE failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::mt19937 rng(1234);
  // cache policy
  for (int policy = 0; policy < 2; ++policy) {
    CacheSim sim(5, policy);
    int hits = 0;
    for (int i = 0; i < 20000; ++i) {
      if (i % 997 == 0) sim.clear_counts();
      if (i % 501 == 0) { int64_t ids[3] = {(int64_t)(rng() % 40), (int64_t)(rng() % 40), (int64_t)(rng() % 40)}; sim.protect(ids, 3); }
      int64_t ev = -1;
      hits += sim.access((int64_t)(rng() % 40), &ev);
      REQUIRE(ev >= -1 && ev < 40);
    }
    REQUIRE(hits > 0);
  }
  // tracer
  {
    const int L = 7, E = 20, cap = 9;
    Tracer tr(L, E, cap);
    std::vector<float> hist((size_t)6 * L * E);
    for (auto& v : hist) v = (float)(rng() % 5);
    tr.load(hist.data(), 6);
    std::vector<float> m((size_t)L * E);
    std::vector<int32_t> ls(L * E), es(L * E);
    std::vector<float> sc(L * E);
    for (int s = 0; s < 12; ++s) {
      const int64_t id = tr.create_entry();
      for (int step = 0; step < 5; ++step)
        for (int l = 0; l < L; ++l) {
          int32_t ex[3] = {(int32_t)(rng() % E), (int32_t)(rng() % E), (int32_t)(rng() % E)};
          const int nearest = tr.predict(id, l, ex, 3, m.data());
          REQUIRE(nearest >= 0 && nearest < cap);
          const int n = tr.prefetch_order(l, m.data(), ls.data(), es.data(), sc.data());
          REQUIRE(n >= 0 && n <= L * E);
          for (int i = 1; i < n; ++i) REQUIRE(sc[i - 1] >= sc[i]);
        }
      tr.finish_entry(id);
      REQUIRE(!tr.has(id));
    }
  }
  // offload store
  {
    const std::string dir = argv[1];
    OffloadStore st;
    REQUIRE(st.open(dir).empty());
    std::vector<std::vector<char>> payloads;
    for (uint32_t id = 0; id < 25; ++id) {
      std::vector<char> p(1 + rng() % 20000);
      for (auto& c : p) c = (char)rng();
      int64_t dims[2] = {(int64_t)p.size(), 1};
      REQUIRE(st.put(id * 3, p.data(), p.size(), dims, 2, 0).empty());
      payloads.push_back(p);
    }
    REQUIRE(st.flush().empty());
    OffloadStore rd;
    REQUIRE(rd.open(dir).empty());
    REQUIRE(rd.count() == 25);
    for (uint32_t id = 0; id < 25; ++id) {
      const TensorMeta* m = rd.find(id * 3);
      REQUIRE(m && m->size == payloads[id].size() && m->offset % 4096 == 0);
      // aligned destination with padded capacity -> O_DIRECT path; unaligned -> buffered path
      void* al = nullptr;
      REQUIRE(posix_memalign(&al, 4096, ((m->size + 4095) / 4096) * 4096) == 0);
      REQUIRE(rd.get(id * 3, al, ((m->size + 4095) / 4096) * 4096).empty());
      REQUIRE(memcmp(al, payloads[id].data(), m->size) == 0);
      free(al);
      std::vector<char> un(m->size + 1);
      REQUIRE(rd.get(id * 3, un.data() + 1, m->size).empty());
      REQUIRE(memcmp(un.data() + 1, payloads[id].data(), m->size) == 0);
    }
    // the same payloads through the two-priority block reader: mixed priorities, promotions, several threads
    {
      PrioAioPool pool(3, 8192);
      struct Pending { PrioAioPool::Handle h; void* buf; uint32_t id; };
      std::vector<Pending> pend;
      for (int round = 0; round < 3; ++round) {
        for (uint32_t id = 0; id < 25; ++id) {
          const TensorMeta* m = rd.find(id * 3);
          const uint64_t cap = ((m->size + 4095) / 4096) * 4096;
          void* al = nullptr;
          REQUIRE(posix_memalign(&al, 4096, cap) == 0);
          OffloadStore::ReadPlan rp;
          REQUIRE(rd.plan_read(id * 3, al, cap, &rp).empty());
          pend.push_back({pool.submit(rp.path, al, (int64_t)rp.size, rp.offset, (rng() & 3) == 0, rp.direct_ok), al, id});
          if ((rng() & 7) == 0) pool.promote(pend[rng() % pend.size()].h);
        }
      }
      for (auto& p : pend) {
        REQUIRE(PrioAioPool::wait(p.h).empty());
        REQUIRE(memcmp(p.buf, payloads[p.id].data(), payloads[p.id].size()) == 0);
        free(p.buf);
      }
      REQUIRE(pool.stats().blocks_high + pool.stats().blocks_low > 75);
      char tmp[16];
      REQUIRE(!PrioAioPool::wait(pool.submit(dir + "/no_such_file", tmp, 16, 0, true, false)).empty());
    }
    REQUIRE(!rd.get(999, payloads[0].data(), 10).empty());
    std::vector<char> small(4);
    REQUIRE(!rd.get(0, small.data(), 1).empty() || payloads[0].size() <= 1);
  }
  printf("host_fuzz ok\n");
  return 0;
}
