// capi_host.cpp — the host-only handles of the C ABI (include/moeinf.h): the offload directory in the reference's format
// (moeinf_store_*: core/aio/archer_tensor_index.cpp, archer_tensor_handle.cpp), the replacement-policy simulator, the
// pending-transfer queue (core/prefetch/task_scheduler.cpp), the two-priority block reader (core/aio/archer_prio_aio_handle.cpp)
// and the activation tracer (moe_infinity/memory/*.py), each a thin wrapper over its header.  No GPU needed except
// moeinf_store_get_device.  (moeinf_register_expert_from_store and moeinf_set_predictor reach into the engine: engine.cpp.)
#include "engine_internal.h"

// ---- disk tier -------------------------------------------------------------------------------
extern "C" int moeinf_store_open(const char* path, moeinf_store** out) {
  if (!path || !out) return fail(MOEINF_ERR_INVALID, "NULL argument");
  moeinf_store* st = new moeinf_store();
  const std::string err = st->s.open(path);
  if (!err.empty()) { delete st; return fail(MOEINF_ERR_INVALID, "%s", err.c_str()); }
  *out = st;
  return MOEINF_OK;
}
extern "C" int moeinf_store_close(moeinf_store* st) {
  if (!st) return MOEINF_OK;
  if (st->s.users > 0) return fail(MOEINF_ERR_STATE, "offload store is still backing %d experts of a live engine: destroy the engine first", st->s.users);
  int rc = MOEINF_OK;
  if (st->s.dirty()) { const std::string err = st->s.flush(); if (!err.empty()) rc = fail(MOEINF_ERR_INVALID, "%s", err.c_str()); }
  for (int i = 0; i < 2; ++i) { if (st->bounce[i]) hipHostFree(st->bounce[i]); if (st->bounce_ev[i]) hipEventDestroy(st->bounce_ev[i]); }
  delete st;
  return rc;
}
extern "C" int moeinf_store_put(moeinf_store* st, uint32_t id, const void* data, uint64_t nbytes, const int64_t* dims, int ndim, int scalar_type) {
  if (!st || !data || ndim < 0 || ndim > 8 || (ndim > 0 && !dims)) return fail(MOEINF_ERR_INVALID, "bad store_put arguments");
  const std::string err = st->s.put(id, data, nbytes, dims, ndim, scalar_type);
  return err.empty() ? MOEINF_OK : fail(MOEINF_ERR_INVALID, "%s", err.c_str());
}
extern "C" int moeinf_store_flush(moeinf_store* st) {
  if (!st) return fail(MOEINF_ERR_INVALID, "store is NULL");
  const std::string err = st->s.flush();
  return err.empty() ? MOEINF_OK : fail(MOEINF_ERR_INVALID, "%s", err.c_str());
}
extern "C" int moeinf_store_count(const moeinf_store* st, int64_t* n) {
  if (!st || !n) return fail(MOEINF_ERR_INVALID, "NULL argument");
  *n = (int64_t)st->s.count();
  return MOEINF_OK;
}
extern "C" int moeinf_store_ids(const moeinf_store* st, uint32_t* ids_out, int64_t capacity) {
  if (!st || !ids_out) return fail(MOEINF_ERR_INVALID, "NULL argument");
  const auto v = st->s.ids();
  if ((int64_t)v.size() > capacity) return fail(MOEINF_ERR_INVALID, "ids_out holds %lld ids, store has %zu", (long long)capacity, v.size());
  std::copy(v.begin(), v.end(), ids_out);
  return MOEINF_OK;
}
extern "C" int moeinf_store_meta(const moeinf_store* st, uint32_t id, int32_t* found, uint64_t* nbytes, int64_t* offset, int32_t* ndim, int64_t* dims_out, int32_t* scalar_type) {
  if (!st || !found) return fail(MOEINF_ERR_INVALID, "NULL argument");
  const TensorMeta* m = st->s.find(id);
  *found = m ? 1 : 0;
  if (!m) return MOEINF_OK;
  if (nbytes) *nbytes = m->size;
  if (offset) *offset = m->offset;
  if (ndim) *ndim = (int32_t)m->shape.size();
  if (dims_out) for (size_t i = 0; i < m->shape.size() && i < 8; ++i) dims_out[i] = m->shape[i];
  if (scalar_type) *scalar_type = m->dtype;
  return MOEINF_OK;
}
extern "C" int moeinf_store_get(const moeinf_store* st, uint32_t id, void* dst, uint64_t capacity) {
  if (!st || !dst) return fail(MOEINF_ERR_INVALID, "NULL argument");
  const std::string err = st->s.get(id, dst, capacity);
  return err.empty() ? MOEINF_OK : fail(MOEINF_ERR_INVALID, "%s", err.c_str());
}
// disk -> device for a DENSE tensor (Node::SetDevice's disk->host->device legs for non-expert nodes,
// model_topology.cpp:76-119; AcquireTensor / FetchTensors, archer_prefetch_handle.cpp:83-130,220-227): the payload is
// read in 32 MiB pieces into two pinned bounce buffers and copied with hipMemcpyAsync on `stream`, the read of piece
// i+1 overlapping the copy of piece i.  Returns when the last copy has been ENQUEUED and the bounce buffers are free
// again (the device data is stream-ordered after the call).
extern "C" int moeinf_store_get_device(moeinf_store* st, uint32_t id, void* dst_dev, uint64_t capacity, void* stream) {
  if (!st || !dst_dev) return fail(MOEINF_ERR_INVALID, "NULL argument");
  const TensorMeta* m = st->s.find(id);
  if (!m) return fail(MOEINF_ERR_INVALID, "tensor %u is not in the offload index", id);
  if (m->size > capacity) return fail(MOEINF_ERR_INVALID, "tensor %u is %llu bytes, destination holds %llu", id, (unsigned long long)m->size, (unsigned long long)capacity);
  constexpr uint64_t kPiece = 32ull << 20;
  if (!st->bounce[0]) {
    for (int i = 0; i < 2; ++i) {
      HIPCHK(hipHostMalloc(&st->bounce[i], kPiece, hipHostMallocDefault));
      HIPCHK(hipEventCreateWithFlags(&st->bounce_ev[i], hipEventDisableTiming));
    }
  }
  hipStream_t s = (hipStream_t)stream;
  int b = 0;
  for (uint64_t off = 0; off < m->size; off += kPiece, b ^= 1) {
    const uint64_t n = std::min<uint64_t>(kPiece, m->size - off);
    if (st->bounce_used[b]) HIPCHK(hipEventSynchronize(st->bounce_ev[b]));
    const std::string err = st->s.get_range(id, off, st->bounce[b], n);
    if (!err.empty()) return fail(MOEINF_ERR_INVALID, "%s", err.c_str());
    HIPCHK(hipMemcpyAsync((char*)dst_dev + off, st->bounce[b], n, hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(st->bounce_ev[b], s));
    st->bounce_used[b] = true;
  }
  for (int i = 0; i < 2; ++i) if (st->bounce_used[i]) HIPCHK(hipEventSynchronize(st->bounce_ev[i]));
  return MOEINF_OK;
}

// ---- cache simulator -----------------------------------------------------------------------
struct moeinf_cache_sim { CacheSim* sim; };
extern "C" int moeinf_cache_sim_create(int num_slots, int policy, moeinf_cache_sim** out) {
  if (!out || num_slots <= 0 || (policy != POLICY_LFU_INCACHE && policy != POLICY_LRU)) return fail(MOEINF_ERR_INVALID, "bad cache_sim arguments");
  *out = new moeinf_cache_sim{new CacheSim(num_slots, policy)};
  return MOEINF_OK;
}
extern "C" int moeinf_cache_sim_destroy(moeinf_cache_sim* s) { if (s) { delete s->sim; delete s; } return MOEINF_OK; }
extern "C" int moeinf_cache_sim_access(moeinf_cache_sim* s, int64_t id, int32_t* hit, int64_t* evicted) {
  if (!s || id < 0) return fail(MOEINF_ERR_INVALID, "bad cache_sim_access arguments");
  int64_t ev = -1;
  const bool h = s->sim->access(id, &ev);
  if (hit) *hit = h ? 1 : 0;
  if (evicted) *evicted = ev;
  return MOEINF_OK;
}
extern "C" int moeinf_cache_sim_protect(moeinf_cache_sim* s, const int64_t* ids, int n) {
  if (!s || n < 0 || (n > 0 && !ids)) return fail(MOEINF_ERR_INVALID, "bad cache_sim_protect arguments");
  s->sim->protect(ids, n);
  return MOEINF_OK;
}
extern "C" int moeinf_cache_sim_clear_counts(moeinf_cache_sim* s) {
  if (!s) return fail(MOEINF_ERR_INVALID, "sim is NULL");
  s->sim->clear_counts();
  return MOEINF_OK;
}

// ---- pending-transfer queue, standalone (host only) -----------------------------------------
struct moeinf_pq { PrefetchQueue q; };
extern "C" int moeinf_pq_create(moeinf_pq** out) {
  if (!out) return fail(MOEINF_ERR_INVALID, "out is NULL");
  *out = new moeinf_pq();
  return MOEINF_OK;
}
extern "C" int moeinf_pq_destroy(moeinf_pq* q) { delete q; return MOEINF_OK; }
extern "C" int moeinf_pq_enqueue(moeinf_pq* q, int64_t node, int layer, int priority, int remove_layer, int32_t* dropped) {
  if (!q || node < 0) return fail(MOEINF_ERR_INVALID, "bad pq_enqueue arguments");
  const int d = q->q.enqueue(node, layer, priority, remove_layer != 0);
  if (dropped) *dropped = d;
  return MOEINF_OK;
}
extern "C" int moeinf_pq_on_demand(moeinf_pq* q, int64_t node, int layer, int32_t* dropped) {
  if (!q) return fail(MOEINF_ERR_INVALID, "queue is NULL");
  const int d = q->q.on_demand(node, layer);
  if (dropped) *dropped = d;
  return MOEINF_OK;
}
extern "C" int moeinf_pq_fetch(moeinf_pq* q, int64_t node, int layer, int already_there, int32_t* dropped) {
  if (!q || node < 0) return fail(MOEINF_ERR_INVALID, "bad pq_fetch arguments");
  const int d = q->q.fetch(node, layer, already_there != 0);
  if (dropped) *dropped = d;
  return MOEINF_OK;
}
extern "C" int moeinf_pq_clear_prefetch(moeinf_pq* q, int32_t* dropped) {
  if (!q) return fail(MOEINF_ERR_INVALID, "queue is NULL");
  const int d = q->q.clear_prefetch();
  if (dropped) *dropped = d;
  return MOEINF_OK;
}
extern "C" int moeinf_pq_pop(moeinf_pq* q, int64_t* node, int32_t* layer, int32_t* priority, int32_t* found) {
  if (!q || !found) return fail(MOEINF_ERR_INVALID, "NULL argument");
  QueuedTask t;
  *found = q->q.pop(&t) ? 1 : 0;
  if (*found) { if (node) *node = t.node; if (layer) *layer = t.layer; if (priority) *priority = t.priority; }
  return MOEINF_OK;
}
extern "C" int moeinf_pq_snapshot(const moeinf_pq* q, int64_t* nodes, int32_t* layers, int32_t* priorities, int capacity, int32_t* n) {
  if (!q || !n) return fail(MOEINF_ERR_INVALID, "NULL argument");
  const auto v = q->q.snapshot();
  if ((int)v.size() > capacity) return fail(MOEINF_ERR_INVALID, "snapshot needs room for %zu tasks", v.size());
  for (size_t i = 0; i < v.size(); ++i) { if (nodes) nodes[i] = v[i].node; if (layers) layers[i] = v[i].layer; if (priorities) priorities[i] = v[i].priority; }
  *n = (int32_t)v.size();
  return MOEINF_OK;
}
extern "C" int moeinf_priority_from_score(float score, int32_t* level) {
  if (!level) return fail(MOEINF_ERR_INVALID, "level is NULL");
  *level = priority_from_score(&score, 0);
  return MOEINF_OK;
}

// ---- priority block reader, standalone (host only) -------------------------------------------
struct moeinf_aio {
  PrioAioPool pool;
  std::mutex mu;
  std::map<int64_t, PrioAioPool::Handle> reqs;
  int64_t next = 1;
  moeinf_aio(int threads, int64_t block) : pool(threads, block) {}
};
extern "C" int moeinf_aio_create(int threads, int64_t block_bytes, moeinf_aio** out) {
  if (!out || threads <= 0 || block_bytes <= 0) return fail(MOEINF_ERR_INVALID, "bad aio arguments");
  *out = new moeinf_aio(threads, block_bytes);
  return MOEINF_OK;
}
extern "C" int moeinf_aio_destroy(moeinf_aio* a) { delete a; return MOEINF_OK; }
extern "C" int moeinf_aio_submit_read(moeinf_aio* a, const char* path, void* dst, int64_t nbytes, int64_t offset, int high_prio, int try_direct, int64_t* request) {
  if (!a || !path || !dst || !request || nbytes < 0 || offset < 0) return fail(MOEINF_ERR_INVALID, "bad aio read arguments");
  auto h = a->pool.submit(path, dst, nbytes, offset, high_prio != 0, try_direct != 0);
  std::lock_guard<std::mutex> lk(a->mu);
  *request = a->next++;
  a->reqs[*request] = h;
  return MOEINF_OK;
}
static PrioAioPool::Handle aio_find(moeinf_aio* a, int64_t request, bool take) {
  std::lock_guard<std::mutex> lk(a->mu);
  auto it = a->reqs.find(request);
  if (it == a->reqs.end()) return nullptr;
  auto h = it->second;
  if (take) a->reqs.erase(it);
  return h;
}
extern "C" int moeinf_aio_promote(moeinf_aio* a, int64_t request) {
  if (!a) return fail(MOEINF_ERR_INVALID, "aio is NULL");
  auto h = aio_find(a, request, false);
  if (!h) return fail(MOEINF_ERR_INVALID, "unknown aio request %lld", (long long)request);
  a->pool.promote(h);
  return MOEINF_OK;
}
extern "C" int moeinf_aio_done(moeinf_aio* a, int64_t request, int32_t* done) {
  if (!a || !done) return fail(MOEINF_ERR_INVALID, "NULL argument");
  auto h = aio_find(a, request, false);
  if (!h) return fail(MOEINF_ERR_INVALID, "unknown aio request %lld", (long long)request);
  *done = PrioAioPool::done(h) ? 1 : 0;
  return MOEINF_OK;
}
extern "C" int moeinf_aio_wait(moeinf_aio* a, int64_t request) {
  if (!a) return fail(MOEINF_ERR_INVALID, "aio is NULL");
  auto h = aio_find(a, request, true);
  if (!h) return fail(MOEINF_ERR_INVALID, "unknown aio request %lld", (long long)request);
  const std::string err = PrioAioPool::wait(h);
  if (!err.empty()) return fail(MOEINF_ERR_INVALID, "%s", err.c_str());
  return MOEINF_OK;
}
extern "C" int moeinf_aio_stats(const moeinf_aio* a, int64_t out[5]) {
  if (!a || !out) return fail(MOEINF_ERR_INVALID, "NULL argument");
  const auto s = a->pool.stats();
  out[0] = s.blocks_high; out[1] = s.blocks_low; out[2] = s.bytes; out[3] = s.promoted; out[4] = s.direct_fallbacks;
  return MOEINF_OK;
}

// ---- tracer --------------------------------------------------------------------------------
extern "C" int moeinf_tracer_create(int L, int E, int capacity, moeinf_tracer** out) {
  if (!out || L <= 0 || E <= 0 || capacity <= 0) return fail(MOEINF_ERR_INVALID, "bad tracer arguments");
  *out = new moeinf_tracer{new Tracer(L, E, capacity)};
  return MOEINF_OK;
}
extern "C" int moeinf_tracer_destroy(moeinf_tracer* t) { if (t) { delete t->t; delete t; } return MOEINF_OK; }
extern "C" int moeinf_tracer_load(moeinf_tracer* t, const float* eams, int n) {
  if (!t || !eams || n < 0 || n > t->t->capacity()) return fail(MOEINF_ERR_INVALID, "tracer_load: n must be in 0..capacity");
  t->t->load(eams, n);
  return MOEINF_OK;
}
extern "C" int moeinf_tracer_create_entry(moeinf_tracer* t, int64_t* seq_id) {
  if (!t || !seq_id) return fail(MOEINF_ERR_INVALID, "NULL argument");
  *seq_id = t->t->create_entry();
  return MOEINF_OK;
}
extern "C" int moeinf_tracer_finish_entry(moeinf_tracer* t, int64_t seq_id) {
  if (!t || !t->t->has(seq_id)) return fail(MOEINF_ERR_INVALID, "unknown seq_id");
  t->t->finish_entry(seq_id);
  return MOEINF_OK;
}
extern "C" int moeinf_tracer_predict(moeinf_tracer* t, int64_t seq_id, int layer, const int32_t* experts, int n, float* matrix_out, int32_t* nearest_out) {
  if (!t || !t->t->has(seq_id) || !matrix_out || n < 0 || (n > 0 && !experts)) return fail(MOEINF_ERR_INVALID, "bad tracer_predict arguments");
  if (layer < 0 || layer >= t->t->layers()) return fail(MOEINF_ERR_INVALID, "layer out of range");
  for (int i = 0; i < n; ++i) if (experts[i] < 0 || experts[i] >= t->t->experts()) return fail(MOEINF_ERR_INVALID, "expert id out of range");
  int nearest = t->t->predict(seq_id, layer, experts, n, matrix_out);
  if (nearest_out) *nearest_out = nearest;
  return MOEINF_OK;
}
extern "C" int moeinf_tracer_prefetch_order(const moeinf_tracer* t, int layer, const float* matrix, int32_t* layers_out, int32_t* experts_out, float* scores_out, int32_t* n_out) {
  if (!t || !matrix || !layers_out || !experts_out || !n_out) return fail(MOEINF_ERR_INVALID, "NULL argument");
  if (layer < 0 || layer >= t->t->layers()) return fail(MOEINF_ERR_INVALID, "layer out of range");
  *n_out = t->t->prefetch_order(layer, matrix, layers_out, experts_out, scores_out);
  return MOEINF_OK;
}
extern "C" int moeinf_tracer_get_eam(moeinf_tracer* t, int64_t seq_id, double* eam_out) {
  if (!t || !t->t->has(seq_id) || !eam_out) return fail(MOEINF_ERR_INVALID, "bad tracer_get_eam arguments");
  t->t->get_eam(seq_id, eam_out);
  return MOEINF_OK;
}

