// engine_ep.cpp — expert parallelism on the host side (include/moeinf.h: moeinf_ep_*): the pack / owner-FFN / combine steps
// around an exchange, RCCL called from inside the engine (ep_comm.h), and the direct peer-store exchange (ep_peer.h) with
// moeinf_ep_moe_forward = one host call per layer.  Replaces the reference's one-process multi-GPU dispatch
// (core/parallel/expert_dispatcher.cpp:284,405 P2P copies; core/prefetch/archer_prefetch_handle.cpp:37-61 peer access).
#include "engine_internal.h"

// ---- expert-parallel helpers ---------------------------------------------------------------
static int ep_alloc(moeinf_engine* g, int cap_rows) {
  const size_t np = (size_t)g->cfg.max_tokens * g->K;
  if (g->d_ep_key && g->ep_alloc_cap >= cap_rows && g->ep_alloc_np >= (int64_t)np) { g->ep_cap_rows = cap_rows; return MOEINF_OK; }
  if (g->d_ep_key) HIPCHK(hipDeviceSynchronize());  // kernels of earlier layers may still use the old buffers
  free_ep_workspace(g);
  const size_t nr = std::max<size_t>(np, (size_t)g->cfg.ep_size * cap_rows);
  const size_t nk = std::max<size_t>((size_t)g->cfg.ep_size, (size_t)g->E) + 2;
  CHK(dmalloc(&g->d_ep_key, nr)); CHK(dmalloc(&g->d_ep_counts, nk)); CHK(dmalloc(&g->d_ep_offsets, nk + 1)); CHK(dmalloc(&g->d_ep_active, nk));
  CHK(dmalloc(&g->d_ep_nactive, 1)); CHK(dmalloc(&g->d_ep_pair_slot, nr)); CHK(dmalloc(&g->d_ep_slot_token, nr + 1)); CHK(dmalloc(&g->d_ep_slot_pair, nr + 1));
  CHK(dmalloc(&g->d_ep_pair_pos, np));
  g->ep_cap_rows = cap_rows;
  g->ep_alloc_cap = cap_rows;
  g->ep_alloc_np = (int64_t)np;
  return MOEINF_OK;
}

static int64_t ep_row_elems(const moeinf_engine* g) { return g->H + 16 / g->es; }
// fewest row slots per peer that can never overflow: a token sends a rank at most one row per expert that rank owns
static int ep_min_cap(const moeinf_engine* g, int T) {
  const int per_rank = (g->E + g->cfg.ep_size - 1) / g->cfg.ep_size;
  return T * std::min(g->K, per_rank);
}
static int ep_pack_fixed(moeinf_engine* g, const void* x_dev, void* send_dev, int32_t* send_counts_dev, int cap_rows, hipStream_t st, const EpPeers* pv = nullptr);
static int ep_peer_forward(moeinf_engine* g, int layer, const void* x_dev, int tokens, int batch_rows, const void* gate_w_dev, void* out_dev, void* stream);

extern "C" int moeinf_ep_row_elems(const moeinf_engine* g, int32_t* elems) {
  if (!g || !elems) return fail(MOEINF_ERR_INVALID, "NULL argument");
  *elems = (int32_t)ep_row_elems(g);
  return MOEINF_OK;
}

extern "C" int moeinf_ep_pack(moeinf_engine* g, const void* x_dev, void* send_dev, int32_t* send_counts_dev, int cap_rows, void* stream) {
  if (!g || !x_dev || !send_dev) return fail(MOEINF_ERR_INVALID, "NULL argument");
  if (g->last_layer < 0) return fail(MOEINF_ERR_STATE, "ep_pack needs a preceding ROUTE_ONLY forward");
  if (cap_rows < ep_min_cap(g, g->last_T)) return fail(MOEINF_ERR_INVALID, "cap_rows %d < %d = tokens * min(K, experts per rank) (worst case: every pair a rank can receive from these tokens)", cap_rows, ep_min_cap(g, g->last_T));
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  hipStream_t st = (hipStream_t)stream;
  CHK(ep_alloc(g, cap_rows));
  return ep_pack_fixed(g, x_dev, send_dev, send_counts_dev, cap_rows, st);
}
static int ep_pack_fixed(moeinf_engine* g, const void* x_dev, void* send_dev, int32_t* send_counts_dev, int cap_rows, hipStream_t st, const EpPeers* pv) {
  const int np = g->last_T * g->K, ep = g->cfg.ep_size;
  if (np <= 64) {  // decode: one launch
    EpPackArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.x = x_dev; pa.send = send_dev; pa.ld_send = ep_row_elems(g); pa.pair_pos = g->d_ep_pair_pos; pa.topk_idx = g->d_topk_idx;
    pa.K = g->K; pa.H = g->H; pa.ep_size = ep; pa.cap_rows = cap_rows; pa.dtype = g->dt;
    HIPCHK(launch_ep_pack_small(pa, g->d_pair_valid, np, send_counts_dev, st, pv));
    return MOEINF_OK;
  }
  HIPCHK(launch_ep_dest_key(g->d_topk_idx, g->d_pair_valid, g->d_ep_key, g->d_ep_pair_pos, np, ep, st));
  IndexArgs ia;
  memset(&ia, 0, sizeof ia);
  ia.topk_idx = g->d_ep_key; ia.pair_valid = nullptr; ia.T = np; ia.K = 1; ia.E = ep; ia.rows = 1; ia.capacity = 0; ia.shared = 0;
  ia.counts = g->d_ep_counts; ia.offsets = g->d_ep_offsets; ia.active = g->d_ep_active; ia.n_active = g->d_ep_nactive;
  ia.pair_slot = g->d_ep_pair_slot; ia.slot_token = g->d_ep_slot_token; ia.slot_pair = g->d_ep_slot_pair; ia.mirror = nullptr;
  HIPCHK(launch_dispatch_index(ia, st));
  EpPackArgs pa;
  memset(&pa, 0, sizeof pa);
  pa.x = x_dev; pa.send = send_dev; pa.ld_send = ep_row_elems(g); pa.pair_pos = g->d_ep_pair_pos; pa.topk_idx = g->d_topk_idx;
  pa.counts = g->d_ep_counts; pa.offsets = g->d_ep_offsets; pa.slot_pair = g->d_ep_slot_pair;
  pa.K = g->K; pa.H = g->H; pa.ep_size = ep; pa.cap_rows = cap_rows; pa.dtype = g->dt;
  HIPCHK(launch_ep_pack(pa, st, pv));
  if (send_counts_dev) HIPCHK(hipMemcpyAsync(send_counts_dev, g->d_ep_counts, (size_t)ep * 4, hipMemcpyDeviceToDevice, st));
  return MOEINF_OK;
}

// Sender side of the fixed-capacity exchange in ONE call: gate (+ stage 1 of a hidden DeepSeek shared expert) -> top-k +
// dispatch index (+ its stage 2) -> send rows.  For decode-sized forwards the send rows are written by the
// single-workgroup router launch itself (EpFuse): two launches per layer before the all-to-all.
static int ep_route_pack_impl(moeinf_engine* g, int layer, const void* x_dev, int tokens, int batch_rows, const void* gate_w_dev,
                              void* send_dev, int32_t* send_counts_dev, int cap_rows, void* stream, const EpPeers* pv);
extern "C" int moeinf_ep_route_pack(moeinf_engine* g, int layer, const void* x_dev, int tokens, int batch_rows, const void* gate_w_dev,
                                    void* send_dev, int32_t* send_counts_dev, int cap_rows, void* stream) {
  if (!g || !send_dev) return fail(MOEINF_ERR_INVALID, "NULL argument");
  return ep_route_pack_impl(g, layer, x_dev, tokens, batch_rows, gate_w_dev, send_dev, send_counts_dev, cap_rows, stream, nullptr);
}
// pv != nullptr: the peer-store exchange — the rows go straight into the destination ranks' windows (send_dev unused)
static int ep_route_pack_impl(moeinf_engine* g, int layer, const void* x_dev, int tokens, int batch_rows, const void* gate_w_dev,
                              void* send_dev, int32_t* send_counts_dev, int cap_rows, void* stream, const EpPeers* pv) {
  if (!g || !x_dev || !gate_w_dev || (!send_dev && !pv)) return fail(MOEINF_ERR_INVALID, "NULL argument");
  if (layer < 0 || layer >= g->L) return fail(MOEINF_ERR_INVALID, "layer %d out of range", layer);
  if (tokens <= 0 || tokens > g->cfg.max_tokens) return fail(MOEINF_ERR_INVALID, "tokens %d not in 1..max_tokens(%d)", tokens, g->cfg.max_tokens);
  if (batch_rows <= 0 || tokens % batch_rows) return fail(MOEINF_ERR_INVALID, "tokens %d not divisible by batch_rows %d", tokens, batch_rows);
  if (cap_rows < ep_min_cap(g, tokens)) return fail(MOEINF_ERR_INVALID, "cap_rows %d < %d = tokens * min(K, experts per rank)", cap_rows, ep_min_cap(g, tokens));
  if (g->has_shared && !g->shared_dev[layer]) return fail(MOEINF_ERR_STATE, "shared expert of layer %d not registered", layer);
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  hipStream_t st = (hipStream_t)stream;
  const int T = tokens, K = g->K, np = T * K;
  CHK(ep_alloc(g, cap_rows));
  RouteArgs ra;
  make_route_args(g, x_dev, gate_w_dev, T, ra);
  if (g->route_v3) ra.e_bias = g->gate_bias[layer];
  IndexArgs ia;
  make_index_args(g, T, batch_rows, nullptr, ia);
  ia.shared = 0;  // the owner-side index is built from the received rows; the shared expert never crosses the fabric
  const bool hide_shared = can_hide_shared(g, T);
  g->last_hidden_shared = hide_shared;
  g->last_selfroute = false;
  // the pack rides in the router's single-workgroup launch while the rows are few KB (one workgroup copies them)
  static const int fuse_kb = getenv("MOEINF_EP_FUSE_PACK_KB") ? atoi(getenv("MOEINF_EP_FUSE_PACK_KB")) : 64;
  const bool fuse = np <= 64 && T <= 64 && (int64_t)np * g->H * g->es <= (int64_t)fuse_kb * 1024;
  EpFuse pk;
  memset(&pk, 0, sizeof pk);
  pk.a.x = x_dev; pk.a.send = send_dev; pk.a.ld_send = ep_row_elems(g); pk.a.pair_pos = g->d_ep_pair_pos; pk.a.topk_idx = g->d_topk_idx;
  pk.a.K = K; pk.a.H = g->H; pk.a.ep_size = g->cfg.ep_size; pk.a.cap_rows = cap_rows; pk.a.dtype = g->dt;
  pk.pair_valid = g->d_pair_valid; pk.send_counts = send_counts_dev; pk.on = 1;
  if (pv) pk.peers = *pv;
  if (hide_shared) {
    FfnStage sh1, sh2;
    hidden_shared_stages(g, layer, x_dev, sh1, sh2);
    HIPCHK(launch_gate_shared1(ra, sh1, st));
    HIPCHK(launch_route_shared2(ra, ia, sh2, st, fuse ? &pk : nullptr));
  } else {
    HIPCHK(launch_gate_logits(ra, st));
    if (T <= 64) {
      HIPCHK(launch_route_index(ra, ia, st, fuse ? &pk : nullptr));
    } else {
      HIPCHK(launch_route_topk(ra, st));
      CHK(launch_index_auto(g, ia, st));
    }
  }
  g->last_T = T; g->last_layer = layer; g->last_stream = st;
  g->st.forwards += 1;
  if (!fuse) CHK(ep_pack_fixed(g, x_dev, send_dev, send_counts_dev, cap_rows, st, pv));
  return MOEINF_OK;
}

extern "C" int moeinf_ep_pack_compact(moeinf_engine* g, const void* x_dev, void* send_dev, int32_t* send_counts_dev, void* stream) {
  if (!g || !x_dev || !send_dev || !send_counts_dev) return fail(MOEINF_ERR_INVALID, "NULL argument");
  if (g->last_layer < 0) return fail(MOEINF_ERR_STATE, "ep_pack_compact needs a preceding ROUTE_ONLY forward");
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  hipStream_t st = (hipStream_t)stream;
  const int np = g->last_T * g->K, ep = g->cfg.ep_size;
  CHK(ep_alloc(g, std::max(1, np)));
  g->ep_cap_rows = 0;  // compact mode: ep_combine takes cap_rows == 0
  HIPCHK(launch_ep_dest_key(g->d_topk_idx, g->d_pair_valid, g->d_ep_key, g->d_ep_pair_pos, np, ep, st));
  IndexArgs ia;
  memset(&ia, 0, sizeof ia);
  ia.topk_idx = g->d_ep_key; ia.pair_valid = nullptr; ia.T = np; ia.K = 1; ia.E = ep; ia.rows = 1; ia.capacity = 0; ia.shared = 0;
  ia.counts = g->d_ep_counts; ia.offsets = g->d_ep_offsets; ia.active = g->d_ep_active; ia.n_active = g->d_ep_nactive;
  ia.pair_slot = g->d_ep_pair_slot; ia.slot_token = g->d_ep_slot_token; ia.slot_pair = g->d_ep_slot_pair; ia.mirror = nullptr;
  CHK(launch_index_auto(g, ia, st));
  EpPackArgs pa;
  memset(&pa, 0, sizeof pa);
  pa.x = x_dev; pa.send = send_dev; pa.ld_send = ep_row_elems(g); pa.pair_pos = g->d_ep_pair_pos; pa.topk_idx = g->d_topk_idx;
  pa.counts = g->d_ep_counts; pa.offsets = g->d_ep_offsets; pa.slot_pair = g->d_ep_slot_pair;
  pa.K = g->K; pa.H = g->H; pa.ep_size = ep; pa.cap_rows = 0; pa.dtype = g->dt;
  HIPCHK(launch_ep_pack_compact(pa, np, st));
  HIPCHK(hipMemcpyAsync(send_counts_dev, g->d_ep_counts, (size_t)ep * 4, hipMemcpyDeviceToDevice, st));
  return MOEINF_OK;
}

static int ep_expert_ffn_rows(moeinf_engine* g, int layer, const void* recv_dev, void* y_dev, int nrows, hipStream_t st, const EpPeers* pv = nullptr);

extern "C" int moeinf_ep_expert_ffn(moeinf_engine* g, int layer, const void* recv_dev, void* y_dev, int cap_rows, void* stream) {
  if (!g || !recv_dev || !y_dev) return fail(MOEINF_ERR_INVALID, "NULL argument");
  if (cap_rows <= 0) return fail(MOEINF_ERR_INVALID, "cap_rows must be > 0");
  return ep_expert_ffn_rows(g, layer, recv_dev, y_dev, g->cfg.ep_size * cap_rows, (hipStream_t)stream);
}
extern "C" int moeinf_ep_expert_ffn_rows(moeinf_engine* g, int layer, const void* recv_dev, void* y_dev, int nrows, void* stream) {
  if (!g || !recv_dev || !y_dev) return fail(MOEINF_ERR_INVALID, "NULL argument");
  if (nrows < 0) return fail(MOEINF_ERR_INVALID, "nrows must be >= 0");
  if (nrows == 0) return MOEINF_OK;  // nothing was routed to this rank
  return ep_expert_ffn_rows(g, layer, recv_dev, y_dev, nrows, (hipStream_t)stream);
}
// pv != nullptr (peer-store exchange): recv_dev is this rank's window; the self-indexing kernels poll the row flags and
// store their outputs into the home ranks' windows themselves, the generic path gets a wait kernel in front and a push
// kernel behind (y_dev = a local staging buffer there)
static int ep_expert_ffn_rows(moeinf_engine* g, int layer, const void* recv_dev, void* y_dev, int nrows, hipStream_t st, const EpPeers* pv) {
  if (layer < 0 || layer >= g->L) return fail(MOEINF_ERR_INVALID, "layer out of range");
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  const int E = g->E;
  if ((int64_t)nrows > (int64_t)g->cfg.max_tokens * g->K) return fail(MOEINF_ERR_INVALID, "ep rows %d exceed workspace (max_tokens*K = %d): create the engine with max_tokens >= ep_size*cap_rows/K", nrows, g->cfg.max_tokens * g->K);
  const int64_t ld = ep_row_elems(g);
  IndexArgs ia;
  memset(&ia, 0, sizeof ia);
  // the expert id of every received row sits in the row's 16-byte tail
  ia.topk_idx = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(recv_dev) + (size_t)g->H * g->es);
  ia.idx_stride = (int)(ld * g->es / 4);
  ia.pair_valid = nullptr; ia.T = nrows; ia.K = 1; ia.E = E; ia.rows = 1; ia.capacity = 0; ia.shared = 0;
  ia.counts = g->d_counts; ia.offsets = g->d_offsets; ia.active = g->d_active; ia.n_active = g->d_n_active;
  MirrorPlan mp;
  drop_stale_prefetches(g, layer);
  CHK(plan_mirror(g, layer, mp));
  const int owned = std::max(1, g->owned_experts);
  // Decode-sized exchange on the sync-free path: both FFN stages index for themselves from the row tails
  // (launch_ffn_ep_stage) — no dispatch-index launch between the all-to-all and the weight stream.
  static const bool selfindex_env = getenv("MOEINF_EP_SELFINDEX") ? atoi(getenv("MOEINF_EP_SELFINDEX")) != 0 : true;
  if (selfindex_env && mp.fast && nrows <= 64 && (E - 1) / g->cfg.ep_size < 64) {
    moeinf_engine::PendingMirror pm;
    pm.buf = mp.target; pm.seq = g->seq + 1; pm.layer = layer; pm.T = nrows; pm.prof = g->profiling; pm.local = false;
    g->pend.push_back(pm);
    for (int e = 0; e < E; ++e) {  // any of the layer's slots may be read by this forward
      const Node& n = g->nodes[node_index(g, layer, e)];
      if (n.slot >= 0) g->slots[n.slot].last_use_seq = g->seq + 1;
    }
    CHK(flush_pokes(g, st));
    FfnStage s1, s2;
    fill_stage(g, layer, 1, s1, ld);
    s1.in = recv_dev; s1.row_map = nullptr;
    fill_stage(g, layer, 2, s2);
    s2.out = y_dev; s2.out_map = nullptr;
    EpOwnArgs o;
    memset(&o, 0, sizeof o);
    o.recv = recv_dev; o.ld_recv = ld; o.H = g->H; o.nrows = nrows; o.ep_size = g->cfg.ep_size; o.ep_rank = g->cfg.ep_rank;
    o.max_active = std::min(owned, nrows);
    o.rec = g->d_ep_rec;
    if (pv) {
      o.peers = *pv;
      o.tile_done = g->d_arrive;
      if (!pv->poll) {  // ranks sharing a GPU: one wave waits, the wide kernel starts when the rows are there
        EpWait w{g->ep_win.recv_flags(), pv->size, pv->epoch, pv->timeout_ticks, pv->err};
        HIPCHK(launch_ep_wait(w, st));
      }
    }
    moeinf_engine::ProfRec pr;
    const bool prof = g->profiling;
    if (prof) {
      for (int i = 0; i < 6; ++i) { pr.ev[i] = get_event(g); if (!pr.ev[i]) return fail(MOEINF_ERR_HIP, "hipEventCreate failed"); }
      record_timing(pr.ev[0], st); record_timing(pr.ev[1], st); record_timing(pr.ev[2], st);
    }
    o.stage = 1; o.mirror = mp.target;
    HIPCHK(launch_ffn_ep_stage(s1, o, st));
    if (prof) record_timing(pr.ev[3], st);
    o.stage = 2; o.mirror = nullptr;
    HIPCHK(launch_ffn_ep_stage(s2, o, st));
    if (prof) { record_timing(pr.ev[4], st); record_timing(pr.ev[5], st); g->prof_pending.push_back(pr); }
    g->st.forwards += 1;
    CHK(end_forward(g, st, false));  // (the sync-free branch)
    return pump_if_pending(g);
  }
  ia.pair_slot = g->d_pair_slot; ia.slot_token = g->d_slot_token; ia.slot_pair = g->d_slot_pair; ia.mirror = mp.target;
  if (pv) {  // the generic kernels know nothing of the exchange: wait in front of them ...
    EpWait w{g->ep_win.recv_flags(), pv->size, pv->epoch, pv->timeout_ticks, pv->err};
    HIPCHK(launch_ep_wait(w, st));
  }
  CHK(launch_index_auto(g, ia, st));
  // stage 2 scatters every output row to its arrival position in y_dev (slot_pair: expert-sorted row -> received
  // row), so the reply needs no un-sort pass
  g->ovr_out = y_dev; g->ovr_map = g->d_slot_pair;
  // profiling: only the two FFN stages are bracketed here (events 2..4); the other intervals are empty
  moeinf_engine::ProfRec pr;
  const bool prof = g->profiling;
  if (prof) {
    for (int i = 0; i < 6; ++i) { pr.ev[i] = get_event(g); if (!pr.ev[i]) { g->ovr_out = nullptr; g->ovr_map = nullptr; return fail(MOEINF_ERR_HIP, "hipEventCreate failed"); } }
    record_timing(pr.ev[0], st); record_timing(pr.ev[1], st);
  }
  const int rc = dispatch_experts(g, layer, recv_dev, ld, nrows, std::min(owned, nrows),
                                  (int)std::min<int64_t>(nrows, ((int64_t)nrows * 3) / (2 * owned) + 1), st, prof, prof ? &pr : nullptr, mp, nullptr, nullptr);
  g->ovr_out = nullptr; g->ovr_map = nullptr;
  if (rc != MOEINF_OK) return rc;
  if (pv) HIPCHK(launch_ep_push(y_dev, recv_dev, ld, g->H, g->dt, *pv, st));  // ... and send their outputs home behind them
  if (prof) { record_timing(pr.ev[5], st); g->prof_pending.push_back(pr); }
  g->st.forwards += 1;
  CHK(end_forward(g, st, !mp.fast));
  return pump_if_pending(g);
}

static int ep_combine_impl(moeinf_engine* g, const void* x_dev, const void* ret_dev, void* out_dev, int cap_rows, void* stream, const EpPeers* pv);
extern "C" int moeinf_ep_combine(moeinf_engine* g, const void* x_dev, const void* ret_dev, void* out_dev, int cap_rows, void* stream) {
  return ep_combine_impl(g, x_dev, ret_dev, out_dev, cap_rows, stream, nullptr);
}
static int ep_combine_impl(moeinf_engine* g, const void* x_dev, const void* ret_dev, void* out_dev, int cap_rows, void* stream, const EpPeers* pv) {
  if (!g || !x_dev || !ret_dev || !out_dev) return fail(MOEINF_ERR_INVALID, "NULL argument");
  if (!g->d_ep_pair_pos || cap_rows != g->ep_cap_rows) return fail(MOEINF_ERR_STATE, "ep_combine needs a preceding ep_pack with the same cap_rows (0 after ep_pack_compact)");
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  hipStream_t st = (hipStream_t)stream;
  if (g->has_shared && !g->last_hidden_shared) {
    // the shared expert (always resident, replicated on every rank) runs on this rank's own tokens; for decode-sized
    // forwards it already ran inside the router launches of moeinf_ep_route_pack, i.e. UNDER the exchange
    const int T = g->last_T;
    if (!g->shared_dev[g->last_layer]) return fail(MOEINF_ERR_STATE, "shared expert of layer %d not registered", g->last_layer);
    IndexArgs ia;
    memset(&ia, 0, sizeof ia);
    ia.T = T; ia.K = 1; ia.E = g->E;
    ia.counts = g->d_counts; ia.offsets = g->d_offsets; ia.active = g->d_active; ia.n_active = g->d_n_active;
    ia.slot_token = g->d_slot_token; ia.slot_pair = g->d_slot_pair;
    HIPCHK(launch_shared_only_index(ia, st));
    FfnStage s1, s2;
    fill_stage(g, g->last_layer, 1, s1);
    s1.in = x_dev;
    fill_stage(g, g->last_layer, 2, s2);
    s1.n_active_host = 1; s2.n_active_host = 1;
    HIPCHK(launch_ffn_stage(s1, 1, T, st));
    HIPCHK(launch_ffn_stage(s2, 1, T, st));
  }
  CombineArgs ca;
  memset(&ca, 0, sizeof ca);
  ca.x = x_dev; ca.y = ret_dev; ca.out = out_dev;
  ca.topk_idx = g->d_topk_idx; ca.topk_w = g->d_topk_w; ca.pair_slot = g->d_ep_pair_pos; ca.pair_order = g->d_pair_order;
  ca.router_prob = g->d_router_prob; ca.y_shared = g->has_shared ? (g->last_hidden_shared ? g->d_y_sh : g->d_y) : nullptr; ca.shared_offsets = nullptr; ca.shared_E = g->E;
  ca.T = g->last_T; ca.H = g->H; ca.K = g->K; ca.kind = g->cfg.router_kind; ca.dtype = g->dt;
  if (pv) {  // the owners' outputs of exchange `epoch` must have landed in this rank's return region
    EpWait w{g->ep_win.ret_flags(), pv->size, pv->epoch, pv->timeout_ticks, pv->err};
    if (pv->poll) { HIPCHK(launch_combine(ca, st, &w)); return MOEINF_OK; }
    HIPCHK(launch_ep_wait(w, st));
  }
  HIPCHK(launch_combine(ca, st));
  return MOEINF_OK;
}

// ---- native transport of the exchange (ep_comm.h) ---------------------------------------------------------------
extern "C" int moeinf_ep_comm_available(int32_t* available) {
  if (!available) return fail(MOEINF_ERR_INVALID, "available is NULL");
  std::string err;
  *available = RcclApi::get(&err) ? 1 : 0;
  if (!*available) g_err = err;
  return MOEINF_OK;
}

extern "C" int moeinf_ep_comm_unique_id(void* id_out, int nbytes) {
  if (!id_out || nbytes != (int)sizeof(RcclUniqueId)) return fail(MOEINF_ERR_INVALID, "id_out must hold %d bytes", (int)sizeof(RcclUniqueId));
  std::string err;
  const RcclApi* api = RcclApi::get(&err);
  if (!api) return fail(MOEINF_ERR_UNSUPPORTED, "%s", err.c_str());
  RcclUniqueId id;
  const int rc = api->GetUniqueId(&id);
  if (rc) return fail(MOEINF_ERR_HIP, "ncclGetUniqueId: %s", api->GetErrorString(rc));
  memcpy(id_out, &id, sizeof id);
  return MOEINF_OK;
}

static void ep_comm_free_buffers(moeinf_engine* g) {
  void** bufs[] = {&g->ep_x_send, &g->ep_x_ret};
  for (void** b : bufs) { if (*b) (void)hipFree(*b); *b = nullptr; }
  if (g->ep_x_recv && !g->ep_win.base) { (void)hipFree(g->ep_x_recv); g->ep_x_recv = nullptr; }  // (shared with the peer-store transport)
  if (g->ep_x_y && !g->ep_win.base) { (void)hipFree(g->ep_x_y); g->ep_x_y = nullptr; }  // (shared with the peer-store transport)
  g->ep_cap_tokens = 0; g->ep_x_cap_rows = 0;
}
// Everything of the RCCL bootstrap that can fail on ONE rank, with no collective inside (round-3 advice: a rank that failed
// here used to return while the others blocked in ncclCommInitRank): validation, library binding, exchange buffers.  The
// host layer agrees on the outcome of this step before any rank enters moeinf_ep_comm_init.
extern "C" int moeinf_ep_comm_prepare(moeinf_engine* g, int cap_tokens) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  if (cap_tokens <= 0 || cap_tokens > g->cfg.max_tokens) return fail(MOEINF_ERR_INVALID, "cap_tokens %d not in 1..max_tokens(%d)", cap_tokens, g->cfg.max_tokens);
  if (g->ep_comm) return fail(MOEINF_ERR_STATE, "the engine already has a communicator");
  std::string err;
  if (!RcclApi::get(&err)) return fail(MOEINF_ERR_UNSUPPORTED, "%s", err.c_str());
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  // exchange buffers: cap_rows row slots per peer, both directions (send/recv rows carry the 16-byte id tail)
  const int cap_rows = ep_min_cap(g, cap_tokens);
  const size_t n = (size_t)g->cfg.ep_size * cap_rows;
  if ((int64_t)n > (int64_t)g->cfg.max_tokens * g->K) return fail(MOEINF_ERR_INVALID, "the owner side needs room for ep_size*cap_rows = %zu rows: create the engine with max_tokens >= %zu", n, (n + g->K - 1) / g->K);
  if (g->ep_x_cap_rows == cap_rows && g->ep_x_send) return MOEINF_OK;  // prepared already
  if (g->ep_win.base && cap_rows != g->ep_win.cap_rows) return fail(MOEINF_ERR_STATE, "the peer-store window was built for another cap_tokens");
  ep_comm_free_buffers(g);
  hipError_t e = hipMalloc(&g->ep_x_send, n * ep_row_elems(g) * g->es);
  if (e == hipSuccess && !g->ep_x_recv) e = hipMalloc(&g->ep_x_recv, n * ep_row_elems(g) * g->es);
  if (e == hipSuccess && !g->ep_x_y) e = hipMalloc(&g->ep_x_y, n * (size_t)g->H * g->es);
  if (e == hipSuccess) e = hipMalloc(&g->ep_x_ret, n * (size_t)g->H * g->es);
  if (e == hipSuccess) e = hipMemset(g->ep_x_y, 0, n * (size_t)g->H * g->es);  // padding rows travel as they are: keep them defined
  if (e != hipSuccess) { ep_comm_free_buffers(g); (void)hipGetLastError(); return fail(MOEINF_ERR_HIP, "exchange buffers: %s", hipGetErrorString(e)); }
  g->ep_cap_tokens = cap_tokens;
  g->ep_x_cap_rows = cap_rows;
  return MOEINF_OK;
}

extern "C" int moeinf_ep_comm_init(moeinf_engine* g, const void* unique_id, int nbytes, int cap_tokens) {
  if (!g || !unique_id || nbytes != (int)sizeof(RcclUniqueId)) return fail(MOEINF_ERR_INVALID, "unique_id must be %d bytes", (int)sizeof(RcclUniqueId));
  CHK(moeinf_ep_comm_prepare(g, cap_tokens));  // (no-op after an explicit prepare with the same cap_tokens)
  const RcclApi* api = RcclApi::get(nullptr);
  RcclUniqueId id;
  memcpy(&id, unique_id, sizeof id);
  const int rc = api->CommInitRank(&g->ep_comm, g->cfg.ep_size, id, g->cfg.ep_rank);
  if (rc) {
    g->ep_comm = nullptr;
    ep_comm_free_buffers(g);
    return fail(MOEINF_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", g->cfg.ep_rank, g->cfg.ep_size, api->GetErrorString(rc));
  }
  g->ep_use_peer = false;
  return MOEINF_OK;
}

extern "C" int moeinf_ep_all_to_all(moeinf_engine* g, const void* send_dev, void* recv_dev, int64_t bytes_per_peer, void* stream) {
  if (!g || !send_dev || !recv_dev || bytes_per_peer <= 0) return fail(MOEINF_ERR_INVALID, "bad all_to_all arguments");
  if (!g->ep_comm) return fail(MOEINF_ERR_STATE, "no communicator: call moeinf_ep_comm_init first");
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  const std::string err = rccl_all_to_all(RcclApi::get(nullptr), g->ep_comm, g->cfg.ep_size, send_dev, recv_dev, (size_t)bytes_per_peer, (hipStream_t)stream);
  if (!err.empty()) return fail(MOEINF_ERR_HIP, "%s", err.c_str());
  return MOEINF_OK;
}

// ---- direct peer-store exchange (ep_peer.h) ------------------------------------------------------------------------
// Bootstrap, every step LOCAL (a rank that fails returns an error and leaves nobody blocked in a collective; the host layer
// agrees on the outcome between the steps): export -> [exchange the blobs] -> attach -> selftest.
extern "C" int moeinf_ep_peer_export(moeinf_engine* g, int cap_tokens, void* blob_out, int nbytes) {
  if (!g || !blob_out || nbytes != kEpPeerBlobBytes) return fail(MOEINF_ERR_INVALID, "blob_out must hold %d bytes", kEpPeerBlobBytes);
  if (cap_tokens <= 0 || cap_tokens > g->cfg.max_tokens) return fail(MOEINF_ERR_INVALID, "cap_tokens %d not in 1..max_tokens(%d)", cap_tokens, g->cfg.max_tokens);
  if (g->ep_win.base) {  // a second host-side exchange object over the same engine: hand out the same window again
    if (cap_tokens != g->ep_win_cap_tokens) return fail(MOEINF_ERR_STATE, "the engine already has an exchange window for cap_tokens %d", g->ep_win_cap_tokens);
    EpPeerBlob b;
    const std::string err = g->ep_win.export_blob(g->cfg.ep_rank, g->cfg.ep_size, g->cfg.device_id, &b);
    if (!err.empty()) return fail(MOEINF_ERR_HIP, "%s", err.c_str());
    memset(blob_out, 0, kEpPeerBlobBytes);
    memcpy(blob_out, &b, sizeof b);
    return MOEINF_OK;
  }
  if (g->cfg.ep_size > EP_MAX_PEERS) return fail(MOEINF_ERR_UNSUPPORTED, "peer-store exchange: ep_size %d > %d", g->cfg.ep_size, EP_MAX_PEERS);
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  const int cap_rows = ep_min_cap(g, cap_tokens);
  const size_t n = (size_t)g->cfg.ep_size * cap_rows;
  if ((int64_t)n > (int64_t)g->cfg.max_tokens * g->K) return fail(MOEINF_ERR_INVALID, "the owner side needs room for ep_size*cap_rows = %zu rows: create the engine with max_tokens >= %zu", n, (n + g->K - 1) / g->K);
  if (g->ep_x_send && g->ep_x_cap_rows != cap_rows)  // staging buffers of a communicator prepared for another capacity would be re-used below
    return fail(MOEINF_ERR_STATE, "the RCCL exchange buffers were built for another cap_tokens (cap_rows %d, wanted %d)", g->ep_x_cap_rows, cap_rows);
  if (!g->ep_err_host) {
    if (hipHostMalloc((void**)&g->ep_err_host, 64, hipHostMallocDefault) != hipSuccess) { g->ep_err_host = nullptr; (void)hipGetLastError(); return fail(MOEINF_ERR_HIP, "pinned error word"); }
    *g->ep_err_host = 0;
    if (const char* ev = getenv("MOEINF_EP_ERR_CHECK_EVERY")) g->ep_err_every = (uint32_t)std::max(1, atoi(ev));
  }
  std::string err = g->ep_win.create(g->cfg.ep_size, cap_rows, ep_row_elems(g) * g->es, (int64_t)g->H * g->es, g->E);
  if (err.empty() && !g->ep_x_recv) {  // routed-form staging of the broadcast form's slow path (launch_ep_bcast_unpack)
    if (hipMalloc(&g->ep_x_recv, n * ep_row_elems(g) * g->es) != hipSuccess) { g->ep_x_recv = nullptr; err = "hipMalloc of the unpack staging buffer failed"; }
  }
  if (err.empty() && !g->ep_x_y) {  // staging of the owner's outputs on the generic path (more rows than the self-indexing kernels take)
    if (hipMalloc(&g->ep_x_y, n * (size_t)g->H * g->es) != hipSuccess) { g->ep_x_y = nullptr; err = "hipMalloc of the output staging buffer failed"; }
  }
  EpPeerBlob b;
  if (err.empty()) err = g->ep_win.export_blob(g->cfg.ep_rank, g->cfg.ep_size, g->cfg.device_id, &b);
  if (!err.empty()) { g->ep_win.destroy(); (void)hipGetLastError(); return fail(MOEINF_ERR_HIP, "%s", err.c_str()); }
  memset(blob_out, 0, kEpPeerBlobBytes);
  memcpy(blob_out, &b, sizeof b);
  g->ep_win_cap_tokens = cap_tokens;
  const char* t = getenv("MOEINF_EP_PEER_TIMEOUT_MS");
  g->ep_peer_timeout_ticks = (int64_t)(t ? atoll(t) : 10000) * 100000;  // wall_clock64: 100 MHz
  return MOEINF_OK;
}

extern "C" int moeinf_ep_peer_attach(moeinf_engine* g, const void* blobs, int nbytes) {
  if (!g || !blobs) return fail(MOEINF_ERR_INVALID, "NULL argument");
  if (nbytes != g->cfg.ep_size * kEpPeerBlobBytes) return fail(MOEINF_ERR_INVALID, "blobs must be ep_size * %d bytes, in rank order", kEpPeerBlobBytes);
  if (!g->ep_win.base) return fail(MOEINF_ERR_STATE, "call moeinf_ep_peer_export first");
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  std::vector<EpPeerBlob> bs(g->cfg.ep_size);
  for (int p = 0; p < g->cfg.ep_size; ++p) memcpy(&bs[p], (const char*)blobs + (size_t)p * kEpPeerBlobBytes, sizeof(EpPeerBlob));
  if (g->ep_win.attached) {  // again (see moeinf_ep_peer_export): the same peers, or an error
    for (int p = 0; p < g->cfg.ep_size; ++p)
      if (bs[p].magic != kEpPeerMagic || bs[p].rank != p || bs[p].pid != g->ep_peer_pids[p] || bs[p].ptr != g->ep_peer_ptrs[p])
        return fail(MOEINF_ERR_STATE, "the engine is attached to other peers already");
    g->ep_use_peer = true;
    return MOEINF_OK;
  }
  const std::string err = g->ep_win.attach(bs.data(), g->cfg.ep_rank, g->cfg.ep_size, g->cfg.device_id);
  if (!err.empty()) { (void)hipGetLastError(); return fail(MOEINF_ERR_HIP, "%s", err.c_str()); }
  // ranks that share a GPU (tests on a one-GPU box) must not spin inside wide kernels — the rank they wait for needs CUs
  // to run on; MOEINF_EP_PEER_POLL=0/1 overrides.  Both, and MOEINF_EP_BCAST, are GROUP decisions: every rank derives them
  // from all ranks' blobs (ep_peer.h: attach), so that no two ranks can end up in different exchange forms.
  g->ep_peer_poll = g->ep_win.poll_agreed;
  g->ep_bcast_ok = g->ep_win.bcast_agreed;
  g->ep_peer_pids.clear(); g->ep_peer_ptrs.clear();
  for (auto& b : bs) { g->ep_peer_pids.push_back(b.pid); g->ep_peer_ptrs.push_back(b.ptr); }
  g->ep_use_peer = true;
  return MOEINF_OK;
}

extern "C" int moeinf_ep_peer_set_timeout_ms(moeinf_engine* g, int ms) {
  if (!g || ms <= 0) return fail(MOEINF_ERR_INVALID, "engine is NULL or ms <= 0");
  g->ep_peer_timeout_ticks = (int64_t)ms * 100000;  // wall_clock64: 100 MHz
  return MOEINF_OK;
}

extern "C" int moeinf_ep_peer_get_timeout_ms(moeinf_engine* g, int* ms) {
  if (!g || !ms) return fail(MOEINF_ERR_INVALID, "engine or ms is NULL");
  *ms = (int)(g->ep_peer_timeout_ticks / 100000);
  return MOEINF_OK;
}

extern "C" int moeinf_ep_peer_release(moeinf_engine* g) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  HIPCHK(hipDeviceSynchronize());  // no kernel of this rank still reads or writes a window
  g->ep_win.destroy();
  g->ep_win_cap_tokens = 0;
  g->ep_use_peer = false;
  g->ep_peer_pids.clear(); g->ep_peer_ptrs.clear();
  if (!g->ep_x_send) {  // the staging buffers belong to this transport alone (no communicator prepared)
    for (void** b : {&g->ep_x_recv, &g->ep_x_y}) if (*b) { (void)hipFree(*b); *b = nullptr; }
  }
  return MOEINF_OK;
}

// which bootstrapped transport moeinf_ep_moe_forward takes (the last one set up is the default)
extern "C" int moeinf_ep_select_transport(moeinf_engine* g, int kind) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  if (kind == MOEINF_EP_TRANSPORT_PEER_STORE) { if (!g->ep_win.attached) return fail(MOEINF_ERR_STATE, "peer-store exchange is not set up"); g->ep_use_peer = true; return MOEINF_OK; }
  if (kind == MOEINF_EP_TRANSPORT_RCCL) { if (!g->ep_comm) return fail(MOEINF_ERR_STATE, "no RCCL communicator"); g->ep_use_peer = false; return MOEINF_OK; }
  return fail(MOEINF_ERR_INVALID, "kind must be MOEINF_EP_TRANSPORT_PEER_STORE or MOEINF_EP_TRANSPORT_RCCL");
}

static void ep_peer_view(moeinf_engine* g, EpPeers* pv) {
  g->ep_win.view(pv, g->cfg.ep_rank, g->cfg.ep_size, g->d_miss, g->ep_peer_timeout_ticks, g->ep_peer_poll);
}

// Collective in effect (every rank must call it the same number of times), but bounded: a rank whose peers never
// show up gets ok = 0 after the poll timeout instead of a hang.
extern "C" int moeinf_ep_peer_selftest(moeinf_engine* g, void* stream, int32_t* ok) {
  if (!g || !ok) return fail(MOEINF_ERR_INVALID, "NULL argument");
  *ok = 0;
  if (!g->ep_win.attached) return fail(MOEINF_ERR_STATE, "call moeinf_ep_peer_attach first");
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  hipStream_t st = (hipStream_t)stream;
  g->ep_win.epoch += 1;
  EpPeers pv;
  ep_peer_view(g, &pv);
  const int words = (int)std::min<int64_t>(1024, std::min(pv.recv_row_bytes, pv.ret_row_bytes) * pv.cap_rows / 4);
  int32_t* ok_dev = g->ep_win.done + 8;  // a spare word of the counter allocation
  HIPCHK(hipMemsetAsync(ok_dev, 0, 4, st));
  // fault injection for the liveness tests (tests/test_gpu_bench_ranks.py): this rank never publishes — what a rank behind a
  // dead link looks like to its peers; they must come out of their self-test with ok = 0 after the bounded wait
  const char* silent = getenv("MOEINF_EP_TEST_SILENT_RANK");
  if (!silent || atoi(silent) != g->cfg.ep_rank) HIPCHK(launch_ep_selftest_send(pv, words, st));
  HIPCHK(launch_ep_selftest_check(pv, words, ok_dev, st));
  HIPCHK(hipMemcpyAsync(ok, ok_dev, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  int32_t f = 0;
  HIPCHK(hipMemcpy(&f, g->d_miss, 4, hipMemcpyDeviceToHost));
  if (f) { HIPCHK(hipMemset(g->d_miss, 0, 4)); *ok = 0; }
  return MOEINF_OK;
}

extern "C" int moeinf_ep_transport(const moeinf_engine* g, int32_t out[4]) {
  if (!g || !out) return fail(MOEINF_ERR_INVALID, "NULL argument");
  out[0] = (g->ep_win.attached && g->ep_use_peer) ? MOEINF_EP_TRANSPORT_PEER_STORE : (g->ep_comm ? MOEINF_EP_TRANSPORT_RCCL : MOEINF_EP_TRANSPORT_NONE);
  out[1] = g->ep_win.shared_device ? 1 : 0;
  out[2] = g->ep_peer_poll ? 1 : 0;
  out[3] = (int32_t)g->ep_win.epoch;
  return MOEINF_OK;
}

// The caller's promise that EVERY rank passes the same token count to every moeinf_ep_moe_forward (decode loops do): with it,
// a one-token forward over the peer-store transport takes the BROADCAST form (kernels.h: EpBcastArgs) — all ranks must then be
// in that form together, which is why it cannot be inferred from this rank's own token count.
extern "C" int moeinf_ep_set_uniform_tokens(moeinf_engine* g, int on) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  g->ep_uniform = on != 0;
  return MOEINF_OK;
}

static bool ep_bcast_eligible(const moeinf_engine* g, int tokens) {
  const bool env = g->ep_bcast_ok;  // (MOEINF_EP_BCAST of EVERY rank, see moeinf_ep_peer_attach)
  const int et = g->cfg.expert_type;
  // consumer kernels must poll for themselves (the broadcast rides in FFN stage 1: no room for a wait kernel in front of it)
  return env && g->ep_uniform && tokens == 1 && g->ep_peer_poll && g->K <= 8 && g->E <= 64 && g->dt != DT_F32 &&
         (g->cfg.router_kind == MOEINF_ROUTER_MIXTRAL || (g->cfg.router_kind == MOEINF_ROUTER_DEEPSEEK && g->cfg.n_group <= 1 && !g->route_v3)) &&
         (et == MOEINF_EXPERT_MIXTRAL || et == MOEINF_EXPERT_DEEPSEEK) && (!g->has_shared || can_hide_shared(g, 1));
}

// Batch-1 decode over the peer-store exchange, broadcast form: gate -> FFN stage 1 (block 0 broadcasts this rank's row +
// logits and routes the home token; the other workgroups wait for every rank's broadcast, route all ep_size tokens and stream
// the experts this rank owns) -> stage 2 (outputs stored into the home ranks' windows) -> combine.  FOUR launches.  If an
// owned expert is not resident the host must see the routing: the broadcast becomes a launch of its own, an unpack kernel
// turns the received (row, logits) pairs into the routed form locally and the generic owner path takes over.
static int ep_peer_forward_bcast(moeinf_engine* g, int layer, const void* x_dev, const void* gate_w_dev, void* out_dev, void* stream) {
  if (!x_dev || !gate_w_dev) return fail(MOEINF_ERR_INVALID, "NULL argument");
  if (layer < 0 || layer >= g->L) return fail(MOEINF_ERR_INVALID, "layer %d out of range", layer);
  if (g->has_shared && !g->shared_dev[layer]) return fail(MOEINF_ERR_STATE, "shared expert of layer %d not registered", layer);
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  hipStream_t st = (hipStream_t)stream;
  const int cap = g->ep_win.cap_rows, G = g->cfg.ep_size;
  const int64_t ld = ep_row_elems(g);
  moeinf_engine::EpProfRec pr;
  const bool prof = g->ep_profiling;
  auto mark = [&](int i) { if (prof) record_timing(pr.ev[i], st); };
  if (prof) for (int i = 0; i < 6; ++i) { pr.ev[i] = get_event(g); if (!pr.ev[i]) return fail(MOEINF_ERR_HIP, "hipEventCreate failed"); }
  CHK(ep_alloc(g, cap));
  EpPeers pv;  // (the exchange number was taken by ep_peer_forward)
  ep_peer_view(g, &pv);
  RouteArgs ra;
  make_route_args(g, x_dev, gate_w_dev, 1, ra);
  if (g->route_v3) ra.e_bias = g->gate_bias[layer];
  drop_stale_prefetches(g, layer);
  MirrorPlan mp;
  CHK(plan_mirror(g, layer, mp));
  EpBcastArgs b;
  memset(&b, 0, sizeof b);
  b.x = x_dev; b.pair_pos = g->d_ep_pair_pos; b.peers = pv;
  g->last_T = 1; g->last_layer = layer; g->last_stream = st; g->last_selfroute = false;
  mark(0);
  if (mp.fast) {
    const bool hide = g->has_shared;  // (eligibility says it can be hidden)
    FfnStage sh1, sh2;
    if (hide) { hidden_shared_stages(g, layer, x_dev, sh1, sh2); HIPCHK(launch_gate_shared1(ra, sh1, st)); }
    else HIPCHK(launch_gate_logits(ra, st));
    g->last_hidden_shared = hide;
    moeinf_engine::PendingMirror pm;
    pm.buf = mp.target; pm.seq = g->seq + 1; pm.layer = layer; pm.T = G; pm.prof = false; pm.local = false;
    g->pend.push_back(pm);
    for (int e = 0; e < g->E; ++e) {  // any of the layer's slots may be read by this forward
      const Node& n = g->nodes[node_index(g, layer, e)];
      if (n.slot >= 0) g->slots[n.slot].last_use_seq = g->seq + 1;
    }
    CHK(flush_pokes(g, st));
    FfnStage s1, s2;
    fill_stage(g, layer, 1, s1, ld);
    s1.in = g->ep_win.recv_region(); s1.row_map = nullptr;
    fill_stage(g, layer, 2, s2);
    s2.out = g->ep_x_y; s2.out_map = nullptr;
    b.mirror = mp.target;
    const int per_rank = (g->E + G - 1) / G;
    const int max_active = std::max(1, std::min(std::max(1, g->owned_experts), G * std::min(g->K, per_rank)));
    mark(1); mark(2);
    HIPCHK(launch_ffn_epb_stage1(ra, s1, hide ? &sh2 : nullptr, b, g->d_ep_rec, max_active, 1, st));
    EpOwnArgs o;
    memset(&o, 0, sizeof o);
    o.recv = g->ep_win.recv_region(); o.ld_recv = ld; o.H = g->H; o.nrows = std::min(64, G * cap); o.ep_size = G; o.ep_rank = g->cfg.ep_rank;
    o.stage = 2; o.max_active = max_active; o.rec = g->d_ep_rec; o.peers = pv; o.tile_done = g->d_arrive;
    HIPCHK(launch_ffn_ep_stage(s2, o, st));
    g->st.forwards += 2;  // (home routing + owner FFN, as the routed form counts them)
    CHK(end_forward(g, st, false));  // (the sync-free branch)
    mark(3);
  } else {
    // plan_mirror handed out the engine's own mirror (nothing pooled to give back); the generic owner path plans again
    HIPCHK(launch_gate_logits(ra, st));
    g->last_hidden_shared = false;  // the shared expert runs on the home rank inside the combine step
    HIPCHK(launch_ep_bcast(ra, b, st));
    g->st.forwards += 1;
    mark(1); mark(2);
    HIPCHK(launch_ep_bcast_unpack(ra, b, g->ep_x_recv, ld, g->dt, st));
    EpPeers pvw = pv;  // (the unpack kernel has waited already; the generic path's wait kernel returns at once)
    CHK(ep_expert_ffn_rows(g, layer, g->ep_x_recv, g->ep_x_y, G * cap, st, &pvw));
    mark(3);
  }
  mark(4);
  CHK(ep_combine_impl(g, x_dev, g->ep_win.ret_region(), out_dev, cap, stream, &pv));
  mark(5);
  if (prof) g->ep_prof_pending.push_back(pr);
  return mp.fast ? pump_if_pending(g) : MOEINF_OK;
}

// One expert-parallel MoE layer over the peer-store exchange: router (+ pack into the destinations' windows) -> owner FFN
// (polls the row flags; stage 2 stores its outputs into the home ranks' windows) -> combine (polls the output flags).
// Five launches on `stream`, no collective, no copy.
static int ep_peer_forward_body(moeinf_engine* g, int layer, const void* x_dev, int tokens, int batch_rows, const void* gate_w_dev, void* out_dev, void* stream);

// Exchange numbers are failure-atomic: EVERY entry takes the next number before anything can fail, so a rank whose call
// returns an error (bad arguments, an allocation) is not one exchange behind its peers for good — its peers' kernels give
// up on the exchange it never published (flag 2 after MOEINF_EP_PEER_TIMEOUT_MS), and a rank that IS out of step is
// caught by the consumers themselves (a flag AHEAD of the exchange they wait for: flag 3, kdev.h ep_poll).  The device flag
// is copied to a pinned word by the stream every ep_err_every exchanges and looked at on entry: a caller that never
// calls moeinf_sync() still gets the error from one of its next forwards instead of silent garbage.
static int ep_peer_forward(moeinf_engine* g, int layer, const void* x_dev, int tokens, int batch_rows, const void* gate_w_dev, void* out_dev, void* stream) {
  g->ep_win.epoch += 1;  // collective discipline: every rank runs the same sequence of exchanges
  if (g->ep_err_host) {
    const int32_t f = *(volatile int32_t*)g->ep_err_host;
    if (f == 2 || f == 3) {
      *g->ep_err_host = 0;
      return fail(MOEINF_ERR_STATE, f == 2 ? "peer-store exchange: a kernel gave up waiting for another rank's rows (MOEINF_EP_PEER_TIMEOUT_MS); the results of the last forwards are invalid"
                                           : "peer-store exchange: another rank is AHEAD of this one (an earlier call failed here or there); the results of the last forwards are invalid");
    }
  }
  const int rc = ep_peer_forward_body(g, layer, x_dev, tokens, batch_rows, gate_w_dev, out_dev, stream);
  if (rc == MOEINF_OK && g->ep_err_host && g->ep_win.epoch % g->ep_err_every == 0)
    HIPCHK(hipMemcpyAsync(g->ep_err_host, g->d_miss, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return rc;
}

static int ep_peer_forward_body(moeinf_engine* g, int layer, const void* x_dev, int tokens, int batch_rows, const void* gate_w_dev, void* out_dev, void* stream) {
  if (tokens > g->ep_win_cap_tokens) return fail(MOEINF_ERR_INVALID, "tokens %d > cap_tokens %d of the exchange window", tokens, g->ep_win_cap_tokens);
  if (ep_bcast_eligible(g, tokens)) return ep_peer_forward_bcast(g, layer, x_dev, gate_w_dev, out_dev, stream);
  hipStream_t st = (hipStream_t)stream;
  const int cap = g->ep_win.cap_rows, G = g->cfg.ep_size;
  moeinf_engine::EpProfRec pr;
  const bool prof = g->ep_profiling;
  auto mark = [&](int i) { if (prof) record_timing(pr.ev[i], st); };
  if (prof) for (int i = 0; i < 6; ++i) { pr.ev[i] = get_event(g); if (!pr.ev[i]) return fail(MOEINF_ERR_HIP, "hipEventCreate failed"); }
  EpPeers pv;
  ep_peer_view(g, &pv);
  mark(0);
  CHK(ep_route_pack_impl(g, layer, x_dev, tokens, batch_rows, gate_w_dev, nullptr, nullptr, cap, stream, &pv));
  mark(1);
  mark(2);  // (no dispatch collective: the rows are already on their way)
  CHK(ep_expert_ffn_rows(g, layer, g->ep_win.recv_region(), g->ep_x_y, G * cap, st, &pv));
  mark(3);
  mark(4);
  CHK(ep_combine_impl(g, x_dev, g->ep_win.ret_region(), out_dev, cap, stream, &pv));
  mark(5);
  if (prof) g->ep_prof_pending.push_back(pr);
  return MOEINF_OK;
}

// One expert-parallel MoE layer in ONE host call (fixed-capacity form): router + send rows -> all-to-all -> owner FFN ->
// all-to-all -> combine, every launch and both collectives enqueued on `stream` from here.
extern "C" int moeinf_ep_moe_forward(moeinf_engine* g, int layer, const void* x_dev, int tokens, int batch_rows, const void* gate_w_dev,
                                     void* out_dev, void* stream) {
  if (!g || !out_dev) return fail(MOEINF_ERR_INVALID, "NULL argument");
  if (g->ep_win.attached && (g->ep_use_peer || !g->ep_comm)) return ep_peer_forward(g, layer, x_dev, tokens, batch_rows, gate_w_dev, out_dev, stream);
  if (!g->ep_comm) return fail(MOEINF_ERR_STATE, "no transport: call moeinf_ep_peer_export + moeinf_ep_peer_attach, or moeinf_ep_comm_init, first");
  if (tokens > g->ep_cap_tokens) return fail(MOEINF_ERR_INVALID, "tokens %d > cap_tokens %d of the communicator's exchange buffers", tokens, g->ep_cap_tokens);
  hipStream_t st = (hipStream_t)stream;
  const RcclApi* api = RcclApi::get(nullptr);
  const int cap = g->ep_x_cap_rows, G = g->cfg.ep_size;
  moeinf_engine::EpProfRec pr;
  const bool prof = g->ep_profiling;
  auto mark = [&](int i) { if (prof) record_timing(pr.ev[i], st); };
  if (prof) for (int i = 0; i < 6; ++i) { pr.ev[i] = get_event(g); if (!pr.ev[i]) return fail(MOEINF_ERR_HIP, "hipEventCreate failed"); }
  mark(0);
  CHK(moeinf_ep_route_pack(g, layer, x_dev, tokens, batch_rows, gate_w_dev, g->ep_x_send, nullptr, cap, stream));
  mark(1);
  std::string err = rccl_all_to_all(api, g->ep_comm, G, g->ep_x_send, g->ep_x_recv, (size_t)cap * ep_row_elems(g) * g->es, st);
  if (!err.empty()) return fail(MOEINF_ERR_HIP, "dispatch all-to-all: %s", err.c_str());
  mark(2);
  CHK(moeinf_ep_expert_ffn(g, layer, g->ep_x_recv, g->ep_x_y, cap, stream));
  mark(3);
  err = rccl_all_to_all(api, g->ep_comm, G, g->ep_x_y, g->ep_x_ret, (size_t)cap * g->H * g->es, st);
  if (!err.empty()) return fail(MOEINF_ERR_HIP, "combine all-to-all: %s", err.c_str());
  mark(4);
  CHK(moeinf_ep_combine(g, x_dev, g->ep_x_ret, out_dev, cap, stream));
  mark(5);
  if (prof) g->ep_prof_pending.push_back(pr);
  return MOEINF_OK;
}

extern "C" int moeinf_ep_get_profile(moeinf_engine* g, moeinf_ep_profile* out) {
  if (!g || !out) return fail(MOEINF_ERR_INVALID, "NULL argument");
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  if (g->last_layer >= 0) HIPCHK(hipStreamSynchronize(g->last_stream));
  for (auto& r : g->ep_prof_pending) {
    float ms = 0.f;
    double* dst[5] = {&g->ep_prof.route_pack_ms, &g->ep_prof.a2a_dispatch_ms, &g->ep_prof.owner_ffn_ms, &g->ep_prof.a2a_combine_ms, &g->ep_prof.combine_ms};
    for (int i = 0; i < 5; ++i) if (hipEventElapsedTime(&ms, r.ev[i], r.ev[i + 1]) == hipSuccess) *dst[i] += ms;
    for (int i = 0; i < 6; ++i) g->event_pool.push_back(r.ev[i]);
    g->ep_prof.calls += 1;
  }
  g->ep_prof_pending.clear();
  *out = g->ep_prof;
  memset(&g->ep_prof, 0, sizeof g->ep_prof);
  return MOEINF_OK;
}
