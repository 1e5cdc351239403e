// engine.cpp — host side of libmoeinf_hip.so: lifecycle, registration, the memory tiers (pinned arena as a cache over the
// offload directory, HBM slots, copy lanes), the local per-layer hot path and its cache control (include/moeinf.h).
// Shared records and the engine object: engine_internal.h; expert parallelism: engine_ep.cpp; host-only handles: capi_host.cpp.
//
// Replaces (reference, /root/reference): core/parallel/expert_dispatcher.cpp (ExpertDispatcher),
// core/model/model_topology.cpp Node::SetDevice (tier mover), core/memory/* (pools, allocators,
// streams), core/prefetch/task_scheduler.cpp (prefetch queue + eviction) — with a different
// architecture: no worker threads, no per-expert stream syncs.  Every H2D copy is a
// hipMemcpyAsync from the pinned arena into a fixed-size HBM slot on a dedicated copy stream and
// is ordered against the compute stream with events (hipStreamWaitEvent), so the host never blocks
// on a copy and a slot is never recycled while a kernel may still read it.
#include "engine_internal.h"

#include <math.h>

// ---- host arena ----------------------------------------------------------------------------
static int arena_alloc(moeinf_engine* g, int64_t bytes, void** out) {
  bytes = align_up(bytes, kAioAlignment);
  if (g->cfg.host_memory_bytes > 0 && g->arena_total + bytes > g->cfg.host_memory_bytes)
    return fail(MOEINF_ERR_OOM, "pinned host arena cap (%lld bytes) exceeded", (long long)g->cfg.host_memory_bytes);
  if (g->arena_chunks.empty() || g->arena_used_in_chunk + bytes > g->arena_chunk_bytes) {
    // chunk = at least 64 experts' worth or 1 GiB, so pinning cost is amortised
    int64_t want = std::max<int64_t>(bytes, std::min<int64_t>(std::max<int64_t>(bytes * 64, 1ll << 30), 8ll << 30));
    want = align_up(want, bytes);  // whole blobs per chunk
    // a capped arena never pins more than the cap: the chunk is clamped to the whole blobs that still fit under it
    if (g->cfg.host_memory_bytes > 0) want = std::max<int64_t>(bytes, std::min<int64_t>(want, (g->cfg.host_memory_bytes - g->arena_total) / bytes * bytes));
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, (size_t)want, hipHostMallocDefault);
    if (e != hipSuccess) {
      want = bytes;
      e = hipHostMalloc(&p, (size_t)want, hipHostMallocDefault);
      if (e != hipSuccess) return fail(MOEINF_ERR_OOM, "hipHostMalloc(%lld) failed: %s", (long long)want, hipGetErrorString(e));
    }
    g->arena_chunks.push_back(p);
    g->arena_chunk_bytes = want;
    g->arena_used_in_chunk = 0;
  }
  *out = (char*)g->arena_chunks.back() + g->arena_used_in_chunk;
  g->arena_used_in_chunk += bytes;
  g->arena_total += bytes;
  g->st.host_arena_bytes = g->arena_total;
  return MOEINF_OK;
}

hipEvent_t get_event(moeinf_engine* g) {
  if (!g->event_pool.empty()) {
    hipEvent_t e = g->event_pool.back();
    g->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

// ---- lifecycle -----------------------------------------------------------------------------
#undef g_err
std::string& moeinf_err_slot() {
  static thread_local std::string err;
  return err;
}
#define g_err (moeinf_err_slot())
extern "C" const char* moeinf_last_error(void) { return moeinf_err_slot().c_str(); }
extern "C" int moeinf_abi_version(void) { return MOEINF_ABI_VERSION; }
// rows of the busiest expert as the sync-free path assumes them (the host does not know the routing there): 1.5 x the mean + 1,
// at most T.  Picks the FORM of the FFN kernels only; an expert with more rows takes more passes.
static inline int rows_estimate(int T, int K, int E) { return (int)std::min<int64_t>(T, ((int64_t)T * K * 3) / (2 * std::max(1, E)) + 1); }
extern "C" int moeinf_rows_estimate(int tokens, int top_k, int num_experts) { return rows_estimate(tokens, top_k, num_experts); }
extern "C" int moeinf_fence_ring(void) { return kFenceRing; }
extern "C" int moeinf_fence_cover_pos(const uint64_t* fence_seq, uint64_t recorded, uint64_t forward) {
  return fence_seq ? fence_cover_pos(fence_seq, recorded, forward) : -1;
}

extern "C" int moeinf_ffn_ring2_form(int dtype, int nmat, int K, int K_sh, int R, int active, int max_rows, int num_cus, int32_t* out5) {
  if (!out5 || (nmat != 1 && nmat != 2) || K <= 0 || R <= 0 || active <= 0) return fail(MOEINF_ERR_INVALID, "moeinf_ffn_ring2_form: bad arguments");
  const bool two_bytes = dtype == MOEINF_DTYPE_BF16 || dtype == MOEINF_DTYPE_F16;
  const moeinf::Ring2Form f = moeinf::ring2_form(two_bytes ? 2 : 4, dtype == MOEINF_DTYPE_F16, nmat, K, K_sh, (R + 15) / 16, active, max_rows, num_cus,
                                                 moeinf::Ring2Knobs::from_env());
  out5[0] = f.ntb; out5[1] = f.tail; out5[2] = f.nblk; out5[3] = f.split; out5[4] = f.blocks;
  return MOEINF_OK;
}

// OCP e4m3fn (torch.float8_e4m3fn: 1-4-3, bias 7, no infinities, S.1111.111 = NaN) -> bf16 bits; exact (3 mantissa bits)
static inline uint16_t f8e4m3_to_bf16_bits(uint8_t v) {
  const uint32_t s = v >> 7, ex = (v >> 3) & 15, m = v & 7;
  float f;
  if (ex == 15 && m == 7) f = NAN;
  else if (ex == 0) f = ldexpf((float)m, -9);
  else f = ldexpf(1.0f + (float)m * 0.125f, (int)ex - 7);
  if (s) f = -f;
  uint32_t bits;
  memcpy(&bits, &f, 4);
  return (uint16_t)(bits >> 16);
}

static int validate(const moeinf_config* c) {
  if (!c) return fail(MOEINF_ERR_INVALID, "cfg is NULL");
  if (c->abi_version != MOEINF_ABI_VERSION) return fail(MOEINF_ERR_INVALID, "abi_version %d != %d", c->abi_version, MOEINF_ABI_VERSION);
  if (c->num_layers <= 0 || c->num_experts <= 0 || c->num_experts > 256) return fail(MOEINF_ERR_INVALID, "num_layers/num_experts out of range (experts <= 256)");
  if (c->dtype != MOEINF_DTYPE_BF16 && c->dtype != MOEINF_DTYPE_F32 && c->dtype != MOEINF_DTYPE_F16) return fail(MOEINF_ERR_UNSUPPORTED, "dtype %d: bf16 (0), fp32 (1) and fp16 (2) are built (fp8, id 3, is not)", c->dtype);
  if (c->gate_dtype != MOEINF_DTYPE_BF16 && c->gate_dtype != MOEINF_DTYPE_F32 && c->gate_dtype != MOEINF_DTYPE_F16) return fail(MOEINF_ERR_UNSUPPORTED, "gate_dtype %d", c->gate_dtype);
  // the gate is either in the model dtype or fp32 (DeepSeek); bf16 <-> fp16 mixes are not built
  if (c->gate_dtype != MOEINF_DTYPE_F32 && c->dtype != MOEINF_DTYPE_F32 && c->gate_dtype != c->dtype) return fail(MOEINF_ERR_UNSUPPORTED, "gate_dtype %d with dtype %d", c->gate_dtype, c->dtype);
  if (c->gate_dtype == MOEINF_DTYPE_F16 && c->dtype == MOEINF_DTYPE_F32) return fail(MOEINF_ERR_UNSUPPORTED, "an fp16 gate with fp32 activations is not built");
  switch (c->expert_type) {
    case MOEINF_EXPERT_SWITCH: case MOEINF_EXPERT_SWITCH_GATED: case MOEINF_EXPERT_NLLB: case MOEINF_EXPERT_FSGPT: case MOEINF_EXPERT_MIXTRAL: case MOEINF_EXPERT_DEEPSEEK: break;
    default: return fail(MOEINF_ERR_UNSUPPORTED, "expert_type %d is not one of the reference's (expert_module.h:13-18)", c->expert_type);
  }
  const int ev = c->dtype == MOEINF_DTYPE_F32 ? 4 : 8;
  if (c->hidden <= 0 || c->inter <= 0 || c->hidden % ev || c->inter % ev) return fail(MOEINF_ERR_INVALID, "hidden/inter must be positive multiples of %d", ev);
  if (c->shared_inter < 0 || c->shared_inter % ev) return fail(MOEINF_ERR_INVALID, "shared_inter must be a multiple of %d", ev);
  if (c->top_k <= 0 || c->top_k > 8 || c->top_k > c->num_experts) return fail(MOEINF_ERR_INVALID, "top_k must be in 1..min(8,E)");
  if (c->router_kind < 0 || c->router_kind > MOEINF_ROUTER_DEEPSEEK_V3) return fail(MOEINF_ERR_INVALID, "router_kind");
  if (c->router_kind == MOEINF_ROUTER_DEEPSEEK_V3 && (c->n_group <= 0 || c->num_experts % c->n_group || c->n_group > 64 || c->topk_group <= 0 || c->topk_group > c->n_group || c->num_experts / c->n_group < 2))
    return fail(MOEINF_ERR_INVALID, "DeepSeek-V3 gate: n_group must divide num_experts into groups of at least two, 1 <= topk_group <= n_group <= 64");
  if (c->router_kind == MOEINF_ROUTER_SWITCH && c->top_k != 1) return fail(MOEINF_ERR_INVALID, "switch router is top-1");
  if (c->router_kind == MOEINF_ROUTER_NLLB && c->top_k != 2) return fail(MOEINF_ERR_INVALID, "nllb router is top-2");
  if (c->router_kind == MOEINF_ROUTER_DEEPSEEK && c->n_group > 1) {
    if (c->num_experts % c->n_group || c->n_group > 64 || c->topk_group <= 0 || c->topk_group > c->n_group)
      return fail(MOEINF_ERR_INVALID, "group_limited_greedy: bad n_group/topk_group");
  }
  if (c->shared_inter > 0 && c->expert_type != MOEINF_EXPERT_DEEPSEEK) return fail(MOEINF_ERR_INVALID, "shared expert only with deepseek experts");
  if (c->ep_size < 1 || c->ep_rank < 0 || c->ep_rank >= c->ep_size) return fail(MOEINF_ERR_INVALID, "ep_rank/ep_size");
  if (c->max_tokens <= 0) return fail(MOEINF_ERR_INVALID, "max_tokens must be > 0");
  if (c->policy != MOEINF_POLICY_LFU_INCACHE && c->policy != MOEINF_POLICY_LRU) return fail(MOEINF_ERR_INVALID, "policy");
  if (c->device_memory_bytes <= 0 && !(c->device_memory_ratio > 0.0 && c->device_memory_ratio <= 1.0)) return fail(MOEINF_ERR_INVALID, "device_memory_ratio must be in (0,1] when no byte budget is given");
  return MOEINF_OK;
}


// the part of the device workspace that is sized by max_tokens (re-allocated by moeinf_reserve_tokens)
static void free_token_workspace(moeinf_engine* g) {
  void** bufs[] = {(void**)&g->d_logits, (void**)&g->d_topk_idx, (void**)&g->d_pair_valid, (void**)&g->d_pair_order, (void**)&g->d_pair_slot,
                   (void**)&g->d_topk_w, (void**)&g->d_router_prob, (void**)&g->d_slot_token, (void**)&g->d_slot_pair, (void**)&g->d_chunk, &g->d_h, &g->d_y};
  for (void** b : bufs) { if (*b) hipFree(*b); *b = nullptr; }
}
void free_ep_workspace(moeinf_engine* g) {
  void** bufs[] = {(void**)&g->d_ep_key, (void**)&g->d_ep_counts, (void**)&g->d_ep_offsets, (void**)&g->d_ep_active, (void**)&g->d_ep_nactive,
                   (void**)&g->d_ep_pair_slot, (void**)&g->d_ep_slot_token, (void**)&g->d_ep_slot_pair, (void**)&g->d_ep_pair_pos};
  for (void** b : bufs) { if (*b) hipFree(*b); *b = nullptr; }
  g->ep_alloc_cap = 0; g->ep_cap_rows = 0; g->ep_alloc_np = 0;
}
static int alloc_token_workspace(moeinf_engine* g, int max_tokens) {
  const size_t T = (size_t)max_tokens, K = (size_t)g->K;
  const size_t rows = T * K + (g->has_shared ? T : 0);
  CHK(dmalloc(&g->d_logits, T * g->E));
  CHK(dmalloc(&g->d_topk_idx, T * K)); CHK(dmalloc(&g->d_pair_valid, T * K)); CHK(dmalloc(&g->d_pair_order, T * K));
  CHK(dmalloc(&g->d_pair_slot, T * K)); CHK(dmalloc(&g->d_topk_w, T * K)); CHK(dmalloc(&g->d_router_prob, T));
  CHK(dmalloc(&g->d_slot_token, rows)); CHK(dmalloc(&g->d_slot_pair, rows));
  CHK(dmalloc(&g->d_chunk, ((T * K + 1023) / 1024 + 1) * (size_t)std::max(g->E, g->cfg.ep_size)));
  HIPCHK(hipMalloc(&g->d_h, rows * (size_t)g->ldh * g->es));
  HIPCHK(hipMalloc(&g->d_y, rows * (size_t)g->H * g->es));
  return MOEINF_OK;
}

static int sync_last(moeinf_engine* g);
static int check_device_flag(moeinf_engine* g);
// wait for the stream of the last forward and report the kernels' error flag (no resident blob for an active expert; a
// peer-store exchange that gave up waiting for another rank)
extern "C" int moeinf_sync(moeinf_engine* g) { return sync_last(g); }

extern "C" int moeinf_destroy(moeinf_engine* g) {
  if (!g) return MOEINF_OK;
  DeviceScope on_dev_(g->cfg.device_id);
  hipDeviceSynchronize();
  if (g->d_layer_trace) {  // debugging aid: "block t0 t1 t2 t3" (100 MHz ticks) of the last one-launch layer
    std::vector<unsigned long long> tr((size_t)g->layer1_trace_blocks * 4);
    if (hipMemcpy(tr.data(), g->d_layer_trace, tr.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
      if (FILE* f = fopen(getenv("MOEINF_LAYER1_TRACE") ? getenv("MOEINF_LAYER1_TRACE") : "/dev/null", "w")) {
        std::vector<int32_t> tb((size_t)g->layer1_trace_blocks, 0);
        {  // role = workgroup id (gate | shared stage 1 | meta | stage 1 | shared stage 2 (front) or stage 2 (Switch form))
          const int n_rg = (g->F + 15) / 16, n_sh1 = g->has_shared ? (g->Fs + 15) / 16 : 0;
          const bool sw = g->cfg.router_kind == MOEINF_ROUTER_SWITCH;
          for (int b = 0; b < g->layer1_trace_blocks; ++b) {
            int c = b;
            if (c < g->E) { tb[b] = (1 << 24) | c; continue; }
            c -= g->E;
            if (c < n_sh1) { tb[b] = (2 << 24) | c; continue; }
            c -= n_sh1;
            if (c == 0) { tb[b] = 3 << 24; continue; }
            c -= 1;
            if (c < g->K * n_rg) { tb[b] = (4 << 24) | c; continue; }
            c -= g->K * n_rg;
            tb[b] = ((sw ? 6 : 5) << 24) | c;
          }
        }
        for (int b = 0; b < g->layer1_trace_blocks; ++b)  // "workgroup item role index t0 t1 t2 t3"
          if (tb[b]) fprintf(f, "%d %d %d %d %llu %llu %llu %llu\n", b, 0, tb[b] >> 24, tb[b] & 0xffffff, tr[b * 4], tr[b * 4 + 1], tr[b * 4 + 2], tr[b * 4 + 3]);
        fclose(f);
      }
    hipFree(g->d_layer_trace);
  }
  for (auto& n : g->nodes) { for (auto& h : n.disk_reqs) PrioAioPool::wait(h); n.disk_reqs.clear(); set_node_store(n, nullptr); }  // reads into the arena
  g->aio.reset();
  if (g->ep_comm) { std::string e; if (const RcclApi* api = RcclApi::get(&e)) api->CommDestroy(g->ep_comm); g->ep_comm = nullptr; }
  for (void* b : {g->ep_x_send, g->ep_x_recv, g->ep_x_y, g->ep_x_ret}) if (b) hipFree(b);
  g->ep_x_y = nullptr;
  if (g->ep_err_host) { hipHostFree(g->ep_err_host); g->ep_err_host = nullptr; }
  g->ep_win.destroy();  // (the caller's ranks have agreed to stop before any of them gets here: a peer may still store into the window)
  for (auto& s : g->slots) if (s.dev) hipFree(s.dev);
  for (auto p : g->shared_dev) if (p) hipFree(p);
  for (auto p : g->arena_chunks) hipHostFree(p);
  for (auto& n : g->nodes) { if (n.ready) hipEventDestroy(n.ready); if (n.ready1) hipEventDestroy(n.ready1); }
  for (auto e : g->event_pool) hipEventDestroy(e);
  if (g->busy_mark) hipEventDestroy(g->busy_mark);
  if (g->d_la_f) hipFree(g->d_la_f);
  if (g->d_la_i) hipFree(g->d_la_i);
  if (g->h_la_idx) hipHostFree(g->h_la_idx);
  if (g->h_la_w) hipHostFree(g->h_la_w);
  if (g->h_keep) hipHostFree(g->h_keep);
  if (g->d_copy_ts) hipFree(g->d_copy_ts);
  for (auto& pr : g->copy_timers) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
  free_token_workspace(g);
  void* bufs[] = {g->d_wptr, g->d_counts, g->d_offsets, g->d_active, g->d_n_active,
                  g->d_arrive, g->d_ep_rec, g->d_miss, g->d_dec_w, g->d_dec_cw, g->d_layer_ctr, g->d_layer_part, g->d_h_sh, g->d_y_sh, g->d_ep_key, g->d_ep_counts, g->d_ep_offsets, g->d_ep_active,
                  g->d_ep_nactive, g->d_ep_pair_slot, g->d_ep_slot_token, g->d_ep_slot_pair, g->d_ep_pair_pos};
  for (void* b : bufs) if (b) hipFree(b);
  if (g->mirror_slab) hipHostFree(g->mirror_slab);
  for (CopyLane* ln : {&g->demand, &g->prefetch}) {
    for (auto& b : ln->ring) {
      if (b.dev) hipFree(b.dev);
      if (b.filled) hipEventDestroy(b.filled);
      if (b.freed) hipEventDestroy(b.freed);
    }
    if (ln->copy) hipStreamDestroy(ln->copy);
    if (ln->retile) hipStreamDestroy(ln->retile);
  }
  if (g->h_mirror) hipHostFree(g->h_mirror);
  if (g->h_miss) hipHostFree(g->h_miss);
  if (g->route_ev) hipEventDestroy(g->route_ev);
  for (int i = 0; i < kFenceRing; ++i) if (g->fence_ev[i]) hipEventDestroy(g->fence_ev[i]);
  delete g;
  return MOEINF_OK;
}

static void prealloc_slots(moeinf_engine* g);
extern "C" int moeinf_create(const moeinf_config* cfg, moeinf_engine** out) {
  if (!out) return fail(MOEINF_ERR_INVALID, "out is NULL");
  *out = nullptr;
  // fp8 experts (the reference's dtype id 3, core/parallel/expert_module.h:23,118-119): e4m3fn bytes in the HOST tier and on the link,
  // up-cast to bf16 when an expert is pulled into its HBM slot; activations, gate and all arithmetic are bf16 — y = FFN(x; W.to(bf16)),
  // what torch::linear over up-cast weights computes.  Everything behind the tier mover sees a bf16 engine.
  moeinf_config cfg_local;
  bool host_f8 = false;
  if (cfg && cfg->dtype == MOEINF_DTYPE_F8E4M3) {
    cfg_local = *cfg; cfg_local.dtype = MOEINF_DTYPE_BF16; host_f8 = true;
    if (cfg_local.gate_dtype == MOEINF_DTYPE_F8E4M3) cfg_local.gate_dtype = MOEINF_DTYPE_BF16;
    cfg = &cfg_local;
  }
  CHK(validate(cfg));
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(MOEINF_ERR_INVALID, "device_id %d but %d HIP devices visible", cfg->device_id, ndev);
  DeviceScope on_dev_(cfg->device_id); HIPCHK(on_dev_.err);
  moeinf_engine* g = new moeinf_engine();
  g->cfg = *cfg;
  // Grok / Arctic (moe_infinity/models/grok.py:38-45): Mixtral's router without the renormalisation.  Everything else about the
  // kind — bf16 gate logits, selection, dispatch, combine, which kernels run — is Mixtral's: the engine keeps ONE kind for it
  // and a flag for the weights.
  if (g->cfg.router_kind == MOEINF_ROUTER_SOFTMAX_TOPK) { g->route_no_renorm = true; g->cfg.router_kind = MOEINF_ROUTER_MIXTRAL; }
  // DeepSeek-V3's gate (modeling_deepseek_v3 MoEGate): DeepSeek's block in everything but the router arithmetic — one kind, a flag;
  // the per-layer e_score_correction_bias comes through moeinf_set_gate_bias
  if (g->cfg.router_kind == MOEINF_ROUTER_DEEPSEEK_V3) { g->route_v3 = true; g->cfg.router_kind = MOEINF_ROUTER_DEEPSEEK; g->gate_bias.assign((size_t)g->cfg.num_layers, nullptr); }
  memset(&g->st, 0, sizeof g->st);
  memset(&g->prof, 0, sizeof g->prof);
  memset(&g->ep_prof, 0, sizeof g->ep_prof);
  for (int i = 0; i < kFenceRing; ++i) g->fence_ev[i] = nullptr;
  g->L = cfg->num_layers; g->E = cfg->num_experts; g->K = cfg->top_k; g->H = cfg->hidden; g->F = cfg->inter; g->Fs = cfg->shared_inter;
  g->has_shared = cfg->shared_inter > 0;
  g->dt = cfg->dtype == MOEINF_DTYPE_BF16 ? DT_BF16 : (cfg->dtype == MOEINF_DTYPE_F16 ? DT_F16 : DT_F32);
  g->es = dt_bytes(g->dt);
  g->host_f8 = host_f8;
  g->host_es = host_f8 ? 1 : g->es;
  g->lay = make_layout(cfg->expert_type, g->H, g->F, g->host_es);
  if (g->has_shared) g->lay_sh = make_layout(cfg->expert_type, g->H, g->Fs, g->host_es);
  g->dlay = make_dev_layout(cfg->expert_type, g->H, g->F, g->dt, g->es);
  if (g->has_shared) g->dlay_sh = make_dev_layout(cfg->expert_type, g->H, g->Fs, g->dt, g->es);
  g->slot_bytes = g->dlay.total;
  g->nodes.resize((size_t)g->L * g->E);
  g->pol.resize((size_t)g->L * g->E);
  g->shared_dev.assign(g->L, nullptr);
  g->resident_per_layer.assign(g->L, 0);

  auto bail = [&](int rc) { std::string keep = g_err; moeinf_destroy(g); g_err = keep; return rc; };
#define TRY(x) do { int r__ = (x); if (r__ != MOEINF_OK) return bail(r__); } while (0)
#define TRYHIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { fail(MOEINF_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e__)); return bail(MOEINF_ERR_HIP); } } while (0)

  // budget: fraction of TOTAL device memory (reference semantics) unless an explicit byte budget is given
  size_t free_b = 0, total_b = 0;
  TRYHIP(hipMemGetInfo(&free_b, &total_b));
  const int64_t budget = cfg->device_memory_bytes > 0 ? cfg->device_memory_bytes : (int64_t)((double)total_b * cfg->device_memory_ratio);
  int64_t owned = 0;
  for (int e = 0; e < g->E; ++e) if (owns(g, e)) ++owned;
  g->owned_experts = (int)owned;
  g->max_slots = std::min<int64_t>(budget / g->slot_bytes, owned * g->L);
  if (g->max_slots < 1) { fail(MOEINF_ERR_OOM, "device budget %lld bytes cannot hold one expert of %lld bytes", (long long)budget, (long long)g->slot_bytes); return bail(MOEINF_ERR_OOM); }
  g->st.slots_total = g->max_slots;
  g->st.slot_bytes = g->slot_bytes;

  int lo = 0, hi = 0;
  TRYHIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  // on-demand misses outrank speculative copies
  TRYHIP(hipStreamCreateWithPriority(&g->demand.copy, hipStreamNonBlocking, hi));
  TRYHIP(hipStreamCreateWithPriority(&g->demand.retile, hipStreamNonBlocking, hi));
  TRYHIP(hipStreamCreateWithPriority(&g->prefetch.copy, hipStreamNonBlocking, lo));
  TRYHIP(hipStreamCreateWithPriority(&g->prefetch.retile, hipStreamNonBlocking, lo));
  if (const char* w = getenv("MOEINF_PREFETCH_WINDOW")) g->prefetch_window = std::max(1, atoi(w));
  TRYHIP(hipEventCreateWithFlags(&g->route_ev, hipEventDisableTiming));
  for (int i = 0; i < kFenceRing; ++i) TRYHIP(hipEventCreateWithFlags(&g->fence_ev[i], hipEventDisableTiming));

  const size_t E1 = (size_t)g->E + 1;
  g->ldh = std::max(g->F, g->Fs);
  TRY(dmalloc(&g->d_wptr, (size_t)g->L * E1));
  TRYHIP(hipMemset(g->d_wptr, 0, (size_t)g->L * E1 * sizeof(uint64_t)));
  TRY(dmalloc(&g->d_counts, E1)); TRY(dmalloc(&g->d_offsets, E1 + 1)); TRY(dmalloc(&g->d_active, E1)); TRY(dmalloc(&g->d_n_active, 1));
  TRY(dmalloc(&g->d_arrive, (g->H + 15) / 16)); TRY(dmalloc(&g->d_miss, 1));
  TRY(dmalloc(&g->d_ep_rec, 64));
  TRYHIP(hipMemset(g->d_arrive, 0, (size_t)((g->H + 15) / 16) * sizeof(int32_t)));
  TRYHIP(hipMemset(g->d_miss, 0, sizeof(int32_t)));
  TRY(dmalloc(&g->d_dec_w, 8)); TRY(dmalloc(&g->d_dec_cw, 8));
  {
    // the counters of the one-launch layer: uncached memory, so that the scalar unit can poll them (no cache level of this GPU
    // keeps a line; MOEINF_LAYER1_POLL=vector: ordinary memory, agent-scope vector loads)
    const bool scalar = !(getenv("MOEINF_LAYER1_POLL") && !strcmp(getenv("MOEINF_LAYER1_POLL"), "vector"));
    const size_t cb = (size_t)LAYER1_CTRS * LAYER1_CTR_STRIDE * sizeof(uint32_t);
    if (scalar && hipExtMallocWithFlags((void**)&g->d_layer_ctr, cb, hipDeviceMallocUncached) == hipSuccess) g->layer1_scalar_poll = true;
    else { (void)hipGetLastError(); g->d_layer_ctr = nullptr; TRY(dmalloc(&g->d_layer_ctr, LAYER1_CTRS * LAYER1_CTR_STRIDE)); }
    TRYHIP(hipMemset(g->d_layer_ctr, 0, cb));
  }
  g->layer1_timeout_ticks = (int64_t)(getenv("MOEINF_LAYER1_TIMEOUT_MS") ? std::max(1, atoi(getenv("MOEINF_LAYER1_TIMEOUT_MS"))) : 2000) * 100000;
  TRYHIP(hipMemset(g->d_dec_w, 0, 8 * sizeof(uint64_t))); TRYHIP(hipMemset(g->d_dec_cw, 0, 8 * sizeof(float)));
  TRYHIP(hipMemset(g->d_active, 0, (size_t)E1 * sizeof(int32_t)));  // the FFN kernels read active[u] before they know n_active
  TRYHIP(hipMemset(g->d_n_active, 0, sizeof(int32_t)));
  TRY(alloc_token_workspace(g, cfg->max_tokens));
  if (g->has_shared) {
    TRYHIP(hipMalloc(&g->d_h_sh, (size_t)kHideSharedMaxTokens * g->Fs * g->es));
    TRYHIP(hipMalloc(&g->d_y_sh, (size_t)kHideSharedMaxTokens * g->H * g->es));
  }
  for (int i = 0; i < 4; ++i) g->stage_bytes = std::max<int64_t>(g->stage_bytes, std::max(align_up(g->lay.size[i], kAioAlignment), align_up(g->lay_sh.size[i] * (g->host_f8 ? 2 : 1), kAioAlignment)));
  {
    // whole-blob transfers for experts up to MOEINF_H2D_WHOLE_BLOB_MB (64; 0 = always tensor by tensor): DeepSeek-V2-Lite's
    // 16.5 MiB expert was three 5.5 MiB copies with an event pair and a re-tile launch each: 46 GB/s against the 54.5 that
    // Mixtral's 112 MiB pieces reach (round 5 offload leg)
    const char* e = getenv("MOEINF_H2D_WHOLE_BLOB_MB");
    const int64_t cap = (e ? atoll(e) : 64) << 20;
    bool vec_ok = true;
    for (int i = 0; i < g->dlay.n; ++i) if (g->dlay.K[i] == 0 && (g->dlay.size[i] % 16) != 0) vec_ok = false;
    g->whole_blob = g->lay.total <= cap && vec_ok;
    // MOEINF_H2D_PULL (default 1): the pull form needs 16-byte vector pieces like the whole-blob re-tile; MOEINF_H2D_PULL_WGS: workgroups per launch
    const char* pe = getenv("MOEINF_H2D_PULL");
    g->h2d_pull = (pe ? atoi(pe) != 0 : true) && vec_ok;
    const char* pw = getenv("MOEINF_H2D_PULL_WGS");
    // 16 workgroups with four 16-byte loads per lane in flight pull at the link's rate, fp8 blobs included (8: -3 %, 32: -2...6 %;
    // profiles/r06_tier_mover_pull_vs_sdma_ab.txt)
    g->h2d_pull_wgs = pw ? std::max(1, atoi(pw)) : 16;
    // MOEINF_FENCE_EVERY (default 16; 1 = a fence behind every forward): how often a sync-free forward records its fence
    const char* fe = getenv("MOEINF_FENCE_EVERY");
    g->fence_every = fe ? std::min(std::max(1, atoi(fe)), kMirrorPool / 4) : 16;
    if (g->host_f8) {  // the fp8 pull loads 16 source bytes (sixteen elements) per lane
      bool ok16 = true;
      for (int i = 0; i < g->dlay.n; ++i) ok16 = ok16 && (g->dlay.K[i] > 0 ? g->dlay.K[i] % 16 == 0 : g->dlay.size[i] % 32 == 0);
      if (!ok16) { fail(MOEINF_ERR_UNSUPPORTED, "fp8 experts (dtype 3): hidden / inter (and bias lengths) must be multiples of 16"); return bail(MOEINF_ERR_UNSUPPORTED); }
    }
    if (g->host_f8 && !g->h2d_pull) { fail(MOEINF_ERR_UNSUPPORTED, "fp8 experts (dtype 3) are up-cast by the PULL tier mover: not with MOEINF_H2D_PULL=0, nor with bias vectors that are not 16-byte multiples"); return bail(MOEINF_ERR_UNSUPPORTED); }
    if (g->h2d_pull) {
      TRYHIP(hipMalloc((void**)&g->d_copy_ts, (size_t)kCopyTsRing * 32));
      TRYHIP(hipMemset(g->d_copy_ts, 0, (size_t)kCopyTsRing * 32));
      g->copy_ts_expect.assign(kCopyTsRing, 0);
    }
    if (g->whole_blob) g->stage_bytes = std::max<int64_t>(g->stage_bytes, align_up(g->lay.total, kAioAlignment));
  }
  for (CopyLane* ln : {&g->demand, &g->prefetch}) {
    for (auto& b : ln->ring) {
      TRYHIP(hipMalloc(&b.dev, (size_t)g->stage_bytes));
      TRYHIP(hipEventCreateWithFlags(&b.filled, hipEventDisableTiming));
      TRYHIP(hipEventCreateWithFlags(&b.freed, hipEventDisableTiming));
    }
  }
  TRYHIP(hipHostMalloc((void**)&g->h_mirror, (1 + 2 * E1) * sizeof(int32_t), hipHostMallocDefault));
  {
    const size_t per = ((size_t)(1 + 2 * E1) + 15) / 16 * 16;  // ints per mirror, 64-byte aligned
    TRYHIP(hipHostMalloc((void**)&g->mirror_slab, per * kMirrorPool * sizeof(int32_t), hipHostMallocDefault));
    memset(g->mirror_slab, 0, per * kMirrorPool * sizeof(int32_t));
    for (int i = 0; i < kMirrorPool; ++i) g->mirror_pool.push_back(g->mirror_slab + per * i);
  }
  TRYHIP(hipHostMalloc((void**)&g->h_miss, sizeof(int32_t), hipHostMallocDefault));
  *g->h_miss = 0;
  prealloc_slots(g);
  if (g->slots.empty() && g->slab_exhausted) { fail(MOEINF_ERR_OOM, "no device memory for a single expert slot of %lld bytes", (long long)g->slot_bytes); return bail(MOEINF_ERR_OOM); }
  *out = g;
  return MOEINF_OK;
#undef TRY
#undef TRYHIP
}

static int retile_tensor(const moeinf_engine* g, const DevLayout& dl, int i, const void* staged, void* slot, hipStream_t cs);
static int alloc_host_block(moeinf_engine* g, int idx, void** out, bool may_block = true);
static int invalidate_resident(moeinf_engine* g, int idx);

// ---- registration --------------------------------------------------------------------------
extern "C" int moeinf_expert_layout(const moeinf_engine* g, int which, int64_t offsets[4], int64_t sizes[4], int32_t* n_tensors, int64_t* total_bytes) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  const BlobLayout& b = which ? g->lay_sh : g->lay;
  if (which && !g->has_shared) return fail(MOEINF_ERR_INVALID, "engine has no shared expert");
  for (int i = 0; i < 4; ++i) { if (offsets) offsets[i] = b.off[i]; if (sizes) sizes[i] = b.size[i]; }
  if (n_tensors) *n_tensors = b.n;
  if (total_bytes) *total_bytes = b.total;
  return MOEINF_OK;
}

static int check_le(const moeinf_engine* g, int layer, int expert) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  if (layer < 0 || layer >= g->L) return fail(MOEINF_ERR_INVALID, "layer %d out of range [0,%d)", layer, g->L);
  if (expert < 0 || expert >= g->E) return fail(MOEINF_ERR_INVALID, "expert %d out of range [0,%d)", expert, g->E);
  return MOEINF_OK;
}

extern "C" int moeinf_register_expert(moeinf_engine* g, int layer, int expert, const void* blob, int64_t nbytes) {
  CHK(check_le(g, layer, expert));
  if (!owns(g, expert)) return fail(MOEINF_ERR_INVALID, "expert %d is not owned by ep_rank %d of %d", expert, g->cfg.ep_rank, g->cfg.ep_size);
  if (blob && nbytes != g->lay.total) return fail(MOEINF_ERR_INVALID, "expert blob is %lld bytes, layout needs %lld", (long long)nbytes, (long long)g->lay.total);
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  const int idx = node_index(g, layer, expert);
  Node& n = g->nodes[idx];
  if (n.host_pending) {  // a background disk read of the old payload is under way: let it finish into its block first
    for (auto& h : n.disk_reqs) PrioAioPool::wait(h);
    n.disk_reqs.clear();
    for (auto it = g->disk_inflight.begin(); it != g->disk_inflight.end(); ++it)
      if ((int)it->node == idx) { g->disk_inflight.erase(it); break; }
    { auto sit = std::find(g->stale_disk.begin(), g->stale_disk.end(), idx); if (sit != g->stale_disk.end()) g->stale_disk.erase(sit); }
    n.host = n.host_pending;
    n.host_pending = nullptr;
  }
  if (!n.host) CHK(alloc_host_block(g, idx, &n.host));
  // the caller's blob is now the authoritative host copy: the expert is no longer re-readable from (and must never be
  // overwritten by) an offload directory it was registered from earlier
  set_node_store(n, nullptr);
  memset(n.store_ids, 0, sizeof n.store_ids);
  if (blob) {
    CHK(invalidate_resident(g, idx));
    memcpy(n.host, blob, (size_t)nbytes);
  }
  return MOEINF_OK;
}

extern "C" int moeinf_expert_host_ptr(moeinf_engine* g, int layer, int expert, void** host_ptr) {
  CHK(check_le(g, layer, expert));
  Node& n = g->nodes[node_index(g, layer, expert)];
  if (!n.host) return fail(MOEINF_ERR_STATE, "expert (%d,%d) is not registered", layer, expert);
  *host_ptr = n.host;
  return MOEINF_OK;
}

extern "C" int moeinf_register_shared(moeinf_engine* g, int layer, const void* blob, int64_t nbytes) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  if (!g->has_shared) return fail(MOEINF_ERR_INVALID, "engine was created without a shared expert");
  if (layer < 0 || layer >= g->L) return fail(MOEINF_ERR_INVALID, "layer out of range");
  if (!blob || nbytes != g->lay_sh.total) return fail(MOEINF_ERR_INVALID, "shared blob is %lld bytes, layout needs %lld", (long long)nbytes, (long long)g->lay_sh.total);
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  if (!g->shared_dev[layer]) HIPCHK(hipMalloc(&g->shared_dev[layer], (size_t)g->dlay_sh.total));
  // one-off, synchronous: tensor by tensor through the demand lane's first staging buffer
  HIPCHK(hipStreamSynchronize(g->demand.copy));
  HIPCHK(hipStreamSynchronize(g->demand.retile));
  std::vector<uint16_t> up;  // fp8 host tier: the (always resident) shared expert is up-cast on the host, once
  for (int i = 0; i < g->dlay_sh.n; ++i) {
    const void* src = (const char*)blob + g->lay_sh.off[i];
    size_t bytes = (size_t)g->lay_sh.size[i];
    if (g->host_f8) {
      up.resize(bytes);
      for (size_t j = 0; j < bytes; ++j) up[j] = f8e4m3_to_bf16_bits(((const uint8_t*)src)[j]);
      src = up.data(); bytes *= 2;
    }
    HIPCHK(hipMemcpyAsync(g->demand.ring[0].dev, src, bytes, hipMemcpyHostToDevice, g->demand.copy));
    CHK(retile_tensor(g, g->dlay_sh, i, g->demand.ring[0].dev, g->shared_dev[layer], g->demand.copy));
    HIPCHK(hipStreamSynchronize(g->demand.copy));
  }
  uint64_t p = (uint64_t)g->shared_dev[layer];
  HIPCHK(hipMemcpy(g->d_wptr + (size_t)layer * (g->E + 1) + g->E, &p, sizeof p, hipMemcpyHostToDevice));
  return MOEINF_OK;
}

// ---- device tier ---------------------------------------------------------------------------
static void queue_poke(moeinf_engine* g, int layer, int expert, uint64_t val) {
  if (g->pending_pokes.empty() || g->pending_pokes.back().n == 16) {
    PokeArgs p;
    p.table = g->d_wptr;
    p.n = 0;
    g->pending_pokes.push_back(p);
  }
  PokeArgs& p = g->pending_pokes.back();
  p.idx[p.n] = layer * (g->E + 1) + expert;
  p.val[p.n] = val;
  ++p.n;
}
int flush_pokes(moeinf_engine* g, hipStream_t st) {
  for (auto& p : g->pending_pokes) HIPCHK(launch_poke(p, st));
  g->pending_pokes.clear();
  return MOEINF_OK;
}

static void drop_ready_count(moeinf_engine* g, int idx) {
  Node& n = g->nodes[idx];
  if (n.slot >= 0 && n.ready_waited) g->resident_per_layer[idx % g->L] -= 1;
}

// Allocate the HBM slots up to the budget NOW (engine creation, budget growth) instead of on the first miss that needs
// one: a hipMalloc of a 336 MiB slot costs milliseconds and would sit on the demand-miss path.  If physical memory runs
// out before the budget does, the cache simply has fewer slots.  MOEINF_PREALLOC=0 restores first-touch allocation.
static void prealloc_slots(moeinf_engine* g) {
  static const bool on = getenv("MOEINF_PREALLOC") ? atoi(getenv("MOEINF_PREALLOC")) != 0 : true;
  if (!on) return;
  while ((int64_t)g->slots.size() < g->max_slots && !g->slab_exhausted) {
    void* p = nullptr;
    if (hipMalloc(&p, (size_t)g->slot_bytes) != hipSuccess) {
      (void)hipGetLastError();
      g->slab_exhausted = true;
      g->st.slots_total = (int64_t)g->slots.size();
      break;
    }
    Slot s;
    s.dev = p;
    g->slots.push_back(s);
    g->free_slots.push_back((int)g->slots.size() - 1);
  }
}

// obtain a device slot for node `idx`; may evict.  Pinned entries (pol[].pinned) are never evicted.
// *victim_out = node index of the evicted tenant (-1: the slot was free / fresh).
static int acquire_slot(moeinf_engine* g, int idx, int* slot_out, bool allow_protected, int* victim_out) {
  *victim_out = -1;
  if (!g->free_slots.empty()) {
    *slot_out = g->free_slots.back();
    g->free_slots.pop_back();
    return MOEINF_OK;
  }
  if ((int64_t)g->slots.size() < g->max_slots && !g->slab_exhausted) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, (size_t)g->slot_bytes);
    if (e == hipSuccess) {
      Slot s;
      s.dev = p;
      g->slots.push_back(s);
      *slot_out = (int)g->slots.size() - 1;
      return MOEINF_OK;
    }
    (void)hipGetLastError();
    g->slab_exhausted = true;  // physical memory ran out before the budget did: cache stops growing
    g->st.slots_total = (int64_t)g->slots.size();
  }
  const int64_t v = pick_victim(g->pol.data(), (int64_t)g->pol.size(), g->cfg.policy, allow_protected);
  if (v < 0) return fail(MOEINF_ERR_OOM, "no evictable expert: %zu slots all pinned or protected", g->slots.size());
  Node& vn = g->nodes[v];
  drop_ready_count(g, (int)v);
  const int slot = vn.slot;
  vn.slot = -1;
  if (vn.prefetched) { g->st.prefetch_wasted += 1; vn.unused += 1; g->gov_score += (0.f - g->gov_score) * 0.125f; g->gov_outcomes += 1; }  // brought in speculatively, evicted before any dispatch used it
  vn.prefetched = false;
  g->pol[v].resident = false;
  g->slots[slot].node = -1;
  g->st.evictions += 1;
  g->st.slots_used -= 1;
  queue_poke(g, (int)(v % g->L), (int)(v / g->L), 0);
  *slot_out = slot;
  *victim_out = (int)v;
  return MOEINF_OK;
}

// the host copy of node idx is about to change: wait for a transfer that may still read it and give up the HBM slot, so
// the next dispatch copies the new payload (the slot's reuse fences cover kernels that may still read the old one)
static int invalidate_resident(moeinf_engine* g, int idx) {
  Node& n = g->nodes[idx];
  if (n.copy_inflight && n.ready) { HIPCHK(hipEventSynchronize(n.ready)); n.copy_inflight = false; }
  if (n.slot < 0) return MOEINF_OK;
  drop_ready_count(g, idx);
  const int slot = n.slot;
  n.slot = -1; n.prefetched = false; n.ready_waited = true; n.waited1 = true;
  g->pol[idx].resident = false;
  g->slots[slot].node = -1;
  g->free_slots.push_back(slot);
  g->st.slots_used -= 1;
  queue_poke(g, idx % g->L, idx / g->L, 0);
  return MOEINF_OK;
}

// staged row-major tensor i (device staging buffer) -> its place in the tiled slot, on stream cs
static int retile_tensor(const moeinf_engine* g, const DevLayout& dl, int i, const void* staged, void* slot, hipStream_t cs) {
  char* dst = (char*)slot + dl.off[i];
  if (dl.K[i] > 0) HIPCHK(launch_retile(staged, dst, dl.R[i], dl.K[i], g->dt, cs));
  else HIPCHK(hipMemcpyAsync(dst, staged, (size_t)dl.size[i], hipMemcpyDeviceToDevice, cs));
  return MOEINF_OK;
}

// Copy order of a blob's tensors: what FFN stage 1 reads first (w1 AND w3 / gate AND up / fc1 + bias / wi), then
// the stage-2 tensors — so the compute stream can start stage 1 while the down projection is still on the link.
// Returns the number of stage-1 tensors.
static int copy_order(int expert_type, int order[4]) {
  switch (expert_type) {
    case MOEINF_EXPERT_MIXTRAL: order[0] = 0; order[1] = 2; order[2] = 1; return 2;   // w1 w3 | w2
    case MOEINF_EXPERT_DEEPSEEK: case MOEINF_EXPERT_SWITCH_GATED: order[0] = 0; order[1] = 1; order[2] = 2; return 2;  // gate up | down
    case MOEINF_EXPERT_NLLB: case MOEINF_EXPERT_FSGPT: order[0] = 0; order[1] = 1; order[2] = 2; order[3] = 3; return 2;  // fc1.w fc1.b | fc2.w fc2.b
    default: order[0] = 0; order[1] = 1; return 1;                                     // wi | wo
  }
}

static int ensure_host(moeinf_engine* g, int idx);

// start the H2D transfer of node idx into a slot on lane `ln` (reference: Node::SetDevice host->device leg,
// model_topology.cpp:102-119, which is a cudaMemcpyAsync + cudaStreamSynchronize of the whole blob)
static int issue_copy(moeinf_engine* g, int idx, CopyLane& ln, bool allow_protected) {
  Node& n = g->nodes[idx];
  if (!n.host && !n.store) return fail(MOEINF_ERR_STATE, "expert (layer %d, expert %d) was dispatched but never registered", idx % g->L, idx / g->L);
  CHK(ensure_host(g, idx));
  int slot = -1, victim = -1;
  CHK(acquire_slot(g, idx, &slot, allow_protected, &victim));
  Slot& s = g->slots[slot];
  if (!n.ready) HIPCHK(hipEventCreateWithFlags(&n.ready, hipEventDisableTiming));
  if (!n.ready1) HIPCHK(hipEventCreateWithFlags(&n.ready1, hipEventDisableTiming));
  // link-busy timers: the pull form times itself inside its kernels (no queue packets); the SDMA forms bracket the copies with events
  hipEvent_t start = g->h2d_pull ? nullptr : get_event(g), stop = g->h2d_pull ? nullptr : get_event(g);
  if (start && stop) HIPCHK(hipEventRecord(start, ln.copy));
  int order[4];
  const int n1 = copy_order(g->cfg.expert_type, order);
  // first write into the slot.  (a) kernels of forward #last_use_seq may still read the previous tenant: wait for a fence
  // that covers it (fence_for records one if the sync-free forwards since have not);
  // (b) the previous tenant's OWN transfer may still be in flight on another lane (a prefetched expert is
  // evictable from the moment its copy is issued): write-after-write on the slot
  auto order_first_write_on = [&](hipStream_t ws) -> int {
    if (s.last_use_seq > 0) {
      hipEvent_t fe = nullptr;
      CHK(fence_for(g, std::min(s.last_use_seq, g->seq), &fe));
      if (hipEventQuery(fe) != hipSuccess) {
        (void)hipGetLastError();
        HIPCHK(hipStreamWaitEvent(ws, fe, 0));
      }
    }
    if (victim >= 0) {
      Node& vn = g->nodes[victim];
      if (!vn.ready_waited && vn.ready && hipEventQuery(vn.ready) != hipSuccess) {
        (void)hipGetLastError();
        HIPCHK(hipStreamWaitEvent(ws, vn.ready, 0));
      }
      vn.ready_waited = true; vn.waited1 = true;
    }
    return MOEINF_OK;
  };
  auto order_first_write = [&]() -> int { return order_first_write_on(ln.retile); };
  if (g->h2d_pull) {
    // PULL (kernels.hip: pull_retile_kernel): the copy stream's own kernel reads the pinned host blob and writes the tiled slot.
    // Small experts: one launch for the whole blob; big ones: the stage-1 tensors first (ready1), then the rest.
    CHK(order_first_write_on(ln.copy));
    unsigned long long* ts = nullptr;
    int ts_slot = -1;
    if (g->d_copy_ts) {
      if (g->copy_ts_head - g->copy_ts_tail >= (uint64_t)kCopyTsRing) g->copy_ts_tail = g->copy_ts_head - kCopyTsRing + 1;  // overflow: the oldest records are given up
      ts_slot = (int)(g->copy_ts_head % kCopyTsRing);
      ts = g->d_copy_ts + (size_t)ts_slot * 4;
    }
    int launches = 0;
    auto pull = [&](int k0, int k1) -> int {
      RetileBlob rb;
      memset(&rb, 0, sizeof rb);
      rb.src = n.host; rb.dst = s.dev; rb.n = 0; rb.src_f8 = g->host_f8 ? 1 : 0;
      for (int k = k0; k < k1; ++k) {
        const int i = order[k], j = rb.n++;
        rb.src_off[j] = g->lay.off[i]; rb.dst_off[j] = g->dlay.off[i];
        rb.K[j] = g->dlay.K[i];
        rb.R[j] = g->dlay.K[i] > 0 ? g->dlay.R[i] : (int)(g->dlay.size[i] / 16);
      }
      HIPCHK(launch_pull_retile(rb, g->dt, g->h2d_pull_wgs, ln.copy, ts, launches == 0 ? 1 : 0));
      launches += 1;
      return MOEINF_OK;
    };
    bool one_event = false;
    if (g->whole_blob || n1 >= g->lay.n) {
      CHK(pull(0, g->lay.n));
      one_event = true;  // stage 1 and stage 2 wait for the same point of the stream: ONE event record serves both
    } else {
      CHK(pull(0, n1));
      HIPCHK(hipEventRecord(n.ready1, ln.copy));
      CHK(pull(n1, g->lay.n));
    }
    if (ts_slot >= 0) {
      g->copy_ts_expect[ts_slot] += (uint64_t)launches * (uint64_t)g->h2d_pull_wgs;
      g->copy_ts_head += 1;
    }
    HIPCHK(hipEventRecord(n.ready, ln.copy));
    n.ready1_is_ready = one_event;
    n.slot = slot;
    n.ready_waited = false;
    n.waited1 = false;
    n.copy_inflight = true;
    n.copy_seq = ++g->copy_seq; n.copy_lane = (&ln == &g->prefetch) ? 1 : 0;
    n.host_clock = ++g->host_clock;
    s.node = idx;
    g->pol[idx].resident = true;
    g->st.slots_used += 1;
    g->st.h2d_bytes += g->lay.total;
    queue_poke(g, idx % g->L, idx / g->L, (uint64_t)s.dev);
    return MOEINF_OK;
  }
  bool sdma_one_event = false;
  if (g->whole_blob) {
    StageBuf& b = ln.ring[ln.next];
    ln.next = (ln.next + 1) % kStageRing;
    if (b.used) HIPCHK(hipStreamWaitEvent(ln.copy, b.freed, 0));  // its previous content has been re-tiled
    HIPCHK(hipMemcpyAsync(b.dev, n.host, (size_t)g->lay.total, hipMemcpyHostToDevice, ln.copy));
    HIPCHK(hipEventRecord(b.filled, ln.copy));
    HIPCHK(hipStreamWaitEvent(ln.retile, b.filled, 0));
    CHK(order_first_write());
    RetileBlob rb;
    memset(&rb, 0, sizeof rb);
    rb.src = b.dev; rb.dst = s.dev; rb.n = g->lay.n;
    for (int i = 0; i < g->lay.n; ++i) {
      rb.src_off[i] = g->lay.off[i]; rb.dst_off[i] = g->dlay.off[i];
      rb.K[i] = g->dlay.K[i];
      rb.R[i] = g->dlay.K[i] > 0 ? g->dlay.R[i] : (int)(g->dlay.size[i] / 16);
    }
    HIPCHK(launch_retile_blob(rb, g->dt, ln.retile));
    HIPCHK(hipEventRecord(b.freed, ln.retile));
    b.used = true;
    sdma_one_event = true;  // (`ready`, recorded below at the same point of the re-tile stream, serves for both)
  } else
  for (int k = 0; k < g->lay.n; ++k) {
    const int i = order[k];
    StageBuf& b = ln.ring[ln.next];
    ln.next = (ln.next + 1) % kStageRing;
    if (b.used) HIPCHK(hipStreamWaitEvent(ln.copy, b.freed, 0));  // its previous content has been re-tiled
    HIPCHK(hipMemcpyAsync(b.dev, (const char*)n.host + g->lay.off[i], (size_t)g->lay.size[i], hipMemcpyHostToDevice, ln.copy));
    HIPCHK(hipEventRecord(b.filled, ln.copy));
    HIPCHK(hipStreamWaitEvent(ln.retile, b.filled, 0));
    if (k == 0) CHK(order_first_write());
    CHK(retile_tensor(g, g->dlay, i, b.dev, s.dev, ln.retile));
    HIPCHK(hipEventRecord(b.freed, ln.retile));
    b.used = true;
    if (k == n1 - 1) HIPCHK(hipEventRecord(n.ready1, ln.retile));
  }
  if (start && stop) {
    HIPCHK(hipEventRecord(stop, ln.copy));  // link-busy interval: first to last hipMemcpyAsync of this expert
    g->copy_timers.push_back({start, stop});
  }
  HIPCHK(hipEventRecord(n.ready, ln.retile));
  n.ready1_is_ready = sdma_one_event;
  n.slot = slot;
  n.ready_waited = false;
  n.waited1 = false;
  n.copy_inflight = true;
  n.copy_seq = ++g->copy_seq; n.copy_lane = (&ln == &g->prefetch) ? 1 : 0;
  n.host_clock = ++g->host_clock;
  s.node = idx;
  g->pol[idx].resident = true;
  g->st.slots_used += 1;
  g->st.h2d_bytes += g->lay.total;
  queue_poke(g, idx % g->L, idx / g->L, (uint64_t)s.dev);
  return MOEINF_OK;
}

// the pull form's timing records (pull_retile_kernel): finished copies' [start, end] ticks (100 MHz) merged into the link-busy time
static void settle_copy_ticks(moeinf_engine* g, bool wait) {
  if (!g->d_copy_ts || g->copy_ts_tail == g->copy_ts_head) return;
  if (wait) { hipStreamSynchronize(g->demand.copy); hipStreamSynchronize(g->prefetch.copy); }
  const uint64_t n = g->copy_ts_head - g->copy_ts_tail;
  std::vector<unsigned long long> rec((size_t)n * 4);
  const uint64_t t0 = g->copy_ts_tail % kCopyTsRing, first = std::min<uint64_t>(n, kCopyTsRing - t0);
  if (hipMemcpy(rec.data(), g->d_copy_ts + t0 * 4, first * 32, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return; }
  if (n > first && hipMemcpy(rec.data() + first * 4, g->d_copy_ts, (n - first) * 32, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return; }
  std::vector<std::pair<unsigned long long, unsigned long long>> iv;
  uint64_t done = 0;
  for (; done < n; ++done) {  // in issue order; stop at the first copy that has not finished
    const int slot = (int)((g->copy_ts_tail + done) % kCopyTsRing);
    if (rec[done * 4 + 2] < g->copy_ts_expect[slot]) break;
    if (rec[done * 4 + 1] > rec[done * 4]) iv.push_back({rec[done * 4], rec[done * 4 + 1]});
  }
  g->copy_ts_tail += done;
  std::sort(iv.begin(), iv.end());
  for (auto& p : iv) {  // union over the lanes: only what extends beyond the accounted time counts
    const unsigned long long lo = std::max(p.first, g->copy_busy_until);
    if (p.second > lo) { g->st.h2d_busy_ms += (double)(p.second - lo) * 1e-5; g->copy_busy_until = p.second; }
  }
}

static void settle_copy_timers(moeinf_engine* g, bool wait) {
  settle_copy_ticks(g, wait);
  size_t keep = 0;
  // link-busy time = the UNION of the lanes' copy intervals (the demand lane and the speculative lane run side by side: a
  // plain sum would count the shared link twice).  Intervals settle in issue order; `busy_mark` is the stop event of the
  // interval that ends latest so far — an interval only adds what it extends beyond that mark.
  for (size_t i = 0; i < g->copy_timers.size(); ++i) {
    auto pr = g->copy_timers[i];
    if (wait) hipEventSynchronize(pr.second);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
      float add = ms, gap = 0.f, ext = 0.f;
      bool new_mark = true;
      if (g->busy_mark && hipEventElapsedTime(&gap, g->busy_mark, pr.first) == hipSuccess && gap < 0.f) {  // starts inside the accounted time
        if (hipEventElapsedTime(&ext, g->busy_mark, pr.second) == hipSuccess && ext > 0.f) add = ext;
        else { add = 0.f; new_mark = false; }
      } else (void)hipGetLastError();
      g->st.h2d_busy_ms += add;
      g->event_pool.push_back(pr.first);
      if (new_mark) {
        if (g->busy_mark) g->event_pool.push_back(g->busy_mark);
        g->busy_mark = pr.second;
      } else g->event_pool.push_back(pr.second);
    } else {
      (void)hipGetLastError();
      g->copy_timers[keep++] = pr;
    }
  }
  g->copy_timers.resize(keep);
  keep = 0;
  for (size_t i = 0; i < g->wait_timers.size(); ++i) {
    auto pr = g->wait_timers[i];
    if (wait) hipEventSynchronize(pr.second);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
      g->st.exposed_wait_ms += ms;
      g->event_pool.push_back(pr.first);
      g->event_pool.push_back(pr.second);
    } else {
      (void)hipGetLastError();
      g->wait_timers[keep++] = pr;
    }
  }
  g->wait_timers.resize(keep);
}

// ---- host tier as a cache over the disk tier -------------------------------------------------
// Experts registered from an offload directory keep (store, tensor ids); when the pinned arena is capped
// (cfg.host_memory_bytes — the reference's host_memory_ratio, memory_pool.cpp:150-158) and full, the least recently
// needed host blob that can be re-read from disk gives up its arena block (reference: Node::SetDevice(DISK),
// model_topology.cpp:76-88) and the wanted expert is read disk -> pinned host (-> HBM by the caller).
static PrioAioPool* aio_pool(moeinf_engine* g) {
  if (!g->aio) {
    const char* t = getenv("MOEINF_AIO_THREADS");
    g->aio.reset(new PrioAioPool(t ? atoi(t) : 4));
  }
  return g->aio.get();
}

// an arena block for node idx's host blob: free list -> grow under the cap -> drop the least recently needed host blob
// that can be re-read from disk
// may_block = false (the speculative pump, which is documented never to block): no waiting for a transfer or a disk read —
// the caller drops the request instead (MOEINF_ERR_OOM)
static int alloc_host_block(moeinf_engine* g, int idx, void** out, bool may_block) {
  void* blk = nullptr;
  if (!g->host_free.empty()) {
    blk = g->host_free.back();
    g->host_free.pop_back();
  } else if (g->cfg.host_memory_bytes <= 0 || g->arena_total + g->lay.total <= g->cfg.host_memory_bytes) {
    CHK(arena_alloc(g, g->lay.total, &blk));
  } else {
    int victim = -1, busy = -1;
    for (int i = 0; i < (int)g->nodes.size(); ++i) {
      Node& v = g->nodes[i];
      if (i == idx || !v.host || !v.store) continue;
      if (v.copy_inflight) {  // its H2D copy may still be reading the blob (ready_waited only says the compute stream is ORDERED after it)
        if (hipEventQuery(v.ready) != hipSuccess) {
          (void)hipGetLastError();
          if (busy < 0 || v.host_clock < g->nodes[busy].host_clock) busy = i;
          continue;
        }
        v.copy_inflight = false;
      }
      if (victim < 0 || v.host_clock < g->nodes[victim].host_clock) victim = i;
    }
    if (victim < 0 && busy >= 0 && may_block) {  // every droppable blob is still being copied: wait for the oldest transfer
      HIPCHK(hipEventSynchronize(g->nodes[busy].ready));
      g->nodes[busy].copy_inflight = false;
      victim = busy;
    }
    if (victim < 0 && may_block) {
      // every block is held by a speculative disk read that has not been adopted yet: the oldest one that can give up a block
      // (stale ones first) is promoted, waited for and gives up its block (its task is dropped — the expert stays on disk).
      // Candidates that cannot (the requester itself, a read that was adopted meanwhile) keep their place in their list
      // (round-3 advice: the first candidate used to be popped, and lost, whether or not it yielded a block).
      int di = -1;
      for (auto it = g->stale_disk.begin(); it != g->stale_disk.end(); ++it)
        if (*it != idx && g->nodes[*it].host_pending) { di = *it; g->stale_disk.erase(it); break; }
      if (di < 0)
        for (auto it = g->disk_inflight.begin(); it != g->disk_inflight.end(); ++it)
          if ((int)it->node != idx && g->nodes[(int)it->node].host_pending) { di = (int)it->node; g->disk_inflight.erase(it); break; }
      if (di >= 0) {
        Node& dn = g->nodes[di];
        for (auto& h : dn.disk_reqs) g->aio->promote(h);
        for (auto& h : dn.disk_reqs) PrioAioPool::wait(h);
        dn.disk_reqs.clear();
        blk = dn.host_pending;
        dn.host_pending = nullptr;
        g->st.prefetch_dropped += 1;
        *out = blk;
        return MOEINF_OK;
      }
    }
    if (victim < 0) return fail(MOEINF_ERR_OOM, "pinned host arena cap (%lld bytes) reached and no host blob can be dropped", (long long)g->cfg.host_memory_bytes);
    blk = g->nodes[victim].host;
    g->nodes[victim].host = nullptr;
    g->st.host_evictions += 1;
  }
  *out = blk;
  return MOEINF_OK;
}

// submit the reads of every tensor of node idx's blob into `blk` (each tensor's region in the blob is 4 KiB aligned and
// padded -> eligible for O_DIRECT)
static int submit_host_read(moeinf_engine* g, int idx, void* blk, bool high) {
  Node& n = g->nodes[idx];
  PrioAioPool* pool = aio_pool(g);
  n.disk_reqs.clear();
  for (int i = 0; i < g->lay.n; ++i) {
    const uint64_t room = (uint64_t)((i + 1 < g->lay.n ? g->lay.off[i + 1] : g->lay.total) - g->lay.off[i]);
    char* dst = (char*)blk + g->lay.off[i];
    OffloadStore::ReadPlan rp;
    const std::string err = n.store->plan_read(n.store_ids[i], dst, room, &rp);
    if (!err.empty()) {
      for (auto& h : n.disk_reqs) PrioAioPool::wait(h);
      n.disk_reqs.clear();
      return fail(MOEINF_ERR_INVALID, "%s", err.c_str());
    }
    n.disk_reqs.push_back(pool->submit(rp.path, dst, (int64_t)rp.size, rp.offset, high, rp.direct_ok));
  }
  n.host_pending = blk;
  return MOEINF_OK;
}

// wait for node idx's pending disk read and adopt the blob
static int finish_host_read(moeinf_engine* g, int idx) {
  Node& n = g->nodes[idx];
  std::string err;
  for (auto& h : n.disk_reqs) {
    const std::string e = PrioAioPool::wait(h);
    if (err.empty()) err = e;
  }
  n.disk_reqs.clear();
  void* blk = n.host_pending;
  n.host_pending = nullptr;
  if (!err.empty()) { g->host_free.push_back(blk); return fail(MOEINF_ERR_INVALID, "%s", err.c_str()); }
  g->st.disk_reads += 1;
  g->st.disk_bytes += g->lay.total;
  n.host = blk;
  return MOEINF_OK;
}
static bool host_read_done(const Node& n) {
  for (auto& h : n.disk_reqs) if (!PrioAioPool::done(h)) return false;
  return true;
}

static int ensure_host(moeinf_engine* g, int idx) {
  Node& n = g->nodes[idx];
  n.host_clock = ++g->host_clock;
  if (n.host) return MOEINF_OK;
  if (!n.store) return fail(MOEINF_ERR_STATE, "expert (layer %d, expert %d) has no host copy and no disk copy", idx % g->L, idx / g->L);
  if (n.host_pending) {  // a speculative read of this blob is under way: it is needed now
    for (auto& h : n.disk_reqs) g->aio->promote(h);
    for (auto it = g->disk_inflight.begin(); it != g->disk_inflight.end(); ++it)
      if ((int)it->node == idx) { g->disk_inflight.erase(it); break; }
    { auto sit = std::find(g->stale_disk.begin(), g->stale_disk.end(), idx); if (sit != g->stale_disk.end()) g->stale_disk.erase(sit); }
    return finish_host_read(g, idx);
  }
  void* blk = nullptr;
  CHK(alloc_host_block(g, idx, &blk));
  const int rc = submit_host_read(g, idx, blk, /*high=*/true);  // all tensors at once: the workers read them in parallel
  if (rc != MOEINF_OK) { g->host_free.push_back(blk); return rc; }
  return finish_host_read(g, idx);
}

// ---- the hot path --------------------------------------------------------------------------
static int pump_prefetch(moeinf_engine* g);
// a layer is being dispatched: speculative transfers queued for layers the pass has already left are stale
// (StartExec drops every queued task with a smaller layer id, task_scheduler.cpp:158-168)
void drop_stale_prefetches(moeinf_engine* g, int layer) {
  if (!g->pq.empty()) g->st.prefetch_cancelled += g->pq.on_demand(-1, layer);
  // speculative disk reads for layers the pass has left: the read itself finishes (the blob is adopted into the host
  // tier by the next demand for it), but its task leaves the pipeline so that no H2D copy is issued for a stale layer
  for (auto it = g->disk_inflight.begin(); it != g->disk_inflight.end();) {
    if (it->layer < layer) { g->stale_disk.push_back((int)it->node); it = g->disk_inflight.erase(it); g->st.prefetch_cancelled += 1; }
    else ++it;
  }
}
int pump_if_pending(moeinf_engine* g) {
  if (g->pq.empty() && g->prefetch_inflight.empty() && g->disk_inflight.empty() && g->stale_disk.empty()) return MOEINF_OK;
  return pump_prefetch(g);
}
// FfnStage::fuse_combine: 1 = the hand-off rows leave as sixteen 2-byte write-through stores; 2 (MOEINF_WIDE_OUT=1) = gathered
// through LDS into 16-byte ones — measured SLOWER (DeepSeek-V2-Lite 1.031/1.038 vs 1.022/1.031 ms/token, stage 2 +0.6 us: the
// extra LDS round trip and barrier cost more than the fabric writes they save), so off by default
static int fuse_mode() {
  static const int m = (getenv("MOEINF_WIDE_OUT") && atoi(getenv("MOEINF_WIDE_OUT")) != 0) ? 2 : 1;
  return m;
}
void fill_stage(const moeinf_engine* g, int layer, int stage, FfnStage& s, int64_t ld_x) {
  const DevLayout& b = g->dlay;
  const DevLayout& bs = g->dlay_sh;
  memset(&s, 0, sizeof s);
  s.wptr = g->d_wptr + (size_t)layer * (g->E + 1);
  s.active = g->d_active; s.n_active = g->d_n_active; s.counts = g->d_counts; s.offsets = g->d_offsets;
  s.miss_flag = g->d_miss;
  s.dec_w = g->d_dec_w; s.dec_cw = g->d_dec_cw;
  s.n_active_host = -1;
  s.E = g->E;
  s.dtype = g->dt;
  s.rows_bound = (int64_t)g->cfg.max_tokens * (g->K + 1);
  const int et = g->cfg.expert_type;
  if (stage == 1) {
    s.K = g->H; s.R = g->F; s.K_sh = g->H; s.R_sh = g->Fs;
    s.ld_in = ld_x > 0 ? ld_x : g->H; s.row_map = g->d_slot_token; s.out = g->d_h; s.ld_out = g->ldh;
    if (et == MOEINF_EXPERT_MIXTRAL) { s.off_a = b.off[0]; s.off_b = b.off[2]; s.epi = EPI_GATED_SILU; }
    else if (et == MOEINF_EXPERT_DEEPSEEK) { s.off_a = b.off[0]; s.off_b = b.off[1]; s.off_a_sh = bs.off[0]; s.off_b_sh = bs.off[1]; s.epi = EPI_GATED_SILU; }
    else if (et == MOEINF_EXPERT_SWITCH_GATED) { s.off_a = b.off[0]; s.off_b = b.off[1]; s.epi = EPI_GATED_GELU; }  // gelu(x wi_0^T) * (x wi_1^T)
    else if (et == MOEINF_EXPERT_SWITCH) { s.off_a = b.off[0]; s.epi = EPI_RELU; }
    else { s.off_a = b.off[0]; s.off_bias = b.off[1]; s.epi = EPI_BIAS_RELU; }
  } else {
    s.K = g->F; s.R = g->H; s.K_sh = g->Fs; s.R_sh = g->H;
    s.in = g->d_h; s.ld_in = g->ldh; s.row_map = nullptr; s.out = g->d_y; s.ld_out = g->H;
    if (g->ovr_out) { s.out = g->ovr_out; s.out_map = g->ovr_map; }
    if (et == MOEINF_EXPERT_MIXTRAL) { s.off_a = b.off[1]; s.epi = EPI_NONE; }
    else if (et == MOEINF_EXPERT_DEEPSEEK) { s.off_a = b.off[2]; s.off_a_sh = bs.off[2]; s.epi = EPI_NONE; }
    else if (et == MOEINF_EXPERT_SWITCH_GATED) { s.off_a = b.off[2]; s.epi = EPI_NONE; }  // wo
    else if (et == MOEINF_EXPERT_SWITCH) { s.off_a = b.off[1]; s.epi = EPI_NONE; }
    else { s.off_a = b.off[2]; s.off_bias = b.off[3]; s.epi = EPI_BIAS; }
  }
}

// algorithmic bytes of one forward (profiling), from its routing mirror
// local = a whole local forward (router, shared expert and combine included); false = the expert-parallel owner-side
// FFN over received rows only
static void account_profile(moeinf_engine* g, const int32_t* mirror, int T, bool local = true) {
  const int E = g->E, K = g->K;
  const moeinf_profile p0 = g->prof;
  int64_t U = 0, rows = 0;
  for (int e = 0; e < E; ++e) { if (mirror[1 + e] > 0) { ++U; rows += mirror[1 + e]; } }
  const bool hidden = g->has_shared && local && g->last_hidden_shared;  // the shared expert ran inside the router launches
  const int64_t es = g->es, H = g->H, F = g->F, Fs = g->Fs, Tsh = (g->has_shared && local && !hidden) ? T : 0;
  // a self-routing forward carries the hidden shared expert's stage 2 inside the FFN stage-1 launch
  const bool sr2 = hidden && g->last_selfroute;
  if (hidden) g->prof.route_bytes += (sr2 ? 2 : 3) * Fs * H * es + (int64_t)T * ((sr2 ? 1 : 2) * Fs + (sr2 ? 1 : 2) * H) * es;
  if (sr2) g->prof.ffn1_bytes += Fs * H * es + (int64_t)T * (Fs + H) * es;
  const int et = g->cfg.expert_type;
  const bool gated = (et == MOEINF_EXPERT_MIXTRAL || et == MOEINF_EXPERT_DEEPSEEK || et == MOEINF_EXPERT_SWITCH_GATED);
  const bool bias = (et == MOEINF_EXPERT_NLLB || et == MOEINF_EXPERT_FSGPT);
  const int64_t b1 = U * ((gated ? 2 : 1) * F * H * es + (bias ? F * es : 0)) + (Tsh ? 2 * Fs * H * es : 0) + (rows + Tsh) * H * es + rows * F * es + Tsh * Fs * es;
  const int64_t b2 = U * (H * F * es + (bias ? H * es : 0)) + (Tsh ? H * Fs * es : 0) + rows * F * es + Tsh * Fs * es + (rows + Tsh) * H * es;
  g->prof.ffn1_bytes += b1;
  g->prof.ffn2_bytes += b2;
  if (local) {
    g->prof.route_bytes += (int64_t)E * H * (g->cfg.gate_dtype == MOEINF_DTYPE_F32 ? 4 : 2) + (int64_t)T * H * es + (int64_t)T * E * 4 * 2 + (int64_t)T * K * 12;
    g->prof.combine_bytes += (rows + Tsh) * H * es + (int64_t)T * H * es;
  }
  g->prof.forwards += 1;
  if (mirror[0] > 0) { g->prof.ffn1_launches += 1; g->prof.ffn2_launches += 1; }
  if (local && g->last_front1 && !g->last_layer1) {
    // the gate (and the hidden shared expert's first stage) ran inside the launch timed as "ffn1": their bytes belong there
    g->prof.ffn1_bytes += g->prof.route_bytes - p0.route_bytes;
    g->prof.route_bytes = p0.route_bytes;
  }
  if (local && g->last_layer1) {
    // the one-launch layer: every byte of the forward moves inside the launch timed as "ffn1" (the other intervals are empty)
    g->prof.ffn1_bytes = p0.ffn1_bytes + (g->prof.ffn1_bytes - p0.ffn1_bytes) + (g->prof.ffn2_bytes - p0.ffn2_bytes) + (g->prof.route_bytes - p0.route_bytes) +
                         (g->prof.combine_bytes - p0.combine_bytes);
    g->prof.ffn2_bytes = p0.ffn2_bytes; g->prof.route_bytes = p0.route_bytes; g->prof.combine_bytes = p0.combine_bytes;
    if (mirror[0] > 0) g->prof.ffn2_launches -= 1;
    g->prof.fused_layers += 1;
  }
}

// ExpertPredictor.predict + ExpertPrefetcher.prefetch_experts (moe_infinity/memory/expert_predictor.py:17-35,
// expert_prefetcher.py:42-59) for the attached sequence, from a routing mirror {n_active, counts[E+1], active[E+1]}:
// update the sequence's EAM with this layer's experts, find the nearest historical EAM, and (prefetch) request the
// experts it predicts for the next `pred_lookahead` layers — only those whose predicted share of their layer's
// activations is >= pred_min_share, best first, at most pred_max per call, never a resident one.  The reference
// requests EVERY predicted expert of every later layer; on a 56 GB/s link that is 5x slower than fetching on demand
// (DESIGN.md section 7.3), hence the threshold and the bounded look-ahead.
static int predictor_observe(moeinf_engine* g, int layer, const int32_t* mirror, bool prefetch) {
  Tracer* tr = g->pred_tracer;
  if (!tr || !tr->has(g->pred_seq)) return MOEINF_OK;
  const int E = g->E, E1 = E + 1, L = g->L;
  std::vector<int32_t> ex;
  for (int i = 0; i < mirror[0]; ++i) {
    const int e = mirror[1 + E1 + i];
    if (e < 0 || e >= E) continue;
    for (int c = 0; c < std::max(1, (int)mirror[1 + e]); ++c) ex.push_back(e);  // one entry per routed (token, k) pair, as expert_index lists them
  }
  g->pred_matrix.resize((size_t)L * E);
  tr->predict(g->pred_seq, layer, ex.data(), (int)ex.size(), g->pred_matrix.data());
  g->pred_calls += 1;
  if (!prefetch || g->pred_lookahead <= 0) return MOEINF_OK;
  struct Cand { float score; int layer, expert; };
  std::vector<Cand> cand;
  for (int l2 = layer + 1; l2 < L && l2 <= layer + g->pred_lookahead; ++l2) {
    const float* row = &g->pred_matrix[(size_t)l2 * E];
    double sum = 0;
    for (int e = 0; e < E; ++e) sum += row[e];
    if (!(sum > 0)) continue;
    for (int e = 0; e < E; ++e) {
      if (!owns(g, e) || row[e] / sum < g->pred_min_share) continue;
      const Node& nd = g->nodes[node_index(g, l2, e)];
      if (nd.slot >= 0 || (!nd.host && !nd.store)) continue;  // resident / in flight / never registered
      cand.push_back({row[e], l2, e});
    }
  }
  std::stable_sort(cand.begin(), cand.end(), [](const Cand& a, const Cand& b) { return a.score > b.score; });
  if ((int)cand.size() > g->pred_max) cand.resize((size_t)std::max(0, g->pred_max));
  const float top = cand.empty() ? 1.f : std::max(cand[0].score, 1e-30f);
  for (const Cand& c : cand) {
    const float sc = c.score / top;
    g->st.prefetch_cancelled += g->pq.enqueue(node_index(g, c.layer, c.expert), c.layer, priority_from_score(&sc, 0));
    g->pred_enqueued += 1;
  }
  return MOEINF_OK;
}

// apply the routing mirrors of finished sync-free forwards to counters / stats (all were hits).
// Entries beyond `max_pending` are waited for (oldest first), the rest are taken only if already complete.
static void drain_mirrors(moeinf_engine* g, size_t max_pending) {
  while (!g->pend.empty()) {
    auto& pm = g->pend.front();
    // a fence that covers the forward: sync-free forwards record one every fence_every-th time, so a mirror is applied up to
    // that many forwards late; only a drain that has to wait records one of its own
    hipEvent_t ev = covering_fence(g, pm.seq);
    if (g->pend.size() > max_pending) {
      if (!ev && fence_for(g, pm.seq, &ev) != MOEINF_OK) break;
      hipEventSynchronize(ev);
    } else if (!ev) {
      break;
    } else if (hipEventQuery(ev) != hipSuccess) {
      (void)hipGetLastError();
      break;
    }
    const int E1 = g->E + 1;
    const int na = pm.buf[0];
    const int32_t* active = pm.buf + 1 + E1;
    for (int i = 0; i < na; ++i) {
      const int e = active[i];
      if (e >= g->E) continue;
      const int idx = node_index(g, pm.layer, e);
      Node& n = g->nodes[idx];
      n.visit += 1; n.hit += 1;
      g->st.expert_hits += 1;
      if (n.prefetched) { g->st.prefetch_useful += 1; n.prefetched = false; g->gov_score += (1.f - g->gov_score) * 0.125f; g->gov_outcomes += 1; }
      g->pol[idx].incache += 1;
      g->pol[idx].last_access = ++g->clock;
    }
    if (pm.prof) account_profile(g, pm.buf, pm.T, pm.local);
    if (g->pred_tracer && pm.local) predictor_observe(g, pm.layer, pm.buf, /*prefetch=*/false);  // resident layer: trace only
    g->mirror_pool.push_back(pm.buf);
    g->pend.pop_front();
  }
}
static void drain_mirrors(moeinf_engine* g, bool block) { drain_mirrors(g, block ? (size_t)0 : (size_t)-1); }

// Where the next index kernel writes its routing mirror, decided BEFORE that launch: a pooled pinned buffer when
// the layer is fully resident (sync-free forward), the engine's h_mirror when the host has to look at it.
static void settle_ready(moeinf_engine* g, int layer);
int plan_mirror(moeinf_engine* g, int layer, MirrorPlan& mp) {
  settle_ready(g, layer);
  mp.fast = g->resident_per_layer[layer] == g->owned_experts;
  if (mp.fast) {
    drain_mirrors(g, (size_t)(kMirrorPool - 4));
    // at most kMirrorPool - 4 mirrors are pending, and the pool was carved out of one pinned slab at creation
    // (hipHostMalloc on the forward path stalls the device for milliseconds)
    if (g->mirror_pool.empty()) return fail(MOEINF_ERR_STATE, "routing-mirror pool exhausted");
    mp.target = g->mirror_pool.back();
    g->mirror_pool.pop_back();
  } else {
    drain_mirrors(g, true);
    mp.target = g->h_mirror;
  }
  // decode-sized index kernels report only the active experts: start from "no expert has rows"
  memset(mp.target, 0, (size_t)(2 + g->E) * sizeof(int32_t));
  return MOEINF_OK;
}

// experts whose copy has already landed need no stream wait any more: count them as ordered
static void settle_ready(moeinf_engine* g, int layer) {
  if (g->resident_per_layer[layer] == g->owned_experts) return;
  for (int e = 0; e < g->E; ++e) {
    Node& n = g->nodes[node_index(g, layer, e)];
    if (n.slot < 0 || n.ready_waited || !n.ready) continue;
    if (hipEventQuery(n.ready) == hipSuccess) {
      n.ready_waited = true;
      n.waited1 = true;
      g->resident_per_layer[layer] += 1;
    } else {
      (void)hipGetLastError();
    }
  }
}

// make every active routed expert of `layer` resident and order the compute stream after the part of its transfer
// that FFN STAGE 1 reads (ready1).  Experts whose stage-2 tensors may still be in flight are appended to `late`:
// the caller orders the compute stream after their `ready` event between the two FFN launches, so stage 1 runs
// while the down projections are still on the link.
// h_mirror = {n_active, counts[E+1], active[E+1]} (already on the host).
static int ensure_resident(moeinf_engine* g, int layer, hipStream_t st, std::vector<int>& late, int a0 = 0, int a1 = -1) {
  const int E1 = g->E + 1;
  const int na = a1 < 0 ? g->h_mirror[0] : a1;
  const int32_t* active = g->h_mirror + 1 + E1;
  // pin this layer's active experts so a miss cannot evict a sibling that the same launch reads
  for (int i = a0; i < na; ++i) { const int e = active[i]; if (e < g->E) g->pol[node_index(g, layer, e)].pinned = true; }
  int rc = MOEINF_OK;
  // if the compute stream will have to wait for a copy, time the stall (exposed copy time)
  bool will_wait = false;
  for (int i = a0; i < na; ++i) { const int e = active[i]; if (e < g->E) { const Node& n = g->nodes[node_index(g, layer, e)]; if (n.slot < 0 || !n.waited1) will_wait = true; } }
  hipEvent_t w0 = nullptr, w1 = nullptr;
  if (will_wait) { w0 = get_event(g); w1 = get_event(g); if (w0 && w1) record_timing(w0, st); }
  std::vector<int> need_one;
  for (int i = a0; i < na && rc == MOEINF_OK; ++i) {
    const int e = active[i];
    if (e >= g->E) continue;  // shared pseudo-expert
    const int idx = node_index(g, layer, e);
    Node& n = g->nodes[idx];
    n.visit += 1;
    if (n.slot >= 0) {
      n.hit += 1;
      g->st.expert_hits += 1;
      if (!n.ready_waited && hipEventQuery(n.ready) != hipSuccess) { (void)hipGetLastError(); g->st.inflight_hits += 1; }
      if (n.prefetched) { g->st.prefetch_useful += 1; n.prefetched = false; g->gov_score += (1.f - g->gov_score) * 0.125f; g->gov_outcomes += 1; }
    } else {
      n.miss += 1;
      g->st.expert_misses += 1;
      // a queued speculative transfer of this expert is overtaken by the demand (StartExec, task_scheduler.cpp:158-168)
      g->st.prefetch_cancelled += g->pq.remove_node(idx);
      // (one demand lane: a layer's misses dealt over TWO concurrent copy streams measured slower — DeepSeek-V2-Lite offload
      // leg 27.9 -> 32.9 ms/token, 52.6 -> 44.2 GB/s, profiles/r06_offload_whole_blob_ab.txt: the streams share one link)
      rc = issue_copy(g, idx, g->demand, true);
      if (rc != MOEINF_OK) break;
      g->demand_inflight.push_back(idx);
    }
    if (!n.waited1 && n.ready1_is_ready) {
      // the whole blob arrives with ONE event (pull form / whole-blob copies): waited for below, once per copy lane — the lane's
      // copies complete in issue order, so its latest one covers the others — and nothing is left for wait_late
      need_one.push_back(idx);
    } else {
      if (!n.waited1) {
        hipError_t he = hipStreamWaitEvent(st, n.ready1, 0);
        if (he != hipSuccess) { rc = fail(MOEINF_ERR_HIP, "hipStreamWaitEvent: %s", hipGetErrorString(he)); break; }
        n.waited1 = true;
      }
      if (!n.ready_waited) late.push_back(idx);
    }
    g->pol[idx].incache += 1;  // incache_visit_count += 1 on every dispatch (expert_dispatcher.cpp:263)
    g->pol[idx].last_access = ++g->clock;
    n.host_clock = ++g->host_clock;
    g->slots[n.slot].last_use_seq = g->seq + 1;
  }
  if (rc == MOEINF_OK && !need_one.empty()) {
    int last[2] = {-1, -1};
    for (int idx : need_one) { const Node& n = g->nodes[idx]; const int ln = n.copy_lane & 1; if (last[ln] < 0 || n.copy_seq > g->nodes[last[ln]].copy_seq) last[ln] = idx; }
    for (int ln = 0; ln < 2 && rc == MOEINF_OK; ++ln)
      if (last[ln] >= 0) {
        hipError_t he = hipStreamWaitEvent(st, g->nodes[last[ln]].ready, 0);
        if (he != hipSuccess) rc = fail(MOEINF_ERR_HIP, "hipStreamWaitEvent: %s", hipGetErrorString(he));
      }
    if (rc == MOEINF_OK)
      for (int idx : need_one) { Node& n = g->nodes[idx]; n.waited1 = true; if (!n.ready_waited) { n.ready_waited = true; g->resident_per_layer[layer] += 1; } }
  }
  for (int i = a0; i < na; ++i) { const int e = active[i]; if (e < g->E) g->pol[node_index(g, layer, e)].pinned = false; }
  if (w0 && w1) { record_timing(w1, st); g->wait_timers.push_back({w0, w1}); }
  return rc;
}

// order the compute stream after the full transfers of `late` (between FFN stage 1 and stage 2)
static int wait_late(moeinf_engine* g, int layer, hipStream_t st, std::vector<int>& late) {
  if (late.empty()) return MOEINF_OK;
  hipEvent_t w0 = get_event(g), w1 = get_event(g);
  if (w0 && w1) record_timing(w0, st);
  for (int idx : late) {
    Node& n = g->nodes[idx];
    if (n.ready_waited || n.slot < 0) continue;
    HIPCHK(hipStreamWaitEvent(st, n.ready, 0));
    n.ready_waited = true;
    g->resident_per_layer[layer] += 1;
  }
  if (w0 && w1) { record_timing(w1, st); g->wait_timers.push_back({w0, w1}); }
  late.clear();
  return MOEINF_OK;
}

// The predicted experts of layer + 1 (moeinf_set_lookahead; g->la_list, best first) go onto the DEMAND lane behind this
// layer's misses: one link, first in first out — the misses keep their full bandwidth, the predictions take the link when it
// would idle (this layer's FFN, the next layer's attention and routing).  This layer's active experts stay pinned
// meanwhile: their forward's fence is not recorded yet, so their slots must not be chosen as victims.
static int lookahead_issue(moeinf_engine* g, int layer) {
  const int E1 = g->E + 1;
  const int na = g->h_mirror[0];
  const int32_t* active = g->h_mirror + 1 + E1;
  for (int i = 0; i < na; ++i) { const int e = active[i]; if (e < g->E) g->pol[node_index(g, layer, e)].pinned = true; }
  int rc = MOEINF_OK;
  for (int idx : g->la_list) {
    Node& nd = g->nodes[idx];
    if (nd.slot >= 0) continue;
    if (!nd.host) continue;  // on the disk tier only: left to the speculative queue's background read (moeinf_prefetch)
    g->st.prefetch_cancelled += g->pq.remove_node(idx);
    g->pol[idx].pinned = true;
    rc = issue_copy(g, idx, g->demand, false);
    g->pol[idx].pinned = false;
    if (rc == MOEINF_ERR_OOM) { g->st.prefetch_dropped += 1; rc = MOEINF_OK; continue; }  // nothing evictable: dropped, as a queued prefetch would be
    if (rc != MOEINF_OK) break;
    nd.prefetched = true;
    nd.prefetch_cnt += 1;
    g->st.prefetch_issued += 1;
    g->demand_inflight.push_back(idx);
  }
  for (int i = 0; i < na; ++i) { const int e = active[i]; if (e < g->E) g->pol[node_index(g, layer, e)].pinned = false; }
  g->la_list.clear();
  return rc;
}

// Launch both FFN stages for the active list in h_mirror.  If the layer needs more experts than
// the device cache holds, the list is processed in chunks (the reference runs experts one at a
// time, so it has no such limit): each chunk is made resident, launched, and fenced so the next
// chunk may recycle its slots once its kernels have drained.
static int run_experts(moeinf_engine* g, int layer, const void* x_in, hipStream_t st, hipEvent_t ev_before,
                       hipEvent_t ev_mid, hipEvent_t ev_after, int64_t ld_x = 0, const CombineArgs* fuse = nullptr,
                       bool* fused = nullptr, int rows_hint = 0) {
  const int E = g->E, E1 = E + 1;
  const int na = g->h_mirror[0];
  const int32_t* active = g->h_mirror + 1 + E1;
  const int64_t cap = g->slab_exhausted ? (int64_t)g->slots.size() : g->max_slots;
  FfnStage s1, s2;
  fill_stage(g, layer, 1, s1, ld_x);
  s1.in = x_in;
  fill_stage(g, layer, 2, s2);
  if (ev_before) HIPCHK(hipEventRecord(ev_before, st));
  int a = 0;
  while (a < na) {
    int b = a;
    int64_t used = 0;
    while (b < na && (active[b] >= E || used < cap)) { if (active[b] < E) ++used; ++b; }
    std::vector<int> late;
    const int64_t misses_before = g->st.expert_misses;
    CHK(ensure_resident(g, layer, st, late, a, b));
    // lookahead copies are issued while the host has nothing better to do: with misses of THIS layer on the link the FFN launches
    // below wait for them anyway (the host calls hide behind the copies); without, the FFN goes first
    const bool la_now = a == 0 && b == na && !g->la_list.empty();
    const bool la_early = la_now && g->st.expert_misses > misses_before;
    if (la_early) CHK(lookahead_issue(g, layer));
    CHK(flush_pokes(g, st));
    s1.active = g->d_active + a; s2.active = g->d_active + a;
    s1.n_active_host = b - a; s2.n_active_host = b - a;
    if (fuse && a == 0 && b == na && na > 0) {  // one chunk: the last column-tile block of stage 2 combines
      s2.fuse_combine = fuse_mode(); s2.tile_done = g->d_arrive; s2.comb = *fuse;
      if (fused) *fused = true;
    }
    // the exact maximum, but never below the estimate the sync-free path passes for the same forward (rows_hint): both paths
    // then pick the same kernel form unless the routing is skewed beyond 1.5 x the mean
    int max_rows = rows_hint;
    for (int i = a; i < b; ++i) max_rows = std::max(max_rows, (int)g->h_mirror[1 + active[i]]);
    HIPCHK(launch_ffn_stage(s1, b - a, max_rows, st));
    if (ev_mid && b == na) HIPCHK(hipEventRecord(ev_mid, st));
    CHK(wait_late(g, layer, st, late));  // stage 2 reads the down projections: wait for the rest of each transfer
    HIPCHK(launch_ffn_stage(s2, b - a, max_rows, st));
    if (la_now && !la_early) CHK(lookahead_issue(g, layer));
    a = b;
    if (a < na) CHK(end_forward(g, st, true));
  }
  if (na == 0 && ev_mid) HIPCHK(hipEventRecord(ev_mid, st));
  if (ev_after) HIPCHK(hipEventRecord(ev_after, st));
  return MOEINF_OK;
}

// Residency + the two FFN launches for the routing result that the index kernel just produced
// (d_mirror/d_active/d_counts/...).  x rows have stride ld_x elements (0: H).
//   Sync-free path: every owned expert of this layer is resident and already ordered before the
//   compute stream, so whatever the router picked is a hit — no host decision is needed and the host
//   does not wait for the routing result (the reference blocks on a D2H sum every layer,
//   expert_executor.py:34-43).  The mirror is applied to the counters lazily.
//   Decision path: some expert may be missing: small pinned D2H + event wait, then fetch/evict.
int dispatch_experts(moeinf_engine* g, int layer, const void* x_in, int64_t ld_x, int T, int max_active, int exp_rows,
                            hipStream_t st, bool prof, moeinf_engine::ProfRec* pr, const MirrorPlan& mp,
                            const CombineArgs* fuse, bool* fused, const SelfRoute* sr) {
  const int E = g->E, E1 = E + 1;
  if (fused) *fused = false;
  if (mp.fast) {
    // a workgroup of an earlier fused launch gave up a wait (flag 4): its kernel also wrote the pinned word, so the forward path
    // sees it without a device round trip — the error surfaces HERE, not only at the next moeinf_sync (as ep_err_host does for the exchange)
    if (g->h_miss && *(volatile int32_t*)g->h_miss == 4) { *g->h_miss = 0; CHK(check_device_flag(g)); }
    moeinf_engine::PendingMirror pm;
    pm.buf = mp.target; pm.seq = g->seq + 1; pm.layer = layer; pm.T = T; pm.prof = prof; pm.local = !g->ovr_out;
    g->pend.push_back(pm);
    for (int e = 0; e < E; ++e) {  // any of the layer's slots may be read by this forward
      const Node& n = g->nodes[node_index(g, layer, e)];
      if (n.slot >= 0) g->slots[n.slot].last_use_seq = g->seq + 1;
    }
    CHK(flush_pokes(g, st));
    FfnStage s1, s2;
    fill_stage(g, layer, 1, s1, ld_x);
    s1.in = x_in;
    fill_stage(g, layer, 2, s2);
    if (fuse) { s2.fuse_combine = fuse_mode(); s2.tile_done = g->d_arrive; s2.comb = *fuse; if (fused) *fused = true; }
    // decode launchers that carry the timer on their own dispatch packet: no event-record packets inside the interval
    // (every FFN-stage launcher is ONE launch and carries the timer; the small-batch self-routing stage 1 is the exception)
    const bool kt1 = prof && !(sr && !sr->layer1_switch && !sr->front1 && T > 1);
    const bool kt2 = prof;
    if (prof && !kt1) HIPCHK(hipEventRecord(pr->ev[2], st));
    if (sr && sr->layer1_switch) {
      LayerSync sy;
      memset(&sy, 0, sizeof sy);
      sy.ctr = g->d_layer_ctr; sy.launch = g->layer1_launches + 1; sy.timeout_ticks = g->layer1_timeout_ticks; sy.err = g->d_miss; sy.err_host = g->h_miss;
      static const int l1_sleep = getenv("MOEINF_LAYER1_SLEEP") ? std::max(1, atoi(getenv("MOEINF_LAYER1_SLEEP"))) : 2;
      sy.sleep = l1_sleep; sy.scalar_poll = g->layer1_scalar_poll ? 1 : 0;
      if (!g->num_cus) (void)hipDeviceGetAttribute(&g->num_cus, hipDeviceAttributeMultiprocessorCount, g->cfg.device_id);  // per engine: engines of one process may sit on different devices
      if (g->layer1_switch_wgs_per_cu < 0) g->layer1_switch_wgs_per_cu = layer1_switch_wgs_per_cu(sr->ra->x_dtype, sr->ra->gate_dtype);  // asked once: registers + LDS of the instantiation
      const int ncu = g->num_cus;
      if (getenv("MOEINF_LAYER1_TRACE")) {
        const int nb = g->E + 1 + (g->F + 15) / 16 + 4 * ((g->H + 15) / 16);
        if (!g->d_layer_trace) { if (hipMalloc((void**)&g->d_layer_trace, (size_t)nb * 32) != hipSuccess) g->d_layer_trace = nullptr; else (void)hipMemset(g->d_layer_trace, 0, (size_t)nb * 32); g->layer1_trace_blocks = nb; }
        sy.trace = g->d_layer_trace;
      }
      if (!g->d_layer_part) { if (hipMalloc((void**)&g->d_layer_part, (size_t)4 * g->H * sizeof(float)) != hipSuccess) { g->d_layer_part = nullptr; (void)hipGetLastError(); } }
      sy.part = g->d_layer_part;
      if (kt1 && fuse) arm_kernel_timer(pr->ev[2], pr->ev[3]);
      const bool one_launch = fuse && launch_moe_layer1_switch(*sr->ra, *sr->ia, s1, s2, sy, ncu, g->layer1_switch_wgs_per_cu, st);
      disarm_kernel_timer();  // (declined or failed before taking it: the self-routing stage 1 below arms its own)
      if (one_launch) {
        g->layer1_launches += 1;
        if (prof) { HIPCHK(hipEventRecord(pr->ev[4], st)); g->prof.kernel_timed_launches += 1; }
        return MOEINF_OK;
      }
      (void)hipGetLastError();
      g->last_layer1 = false;  // declined (fewer CUs than workgroups, ...): the three launches, starting with the gate the caller left out
      HIPCHK(launch_gate_logits(*sr->ra, st));
    }
    if (sr && sr->front1) {
      LayerSync sy;
      memset(&sy, 0, sizeof sy);
      sy.ctr = g->d_layer_ctr; sy.launch = g->layer1_launches + 1; sy.timeout_ticks = g->layer1_timeout_ticks; sy.err = g->d_miss; sy.err_host = g->h_miss;
      static const int f1_sleep = getenv("MOEINF_LAYER1_SLEEP") ? std::max(1, atoi(getenv("MOEINF_LAYER1_SLEEP"))) : 2;
      sy.sleep = f1_sleep; sy.scalar_poll = g->layer1_scalar_poll ? 1 : 0;
      if (getenv("MOEINF_LAYER1_TRACE")) {
        const int nb = g->E + (sr->sh1 ? (g->Fs + 15) / 16 : 0) + 1 + g->K * ((g->F + 15) / 16) + (sr->sh2 ? (g->H + 15) / 16 : 0);
        if (!g->d_layer_trace) { if (hipMalloc((void**)&g->d_layer_trace, (size_t)nb * 32) != hipSuccess) g->d_layer_trace = nullptr; else (void)hipMemset(g->d_layer_trace, 0, (size_t)nb * 32); g->layer1_trace_blocks = nb; }
        sy.trace = g->d_layer_trace;
      }
      // (a physical workgroup order that evens out the bytes per CU was measured in round 6: slower, profiles/r06_deepseek_front1_balanced_order_rejected.txt)
      if (kt1) arm_kernel_timer(pr->ev[2], pr->ev[3]);
      const hipError_t le = launch_moe_front1(*sr->ra, *sr->ia, sr->sh1, sr->sh2, s1, sy, st);
      disarm_kernel_timer();
      HIPCHK(le);
      g->layer1_launches += 1;  // only a launch that went out moves the grow-only counters' target (a failed one must not leave them out of step)
    } else if (sr && T > 1) HIPCHK(launch_ffn1_selfroute_multi(*sr->ra, *sr->ia, s1, sr->sh2, std::min(E, T * g->K), st));
    else if (sr) {
      if (kt1) arm_kernel_timer(pr->ev[2], pr->ev[3]);
      const hipError_t le = launch_ffn1_selfroute(*sr->ra, *sr->ia, s1, sr->sh2, st);
      disarm_kernel_timer();
      HIPCHK(le);
    }
    else {
      if (kt1) arm_kernel_timer(pr->ev[2], pr->ev[3]);
      const hipError_t le = launch_ffn_stage(s1, max_active, exp_rows, st);
      disarm_kernel_timer();
      HIPCHK(le);
    }
    if (prof && !kt1) HIPCHK(hipEventRecord(pr->ev[3], st));
    {
      if (kt2 && (pr->k2 = get_event(g))) arm_kernel_timer(pr->k2, pr->ev[4]);
      const hipError_t le = (sr && fuse && T == 1) ? launch_ffn2_decode1(s2, st) : launch_ffn_stage(s2, max_active, exp_rows, st);
      disarm_kernel_timer();
      HIPCHK(le);
    }
    if (prof && !(kt2 && pr->k2)) HIPCHK(hipEventRecord(pr->ev[4], st));
    if (prof) g->prof.kernel_timed_launches += (kt1 ? 1 : 0) + ((kt2 && pr->k2) ? 1 : 0);
  } else {
    // the index kernel wrote h_mirror itself (pinned, device-visible): wait for it, no copy
    HIPCHK(hipEventRecord(g->route_ev, st));
    const auto tw0 = std::chrono::steady_clock::now();
    HIPCHK(hipEventSynchronize(g->route_ev));
    if (g->profiling) g->prof.host_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
    for (int i = 0; i < g->h_mirror[0]; ++i) {
      const int e = g->h_mirror[1 + E1 + i];
      if (e < E && !owns(g, e)) return fail(MOEINF_ERR_STATE, "rank %d was handed rows for expert %d it does not own", g->cfg.ep_rank, e);
    }
    if (prof) account_profile(g, g->h_mirror, T, !g->ovr_out);
    // the host holds this layer's routing right now: predict and request the next layers' experts BEFORE serving this
    // layer's misses, so the speculative queue is ordered and the copies start as soon as the link is free
    if (g->pred_tracer && !g->ovr_out) CHK(predictor_observe(g, layer, g->h_mirror, /*prefetch=*/true));
    g->la_list.clear();
    if (g->la_armed_T > 0) {  // the lookahead route of this forward has landed with the routing mirror (same stream, same wait)
      // the la_max most confident predictions (largest gate weight over the tokens), of which the non-resident ones are issued
      std::vector<std::pair<float, int>> cand;
      for (int i = 0; i < g->la_armed_T * g->K; ++i) {
        const int e = g->h_la_idx[i];
        if (e < 0 || e >= E || !owns(g, e)) continue;
        const int idx = node_index(g, layer + 1, e);
        bool dup = false;
        for (auto& c : cand) if (c.second == idx) { c.first = std::max(c.first, g->h_la_w[i]); dup = true; }
        if (!dup) cand.push_back({g->h_la_w[i], idx});
      }
      std::stable_sort(cand.begin(), cand.end(), [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first > b.first; });
      static const bool la_dry = getenv("MOEINF_LA_DRY") && atoi(getenv("MOEINF_LA_DRY")) != 0;  // (measurement: the route runs, nothing is issued)
      for (size_t i = 0; i < cand.size() && (int)i < g->la_max && !la_dry; ++i) {
        const Node& n = g->nodes[cand[i].second];
        if (n.slot >= 0 || (!n.host && !n.store)) continue;  // resident / in flight already, or never registered
        g->la_list.push_back(cand[i].second);
      }
      g->la_armed_T = 0;
    }
    CHK(run_experts(g, layer, x_in, st, prof ? pr->ev[2] : nullptr, prof ? pr->ev[3] : nullptr, prof ? pr->ev[4] : nullptr, ld_x, fuse, fused, exp_rows));
  }
  return MOEINF_OK;
}

void make_route_args(const moeinf_engine* g, const void* x_dev, const void* gate_w_dev, int T, RouteArgs& ra) {
  memset(&ra, 0, sizeof ra);
  ra.x = x_dev; ra.gate_w = gate_w_dev; ra.logits = g->d_logits;
  ra.T = T; ra.H = g->H; ra.E = g->E; ra.K = g->K;
  ra.x_dtype = g->dt; ra.gate_dtype = g->cfg.gate_dtype == MOEINF_DTYPE_BF16 ? DT_BF16 : (g->cfg.gate_dtype == MOEINF_DTYPE_F16 ? DT_F16 : DT_F32);
  ra.kind = g->cfg.router_kind; ra.norm_topk_prob = g->cfg.norm_topk_prob; ra.scale = g->cfg.routed_scaling_factor;
  ra.no_renorm = g->route_no_renorm ? 1 : 0;
  ra.v3 = g->route_v3 ? 1 : 0;  // (e_bias: per layer, set by the callers that know the layer)
  ra.n_group = g->cfg.n_group; ra.topk_group = g->cfg.topk_group;
  ra.topk_idx = g->d_topk_idx; ra.topk_w = g->d_topk_w; ra.pair_valid = g->d_pair_valid; ra.pair_order = g->d_pair_order;
  ra.router_prob = g->d_router_prob;
}
void make_index_args(const moeinf_engine* g, int T, int batch_rows, int32_t* mirror, IndexArgs& ia) {
  memset(&ia, 0, sizeof ia);
  ia.topk_idx = g->d_topk_idx; ia.pair_valid = g->d_pair_valid; ia.T = T; ia.K = g->K; ia.E = g->E;
  ia.rows = batch_rows;
  ia.capacity = g->cfg.router_kind == MOEINF_ROUTER_SWITCH ? g->cfg.expert_capacity : 0;
  ia.shared = g->has_shared ? 1 : 0;
  ia.counts = g->d_counts; ia.offsets = g->d_offsets; ia.active = g->d_active; ia.n_active = g->d_n_active;
  ia.pair_slot = g->d_pair_slot; ia.slot_token = g->d_slot_token; ia.slot_pair = g->d_slot_pair; ia.mirror = mirror;
}
// decode-sized DeepSeek forwards: the shared expert (routing-independent, always resident) runs INSIDE the two router launches
bool can_hide_shared(const moeinf_engine* g, int T) {
  static const bool hide_env = getenv("MOEINF_HIDE_SHARED") ? atoi(getenv("MOEINF_HIDE_SHARED")) != 0 : true;
  // (fp16 since round 5: gate_shared1 / route_shared2 / moe_front1 on half_t; fp32 experts keep the shared expert behind the router)
  // (the gate is in the model dtype or fp32: moeinf_create refuses the mixed pairs)
  return hide_env && g->has_shared && (g->dt == DT_BF16 || g->dt == DT_F16) && T <= kHideSharedMaxTokens && T * g->K <= 64 && g->cfg.router_kind == MOEINF_ROUTER_DEEPSEEK;
}
void hidden_shared_stages(const moeinf_engine* g, int layer, const void* x_dev, FfnStage& sh1, FfnStage& sh2) {
  fill_stage(g, layer, 1, sh1, 0);
  sh1.in = x_dev; sh1.row_map = nullptr; sh1.out = g->d_h_sh; sh1.ld_out = g->Fs;
  fill_stage(g, layer, 2, sh2, 0);
  sh2.in = g->d_h_sh; sh2.ld_in = g->Fs; sh2.out = g->d_y_sh; sh2.out_map = nullptr;
}

// MOEINF_STALL_TRACE=<ms>: report (stderr) every forward whose HOST side took longer than <ms>, with the time between
// checkpoints — for hunting one-off stalls in HIP runtime calls (debugging aid, off by default)
struct StallTrace {
  double limit_ms;
  int n = 0;
  const char* name[16];
  std::chrono::steady_clock::time_point t[16];
  int layer, tokens;
  StallTrace(int layer_, int tokens_) : layer(layer_), tokens(tokens_) {
    static const double lim = getenv("MOEINF_STALL_TRACE") ? atof(getenv("MOEINF_STALL_TRACE")) : 0.0;
    limit_ms = lim;
    if (limit_ms > 0) mark("enter");
  }
  void mark(const char* what) {
    if (limit_ms > 0 && n < 16) { name[n] = what; t[n] = std::chrono::steady_clock::now(); ++n; }
  }
  ~StallTrace() {
    if (limit_ms <= 0 || n < 1) return;
    mark("exit");
    const double total = std::chrono::duration<double, std::milli>(t[n - 1] - t[0]).count();
    if (total < limit_ms) return;
    fprintf(stderr, "[moeinf stall] forward layer %d tokens %d: %.2f ms on the host:", layer, tokens, total);
    for (int i = 1; i < n; ++i) fprintf(stderr, " %s +%.2f", name[i], std::chrono::duration<double, std::milli>(t[i] - t[i - 1]).count());
    fprintf(stderr, "\n");
  }
};

extern "C" int moeinf_moe_forward(moeinf_engine* g, int layer, const void* x_dev, int tokens, int batch_rows, const void* gate_w_dev,
                                  void* out_dev, void* stream, uint32_t flags) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  if (layer < 0 || layer >= g->L) return fail(MOEINF_ERR_INVALID, "layer %d out of range", layer);
  if (tokens <= 0 || tokens > g->cfg.max_tokens) return fail(MOEINF_ERR_INVALID, "tokens %d not in 1..max_tokens(%d)", tokens, g->cfg.max_tokens);
  if (!x_dev || !gate_w_dev) return fail(MOEINF_ERR_INVALID, "x_dev/gate_w_dev is NULL");
  if (batch_rows <= 0 || tokens % batch_rows) return fail(MOEINF_ERR_INVALID, "tokens %d not divisible by batch_rows %d", tokens, batch_rows);
  if (g->has_shared && !g->shared_dev[layer]) return fail(MOEINF_ERR_STATE, "shared expert of layer %d not registered", layer);
  const bool route_only = flags & MOEINF_FWD_ROUTE_ONLY;
  if (!route_only && g->cfg.ep_size > 1) return fail(MOEINF_ERR_STATE, "engine is expert-parallel (ep_size %d): run ROUTE_ONLY here and moeinf_ep_pack / ep_expert_ffn / ep_combine around the all-to-alls", g->cfg.ep_size);
  if (!route_only && !(flags & MOEINF_FWD_NO_COMBINE) && !out_dev) return fail(MOEINF_ERR_INVALID, "out_dev is NULL");
  StallTrace strace(layer, tokens);
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  strace.mark("set_device");
  hipStream_t st = (hipStream_t)stream;
  const int T = tokens, K = g->K, E = g->E;

  RouteArgs ra;
  make_route_args(g, x_dev, gate_w_dev, T, ra);
  if (g->route_v3) ra.e_bias = g->gate_bias[layer];
  moeinf_engine::ProfRec pr;
  const bool prof = g->profiling && !route_only;
  if (prof) { for (int i = 0; i < 6; ++i) { pr.ev[i] = get_event(g); if (!pr.ev[i]) return fail(MOEINF_ERR_HIP, "hipEventCreate failed"); } HIPCHK(hipEventRecord(pr.ev[0], st)); }
  MirrorPlan mp;
  if (!route_only) {
    drop_stale_prefetches(g, layer);
    CHK(plan_mirror(g, layer, mp));
  }
  strace.mark("plan_mirror");

  IndexArgs ia;
  make_index_args(g, T, batch_rows, mp.target, ia);
  // decode-sized DeepSeek forwards: the shared expert (routing-independent, always resident) runs INSIDE the two router
  // launches instead of behind them
  const bool hide_shared = !route_only && can_hide_shared(g, T);
  g->last_hidden_shared = hide_shared;
  // batch-1 decode on the sync-free path (gated families, bf16): no top-k/index launch at all — FFN stage 1 routes for
  // itself from the gate logits (ffn1_selfroute_kernel) and one extra block of it writes the routing outputs
  static const bool selfroute_env = getenv("MOEINF_SELFROUTE") ? atoi(getenv("MOEINF_SELFROUTE")) != 0 : true;
  const int et_ = g->cfg.expert_type;
  const bool sr_gated = g->dt != DT_F32 &&
                        (g->cfg.router_kind == MOEINF_ROUTER_MIXTRAL || (g->cfg.router_kind == MOEINF_ROUTER_DEEPSEEK && g->cfg.n_group <= 1 && !g->route_v3)) &&
                        (et_ == MOEINF_EXPERT_MIXTRAL || et_ == MOEINF_EXPERT_DEEPSEEK) && (!g->has_shared || hide_shared);
  // (round 4) Switch: top-1, plain ReLU experts, bf16 or fp32; a single token can never exceed the per-row capacity
  const bool sr_switch = g->cfg.router_kind == MOEINF_ROUTER_SWITCH && et_ == MOEINF_EXPERT_SWITCH && K == 1 && !g->has_shared &&
                         (flags & MOEINF_FWD_NO_COMBINE) == 0 && ia.capacity != 0 && fuse_mode() != 0;
  // (round 4) decode batches of 2..8 tokens of the gated families: the same idea, every workgroup routes every token
  // Measured (profiles/r04_small_batch_selfroute.txt): DeepSeek-V2-Lite batch 2 / 4: 1.530 -> 1.373 / 2.163 -> 2.056 ms per step;
  // batch 8 (48 pairs over 64 experts): 3.17 -> 3.61 — every workgroup of the worst-case grid (48 expert slots) pays eight
  // routings before it knows that its slot is empty.  Hence at most 24 (token, expert) pairs; Mixtral (8 experts, all of them
  // active from batch 4 on) gains 2.3 / 0.9 / 0.6 % at batch 2 / 4 / 8.
  static const int sr_multi_env = getenv("MOEINF_SELFROUTE_MULTI") ? atoi(getenv("MOEINF_SELFROUTE_MULTI")) : 8;
  static const int sr_multi_pairs = getenv("MOEINF_SELFROUTE_MULTI_PAIRS") ? atoi(getenv("MOEINF_SELFROUTE_MULTI_PAIRS")) : 24;
  const bool sr_multi = T >= 2 && T <= std::min(8, sr_multi_env) && T * K <= std::min(64, sr_multi_pairs) && sr_gated;
  const bool selfroute = selfroute_env && !route_only && mp.fast && K <= 8 && E <= 64 && !g->ovr_out &&
                         ((T == 1 && (sr_gated || sr_switch)) || sr_multi);
  g->last_selfroute = selfroute;
  // (the whole DeepSeek layer as ONE persistent launch was built in round 5, measured slower — 1.09 vs 0.958 ms/token — and
  // removed in round 6: DESIGN.md section 4.5.1 keeps the analysis)
  // Switch (top-1, no shared expert): the one-launch form is the DEFAULT — three launches of 3-10 us for 18.9 MB are pure fixed
  // cost, and with hardly any traffic in flight a flag costs ~1 us (MOEINF_LAYER1_SWITCH=0: the three launches)
  static const bool layer1s_env = getenv("MOEINF_LAYER1_SWITCH") ? atoi(getenv("MOEINF_LAYER1_SWITCH")) != 0 : true;
  const bool layer1_switch = layer1s_env && selfroute && T == 1 && sr_switch && !sr_gated && !(flags & MOEINF_FWD_NO_COMBINE) && g->dt != DT_F16;
  g->last_layer1 = layer1_switch;
  // the gated families: the gate (and the hidden shared expert) can ride in FRONT of the self-routing stage 1, in the same
  // launch (round 5, launch_moe_front1).  Measured A/B/A/B (profiles/r05_front1_gate_and_stage1_in_one_launch.txt): DeepSeek-V2-Lite
  // 0.949-0.967 -> 0.937 ms/token (two launches per layer instead of three) = the default with a hidden shared expert; Mixtral
  // 3.708-3.726 -> 3.723-3.728 (nothing: the hop costs what the gate launch cost) = off unless MOEINF_FRONT1=1; =0: never
  static const int front1_env = getenv("MOEINF_FRONT1") ? atoi(getenv("MOEINF_FRONT1")) : -1;
  g->last_front1 = false;
  const bool front1 = (front1_env < 0 ? hide_shared : front1_env != 0) && selfroute && T == 1 && sr_gated && g->dt != DT_F32 &&
                      (hide_shared || !g->has_shared) && (ra.gate_dtype == ra.x_dtype || ra.gate_dtype == DT_F32);
  FfnStage sh1, sh2;
  if (hide_shared) {
    hidden_shared_stages(g, layer, x_dev, sh1, sh2);
    ia.shared = 0;  // the index lists routed experts only
  }
  g->last_front1 = front1;
  if (layer1_switch || front1) {
    // nothing here: dispatch_experts launches the layer / the launch that carries the gate
  } else if (selfroute) {
    if (hide_shared) HIPCHK(launch_gate_shared1(ra, sh1, st));
    else HIPCHK(launch_gate_logits(ra, st));
  } else if (hide_shared) {
    HIPCHK(launch_gate_shared1(ra, sh1, st));
    HIPCHK(launch_route_shared2(ra, ia, sh2, st));
  } else {
    HIPCHK(launch_gate_logits(ra, st));
    if (T <= 64) {
      HIPCHK(launch_route_index(ra, ia, st));  // decode: top-k + dispatch index in one launch
    } else {
      HIPCHK(launch_route_topk(ra, st));
      // long prefills: the index over many workgroups (one workgroup walks 1024-pair chunks serially, ~12 us each)
      static const int wide_pairs = getenv("MOEINF_INDEX_WIDE_PAIRS") ? atoi(getenv("MOEINF_INDEX_WIDE_PAIRS")) : 2048;
      if (ia.capacity <= 0 && (int64_t)T * K > wide_pairs) HIPCHK(launch_dispatch_index_wide(ia, g->d_chunk, st));
      else HIPCHK(launch_dispatch_index(ia, st));
    }
  }
  // next-layer gate lookahead: the decision path waits for this layer's routing anyway — layer l+1's gate over the same
  // rows rides in front of that wait (two small launches; the top-k lands in pinned memory)
  g->la_armed_T = 0;
  if (!route_only && !mp.fast && !g->la_gates.empty() && T <= kLaTokens && layer + 1 < g->L && !g->ovr_out &&
      g->resident_per_layer[layer + 1] < g->owned_experts) {
    RouteArgs la = ra;
    la.gate_w = g->la_gates[layer + 1];
    if (g->route_v3) la.e_bias = g->gate_bias[layer + 1];
    la.logits = g->d_la_f; la.router_prob = g->d_la_f + (size_t)kLaTokens * g->E;
    la.topk_idx = g->h_la_idx; la.topk_w = g->h_la_w;
    la.pair_valid = g->d_la_i; la.pair_order = g->d_la_i + (size_t)kLaTokens * g->K;
    HIPCHK(launch_gate_logits(la, st));
    HIPCHK(launch_route_topk(la, st));
    g->la_armed_T = T;
  }
  g->last_T = T; g->last_layer = layer; g->last_stream = st;
  g->st.forwards += 1;
  strace.mark("router_launches");
  if (route_only) return MOEINF_OK;
  if (prof) HIPCHK(hipEventRecord(pr.ev[1], st));

  const bool want_combine = !(flags & MOEINF_FWD_NO_COMBINE);
  CombineArgs ca;
  memset(&ca, 0, sizeof ca);
  ca.x = x_dev; ca.y = g->d_y; ca.out = out_dev;
  ca.topk_idx = g->d_topk_idx; ca.topk_w = g->d_topk_w; ca.pair_slot = g->d_pair_slot; ca.pair_order = g->d_pair_order;
  ca.router_prob = g->d_router_prob;
  ca.y_shared = g->has_shared ? (hide_shared ? g->d_y_sh : g->d_y) : nullptr;
  ca.shared_offsets = (g->has_shared && !hide_shared) ? g->d_offsets : nullptr;  // shared rows start at offsets[E] (hidden: row 0 of y_shared)
  ca.shared_E = E;
  ca.T = T; ca.H = g->H; ca.K = K; ca.kind = g->cfg.router_kind; ca.dtype = g->dt;
  // decode-sized Mixtral/DeepSeek forwards (every token keeps K experts, so stage 2 always runs): the combine
  // rides in the epilogue of FFN stage 2
  static const bool fuse_combine = getenv("MOEINF_FUSE_COMBINE") ? atoi(getenv("MOEINF_FUSE_COMBINE")) != 0 : true;
  const bool can_fuse = fuse_combine && want_combine && T <= 16 &&
                        (g->cfg.router_kind == MOEINF_ROUTER_MIXTRAL || g->cfg.router_kind == MOEINF_ROUTER_DEEPSEEK ||
                         (selfroute && sr_switch));  // (Switch: only the batch-1 stage 2 knows its combine)
  bool fused = false;
  SelfRoute sr{&ra, &ia, hide_shared ? &sh2 : nullptr, hide_shared ? &sh1 : nullptr, front1, layer1_switch};
  CHK(dispatch_experts(g, layer, x_dev, 0, T, std::min(E, T * K) + ((g->has_shared && !hide_shared) ? 1 : 0),
                       rows_estimate(T, K, E), st, prof, prof ? &pr : nullptr,
                       mp, can_fuse ? &ca : nullptr, &fused, selfroute ? &sr : nullptr));
  strace.mark("dispatch_experts");
  if (want_combine && !fused) HIPCHK(launch_combine(ca, st));
  if (prof) { HIPCHK(hipEventRecord(pr.ev[5], st)); g->prof_pending.push_back(pr); }
  // fence: slots used by this forward may be recycled only after this point of the stream
  CHK(end_forward(g, st, !mp.fast));
  strace.mark("combine_fence");
  return pump_if_pending(g);  // the host is idle until the next layer: serve the speculative queue now
}

extern "C" int moeinf_dispatch_mask(moeinf_engine* g, int layer, const void* x_dev, int tokens, const void* mask_dev, int mask_elem_bytes,
                                    void* y_dev, int32_t* counts_host, int32_t* hit_host, void* stream) {
  return moeinf_dispatch_mask_subset(g, layer, x_dev, tokens, mask_dev, mask_elem_bytes, y_dev, counts_host, hit_host, stream, nullptr, 0);
}

extern "C" int moeinf_dispatch_mask_subset(moeinf_engine* g, int layer, const void* x_dev, int tokens, const void* mask_dev, int mask_elem_bytes,
                                           void* y_dev, int32_t* counts_host, int32_t* hit_host, void* stream, const int32_t* expert_ids, int n_ids) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  if (n_ids < 0 || (n_ids > 0 && !expert_ids)) return fail(MOEINF_ERR_INVALID, "expert_ids is NULL");
  if (layer < 0 || layer >= g->L) return fail(MOEINF_ERR_INVALID, "layer %d out of range", layer);
  if (!x_dev || !mask_dev || !y_dev) return fail(MOEINF_ERR_INVALID, "x_dev/router_mask_dev/y_dev is NULL");
  if (mask_elem_bytes != 1 && mask_elem_bytes != 4 && mask_elem_bytes != 8) return fail(MOEINF_ERR_INVALID, "mask_elem_bytes must be 1, 4 or 8");
  // a token may be routed to up to E experts in a mask; the workspace holds max_tokens*K rows
  if (tokens <= 0 || tokens > g->cfg.max_tokens) return fail(MOEINF_ERR_INVALID, "tokens %d not in 1..max_tokens(%d)", tokens, g->cfg.max_tokens);
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  hipStream_t st = (hipStream_t)stream;
  const int E = g->E;
  drain_mirrors(g, true);
  IndexArgs ia;
  memset(&ia, 0, sizeof ia);
  ia.T = tokens; ia.K = 1; ia.E = E;
  ia.counts = g->d_counts; ia.offsets = g->d_offsets; ia.active = g->d_active; ia.n_active = g->d_n_active;
  ia.pair_slot = g->d_pair_slot; ia.slot_token = g->d_slot_token; ia.slot_pair = g->d_slot_pair; ia.mirror = g->h_mirror;
  ia.slot_cap = g->cfg.max_tokens * g->K;  // rows the workspace holds: the kernel never writes past them
  drop_stale_prefetches(g, layer);
  const uint8_t* keep = nullptr;
  if (expert_ids) {  // only the enqueued experts run (expert_dispatcher.wait_expert: the queue of THIS device): a pinned byte per expert
    if (!g->h_keep) HIPCHK(hipHostMalloc((void**)&g->h_keep, (size_t)E, hipHostMallocDefault));
    memset(g->h_keep, 0, (size_t)E);
    for (int i = 0; i < n_ids; ++i) {
      if (expert_ids[i] < 0 || expert_ids[i] >= E) return fail(MOEINF_ERR_INVALID, "expert id %d out of range", expert_ids[i]);
      g->h_keep[expert_ids[i]] = 1;
    }
    keep = g->h_keep;  // read by the kernel below, which this call waits for before it returns
  }
  HIPCHK(launch_mask_index(mask_dev, mask_elem_bytes, tokens, E, ia, st, keep));
  HIPCHK(hipEventRecord(g->route_ev, st));
  HIPCHK(hipEventSynchronize(g->route_ev));
  int64_t rows = 0;
  for (int e = 0; e < E; ++e) rows += g->h_mirror[1 + e];
  if (rows > (int64_t)g->cfg.max_tokens * g->K) return fail(MOEINF_ERR_INVALID, "mask routes %lld rows but the workspace holds max_tokens*K = %lld", (long long)rows, (long long)g->cfg.max_tokens * g->K);
  if (hit_host) {
    for (int e = 0; e < E; ++e) hit_host[e] = g->h_mirror[1 + e] > 0 ? (g->nodes[node_index(g, layer, e)].slot >= 0 ? 1 : 0) : -1;
  }
  if (counts_host) memcpy(counts_host, g->h_mirror + 1, (size_t)E * sizeof(int32_t));
  // the shared pseudo-expert is not part of a mask dispatch (the reference runs it in Python, deepseek.py:133-136)
  g->la_list.clear();  // (gate-lookahead predictions belong to the fused forward that made them)
  CHK(run_experts(g, layer, x_dev, st, nullptr, nullptr, nullptr));
  if (rows > 0) HIPCHK(hipMemcpyAsync(y_dev, g->d_y, (size_t)rows * g->H * g->es, hipMemcpyDeviceToDevice, st));
  g->last_T = tokens; g->last_layer = layer; g->last_stream = st;
  g->st.forwards += 1;
  CHK(end_forward(g, st, true));
  return pump_if_pending(g);
}

// Standalone combine for callers that keep the reference's PYTHON router: the weighted
// scatter-add loop of the blocks (mixtral.py:96-101, deepseek.py:123-131, switch_transformers.py:99-109,
// nllb_moe.py:84-104) over the expert outputs moeinf_dispatch_mask returned.
extern "C" int moeinf_combine(moeinf_engine* g, const void* x_dev, const void* y_dev, const int32_t* topk_idx_dev, const float* topk_w_dev,
                              const float* router_prob_dev, int tokens, void* out_dev, void* stream) {
  if (!g || !y_dev || !topk_idx_dev || !topk_w_dev || !out_dev) return fail(MOEINF_ERR_INVALID, "NULL argument");
  if (tokens <= 0 || tokens > g->cfg.max_tokens) return fail(MOEINF_ERR_INVALID, "tokens %d not in 1..max_tokens(%d)", tokens, g->cfg.max_tokens);
  const int kind = g->cfg.router_kind;
  if ((kind == MOEINF_ROUTER_SWITCH || kind == MOEINF_ROUTER_NLLB) && !x_dev) return fail(MOEINF_ERR_INVALID, "x_dev is needed for the Switch/NLLB passthrough rules");
  if (kind == MOEINF_ROUTER_SWITCH && !router_prob_dev) return fail(MOEINF_ERR_INVALID, "router_prob_dev is needed for the Switch block");
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  hipStream_t st = (hipStream_t)stream;
  drain_mirrors(g, true);
  const int T = tokens, K = g->K;
  HIPCHK(launch_prep_pairs(topk_idx_dev, topk_w_dev, T, K, g->d_topk_idx, g->d_topk_w, g->d_pair_valid, g->d_pair_order, st));
  // expert-sorted row of every pair = the row order moeinf_dispatch_mask / wait_expert() produce (expert ascending,
  // tokens ascending inside an expert)
  IndexArgs ia;
  make_index_args(g, T, 1, nullptr, ia);
  ia.capacity = 0;  // the caller's router already applied its capacity rule (dropped pairs arrive as -1)
  ia.shared = 0;
  CHK(launch_index_auto(g, ia, st));
  CombineArgs ca;
  memset(&ca, 0, sizeof ca);
  ca.x = x_dev; ca.y = y_dev; ca.out = out_dev;
  ca.topk_idx = g->d_topk_idx; ca.topk_w = g->d_topk_w; ca.pair_slot = g->d_pair_slot; ca.pair_order = g->d_pair_order;
  ca.router_prob = router_prob_dev; ca.y_shared = nullptr; ca.shared_offsets = nullptr; ca.shared_E = g->E;
  ca.T = T; ca.H = g->H; ca.K = K; ca.kind = kind; ca.dtype = g->dt;
  HIPCHK(launch_combine(ca, st));
  g->last_T = T; g->last_stream = st;
  return MOEINF_OK;
}

extern "C" int moeinf_copy_routing_dev(moeinf_engine* g, float* logits_dev, int32_t* topk_idx_dev, float* topk_w_dev, void* stream) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  if (g->last_layer < 0) return fail(MOEINF_ERR_STATE, "no forward has run yet");
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  hipStream_t st = (hipStream_t)stream;
  const size_t T = (size_t)g->last_T;
  if (logits_dev) HIPCHK(hipMemcpyAsync(logits_dev, g->d_logits, T * g->E * 4, hipMemcpyDeviceToDevice, st));
  // capacity-dropped / unrouted pairs are reported as -1, like moeinf_get_routing does on the host
  if (topk_idx_dev) HIPCHK(launch_masked_idx(g->d_topk_idx, g->d_pair_valid, topk_idx_dev, (int)(T * g->K), st));
  if (topk_w_dev) HIPCHK(hipMemcpyAsync(topk_w_dev, g->d_topk_w, T * g->K * 4, hipMemcpyDeviceToDevice, st));
  return MOEINF_OK;
}

extern "C" int moeinf_set_profiling(moeinf_engine* g, int enabled) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  g->profiling = (enabled & 1) != 0;     // per-kernel events of the forwards
  g->ep_profiling = (enabled & 2) != 0;  // per-phase events of moeinf_ep_moe_forward
  return MOEINF_OK;
}

static int check_device_flag(moeinf_engine* g);
extern "C" int moeinf_get_profile(moeinf_engine* g, moeinf_profile* out) {
  if (!g || !out) return fail(MOEINF_ERR_INVALID, "NULL argument");
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  if (g->last_stream || g->last_layer >= 0) HIPCHK(hipStreamSynchronize(g->last_stream));
  CHK(check_device_flag(g));
  drain_mirrors(g, true);
  for (auto& r : g->prof_pending) {
    float ms = 0.f;
    // (an interval whose events were never recorded fails here and is skipped; its error must not stay in the sticky slot)
    if (hipEventElapsedTime(&ms, r.ev[0], r.ev[1]) == hipSuccess) g->prof.route_ms += ms;
    if (hipEventElapsedTime(&ms, r.ev[2], r.ev[3]) == hipSuccess) g->prof.ffn1_ms += ms;
    if (hipEventElapsedTime(&ms, r.k2 ? r.k2 : r.ev[3], r.ev[4]) == hipSuccess) g->prof.ffn2_ms += ms;
    if (hipEventElapsedTime(&ms, r.ev[4], r.ev[5]) == hipSuccess) g->prof.combine_ms += ms;
    (void)hipGetLastError();
    for (int i = 0; i < 6; ++i) g->event_pool.push_back(r.ev[i]);
    if (r.k2) g->event_pool.push_back(r.k2);
  }
  g->prof_pending.clear();
  *out = g->prof;
  memset(&g->prof, 0, sizeof g->prof);
  return MOEINF_OK;
}

// ---- getters -------------------------------------------------------------------------------
// the kernels' error flag (FfnStage::miss_flag): an FFN workgroup found no blob pointer for an active expert.  Read at
// host sync points only (never on the forward path).
static int check_device_flag(moeinf_engine* g) {
  int32_t f = 0;
  HIPCHK(hipMemcpy(&f, g->d_miss, sizeof f, hipMemcpyDeviceToHost));
  if (f == 0) return MOEINF_OK;
  HIPCHK(hipMemset(g->d_miss, 0, sizeof f));
  if (g->ep_err_host) *g->ep_err_host = 0;
  if (g->h_miss) *g->h_miss = 0;
  if (f == 4) {  // the counters of the one-launch layer may be out of step now: start them again
    HIPCHK(hipMemset(g->d_layer_ctr, 0, LAYER1_CTRS * LAYER1_CTR_STRIDE * sizeof(uint32_t)));
    HIPCHK(hipMemset(g->d_arrive, 0, (size_t)((g->H + 15) / 16) * sizeof(int32_t)));
    g->layer1_launches = 0;
    return fail(MOEINF_ERR_STATE, "device error flag 4: a workgroup of the one-launch decode layer gave up waiting for another one (MOEINF_LAYER1_TIMEOUT_MS), results of the last forwards are invalid");
  }
  if (f == 3) return fail(MOEINF_ERR_STATE, "device error flag 3: a kernel of the peer-store exchange found another rank AHEAD of this one (an earlier call failed on one side), results of the last forwards are invalid");
  if (f == 2) return fail(MOEINF_ERR_STATE, "device error flag 2: a kernel of the peer-store exchange gave up waiting for another rank's rows (MOEINF_EP_PEER_TIMEOUT_MS), results of the last forwards are invalid");
  return fail(MOEINF_ERR_STATE, "device error flag %d: an FFN workgroup found no resident blob for an active expert, results of the last forwards are invalid", f);
}

static int sync_last(moeinf_engine* g) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  if (g->last_layer < 0) return fail(MOEINF_ERR_STATE, "no forward has run yet");
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  HIPCHK(hipStreamSynchronize(g->last_stream));
  return check_device_flag(g);
}

extern "C" int moeinf_get_routing(moeinf_engine* g, int32_t* topk_idx, float* topk_w, int32_t* counts, int32_t* offsets, int32_t* slot_token, int32_t* pair_slot) {
  CHK(sync_last(g));
  const size_t n = (size_t)g->last_T * g->K;
  std::vector<int32_t> valid(n), idx(n);
  HIPCHK(hipMemcpy(idx.data(), g->d_topk_idx, n * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(valid.data(), g->d_pair_valid, n * 4, hipMemcpyDeviceToHost));
  if (topk_idx) {
    for (size_t i = 0; i < n; ++i) topk_idx[i] = valid[i] ? idx[i] : -1;
  }
  if (topk_w) HIPCHK(hipMemcpy(topk_w, g->d_topk_w, n * 4, hipMemcpyDeviceToHost));
  if (counts) HIPCHK(hipMemcpy(counts, g->d_counts, (size_t)g->E * 4, hipMemcpyDeviceToHost));
  if (offsets) HIPCHK(hipMemcpy(offsets, g->d_offsets, (size_t)(g->E + 1) * 4, hipMemcpyDeviceToHost));
  if (slot_token) HIPCHK(hipMemcpy(slot_token, g->d_slot_token, n * 4, hipMemcpyDeviceToHost));
  if (pair_slot) HIPCHK(hipMemcpy(pair_slot, g->d_pair_slot, n * 4, hipMemcpyDeviceToHost));
  return MOEINF_OK;
}

extern "C" int moeinf_get_expert_outputs(moeinf_engine* g, void* host_out, int64_t nbytes) {
  CHK(sync_last(g));
  const int64_t rows = (int64_t)g->last_T * g->K + (g->has_shared ? g->last_T : 0);
  const int64_t need = rows * g->H * g->es;
  if (!host_out || nbytes > need || nbytes <= 0) return fail(MOEINF_ERR_INVALID, "nbytes must be in 1..%lld", (long long)need);
  if (g->has_shared && g->last_hidden_shared) {
    // routed rows sit in y, the shared expert's rows (computed inside the router launches) in their own buffer
    int32_t routed = 0;
    HIPCHK(hipMemcpy(&routed, g->d_offsets + g->E, sizeof routed, hipMemcpyDeviceToHost));
    const int64_t rb = std::min<int64_t>(nbytes, (int64_t)routed * g->H * g->es);
    if (rb > 0) HIPCHK(hipMemcpy(host_out, g->d_y, (size_t)rb, hipMemcpyDeviceToHost));
    if (nbytes > rb) HIPCHK(hipMemcpy((char*)host_out + rb, g->d_y_sh, (size_t)std::min<int64_t>(nbytes - rb, (int64_t)g->last_T * g->H * g->es), hipMemcpyDeviceToHost));
    return MOEINF_OK;
  }
  HIPCHK(hipMemcpy(host_out, g->d_y, (size_t)nbytes, hipMemcpyDeviceToHost));
  return MOEINF_OK;
}

extern "C" int moeinf_get_logits(moeinf_engine* g, float* host_out, int64_t n_floats) {
  CHK(sync_last(g));
  if (!host_out || n_floats != (int64_t)g->last_T * g->E) return fail(MOEINF_ERR_INVALID, "n_floats must be tokens*E");
  HIPCHK(hipMemcpy(host_out, g->d_logits, (size_t)n_floats * 4, hipMemcpyDeviceToHost));
  return MOEINF_OK;
}

// ---- prefetch / cache control ----------------------------------------------------------------
// Serve the pending-transfer queue: keep at most `prefetch_window` speculative copies in flight on the low-priority
// lane (reference: one worker thread per GPU popping ArcherTaskPool's queue, task_scheduler.cpp:451-517).  Called
// from every entry point that may have freed the lane or added work; never blocks.
static int pump_prefetch(moeinf_engine* g) {
  for (size_t i = 0; i < g->stale_disk.size();) {  // disk reads that outlived their task: adopt the blob once it has landed
    const int di = g->stale_disk[i];
    Node& dn = g->nodes[di];
    if (dn.host_pending && !host_read_done(dn)) { ++i; continue; }
    if (dn.host_pending) CHK(finish_host_read(g, di));
    g->stale_disk.erase(g->stale_disk.begin() + (long)i);
  }
  while (!g->prefetch_inflight.empty()) {  // retire finished copies, oldest first
    const int idx = g->prefetch_inflight.front();
    Node& n = g->nodes[idx];
    if (n.slot >= 0 && !n.ready_waited && n.ready && hipEventQuery(n.ready) != hipSuccess) { (void)hipGetLastError(); break; }
    g->prefetch_inflight.pop_front();
  }
  while (!g->demand_inflight.empty()) {  // retire on-demand copies that have landed
    const Node& dn = g->nodes[g->demand_inflight.front()];
    if (dn.slot >= 0 && dn.ready && hipEventQuery(dn.ready) != hipSuccess) { (void)hipGetLastError(); break; }
    g->demand_inflight.pop_front();
  }
  while ((int)g->prefetch_inflight.size() < g->prefetch_window) {
    // an on-demand copy is on the link: a speculative copy started now would take half of its bandwidth away — the
    // queue keeps its tasks (they go stale with the layer counter if the pass moves on), the next pump tries again
    if (!g->demand_inflight.empty() && !g->draining) break;
    QueuedTask t;
    bool have = false;
    // speculative tasks whose host blob has arrived from disk go first (oldest first); unfinished ones stay parked
    for (auto it = g->disk_inflight.begin(); it != g->disk_inflight.end(); ++it) {
      Node& dn = g->nodes[(int)it->node];
      if (dn.host) { t = *it; g->disk_inflight.erase(it); have = true; break; }  // adopted by a demand meanwhile
      if (dn.host_pending && host_read_done(dn)) {
        t = *it;
        g->disk_inflight.erase(it);
        CHK(finish_host_read(g, (int)t.node));
        have = true;
        break;
      }
    }
    if (!have && !g->pq.pop(&t)) break;
    const int idx = (int)t.node;
    Node& nd = g->nodes[idx];
    if (nd.slot >= 0) continue;  // became resident (demand fetch) while it waited
    if (!nd.host && nd.store && !nd.host_pending) {
      // the blob is on disk only: read it in the background at LOW priority (the reference's prefetch thread does the
      // disk leg at low priority too, archer_prio_aio_handle.cpp:150-166) instead of stalling this call on a pread;
      // the H2D copy is issued by a later pump once the blob has landed
      if ((int)g->disk_inflight.size() >= g->disk_window) { g->pq.enqueue(t.node, t.layer, t.priority); break; }
      void* blk = nullptr;
      const int arc = alloc_host_block(g, idx, &blk, /*may_block=*/false);  // the pump never waits: no block free right now = the request is dropped
      if (arc == MOEINF_ERR_OOM) { g->st.prefetch_dropped += 1; continue; }
      if (arc != MOEINF_OK) return arc;
      const int src = submit_host_read(g, idx, blk, /*high=*/false);
      if (src != MOEINF_OK) { g->host_free.push_back(blk); return src; }
      g->disk_inflight.push_back(t);
      g->st.disk_reads_async += 1;
      continue;
    }
    if (nd.host_pending) {  // its disk read is under way: parked in disk_inflight by an earlier request, or stale — then this request re-parks it
      auto sit = std::find(g->stale_disk.begin(), g->stale_disk.end(), idx);
      if (sit != g->stale_disk.end()) { g->stale_disk.erase(sit); g->disk_inflight.push_back(t); }
      continue;
    }
    // speculation governor: once enough speculative copies have finished and too few of them were ever dispatched,
    // stop issuing — all but one probe in `gov_probe_every`, so that a workload whose predictions become good again
    // is noticed
    if (g->gov_min_useful > 0.f && g->gov_outcomes >= 8 && g->gov_score < g->gov_min_useful && (g->gov_skipped++ % g->gov_probe_every) != 0) {
      g->st.prefetch_throttled += 1;
      continue;
    }
    // a speculative copy never evicts the protected set (candidates_, task_scheduler.cpp:292-297) nor an expert of
    // the layer being dispatched; if nothing can be freed the task is dropped ("evict failed", :505-510)
    g->pol[idx].pinned = true;
    const int rc = issue_copy(g, idx, g->prefetch, false);
    g->pol[idx].pinned = false;
    if (rc == MOEINF_ERR_OOM) { g->st.prefetch_dropped += 1; continue; }
    if (rc != MOEINF_OK) return rc;
    nd.prefetched = true;
    nd.prefetch_cnt += 1;
    g->st.prefetch_issued += 1;
    g->prefetch_inflight.push_back(idx);
  }
  g->st.prefetch_queued = (int64_t)g->pq.size();
  return MOEINF_OK;
}

extern "C" int moeinf_prefetch(moeinf_engine* g, int layer, const int32_t* experts, const float* scores, int n) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  drain_mirrors(g, true);
  if (layer < 0 || layer >= g->L) return fail(MOEINF_ERR_INVALID, "layer out of range");
  if (n < 0 || (n > 0 && !experts)) return fail(MOEINF_ERR_INVALID, "experts is NULL");
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  for (int i = 0; i < n; ++i) {
    const int e = experts[i];
    if (e < 0 || e >= g->E) return fail(MOEINF_ERR_INVALID, "expert id %d out of range", e);
    if (!owns(g, e)) continue;
    const int idx = node_index(g, layer, e);
    Node& nd = g->nodes[idx];
    if (nd.slot >= 0) continue;  // resident or already in flight (src == dst device: EnqueueTask returns, :105-110)
    if (!nd.host && !nd.store) return fail(MOEINF_ERR_STATE, "expert (%d,%d) not registered", layer, e);
    // EnqueueTask: same-node tasks of equal or lower urgency are displaced (dedup / upgrade)
    g->st.prefetch_cancelled += g->pq.enqueue(idx, layer, priority_from_score(scores, i));
  }
  return pump_prefetch(g);
}

extern "C" int moeinf_set_gate_bias(moeinf_engine* g, int layer, const float* bias_dev) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  if (!g->route_v3) return fail(MOEINF_ERR_STATE, "moeinf_set_gate_bias: the engine's router is not MOEINF_ROUTER_DEEPSEEK_V3");
  if (layer < 0 || layer >= g->L) return fail(MOEINF_ERR_INVALID, "layer %d out of range", layer);
  g->gate_bias[layer] = bias_dev;  // borrowed; NULL = zeros
  return MOEINF_OK;
}

extern "C" int moeinf_set_lookahead(moeinf_engine* g, const void* const* gate_w_dev, int n_layers, int max_experts) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  if (!gate_w_dev || n_layers == 0) { g->la_gates.clear(); g->la_armed_T = 0; g->la_list.clear(); return MOEINF_OK; }
  if (n_layers != g->L) return fail(MOEINF_ERR_INVALID, "moeinf_set_lookahead: %d gate pointers for %d layers", n_layers, g->L);
  if (max_experts <= 0) return fail(MOEINF_ERR_INVALID, "moeinf_set_lookahead: max_experts must be positive");
  for (int l = 0; l < n_layers; ++l) if (!gate_w_dev[l]) return fail(MOEINF_ERR_INVALID, "moeinf_set_lookahead: gate of layer %d is NULL", l);
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  if (!g->d_la_f) {
    HIPCHK(hipMalloc((void**)&g->d_la_f, (size_t)kLaTokens * (g->E + 1) * sizeof(float)));
    HIPCHK(hipMalloc((void**)&g->d_la_i, (size_t)2 * kLaTokens * g->K * sizeof(int32_t)));
    HIPCHK(hipHostMalloc((void**)&g->h_la_idx, (size_t)kLaTokens * g->K * sizeof(int32_t), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&g->h_la_w, (size_t)kLaTokens * g->K * sizeof(float), hipHostMallocDefault));
  }
  g->la_gates.assign(gate_w_dev, gate_w_dev + n_layers);
  g->la_max = max_experts;
  return MOEINF_OK;
}

extern "C" int moeinf_protect(moeinf_engine* g, const int32_t* layers, const int32_t* experts, int n) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  if (n < 0 || (n > 0 && (!layers || !experts))) return fail(MOEINF_ERR_INVALID, "NULL id arrays");
  for (int i = 0; i < n; ++i) CHK(check_le(g, layers[i], experts[i]));
  for (auto& p : g->pol) p.is_protected = false;
  for (int i = 0; i < n; ++i) g->pol[node_index(g, layers[i], experts[i])].is_protected = true;
  // the new candidate set also empties the speculative levels of the queue (task_scheduler.h:66-79)
  g->st.prefetch_cancelled += g->pq.clear_prefetch();
  g->st.prefetch_queued = 0;
  return MOEINF_OK;
}

extern "C" int moeinf_clear_cache_counts(moeinf_engine* g) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  drain_mirrors(g, true);
  for (auto& p : g->pol) p.incache = 0;
  return MOEINF_OK;
}

extern "C" int moeinf_is_resident(moeinf_engine* g, int layer, int expert, int32_t* resident) {
  CHK(check_le(g, layer, expert));
  if (!resident) return fail(MOEINF_ERR_INVALID, "resident is NULL");
  *resident = g->nodes[node_index(g, layer, expert)].slot >= 0 ? 1 : 0;
  return MOEINF_OK;
}

extern "C" int moeinf_sync_copies(moeinf_engine* g) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  g->draining = true;
  struct Undrain { moeinf_engine* g; ~Undrain() { g->draining = false; } } undrain{g};
  for (;;) {  // serve the whole pending queue, a window at a time
    CHK(pump_prefetch(g));
    for (CopyLane* ln : {&g->demand, &g->prefetch}) {
      HIPCHK(hipStreamSynchronize(ln->copy));
      HIPCHK(hipStreamSynchronize(ln->retile));
    }
    if (g->pq.empty() && g->disk_inflight.empty()) break;
    // speculative blobs still on their way from disk: wait for the oldest, the next pump issues its H2D copy
    if (!g->disk_inflight.empty()) { Node& dn = g->nodes[(int)g->disk_inflight.front().node]; for (auto& h : dn.disk_reqs) PrioAioPool::wait(h); }
  }
  for (int di : g->stale_disk) { Node& dn = g->nodes[di]; for (auto& h : dn.disk_reqs) PrioAioPool::wait(h); }
  CHK(pump_prefetch(g));
  settle_copy_timers(g, true);
  return MOEINF_OK;
}

extern "C" int moeinf_get_expert_counters(moeinf_engine* g, int64_t* out, int64_t n_int64) {
  if (!g || !out) return fail(MOEINF_ERR_INVALID, "NULL argument");
  drain_mirrors(g, true);
  if (n_int64 != (int64_t)g->L * g->E * 7) return fail(MOEINF_ERR_INVALID, "n_int64 must be L*E*7");
  for (int l = 0; l < g->L; ++l)
    for (int e = 0; e < g->E; ++e) {
      const int idx = node_index(g, l, e);
      int64_t* o = out + ((int64_t)l * g->E + e) * 7;
      o[0] = g->nodes[idx].visit; o[1] = g->nodes[idx].hit; o[2] = g->nodes[idx].miss; o[3] = g->nodes[idx].prefetch_cnt;
      o[4] = g->pol[idx].incache; o[5] = g->nodes[idx].slot >= 0 ? 1 : 0; o[6] = g->nodes[idx].unused;
    }
  return MOEINF_OK;
}

extern "C" int moeinf_get_stats(moeinf_engine* g, moeinf_stats* out) {
  if (!g || !out) return fail(MOEINF_ERR_INVALID, "NULL argument");
  drain_mirrors(g, true);
  settle_copy_timers(g, false);
  CHK(pump_if_pending(g));
  g->st.prefetch_queued = (int64_t)g->pq.size();
  *out = g->st;
  return MOEINF_OK;
}

extern "C" int moeinf_reset_stats(moeinf_engine* g) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  drain_mirrors(g, true);
  settle_copy_timers(g, false);
  const int64_t st = g->st.slots_total, su = g->st.slots_used, sb = g->st.slot_bytes, ha = g->st.host_arena_bytes;
  memset(&g->st, 0, sizeof g->st);
  g->st.slots_total = st; g->st.slots_used = su; g->st.slot_bytes = sb; g->st.host_arena_bytes = ha;
  g->st.prefetch_queued = (int64_t)g->pq.size();
  return MOEINF_OK;
}

extern "C" int moeinf_set_prefetch_governor(moeinf_engine* g, float min_useful_fraction, int probe_every) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  if (!(min_useful_fraction >= 0.f && min_useful_fraction <= 1.f) || probe_every < 1) return fail(MOEINF_ERR_INVALID, "min_useful_fraction must be in [0,1], probe_every >= 1");
  g->gov_min_useful = min_useful_fraction;
  g->gov_probe_every = probe_every;
  g->gov_score = 1.f; g->gov_outcomes = 0; g->gov_skipped = 0;
  return MOEINF_OK;
}

// DeviceMemoryPool::SetMemoryRatio (core/memory/memory_pool.cpp:150-158) at run time, in bytes: shrink or grow the
// expert cache.  Shrinking evicts by the replacement policy until the resident set fits and returns the freed slots'
// memory to the device; the host copies are authoritative, nothing is written back.
extern "C" int moeinf_set_cache_policy(moeinf_engine* g, int policy) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  if (policy != MOEINF_POLICY_LFU_INCACHE && policy != MOEINF_POLICY_LRU) return fail(MOEINF_ERR_INVALID, "policy must be MOEINF_POLICY_LFU_INCACHE or MOEINF_POLICY_LRU");
  g->cfg.policy = policy;
  return MOEINF_OK;
}

extern "C" int moeinf_set_cache_budget(moeinf_engine* g, int64_t device_memory_bytes) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  if (device_memory_bytes < g->slot_bytes) return fail(MOEINF_ERR_OOM, "budget %lld bytes cannot hold one expert of %lld bytes", (long long)device_memory_bytes, (long long)g->slot_bytes);
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  drain_mirrors(g, true);
  g->st.prefetch_cancelled += g->pq.clear_prefetch();
  HIPCHK(hipDeviceSynchronize());  // no kernel or copy may still touch a slot that is about to be freed
  g->prefetch_inflight.clear();
  for (int l = 0; l < g->L; ++l) settle_ready(g, l);
  const int64_t new_max = std::min<int64_t>(device_memory_bytes / g->slot_bytes, (int64_t)g->owned_experts * g->L);
  while (g->st.slots_used > new_max) {
    const int64_t v = pick_victim(g->pol.data(), (int64_t)g->pol.size(), g->cfg.policy, true);
    if (v < 0) return fail(MOEINF_ERR_STATE, "cannot shrink the cache: nothing evictable");
    Node& vn = g->nodes[v];
    drop_ready_count(g, (int)v);
    g->free_slots.push_back(vn.slot);
    g->slots[vn.slot].node = -1;
    g->slots[vn.slot].last_use_seq = 0;
    vn.slot = -1; vn.prefetched = false;
    g->pol[v].resident = false;
    g->st.evictions += 1;
    g->st.slots_used -= 1;
    queue_poke(g, (int)(v % g->L), (int)(v / g->L), 0);
  }
  // free slots beyond the new capacity give their memory back
  while ((int64_t)g->slots.size() > new_max && !g->free_slots.empty()) {
    // compact: move the last slot's tenant bookkeeping is not needed — only FREE slots are released, and only from
    // the tail of the slot array so indices of live slots stay valid
    const int last = (int)g->slots.size() - 1;
    auto it = std::find(g->free_slots.begin(), g->free_slots.end(), last);
    if (it == g->free_slots.end()) {
      // the tail slot is live: move its tenant into a free slot with a lower index (device-to-device copy)
      const int dst = *std::min_element(g->free_slots.begin(), g->free_slots.end());
      Slot& from = g->slots[last];
      Slot& to = g->slots[dst];
      HIPCHK(hipMemcpy(to.dev, from.dev, (size_t)g->slot_bytes, hipMemcpyDeviceToDevice));
      to.node = from.node; to.last_use_seq = from.last_use_seq;
      g->nodes[to.node].slot = dst;
      queue_poke(g, to.node % g->L, to.node / g->L, (uint64_t)to.dev);
      from.node = -1;
      g->free_slots.erase(std::find(g->free_slots.begin(), g->free_slots.end(), dst));
      g->free_slots.push_back(last);
      it = std::find(g->free_slots.begin(), g->free_slots.end(), last);
    }
    g->free_slots.erase(it);
    HIPCHK(hipFree(g->slots[last].dev));
    g->slots.pop_back();
  }
  CHK(flush_pokes(g, nullptr));
  HIPCHK(hipDeviceSynchronize());
  g->max_slots = new_max;
  g->slab_exhausted = false;
  g->st.slots_total = new_max;
  prealloc_slots(g);  // a grown budget gets its slots here, not on the misses that will use them
  return MOEINF_OK;
}

// Grow the token-sized workspace (the reference's dispatcher has no such limit: it allocates per call).  Only grows;
// synchronises the device.
extern "C" int moeinf_reserve_tokens(moeinf_engine* g, int max_tokens) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  if (max_tokens <= g->cfg.max_tokens) return MOEINF_OK;
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  drain_mirrors(g, true);
  HIPCHK(hipDeviceSynchronize());
  free_token_workspace(g);
  const int rc = alloc_token_workspace(g, max_tokens);
  if (rc != MOEINF_OK) {  // fall back to the old size so the engine stays usable
    free_token_workspace(g);
    const std::string keep = g_err;
    if (alloc_token_workspace(g, g->cfg.max_tokens) != MOEINF_OK) return fail(MOEINF_ERR_OOM, "workspace lost: %s", keep.c_str());
    g_err = keep;
    return rc;
  }
  g->cfg.max_tokens = max_tokens;
  g->last_layer = -1;  // routing results of the previous forward lived in the old buffers
  // the expert-parallel workspace (keys, pair positions, slot maps) is sized by max_tokens*K as well: drop it, the next
  // ep_pack re-allocates it for the new size (ep_alloc)
  free_ep_workspace(g);
  return MOEINF_OK;
}


int launch_index_auto(moeinf_engine* g, const IndexArgs& ia, hipStream_t st) {
  if (ia.capacity <= 0 && (int64_t)ia.T * ia.K > 2048) HIPCHK(launch_dispatch_index_wide(ia, g->d_chunk, st));
  else HIPCHK(launch_dispatch_index(ia, st));
  return MOEINF_OK;
}

extern "C" int moeinf_register_expert_from_store(moeinf_engine* g, int layer, int expert, const moeinf_store* st, const uint32_t* tensor_ids, int n) {
  CHK(check_le(g, layer, expert));
  if (!st || !tensor_ids) return fail(MOEINF_ERR_INVALID, "NULL argument");
  if (!owns(g, expert)) return fail(MOEINF_ERR_INVALID, "expert %d is not owned by ep_rank %d of %d", expert, g->cfg.ep_rank, g->cfg.ep_size);
  if (n != g->lay.n) return fail(MOEINF_ERR_INVALID, "expert type %d has %d tensors, got %d ids", g->cfg.expert_type, g->lay.n, n);
  for (int i = 0; i < n; ++i) {
    const TensorMeta* m = st->s.find(tensor_ids[i]);
    if (!m) return fail(MOEINF_ERR_INVALID, "tensor %u is not in the offload index", tensor_ids[i]);
    if ((int64_t)m->size != g->lay.size[i]) return fail(MOEINF_ERR_INVALID, "tensor %u is %llu bytes on disk, blob slot %d needs %lld", tensor_ids[i], (unsigned long long)m->size, i, (long long)g->lay.size[i]);
  }
  DeviceScope on_dev_(g->cfg.device_id); HIPCHK(on_dev_.err);
  const int idx = node_index(g, layer, expert);
  Node& nd = g->nodes[idx];
  set_node_store(nd, &st->s);  // counted: moeinf_store_close refuses while an engine may still re-read this expert
  for (int i = 0; i < n; ++i) nd.store_ids[i] = tensor_ids[i];
  // disk -> pinned host now while the arena has room; once the cap is reached the expert stays on disk and is read on
  // its first miss (the host tier then works as an LRU cache over the offload directory)
  const bool room = g->cfg.host_memory_bytes <= 0 || !g->host_free.empty() || g->arena_total + g->lay.total <= g->cfg.host_memory_bytes;
  if (nd.host_pending) CHK(ensure_host(g, idx));  // a background read of the previous registration: finish it first
  if (nd.host) {
    // re-registration: the directory's payload replaces whatever the host blob held (and any resident copy of it)
    CHK(invalidate_resident(g, idx));
    void* blk = nd.host;
    nd.host = nullptr;
    const int rc = submit_host_read(g, idx, blk, /*high=*/true);
    if (rc != MOEINF_OK) { g->host_free.push_back(blk); return rc; }
    CHK(finish_host_read(g, idx));
  } else if (room) {
    CHK(ensure_host(g, idx));
  }
  return MOEINF_OK;
}


extern "C" int moeinf_set_predictor(moeinf_engine* g, moeinf_tracer* tr, int64_t seq_id, int lookahead_layers, float min_share, int max_experts) {
  if (!g) return fail(MOEINF_ERR_INVALID, "engine is NULL");
  drain_mirrors(g, true);  // mirrors of earlier forwards belong to the previous sequence
  if (!tr || seq_id < 0) { g->pred_tracer = nullptr; g->pred_seq = -1; return MOEINF_OK; }
  if (tr->t->layers() != g->L || tr->t->experts() != g->E) return fail(MOEINF_ERR_INVALID, "tracer is [%d,%d], engine is [%d,%d]", tr->t->layers(), tr->t->experts(), g->L, g->E);
  if (!tr->t->has(seq_id)) return fail(MOEINF_ERR_INVALID, "unknown seq_id");
  if (lookahead_layers < 0 || max_experts < 0 || !(min_share >= 0.f && min_share <= 1.f)) return fail(MOEINF_ERR_INVALID, "bad predictor options");
  g->pred_tracer = tr->t; g->pred_seq = seq_id; g->pred_lookahead = lookahead_layers; g->pred_min_share = min_share; g->pred_max = max_experts;
  return MOEINF_OK;
}

