// engine_internal.h — what the host-side translation units of libmoeinf_hip.so share: the engine object, its tiers' records,
// the error slot, and the handful of hot-path helpers the expert-parallel unit calls.  Round 5 split of the former single
// 3 000-line engine.cpp (no behaviour change):
//   engine.cpp       lifecycle, registration, memory tiers, the local hot path, cache control, getters
//   engine_ep.cpp    expert parallelism: pack / owner FFN / combine steps, RCCL transport, direct peer-store exchange
//   capi_host.cpp    the host-only handles of the C ABI (offload store, cache simulator, pending-transfer queue, block reader, tracer)
//
// The host side of libmoeinf_hip.so: memory tiers, residency, the per-layer hot path orchestration, and the C ABI
// (include/moeinf.h).
//
// Replaces (reference, /root/reference): core/parallel/expert_dispatcher.cpp (ExpertDispatcher),
// core/model/model_topology.cpp Node::SetDevice (tier mover), core/memory/* (pools, allocators,
// streams), core/prefetch/task_scheduler.cpp (prefetch queue + eviction) — with a different
// architecture: no worker threads, no per-expert stream syncs.  Every H2D copy is a
// hipMemcpyAsync from the pinned arena into a fixed-size HBM slot on a dedicated copy stream and
// is ordered against the compute stream with events (hipStreamWaitEvent), so the host never blocks
// on a copy and a slot is never recycled while a kernel may still read it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <deque>
#include <string>
#include <vector>

#include "../../include/moeinf.h"
#include "cache_policy.h"
#include "kernels.h"
#include <map>
#include <mutex>
#include "aio_pool.h"
#include "ep_comm.h"
#include "ep_peer.h"
#include "offload_store.h"
#include "prefetch_queue.h"
#include "tracer.h"

using namespace moeinf;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
std::string& moeinf_err_slot();  // the calling thread's last error text (engine.cpp; moeinf_last_error returns it)
static inline int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  moeinf_err_slot() = buf;
  return code;
}
#define g_err (moeinf_err_slot())
#define HIPCHK(call)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) return fail(MOEINF_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
// Every C entry point works on its engine's device and leaves the calling thread's current device as it found it: a
// prefetch_handle drives one engine per device from ONE Python thread, and torch's notion of the current device must
// not move under it (device='cuda' allocations, torch.cuda.current_stream()).  hipGetDevice is a thread-local read.
struct DeviceScope {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceScope(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) err = hipSetDevice(dev);
    else prev = -1;  // nothing to restore
  }
  ~DeviceScope() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};
#define CHK(call)              \
  do {                         \
    int r_ = (call);           \
    if (r_ != MOEINF_OK) return r_; \
  } while (0)

static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
static constexpr int64_t kAioAlignment = 4096;  // core/aio/archer_aio_utils.h kAioAlignment (model_topology.cpp:429-431)

// ------------------------------------------------------------------------------------------------
// blob layout
// ------------------------------------------------------------------------------------------------
struct BlobLayout {
  int n = 0;
  int64_t off[4] = {0, 0, 0, 0}, size[4] = {0, 0, 0, 0};
  int64_t total = 0;
};
static BlobLayout make_layout(int expert_type, int64_t H, int64_t F, int64_t es) {
  BlobLayout b;
  auto add = [&](int64_t bytes) {
    b.off[b.n] = b.total;
    b.size[b.n] = bytes;
    b.total += align_up(bytes, kAioAlignment);
    ++b.n;
  };
  switch (expert_type) {
    case MOEINF_EXPERT_MIXTRAL:   // w1[F,H] w2[H,F] w3[F,H]
    case MOEINF_EXPERT_DEEPSEEK:  // gate[F,H] up[F,H] down[H,F]
    case MOEINF_EXPERT_SWITCH_GATED:  // wi_0[F,H] wi_1[F,H] wo[H,F] (expert_module.cpp:46-52)
      add(F * H * es); add(F * H * es); add(F * H * es);
      break;
    case MOEINF_EXPERT_NLLB:
    case MOEINF_EXPERT_FSGPT:  // fc1.w fc1.b fc2.w fc2.b
      add(F * H * es); add(F * es); add(H * F * es); add(H * es);
      break;
    case MOEINF_EXPERT_SWITCH:  // wi wo
      add(F * H * es); add(H * F * es);
      break;
    default: break;
  }
  return b;
}

// Device-side (HBM slot) layout: matrices in MFMA-tile order (kernels.hip), biases raw; 4 KiB aligned.
struct DevLayout {
  int n = 0;
  int64_t off[4] = {0, 0, 0, 0}, size[4] = {0, 0, 0, 0};
  int R[4] = {0, 0, 0, 0}, K[4] = {0, 0, 0, 0};  // K == 0: not a matrix (bias vector, copied as is)
  int64_t total = 0;
};
static DevLayout make_dev_layout(int expert_type, int64_t H, int64_t F, int dt, int64_t es) {
  DevLayout d;
  auto mat = [&](int64_t R, int64_t K) {
    d.off[d.n] = d.total; d.R[d.n] = (int)R; d.K[d.n] = (int)K; d.size[d.n] = tiled_bytes(R, K, dt);
    d.total += align_up(d.size[d.n], kAioAlignment); ++d.n;
  };
  auto vec = [&](int64_t n) {
    d.off[d.n] = d.total; d.R[d.n] = (int)n; d.K[d.n] = 0; d.size[d.n] = n * es;
    d.total += align_up(d.size[d.n], kAioAlignment); ++d.n;
  };
  switch (expert_type) {
    case MOEINF_EXPERT_MIXTRAL: mat(F, H); mat(H, F); mat(F, H); break;
    case MOEINF_EXPERT_DEEPSEEK: case MOEINF_EXPERT_SWITCH_GATED: mat(F, H); mat(F, H); mat(H, F); break;
    case MOEINF_EXPERT_NLLB: case MOEINF_EXPERT_FSGPT: mat(F, H); vec(F); mat(H, F); vec(H); break;
    case MOEINF_EXPERT_SWITCH: mat(F, H); mat(H, F); break;
    default: break;
  }
  return d;
}

// ------------------------------------------------------------------------------------------------
// engine
// ------------------------------------------------------------------------------------------------
struct Node {
  void* host = nullptr;        // pinned arena blob (reference layout); nullptr + store != nullptr: on disk only
  int slot = -1;
  hipEvent_t ready1 = nullptr; // recorded once the tensors FFN stage 1 reads are in the slot
  uint64_t copy_seq = 0;        // issue number of the node's latest transfer (per engine, grows)
  int8_t copy_lane = 0;         // 0 demand, 1 speculative
  bool ready1_is_ready = false; // the whole blob arrives with one launch: `ready` serves for both (one event record less per copy)
  hipEvent_t ready = nullptr;  // recorded once the whole expert is in the slot
  bool waited1 = true;         // compute stream already ordered after `ready1`
  bool ready_waited = true;    // compute stream already ordered after `ready`
  bool copy_inflight = false;  // an H2D transfer out of `host` was issued and has not been OBSERVED complete yet
  bool prefetched = false;     // resident because of a prefetch, not yet dispatched
  int64_t visit = 0, hit = 0, miss = 0, prefetch_cnt = 0;
  int64_t unused = 0;          // evicted after a speculative copy that no dispatch ever used (Node::unused_count, task_scheduler.cpp:304)
  // disk tier (register_expert_from_store): where the host blob can be re-read from when the arena evicted it
  const OffloadStore* store = nullptr;
  uint32_t store_ids[4] = {0, 0, 0, 0};
  uint64_t host_clock = 0;     // last time the host blob was needed (host-tier LRU)
  // a disk -> pinned-host read in flight on the priority block reader (speculative requests read at LOW priority in
  // the background; a demand promotes and waits): the blob becomes `host` once every tensor's request has finished
  void* host_pending = nullptr;
  std::vector<PrioAioPool::Handle> disk_reqs;
};
struct Slot {
  void* dev = nullptr;
  int node = -1;
  uint64_t last_use_seq = 0;  // sequence number of the last forward whose kernels read this slot
};

static constexpr int kFenceRing = 64;
static constexpr int kMirrorPool = 128;  // pooled pinned routing mirrors of sync-free forwards not yet applied to the counters
static constexpr int kHideSharedMaxTokens = 16;  // forwards up to this many tokens hide the shared expert under the router

// One H2D lane = a copy stream (hipMemcpyAsync, served by an SDMA engine) + a re-tile stream (kernels) + a ring of two
// staging buffers, each large enough for the biggest tensor of a blob.  Tensor i+1 is copied into the other
// buffer while tensor i is re-tiled into its slot, so the link never waits for a kernel; the slot-reuse fences are
// waited for by the RE-TILE stream only (the copy into staging does not touch the slot).
struct StageBuf {
  void* dev = nullptr;
  hipEvent_t filled = nullptr, freed = nullptr;
  bool used = false;
};
constexpr int kCopyTsRing = 8192;
constexpr int kStageRing = 4;  // staging buffers per lane: the link never waits for a re-tile launch two copies back
struct CopyLane {
  hipStream_t copy = nullptr, retile = nullptr;
  StageBuf ring[kStageRing];
  int next = 0;
};

constexpr int kLaTokens = 8;  // the lookahead route runs for decode-sized forwards only
struct moeinf_engine {
  moeinf_config cfg;
  int64_t es = 2;  // element size
  int dt = DT_BF16;
  BlobLayout lay, lay_sh;   // host blob (reference layout)
  DevLayout dlay, dlay_sh;  // HBM slot (tiled)
  CopyLane demand, prefetch;  // on-demand misses (high priority) / speculative copies (low priority)
  int64_t stage_bytes = 0;
  // Small experts travel as ONE hipMemcpyAsync of the whole contiguous host blob (as the reference copies it,
  // model_topology.cpp:102-119) into a staging buffer sized for an expert, re-tiled by one launch; big ones (Mixtral: 336 MiB)
  // tensor by tensor, so FFN stage 1 can start while the down projection is still on the link.
  bool whole_blob = false;
  unsigned long long* d_copy_ts = nullptr;  // [kCopyTsRing][4] timing records of the pull form (kernels.hip: pull_retile_kernel)
  std::vector<uint64_t> copy_ts_expect;     // per ring slot: finished-workgroup count that means "this use of the slot is complete" (grows)
  uint64_t copy_ts_head = 0, copy_ts_tail = 0;
  uint64_t copy_seq = 0;
  unsigned long long copy_busy_until = 0;   // tick up to which link-busy time has been accounted
  bool h2d_pull = false;   // experts are PULLED by a kernel of the copy stream straight from the pinned host blob into the tiled slot
  int h2d_pull_wgs = 16;
  // next-layer gate lookahead (moeinf_set_lookahead)
  std::vector<const void*> la_gates;   // [L] borrowed device pointers; empty: off
  int la_max = 0;
  float* d_la_f = nullptr;             // logits [kLaTokens * E] + router_prob [kLaTokens]
  int32_t* d_la_i = nullptr;           // pair_valid + pair_order [2 * kLaTokens * K]
  int32_t* h_la_idx = nullptr;         // pinned, written by the route kernel itself: top-k ids [kLaTokens * K]
  float* h_la_w = nullptr;             // ... and weights
  int la_armed_T = 0;                  // > 0: the running forward launched a lookahead route over this many tokens
  std::vector<int> la_list;            // node indices predicted for the next layer, best first
  int num_cus = 0;                     // of THIS engine's device
  int layer1_switch_wgs_per_cu = -1;   // occupancy of the one-launch Switch kernel (asked once per engine)
  bool host_f8 = false;                // dtype id 3: fp8 (e4m3fn) experts in the host tier, bf16 slots and arithmetic
  int64_t host_es = 2;                 // bytes per element of the HOST blob (1 with host_f8, else es)
  bool route_v3 = false;               // MOEINF_ROUTER_DEEPSEEK_V3: cfg.router_kind is stored as DEEPSEEK
  std::vector<const float*> gate_bias; // ... per layer: e_score_correction_bias (borrowed device pointers)
  bool route_no_renorm = false;        // MOEINF_ROUTER_SOFTMAX_TOPK (Grok / Arctic): cfg.router_kind is stored as MIXTRAL
  uint8_t* h_keep = nullptr;           // moeinf_dispatch_mask_subset: pinned byte per expert
  hipEvent_t busy_mark = nullptr;  // stop event of the latest-ending copy interval accounted so far (union of the lanes' busy time)
  int64_t slot_bytes = 0;
  int L = 0, E = 0, K = 0, H = 0, F = 0, Fs = 0;
  bool has_shared = false;

  // host tier: pinned arena in chunks
  std::vector<void*> arena_chunks;
  int64_t arena_chunk_bytes = 0, arena_used_in_chunk = 0, arena_total = 0;

  // device tier
  std::vector<Slot> slots;
  std::vector<int> free_slots;
  int64_t max_slots = 0;
  bool slab_exhausted = false;
  std::vector<Node> nodes;           // [e*L + l]  (expert-major: the reference's eviction scan order)
  std::vector<PolicyEntry> pol;      // same indexing
  std::vector<void*> shared_dev;     // [L]
  uint64_t clock = 0;
  std::vector<int> resident_per_layer;  // #experts of layer l resident AND ordered (ready_waited)

  // pending speculative transfers (reference: ArcherTaskPool's unified_queue_) and the copies in flight
  PrefetchQueue pq;
  std::unique_ptr<PrioAioPool> aio;   // disk tier reader (created with the first expert registered from a store)
  std::deque<QueuedTask> disk_inflight;  // speculative tasks whose host blob is being read from disk (low priority)
  std::vector<int> stale_disk;        // nodes whose speculative disk read outlived its task (stale layer): adopted when done
  int disk_window = 2;                // such reads in flight at most
  bool draining = false;              // moeinf_sync_copies: serve the queue even while demand copies are in flight
  std::deque<int> demand_inflight;    // node indices whose copy was issued on the demand lane and not yet observed complete
  // speculation governor (moeinf_set_prefetch_governor): running usefulness of finished speculative copies
  float gov_min_useful = 0.f;         // 0 = off
  int gov_probe_every = 16;
  float gov_score = 1.f;              // exponential average of outcomes (1 = dispatched before eviction, 0 = evicted unused)
  int gov_outcomes = 0, gov_skipped = 0;
  std::deque<int> prefetch_inflight;  // node indices whose copy was issued on the prefetch lane, oldest first
  int prefetch_window = 2;            // experts in flight on the prefetch lane at most
  std::vector<void*> host_free;       // arena blocks returned by host-tier eviction
  uint64_t host_clock = 0;

  // streams / events
  hipEvent_t route_ev = nullptr;
  // fences: an event on the compute stream after forward #fence_seq[i].  Recorded after EVERY forward that took the decision path
  // (copies follow, and they should wait for no more compute than they must) but only after every fence_every-th sync-free
  // forward (a record between two kernels costs the stream 2.7-4 us: profiles/r06_fence_every.md); whoever needs a forward that
  // no fence covers yet records one then (fence_for)
  hipEvent_t fence_ev[kFenceRing];
  uint64_t fence_seq[kFenceRing] = {};
  uint64_t fence_head = 0;   // fences recorded so far (ring position = fence_head % kFenceRing)
  uint64_t fenced_seq = 0;   // the newest forward a recorded fence covers
  hipStream_t unfenced_stream = nullptr;  // the stream the forwards after fenced_seq were launched on
  int fence_every = 16;
  uint64_t seq = 0;  // forwards issued
  std::vector<std::pair<hipEvent_t, hipEvent_t>> copy_timers;  // (start, end) pairs not yet accumulated
  std::vector<std::pair<hipEvent_t, hipEvent_t>> wait_timers;  // compute-stream stalls on copies
  std::vector<hipEvent_t> event_pool;

  // device workspace
  uint64_t* d_wptr = nullptr;  // [L][E+1]
  float* d_logits = nullptr;
  int32_t *d_topk_idx = nullptr, *d_pair_valid = nullptr, *d_pair_order = nullptr, *d_pair_slot = nullptr;
  float *d_topk_w = nullptr, *d_router_prob = nullptr;
  int32_t *d_counts = nullptr, *d_offsets = nullptr, *d_active = nullptr, *d_n_active = nullptr;
  int32_t *d_slot_token = nullptr, *d_slot_pair = nullptr, *d_miss = nullptr;
  int32_t* d_arrive = nullptr;  // [ceil(H/16)] zeroed arrival counters of the fused combine's column tiles
  int32_t* d_chunk = nullptr;   // [ceil(rows/1024) * E] scratch of the many-workgroup dispatch index
  void *d_h = nullptr, *d_y = nullptr;
  uint64_t* d_dec_w = nullptr;  // [8] batch-1 decode records written by the self-routing FFN stage 1 (kernels.h FfnStage::dec_w)
  float* d_dec_cw = nullptr;    // [8]
  void *d_h_sh = nullptr, *d_y_sh = nullptr;  // decode-sized DeepSeek: shared expert's h / y (its FFN rides with the router)
  int64_t ldh = 0;
  int32_t* h_mirror = nullptr;  // pinned, written by the index kernels themselves: {n_active, counts[E+1], active[E+1]}
  int32_t* h_miss = nullptr;
  std::vector<PokeArgs> pending_pokes;
  // sync-free forwards: the index kernel writes the routing mirror straight into a pooled pinned buffer
  // (no copy command on the compute stream); it is applied to the counters/statistics lazily, once the
  // forward's end-of-forward fence has passed (no residency decision depends on it while every expert of
  // the layer is resident)
  struct PendingMirror { uint64_t seq; int32_t* buf; int layer; int T; bool prof; bool local; };
  std::deque<PendingMirror> pend;
  std::vector<int32_t*> mirror_pool;
  int32_t* mirror_slab = nullptr;  // one pinned allocation holding every pooled mirror
  int owned_experts = 0;

  // EP workspace (lazily allocated)
  EpOwnArgs::Rec* d_ep_rec = nullptr;  // [64] stage 1 -> stage 2 records of the self-indexing owner kernels
  int32_t *d_ep_key = nullptr, *d_ep_counts = nullptr, *d_ep_offsets = nullptr, *d_ep_active = nullptr,
          *d_ep_nactive = nullptr, *d_ep_pair_slot = nullptr, *d_ep_slot_token = nullptr, *d_ep_slot_pair = nullptr,
          *d_ep_pair_pos = nullptr;
  int ep_cap_rows = 0;   // per-peer capacity of the last ep_pack (0: compact, variable-split exchange)
  int ep_alloc_cap = 0;  // what the EP workspace is sized for
  int64_t ep_alloc_np = 0;  // ... and the max_tokens*K it was built for

  // activation-aware speculation inside the engine (moeinf_set_predictor): the attached tracer is fed from the routing
  // mirrors the index kernels write — no read-back, no Python between "layer l routed" and "layer l+k experts requested"
  Tracer* pred_tracer = nullptr;
  int64_t pred_seq = -1;
  int pred_lookahead = 0, pred_max = 0;
  float pred_min_share = 0.f;
  std::vector<float> pred_matrix;
  int64_t pred_calls = 0, pred_enqueued = 0;

  // native transport of the exchange (moeinf_ep_comm_init): RCCL communicator + engine-owned exchange buffers
  RcclComm ep_comm = nullptr;
  int ep_cap_tokens = 0, ep_x_cap_rows = 0;
  void *ep_x_send = nullptr, *ep_x_recv = nullptr, *ep_x_y = nullptr, *ep_x_ret = nullptr;
  // ... or the direct peer-store exchange (moeinf_ep_peer_export / _attach, ep_peer.h): no collective at all
  EpPeerWindow ep_win;
  int ep_win_cap_tokens = 0;
  bool ep_use_peer = false;        // moeinf_ep_moe_forward takes this transport (moeinf_ep_select_transport)
  bool ep_uniform = false;         // the caller guarantees equal token counts on every rank (moeinf_ep_set_uniform_tokens): batch 1 = broadcast form
  std::vector<int32_t> ep_peer_pids;
  std::vector<uint64_t> ep_peer_ptrs;
  bool ep_peer_poll = true;        // consumer kernels poll their flags themselves (false: a one-wave wait kernel in front); agreed by all ranks (ep_peer.h: poll_agreed)
  bool ep_bcast_ok = true;         // the broadcast form may be taken: agreed by all ranks (ep_peer.h: bcast_agreed)
  int32_t* ep_err_host = nullptr;  // pinned copy of the device error flag, refreshed by the exchange itself (see ep_peer_forward)
  uint32_t ep_err_every = 16;      // ... every so many exchanges (MOEINF_EP_ERR_CHECK_EVERY)
  int64_t ep_peer_timeout_ticks = 0;
  struct EpProfRec { hipEvent_t ev[6]; };
  std::vector<EpProfRec> ep_prof_pending;
  moeinf_ep_profile ep_prof;

  // stage-2 output override of the expert-parallel FFN: rows go straight to the reply buffer, in arrival order
  void* ovr_out = nullptr;
  const int32_t* ovr_map = nullptr;

  // last forward
  bool last_hidden_shared = false;
  bool last_selfroute = false;      // the last forward used the self-routing FFN stage 1 (batch-1 decode)
  bool last_front1 = false;         // ... or its front (gate, hidden shared expert, stage 1) as one launch (moe_front1_kernel)
  bool last_layer1 = false;         // ... and ran as ONE launch (layer_fused.hip)
  uint32_t* d_layer_ctr = nullptr;  // its counters (kernels.h LayerSync): only grow, zeroed at creation and after an error
  uint32_t layer1_launches = 0;
  float* d_layer_part = nullptr;    // [4][H] partial sums of the Switch form's split stage 2
  bool layer1_scalar_poll = false;
  unsigned long long* d_layer_trace = nullptr;  // MOEINF_LAYER1_TRACE=<file>: per-workgroup timestamps of the last one-launch layer, written out at destroy
  int layer1_trace_blocks = 0;
  int64_t layer1_timeout_ticks = 0;
  int last_T = 0, last_layer = -1;
  hipStream_t last_stream = nullptr;
  int last_rows = 0;

  moeinf_stats st;

  // profiling
  bool profiling = false, ep_profiling = false;
  // ev[0..5]: forward start, after the router, before stage 1, between the stages, after stage 2, end.  Launchers that carry a timer
  // (kernels.h arm_kernel_timer) get ev[2] / ev[3] as the stage-1 kernel's own begin / end and k2 / ev[4] as stage 2's
  struct ProfRec { hipEvent_t ev[6]; hipEvent_t k2 = nullptr; };
  std::vector<ProfRec> prof_pending;
  moeinf_profile prof;
};

// Events that only TIME things (profiling intervals, exposed-wait timers): a record that fails must not fail the forward it
// brackets — the interval is simply lost (hipEventElapsedTime on it fails and the reader skips it) — but its error must not
// linger in the runtime's sticky last-error slot either.
static inline void record_timing(hipEvent_t ev, hipStream_t st) {
  if (hipEventRecord(ev, st) != hipSuccess) (void)hipGetLastError();
}

// ---- fences (the fields are described in moeinf_engine) ----
static inline int record_fence(moeinf_engine* g, hipStream_t st) {
  const int i = (int)(g->fence_head % kFenceRing);
  HIPCHK(hipEventRecord(g->fence_ev[i], st));
  g->fence_seq[i] = g->seq;
  g->fence_head += 1;
  g->fenced_seq = g->seq;
  return MOEINF_OK;
}
// ring position of the oldest recorded fence that covers forward #s, -1 if none does.  fence_seq[kFenceRing]: the forward each
// ring entry was recorded behind, head: fences recorded so far (entry i lives at i % kFenceRing; an entry that left the ring is
// covered by every entry still in it).  Pure: tests/test_kernel_selection_cpu.py drives it through moeinf_fence_cover_pos
static inline int fence_cover_pos(const uint64_t* fence_seq, uint64_t head, uint64_t s) {
  if (head == 0 || fence_seq[(head - 1) % kFenceRing] < s) return -1;
  const uint64_t lo = head > (uint64_t)kFenceRing ? head - kFenceRing : 0;
  uint64_t i = head;  // newest first: the covering entries are a suffix of the ring
  while (i > lo && fence_seq[(i - 1) % kFenceRing] >= s) --i;
  return (int)(i % kFenceRing);
}
static inline hipEvent_t covering_fence(const moeinf_engine* g, uint64_t s) {
  const int pos = fence_cover_pos(g->fence_seq, g->fence_head, s);
  return pos < 0 ? nullptr : g->fence_ev[pos];
}
// a fence for forward #s (already launched: s <= seq); one is recorded now, behind everything launched so far, if none covers it
static inline int fence_for(moeinf_engine* g, uint64_t s, hipEvent_t* out) {
  if (s > g->fenced_seq) {
    if (s > g->seq) return fail(MOEINF_ERR_STATE, "fence for forward %llu asked before it was launched (%llu)",
                                (unsigned long long)s, (unsigned long long)g->seq);
    CHK(record_fence(g, g->unfenced_stream));
  }
  *out = covering_fence(g, s);
  return MOEINF_OK;
}
// the end of a forward (or of one chunk of it) launched on st.  must: the decision path, whose copies should see the tightest fence
static inline int end_forward(moeinf_engine* g, hipStream_t st, bool must) {
  // forwards on another stream are still unfenced: close them there first (a fence on st says nothing about them)
  if (g->fenced_seq < g->seq && g->unfenced_stream != st) CHK(record_fence(g, g->unfenced_stream));
  g->seq += 1;
  g->unfenced_stream = st;
  if (must || g->seq - g->fenced_seq >= (uint64_t)g->fence_every) CHK(record_fence(g, st));
  return MOEINF_OK;
}

static int node_index(const moeinf_engine* g, int layer, int expert) { return expert * g->L + layer; }
// a node's disk backing, reference-counted on the store so that it cannot be closed under a live engine
static void set_node_store(Node& n, const OffloadStore* st) {
  if (n.store == st) return;
  if (n.store) n.store->users -= 1;
  n.store = st;
  if (st) st->users += 1;
}
static bool owns(const moeinf_engine* g, int expert) { return g->cfg.ep_size <= 1 || (expert % g->cfg.ep_size) == g->cfg.ep_rank; }


// ---- handles of the host-only part of the C ABI that the engine also looks into (capi_host.cpp owns their entry points)
struct moeinf_store {
  OffloadStore s;
  void* bounce[2] = {nullptr, nullptr};  // pinned pieces of moeinf_store_get_device
  hipEvent_t bounce_ev[2] = {nullptr, nullptr};
  bool bounce_used[2] = {false, false};
};
struct moeinf_tracer { Tracer* t; };

// ---- hot-path pieces shared between engine.cpp and engine_ep.cpp (defined in engine.cpp) ---------------------------
struct MirrorPlan { bool fast = false; int32_t* target = nullptr; };
struct SelfRoute {  // batch-1 decode: FFN stage 1 routes for itself (launch_ffn1_selfroute)
  const RouteArgs* ra;
  const IndexArgs* ia;
  const FfnStage* sh2;  // hidden shared expert's stage 2, or nullptr
  const FfnStage* sh1 = nullptr;  // front1: the hidden shared expert's stage 1
  bool front1 = false;  // gate + stage 1 (+ the hidden shared expert) as ONE launch (launch_moe_front1): the caller has NOT launched the gate
  bool layer1_switch = false;  // ... its Switch form (launch_moe_layer1_switch); if that declines, dispatch_experts launches the gate itself
};

template <typename T>
static inline int dmalloc(T** p, size_t n) {
  HIPCHK(hipMalloc((void**)p, n * sizeof(T)));
  return MOEINF_OK;
}
bool can_hide_shared(const moeinf_engine* g, int T);
int dispatch_experts(moeinf_engine* g, int layer, const void* x_in, int64_t ld_x, int T, int max_active, int exp_rows,
                     hipStream_t st, bool prof, moeinf_engine::ProfRec* pr, const MirrorPlan& mp,
                     const CombineArgs* fuse, bool* fused, const SelfRoute* sr = nullptr);
void drop_stale_prefetches(moeinf_engine* g, int layer);
void fill_stage(const moeinf_engine* g, int layer, int stage, FfnStage& s, int64_t ld_x = 0);
int flush_pokes(moeinf_engine* g, hipStream_t st);
void free_ep_workspace(moeinf_engine* g);
hipEvent_t get_event(moeinf_engine* g);
void hidden_shared_stages(const moeinf_engine* g, int layer, const void* x_dev, FfnStage& sh1, FfnStage& sh2);
int launch_index_auto(moeinf_engine* g, const IndexArgs& ia, hipStream_t st);
void make_index_args(const moeinf_engine* g, int T, int batch_rows, int32_t* mirror, IndexArgs& ia);
void make_route_args(const moeinf_engine* g, const void* x_dev, const void* gate_w_dev, int T, RouteArgs& ra);
int plan_mirror(moeinf_engine* g, int layer, MirrorPlan& mp);
int pump_if_pending(moeinf_engine* g);
