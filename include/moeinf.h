/* moeinf.h — C ABI of libmoeinf_hip.so, the MI355X-native expert-offload engine.
 *
 * This is the drop-in boundary for MoE-Infinity's native core.  In the reference the boundary is
 * the pybind11 module `prefetch_op` (core/python/py_archer_prefetch.cpp:10-93) exposing two
 * classes, `prefetch_handle` (ArcherPrefetchHandle) and `expert_dispatcher` (ExpertDispatcher).
 * Every entry point below names the reference interface it replaces.  No torch types cross
 * this boundary: plain pointers, sizes and a HIP stream handle (as void*).
 *
 * Conventions
 *   - every function returns an int status (MOEINF_OK == 0); on failure
 *     moeinf_last_error() returns a thread-local message.  Nothing aborts the process
 *     (the reference DLOG_FATAL -> abort(), core/base/logging.cc:168-176).
 *   - one engine per (process, GPU).  Calls on one engine must come from one thread at a time.
 *   - "dev" pointers are HIP device pointers on cfg.device_id; "host" pointers are ordinary
 *     host memory unless stated pinned.
 *   - row-major tensors; nn.Linear layout [out, in] for every weight matrix.
 */
#ifndef MOEINF_H_
#define MOEINF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOEINF_ABI_VERSION 4

/* status codes */
enum {
  MOEINF_OK = 0,
  MOEINF_ERR_INVALID = 1,     /* bad argument / unsupported shape */
  MOEINF_ERR_HIP = 2,         /* a HIP runtime call failed (message has hipGetErrorString) */
  MOEINF_ERR_OOM = 3,         /* host arena or device slab exhausted */
  MOEINF_ERR_STATE = 4,       /* call order violated (e.g. forward before experts registered) */
  MOEINF_ERR_UNSUPPORTED = 5  /* feature of the reference not (yet) built */
};

/* dtype ids: core/parallel/expert_module.h:20-23 (DTYPE_BFLOAT16 0, DTYPE_FLOAT32 1, DTYPE_FLOAT16 2) */
enum { MOEINF_DTYPE_BF16 = 0, MOEINF_DTYPE_F32 = 1, MOEINF_DTYPE_F16 = 2,
       /* the reference's id 3 (core/parallel/expert_module.h:23,118-119 -> torch::kFloat8_e4m3fn): expert blobs are e4m3fn bytes in
        * the HOST tier and on the link (half of bf16's: half the copy time of every miss), up-cast to bf16 when pulled into their HBM
        * slot; activations, gate and arithmetic are bf16 — y = FFN(x; W.to(bf16)) (round 6) */
       MOEINF_DTYPE_F8E4M3 = 3 };

/* expert_type ids: core/parallel/expert_module.h:13-18, moe_infinity/common/constants.py:29-37 */
enum {
  MOEINF_EXPERT_SWITCH = 0,       /* relu(x wi^T) wo^T                      expert_module.cpp:24-36   */
  MOEINF_EXPERT_SWITCH_GATED = 1, /* (gelu(x wi_0^T) * (x wi_1^T)) wo^T          expert_module.cpp:46-59   */
  MOEINF_EXPERT_NLLB = 2,         /* relu(x fc1^T + b1) fc2^T + b2          expert_module.cpp:79-93   */
  MOEINF_EXPERT_FSGPT = 3,        /* same math as NLLB                      expert_module.cpp:113-129 */
  MOEINF_EXPERT_MIXTRAL = 4,      /* (silu(x w1^T) * (x w3^T)) w2^T         expert_module.cpp:147-175 */
  MOEINF_EXPERT_DEEPSEEK = 5      /* (silu(x g^T) * (x u^T)) d^T            expert_module.cpp:193-204 */
};

/* router kinds (the Python blocks of the reference) */
enum {
  MOEINF_ROUTER_MIXTRAL = 0,  /* moe_infinity/models/mixtral.py:46-54 */
  MOEINF_ROUTER_DEEPSEEK = 1, /* models/modeling_deepseek/modeling_deepseek.py:463-512 (MoEGate) */
  MOEINF_ROUTER_SWITCH = 2,   /* HF SwitchTransformersTop1Router @ models/switch_transformers.py:76 */
  MOEINF_ROUTER_NLLB = 3,     /* HF NllbMoeTop2Router @ models/nllb_moe.py:53 */
  MOEINF_ROUTER_DEEPSEEK_V3 = 5,  /* DeepSeek-V3's MoEGate: sigmoid scores + e_score_correction_bias, groups ranked by the sum of their two best, weights normalised
                                   * then scaled (models/modeling_deepseek_v3/modeling_deepseek.py:466-528); n_group / topk_group / norm_topk_prob /
                                   * routed_scaling_factor as for DeepSeek; the per-layer bias: moeinf_set_gate_bias */
  MOEINF_ROUTER_SOFTMAX_TOPK = 4 /* Grok / Arctic: softmax -> top-k, NO renormalisation (models/grok.py:38-45, arctic.py:39-45); otherwise Mixtral's */
};

/* cache replacement policy. LFU-in-cache is what the reference actually runs
 * (core/parallel/expert_dispatcher.cpp:227-258, core/prefetch/task_scheduler.cpp:276-310). */
enum { MOEINF_POLICY_LFU_INCACHE = 0, MOEINF_POLICY_LRU = 1 };

typedef struct moeinf_engine moeinf_engine;

typedef struct moeinf_config {
  int32_t abi_version; /* MOEINF_ABI_VERSION */
  int32_t device_id;   /* HIP device ordinal of this process's GPU */
  /* model: expert_dispatcher(num_experts, num_layers, dtype, expert_type, num_threads)
   * (core/python/py_archer_prefetch.cpp:84-85); num_threads has no equivalent here. */
  int32_t num_layers;  /* MoE layers (parse_moe_param, moe_infinity/utils/hf_config.py:22-55) */
  int32_t num_experts; /* routed experts per layer */
  int32_t expert_type; /* MOEINF_EXPERT_* */
  int32_t dtype;       /* MOEINF_DTYPE_* of expert weights AND activations */
  int32_t hidden;      /* H */
  int32_t inter;       /* F (routed expert intermediate size) */
  int32_t shared_inter;/* DeepSeek shared expert F (n_shared_experts * moe_intermediate_size), 0 = none
                          (moe_infinity/models/deepseek.py:39-46,133-136) */
  int32_t top_k;       /* K */
  /* router */
  int32_t router_kind;    /* MOEINF_ROUTER_* */
  int32_t gate_dtype;     /* dtype of the gate/classifier weight handed to moe_forward */
  int32_t norm_topk_prob; /* DeepSeek norm_topk_prob; NLLB normalize_router_prob_before_dropping */
  float routed_scaling_factor; /* DeepSeek */
  int32_t n_group;        /* DeepSeek group_limited_greedy: >1 enables it */
  int32_t topk_group;
  int32_t expert_capacity;/* Switch expert_capacity (tokens per expert per batch row) */
  /* memory tiers: prefetch_handle(prefix, device_memory_ratio) (py_archer_prefetch.cpp:12) */
  double device_memory_ratio;  /* fraction of TOTAL device memory (core/memory/memory_pool.cpp:150-158) */
  int64_t device_memory_bytes; /* >0: explicit expert-cache byte budget overriding the ratio */
  int64_t host_memory_bytes;   /* >0: cap of the pinned host arena; 0 = sized on demand */
  int32_t policy;              /* MOEINF_POLICY_* */
  /* expert parallelism: this engine owns experts with (e % ep_size) == ep_rank
   * (core/model/model_topology.cpp:533-536, distributed/expert_executor.py:49-54) */
  int32_t ep_rank;
  int32_t ep_size;
  int32_t max_tokens; /* largest T (tokens per forward call) the workspace is sized for */
} moeinf_config;

typedef struct moeinf_stats {
  int64_t forwards;        /* moe_forward calls */
  int64_t expert_hits;     /* dispatches that found the expert resident */
  int64_t expert_misses;   /* dispatches that had to fetch on demand */
  int64_t prefetch_issued; /* H2D copies started by moeinf_prefetch */
  int64_t prefetch_useful; /* prefetched experts that were later dispatched before eviction */
  int64_t evictions;
  int64_t h2d_bytes;       /* bytes copied host->device (demand + prefetch) */
  int64_t slots_total;     /* device cache capacity in experts */
  int64_t slots_used;
  int64_t slot_bytes;      /* bytes per slot (= expert blob size rounded to 4 KiB) */
  int64_t host_arena_bytes;
  double h2d_busy_ms;      /* link-busy time: first to last hipMemcpyAsync of every transfer, HIP events on the copy streams */
  double exposed_wait_ms;  /* time the compute stream spent waiting on copies (event-timed) */
  /* pending-transfer queue (reference: ArcherTaskPool, core/prefetch/task_scheduler.cpp) */
  int64_t prefetch_queued;    /* speculative transfers waiting in the queue right now */
  int64_t prefetch_cancelled; /* queued transfers dropped: stale layer, displaced by a more urgent request, overtaken by a demand fetch, queue cleared */
  int64_t prefetch_dropped;   /* popped but not started: no evictable slot ("evict failed", task_scheduler.cpp:505-510) */
  int64_t prefetch_wasted;    /* prefetched experts evicted before any dispatch used them */
  int64_t inflight_hits;      /* dispatches that found the expert's transfer still in flight (counted in expert_hits) */
  /* host tier as a cache over the disk tier */
  int64_t host_evictions;     /* host blobs dropped from the pinned arena (re-readable from the offload directory) */
  int64_t disk_reads;         /* experts read disk -> pinned host */
  int64_t disk_bytes;
  int64_t disk_reads_async;   /* of those, started in the background at low priority by a speculative request */
  int64_t prefetch_throttled; /* speculative requests not started because the governor found speculation unprofitable */
} moeinf_stats;

/* ---- errors ------------------------------------------------------------------------------ */
const char* moeinf_last_error(void);
int moeinf_abi_version(void);

/* ---- introspection for tests (no reference counterpart, no GPU needed): which form of the register-ring GEMM
 * (csrc/ffn_ring2_kernel.h: ffn_gemm_ring2) the launcher picks for an FFN stage.  dtype: MOEINF_DTYPE_*; nmat: 2 = gated stage,
 * 1 = plain; K / K_sh: reduction length of the routed / shared experts (0: no shared expert); R: output rows; active: experts
 * with rows (the grid's upper bound); max_rows: rows of the busiest expert as the engine passes it (1.5 x the mean + 1 on the
 * sync-free path); num_cus: compute units.  out[0] = token groups of 16 per pass (0: another kernel runs), out[1] = 1 when the
 * last round of workgroups is split into half workgroups, out[2] = row blocks per expert, out[3] = first split unit,
 * out[4] = workgroups launched.  Environment knobs are honoured as in the launcher. */
/* the row estimate moeinf_moe_forward passes to the FFN launchers when every expert of the layer is resident (sync-free path) */
int moeinf_rows_estimate(int tokens, int top_k, int num_experts);
int moeinf_ffn_ring2_form(int dtype, int nmat, int K, int K_sh, int R, int active, int max_rows, int num_cus, int32_t* out5);
/* The fence ring (csrc/engine_internal.h): sync-free forwards record a fence event only every MOEINF_FENCE_EVERY-th time; a copy
 * that recycles a slot waits for the OLDEST recorded fence that covers the slot's last reader.  moeinf_fence_ring: entries in the
 * ring; moeinf_fence_cover_pos: the ring position that lookup returns (-1: no recorded fence covers `forward` yet) for
 * fence_seq[ring] = the forward each entry was recorded behind and `recorded` = fences recorded so far (entry i at i % ring).
 * Pure host logic, exported so that the wrap-around cases are tested without a GPU. */
int moeinf_fence_ring(void);
int moeinf_fence_cover_pos(const uint64_t* fence_seq, uint64_t recorded, uint64_t forward);

/* ---- lifecycle: prefetch_handle.__init__ / clean_up_resources ------------------------------
 * (core/prefetch/archer_prefetch_handle.cpp:18-64,73-81) */
int moeinf_create(const moeinf_config* cfg, moeinf_engine** out);
int moeinf_destroy(moeinf_engine* eng);

/* ---- expert blobs --------------------------------------------------------------------------
 * The reference keeps one contiguous, 4 KiB-aligned blob per expert with the tensors in
 * `tensor_ids` order (core/model/model_topology.cpp:429-431,677-700):
 *   mixtral  w1[F,H] w2[H,F] w3[F,H]          deepseek gate[F,H] up[F,H] down[H,F]
 *   nllb     fc1.w[F,H] fc1.b[F] fc2.w[H,F] fc2.b[H]      switch wi[F,H] wo[H,F]
 * moeinf_expert_layout reports that layout (byte offsets/sizes per tensor, total bytes).
 * which = 0 routed expert, 1 shared expert (DeepSeek). */
int moeinf_expert_layout(const moeinf_engine* eng, int which, int64_t offsets[4], int64_t sizes[4],
                         int32_t* n_tensors, int64_t* total_bytes);

/* expert_dispatcher.register_expert(layer, expert, tensor_ids) + prefetch_handle.offload/register
 * (core/parallel/expert_dispatcher.cpp:160-173, core/prefetch/archer_prefetch_handle.cpp:229-237):
 * hands the engine the host copy of one expert.  blob != NULL: `nbytes` bytes are copied into the
 * engine's pinned arena.  blob == NULL: an uninitialised arena block is reserved and the caller
 * fills it through moeinf_expert_host_ptr (zero-copy registration). */
int moeinf_register_expert(moeinf_engine* eng, int layer, int expert, const void* blob, int64_t nbytes);
int moeinf_expert_host_ptr(moeinf_engine* eng, int layer, int expert, void** host_ptr);
/* DeepSeek shared expert: always device-resident ("shared" params are exempt from offloading,
 * moe_infinity/runtime/model_offload.py:758-759,824-826).  Host blob is copied to the device once. */
int moeinf_register_shared(moeinf_engine* eng, int layer, const void* blob, int64_t nbytes);

/* ---- the hot path --------------------------------------------------------------------------
 * moeinf_moe_forward replaces, for one MoE layer and one step:
 *   Sync*MoeBlock.forward router+mask   (moe_infinity/models/{mixtral,deepseek,switch_transformers,nllb_moe}.py)
 *   DistributedExpertExecutor.dispatch_local (moe_infinity/distributed/expert_executor.py:32-58)
 *   ExpertDispatcher set_inputs/enqueue_expert/wait_expert (core/parallel/expert_dispatcher.cpp:111-450)
 *   Node::SetDevice on-demand fetch + LFU eviction (core/model/model_topology.cpp:53-136,
 *                                                   expert_dispatcher.cpp:227-266)
 *   the combine loop of the block (e.g. mixtral.py:96-101).
 * x_dev:  [tokens, H] activations (cfg.dtype).  batch_rows: B (tokens = B*S; only Switch's
 * per-row capacity uses it; pass 1 otherwise).  gate_w_dev: [E, H] (cfg.gate_dtype).
 * out_dev: [tokens, H] (cfg.dtype).  stream: hipStream_t the caller's work is ordered on.
 * The call enqueues work and returns; it blocks the host only to read the routing counts
 * when a residency decision is needed.
 * Stream lifetime: the engine keeps the handle of the stream its last forwards ran on (moeinf_sync waits on it; a later
 * copy that recycles one of their slots, or the first forward on a different stream, records an event on it): do not
 * destroy a stream that has carried a forward until moeinf_sync (or moeinf_destroy) has returned or the engine has run a
 * forward on another stream. */
#define MOEINF_FWD_DEFAULT 0u
#define MOEINF_FWD_ROUTE_ONLY 1u  /* stop after router + dispatch-index (parity tests) */
#define MOEINF_FWD_NO_COMBINE 2u  /* stop after the expert FFN (parity tests) */
int moeinf_moe_forward(moeinf_engine* eng, int layer, const void* x_dev, int tokens, int batch_rows,
                       const void* gate_w_dev, void* out_dev, void* stream, uint32_t flags);

/* expert_dispatcher.set_inputs + enqueue_expert(...) for every active expert + wait_expert
 * (core/parallel/expert_dispatcher.cpp:111-158,436-450; driven by dispatch_local,
 * moe_infinity/distributed/expert_executor.py:32-58) for callers that keep the reference's PYTHON
 * router and combine: router_mask_dev is the dense [tokens, E] mask (mask_elem_bytes = 1 for
 * bool/uint8, 4 for int32, 8 for int64; non-zero = token routed to expert).  Runs residency + the
 * grouped expert FFN; y_dev receives the expert outputs as expert-sorted rows (expert ascending,
 * tokens ascending inside an expert), i.e. the concatenation of wait_expert()'s tensors.
 * Host outputs (any may be NULL): counts_host[E] tokens per expert, hit_host[E] = 1 if the expert was
 * already resident when dispatched (wait_expert's 4th tuple field), 0 if fetched on demand, -1 if idle. */
int moeinf_dispatch_mask(moeinf_engine* eng, int layer, const void* x_dev, int tokens, const void* router_mask_dev,
                         int mask_elem_bytes, void* y_dev, int32_t* counts_host, int32_t* hit_host, void* stream);
/* ... the same with only the mask columns of `expert_ids` taken (the others count as all-false): what
 * expert_dispatcher.wait_expert() needs when the experts enqueued on THIS device are a subset of the mask's columns
 * (core/parallel/expert_dispatcher.cpp:121-158 queues experts one by one), without building a second mask on the host side.
 * expert_ids == NULL: every column (= moeinf_dispatch_mask). */
int moeinf_dispatch_mask_subset(moeinf_engine* eng, int layer, const void* x_dev, int tokens, const void* router_mask_dev, int mask_elem_bytes,
                                void* y_dev, int32_t* counts_host, int32_t* hit_host, void* stream, const int32_t* expert_ids, int n_ids);

/* The combine loop of the reference's blocks on its own (mixtral.py:96-101 `final_hidden_states[token_indices] +=
 * output * routing_weights_mask[...]`, deepseek.py:123-131, switch_transformers.py:99-109 incl. the `router_probs *`
 * scale and the passthrough of dropped tokens, nllb_moe.py:84-104 incl. `next_states[next_states == 0] = hidden`), for
 * callers that keep the Python router and use moeinf_dispatch_mask for the experts.
 *   y_dev         expert outputs as expert-sorted rows (what moeinf_dispatch_mask wrote / wait_expert()'s tensors
 *                 concatenated in ascending expert order)
 *   topk_idx_dev  [tokens, K] int32 expert of every (token, k) pair, -1 = pair not dispatched (capacity / zero weight)
 *   topk_w_dev    [tokens, K] float combine weights (the values of routing_weights_mask at the chosen experts)
 *   router_prob_dev [tokens] float, Switch only (router_probs); x_dev [tokens, H]: Switch/NLLB passthrough (else may be NULL)
 * Summation order = ascending expert id (the order dispatch_local enqueues and wait_expert returns), rounding points
 * of the block's dtype arithmetic as in moeinf_moe_forward.  DeepSeek's shared expert is NOT added (the reference adds it
 * in Python after the loop, deepseek.py:133-136). */
int moeinf_combine(moeinf_engine* eng, const void* x_dev, const void* y_dev, const int32_t* topk_idx_dev, const float* topk_w_dev,
                   const float* router_prob_dev, int tokens, void* out_dev, void* stream);

/* Device-side copies of the LAST forward's router results into caller tensors (async on `stream`;
 * any pointer may be NULL): logits [tokens,E] f32, topk_idx [tokens,K] i32, topk_w [tokens,K] f32.
 * These are the tensors the reference's blocks return to HF (e.g. router_logits, mixtral.py:118). */
int moeinf_copy_routing_dev(moeinf_engine* eng, float* logits_dev, int32_t* topk_idx_dev, float* topk_w_dev, void* stream);

/* Decomposed results of the LAST forward, copied to host (synchronises `stream`).
 * Any output pointer may be NULL.  Replaces the values the reference's blocks hold in Python
 * locals: selected_experts/routing_weights (mixtral.py:48-54), topk_idx/topk_weight
 * (deepseek.py:55-60), and dispatch_local's expert_count (expert_executor.py:34-43).
 *   topk_idx  [tokens*K] int32 expert ids, descending weight (Switch: -1 = dropped token)
 *   topk_w    [tokens*K] float  (values already rounded to the dtype the reference holds them in)
 *   counts    [E] int32 tokens per expert;  offsets [E+1] exclusive scan
 *   slot_token[tokens*K] int32 token id of every expert-sorted row (first offsets[E] valid)
 *   pair_slot [tokens*K] int32 expert-sorted row of every (token,k) pair, -1 if not dispatched */
int moeinf_get_routing(moeinf_engine* eng, int32_t* topk_idx, float* topk_w, int32_t* counts,
                       int32_t* offsets, int32_t* slot_token, int32_t* pair_slot);
/* Per-expert FFN outputs of the last forward, expert-sorted rows [offsets[E], H] in cfg.dtype:
 * what wait_expert() returns as a list of tensors (core/parallel/expert_dispatcher.cpp:436-450). */
int moeinf_get_expert_outputs(moeinf_engine* eng, void* host_out, int64_t nbytes);
/* router logits [tokens, E] fp32 of the last forward (Mixtral: bf16-rounded values) */
/* wait for the stream of the last forward and report the kernels' error flag (MOEINF_ERR_STATE: an FFN workgroup found no
 * resident blob for an active expert, or a kernel of the peer-store exchange gave up waiting for another rank) */
int moeinf_sync(moeinf_engine* eng);
int moeinf_get_logits(moeinf_engine* eng, float* host_out, int64_t n_floats);

/* ---- prefetch / cache control --------------------------------------------------------------
 * prefetch_handle.enqueue_prefetch(tensor_id, gpu) (archer_prefetch_handle.cpp:206-218) for n
 * experts of one layer, in priority order (highest score first).  Copies run on the engine's
 * prefetch stream and never block the caller. */
int moeinf_prefetch(moeinf_engine* eng, int layer, const int32_t* experts, const float* scores, int n);
/* prefetch_handle.replace_cache_candidates(ids) (archer_prefetch_handle.cpp:195-204,
 * core/prefetch/task_scheduler.h:66-79): the new protected set replaces the old one;
 * protected experts are skipped by eviction. */
int moeinf_protect(moeinf_engine* eng, const int32_t* layers, const int32_t* experts, int n);
/* expert_dispatcher.clear_expert_cache_counts() (expert_dispatcher.cpp:175-184) */
int moeinf_clear_cache_counts(moeinf_engine* eng);
/* 1 if resident on device, 0 otherwise: prefetch_handle.is_tensor_on_device (py_archer_prefetch.cpp:64-69) */
int moeinf_is_resident(moeinf_engine* eng, int layer, int expert, int32_t* resident);
/* serves the whole pending queue and blocks until every H2D copy has landed (tests/bench warm-up) */
int moeinf_sync_copies(moeinf_engine* eng);
/* DeviceMemoryPool::SetMemoryRatio (core/memory/memory_pool.cpp:150-158) at run time, in bytes: shrink (evicting by
 * the replacement policy, freeing the slots' memory) or grow the expert cache.  Synchronises the device. */
int moeinf_set_cache_budget(moeinf_engine* eng, int64_t device_memory_bytes);
/* The replacement policy (MOEINF_POLICY_*) at run time: which resident expert the next miss evicts — the reference's
 * LFU over in-cache visit counts (core/parallel/expert_dispatcher.cpp:227-266, core/prefetch/task_scheduler.cpp:236-317) or
 * LRU.  Takes effect with the next eviction; nothing is evicted by the call (bench.py compares the two on ONE warm engine). */
int moeinf_set_cache_policy(moeinf_engine* eng, int policy);
/* Speculation governor (no counterpart in the reference, whose prefetcher enqueues everything it predicts).  The engine
 * keeps a running average of how speculative copies END — dispatched before eviction (1) or evicted unused (0); once 8
 * have ended and the average is below min_useful_fraction, speculative requests are dropped except one probe in
 * probe_every.  0 switches it off (default).  Independent of it, a speculative copy is never STARTED while an
 * on-demand copy is on the link. */
int moeinf_set_prefetch_governor(moeinf_engine* eng, float min_useful_fraction, int probe_every);
/* Grow the token-sized workspace to hold forwards of up to max_tokens tokens (the reference's dispatcher allocates per
 * call and has no such limit; expert_dispatcher.set_inputs grows it on demand).  Only grows; synchronises the device. */
int moeinf_reserve_tokens(moeinf_engine* eng, int max_tokens);

/* prefetch_handle.get_hit_rate() (archer_prefetch_handle.cpp:281-297; columns model_topology.cpp:253-263): per-expert
 * counters, out[L*E][7] = {visit_cnt, hit_cnt, miss_cnt, prefetch_cnt, incache_visit_count, resident, unused_count}
 * (unused_count: times the expert was evicted after a speculative copy no dispatch ever used, task_scheduler.cpp:304) */
int moeinf_get_expert_counters(moeinf_engine* eng, int64_t* out, int64_t n_int64);
int moeinf_get_stats(moeinf_engine* eng, moeinf_stats* out);
int moeinf_reset_stats(moeinf_engine* eng);

/* ---- per-kernel timing (bench.py's roofline leg) ----------------------------------------------
 * With profiling on, every forward brackets its kernels with HIP events ON THE CALLER'S STREAM
 * (the stream the kernels are launched on) and accumulates durations plus the ALGORITHMIC bytes
 * each launch had to move (DESIGN.md section "algorithmic bytes"):
 *   ffn1: active experts' gate/up (or fc1/wi) weights + gathered x rows read + h rows written
 *   ffn2: active experts' down (fc2/wo) weights + h rows read + y rows written
 *   route: gate weight + x read, logits/top-k written;  combine: y rows + x read, out written */
typedef struct moeinf_profile {
  int64_t forwards;
  int64_t ffn1_launches, ffn2_launches;
  int64_t ffn1_bytes, ffn2_bytes, route_bytes, combine_bytes;
  double route_ms, ffn1_ms, ffn2_ms, combine_ms;
  double host_wait_ms; /* wall-clock time the host spent blocked on the routing D2H */
  int64_t fused_layers; /* forwards that ran as ONE launch (csrc/layer_fused.hip): all their bytes and time are under ffn1 */
  int64_t kernel_timed_launches; /* ffn1 / ffn2 launches whose interval is the kernel's OWN begin..end (start / stop events on its
                                  * dispatch packet, hipExtLaunchKernel: the batch-1 decode launchers) — no event-record packets
                                  * of the command processor inside it, so it agrees with rocprofv3's duration */
} moeinf_profile;
int moeinf_set_profiling(moeinf_engine* eng, int enabled); /* bit 0: per-kernel events; bit 1: per-phase events of moeinf_ep_moe_forward */
/* synchronises the last stream, returns the accumulated numbers and resets them */
int moeinf_get_profile(moeinf_engine* eng, moeinf_profile* out);

/* ---- activation-aware tracer / predictor ---------------------------------------------------
 * Host-side restatement of moe_infinity/memory/expert_tracer.py, expert_predictor.py,
 * expert_prefetcher.py (EAM = expert activation matrix [L,E]). */
typedef struct moeinf_tracer moeinf_tracer;
int moeinf_tracer_create(int num_layers, int num_experts, int capacity, moeinf_tracer** out);
int moeinf_tracer_destroy(moeinf_tracer* tr);
/* load historical EAMs: n x [L,E] float (expert_tracer.py:40-52 load_trace) */
int moeinf_tracer_load(moeinf_tracer* tr, const float* eams, int n);
/* ExpertTracer.create_entry (expert_tracer.py:54-59): returns a sequence handle */
int moeinf_tracer_create_entry(moeinf_tracer* tr, int64_t* seq_id);
/* ExpertTracer.finish_entry (expert_tracer.py:61-76): fold the sequence's EAM into the collection */
int moeinf_tracer_finish_entry(moeinf_tracer* tr, int64_t seq_id);
/* ExpertPredictor.predict (expert_predictor.py:17-35) = update_entry + find_most_similar + decay.
 * experts: the n expert ids activated at `layer` in this step.  matrix_out: [L,E] float scores.
 * nearest_out (optional): index of the most similar historical EAM. */
int moeinf_tracer_predict(moeinf_tracer* tr, int64_t seq_id, int layer, const int32_t* experts, int n,
                          float* matrix_out, int32_t* nearest_out);
/* ExpertPrefetcher.prefetch_experts ordering (expert_prefetcher.py:42-59): from a predicted
 * matrix, the (layer, expert) list for layers >= `layer` with score > 0, descending score,
 * ties in (layer, expert) order.  Returns count in *n_out (<= L*E). */
int moeinf_tracer_prefetch_order(const moeinf_tracer* tr, int layer, const float* matrix, int32_t* layers_out,
                                 int32_t* experts_out, float* scores_out, int32_t* n_out);
int moeinf_tracer_get_eam(moeinf_tracer* tr, int64_t seq_id, double* eam_out);
/* Activation-aware speculation INSIDE the engine.  With a tracer attached, every forward feeds the running sequence's EAM
 * from the routing mirror its index kernel wrote (ExpertTracer.update_entry) and — on forwards that take the decision
 * path, where the host holds the layer's routing anyway — runs ExpertPredictor.predict + ExpertPrefetcher.prefetch_experts
 * for the next `lookahead_layers` layers: experts whose predicted share of their layer's activations is >= min_share,
 * best first, at most max_experts per layer call, go into the pending-transfer queue (moeinf_prefetch semantics).  No
 * device synchronisation and no host code between "layer l routed" and "layer l+k requested" (the reference does this in
 * Python with a D2H read per layer, expert_tracer.py:94-125, and requests EVERY predicted expert).  tr == NULL or
 * seq_id < 0 detaches.  The tracer must outlive the attachment. */
int moeinf_set_predictor(moeinf_engine* eng, moeinf_tracer* tr, int64_t seq_id, int lookahead_layers, float min_share, int max_experts);
/* Next-layer gate lookahead (speculation from the activations instead of from history).  With the gate weights of every
 * layer registered (borrowed device pointers, [E, H] each, cfg.gate_dtype; n_layers = cfg.num_layers), a decode-sized
 * forward (tokens <= 8) of layer l that takes the decision path also applies layer l+1's gate to ITS OWN input rows — in
 * a transformer the residual stream changes little from one layer to the next, so that top-k is a prediction of layer
 * l+1's routing — and, once layer l's misses are on the link and its FFN is launched, issues the predicted experts that
 * are not resident BEHIND those misses on the same copy stream (of the max_experts most confident predictions — largest
 * gate weight first; the link idles for about one small expert per layer, so 1-2 is the useful range —; never
 * evicting the protected set or the experts of the layer in flight).  The link then stays busy while layer l computes and
 * layer l+1's attention runs, instead of idling until layer l+1 routes.  Counted as speculative copies (prefetch_issued /
 * prefetch_useful / prefetch_wasted of moeinf_stats).  Results of the forward are unchanged.  gate_w_dev == NULL or
 * n_layers == 0 turns it off.  The reference has no counterpart: its prefetcher predicts from the EAM history only
 * (moe_infinity/memory/expert_prefetcher.py:28-59) and is dormant for these models (mixtral.py:69-85 commented out). */
int moeinf_set_lookahead(moeinf_engine* eng, const void* const* gate_w_dev, int n_layers, int max_experts);
/* MOEINF_ROUTER_DEEPSEEK_V3 only: layer `layer`'s e_score_correction_bias — num_experts fp32 values in device memory, borrowed
 * (modeling_deepseek_v3/modeling_deepseek.py:455-458,484-487: added to the sigmoid scores for the CHOICE of experts, not to their
 * weights).  NULL = zeros.  Takes effect with the next forward of that layer. */
int moeinf_set_gate_bias(moeinf_engine* eng, int layer, const float* bias_dev);

/* ---- disk tier: the reference's offload directory (host only) ----------------
 * Reads and writes `<offload_path>/archer_index` + `archer_param_<n>` in the reference's own format
 * (core/aio/archer_tensor_index.cpp:11-25,51-67,101-132; archer_tensor_handle.cpp:53-86), so directories
 * produced by MoE-Infinity load here and vice versa.  scalar_type = c10::ScalarType code
 * (Float 6, Half 5, BFloat16 15, ...). */
typedef struct moeinf_store moeinf_store;
int moeinf_store_open(const char* offload_path, moeinf_store** out); /* ArcherTensorHandle ctor (archer_tensor_handle.cpp:23-51) */
int moeinf_store_close(moeinf_store* st);                            /* flushes the index if it changed */
/* prefetch_handle.offload(tensor, id) = StoreTensor (archer_tensor_handle.cpp:53-86) */
int moeinf_store_put(moeinf_store* st, uint32_t tensor_id, const void* data, uint64_t nbytes, const int64_t* dims,
                     int ndim, int scalar_type);
int moeinf_store_flush(moeinf_store* st); /* ArcherTensorIndex::Serialize */
int moeinf_store_count(const moeinf_store* st, int64_t* n);
int moeinf_store_ids(const moeinf_store* st, uint32_t* ids_out, int64_t capacity);
/* prefetch_handle.is_tensor_offloaded(id) + index lookup: *found = 0/1; dims_out has room for 8 dims */
int moeinf_store_meta(const moeinf_store* st, uint32_t tensor_id, int32_t* found, uint64_t* nbytes, int64_t* offset,
                      int32_t* ndim, int64_t* dims_out, int32_t* scalar_type);
/* ReadTensor (archer_tensor_handle.cpp:189-201): payload -> dst (O_DIRECT when dst is 4 KiB-aligned and
 * has room for the 4 KiB-padded size) */
int moeinf_store_get(const moeinf_store* st, uint32_t tensor_id, void* dst, uint64_t capacity);
/* DENSE (non-expert) tensors: disk -> device in one call — the data movement behind prefetch_handle.begin /
 * fetch_tensors / set_topology for dense nodes (AcquireTensor, FetchTensors, InitializeTopology:
 * archer_prefetch_handle.cpp:83-130,220-227, model_topology.cpp:402-548; Node::SetDevice :76-119).  The payload of
 * `tensor_id` is read in pinned pieces and copied with hipMemcpyAsync on `stream` into dst_dev (caller-owned device
 * memory, e.g. a slice of the node's slab).  Device data is stream-ordered after the call. */
int moeinf_store_get_device(moeinf_store* st, uint32_t tensor_id, void* dst_dev, uint64_t capacity, void* stream);
/* Node::SetDevice disk->host leg (model_topology.cpp:76-100, SetModuleMemoryFromDisk :647-674):
 * read the n tensors of one expert (tensor_ids in blob order, as expert_dispatcher.register_expert
 * receives them) from the store straight into the expert's pinned arena blob. */
int moeinf_register_expert_from_store(moeinf_engine* eng, int layer, int expert, const moeinf_store* st,
                                      const uint32_t* tensor_ids, int n);
/* With cfg.host_memory_bytes > 0 the pinned arena is an LRU cache over the offload directory: experts registered
 * after the cap is reached stay on disk and are read on their first miss, and a full arena drops the least recently
 * needed re-readable blob (Node::SetDevice(DISK), model_topology.cpp:76-88).  `st` must stay open for the
 * engine's lifetime. */

/* ---- cache-policy simulator (host only, no GPU) --------------------------------------------
 * The engine's replacement policy as a standalone object, so the policy can be checked against
 * the oracle without a device (tests -m "not gpu"). ids are arbitrary non-negative ints. */
typedef struct moeinf_cache_sim moeinf_cache_sim;
int moeinf_cache_sim_create(int num_slots, int policy, moeinf_cache_sim** out);
int moeinf_cache_sim_destroy(moeinf_cache_sim* sim);
/* access one id: *hit = 1/0, *evicted = evicted id or -1 */
int moeinf_cache_sim_access(moeinf_cache_sim* sim, int64_t id, int32_t* hit, int64_t* evicted);
int moeinf_cache_sim_protect(moeinf_cache_sim* sim, const int64_t* ids, int n);
int moeinf_cache_sim_clear_counts(moeinf_cache_sim* sim);

/* ---- pending-transfer queue, standalone (host only, no GPU) ------------------------------------
 * The engine's queue discipline as an object of its own, so it can be checked against the oracle's restatement of
 * ArcherTaskPool (core/prefetch/task_scheduler.cpp) without a device.  node ids are arbitrary non-negative ints,
 * `layer` is the reference's corr_id & 0xffffffff, level 0 is the most urgent of 20. */
typedef struct moeinf_pq moeinf_pq;
int moeinf_pq_create(moeinf_pq** out);
int moeinf_pq_destroy(moeinf_pq* q);
/* EnqueueTask (task_scheduler.cpp:82-118) */
int moeinf_pq_enqueue(moeinf_pq* q, int64_t node, int layer, int priority, int remove_layer, int32_t* dropped);
/* StartExec's queue step (:158-168); node < 0: only the stale-layer rule */
int moeinf_pq_on_demand(moeinf_pq* q, int64_t node, int layer, int32_t* dropped);
/* FetchExec (:44-80); already_there: the node is on its target device (not queued) */
int moeinf_pq_fetch(moeinf_pq* q, int64_t node, int layer, int already_there, int32_t* dropped);
/* ReplaceCacheCandidates / ClearQueue (task_scheduler.h:55-79) */
int moeinf_pq_clear_prefetch(moeinf_pq* q, int32_t* dropped);
/* GPUThreadFunc's pick (:451-497): *found = 0 when empty */
int moeinf_pq_pop(moeinf_pq* q, int64_t* node, int32_t* layer, int32_t* priority, int32_t* found);
int moeinf_pq_snapshot(const moeinf_pq* q, int64_t* nodes, int32_t* layers, int32_t* priorities, int capacity, int32_t* n);
/* prefetch score in (0,1] -> queue level 1..19 (moeinf_prefetch with scores) */
int moeinf_priority_from_score(float score, int32_t* level);

/* ---- two-priority block reader of the disk tier, standalone (host only, no GPU) ----------------
 * ArcherPrioAioHandle::Read(filename, buffer, high_prio, num_bytes, offset) (core/aio/archer_prio_aio_handle.cpp:37-71):
 * reads are cut into blocks (the reference: 1 MiB, :13); worker threads serve HIGH-priority blocks before any LOW one
 * (:123-169), so an on-demand read waits for at most the low blocks already in flight.  submit never blocks (the
 * reference's Read does: call moeinf_aio_wait right after it for that behaviour); promote moves a low request's
 * unstarted blocks to the high queue (a speculative read whose expert is demanded).  try_direct: O_DIRECT when dst and
 * offset are 4 KiB-aligned (the read is then rounded up to 4 KiB: dst must have room), buffered where the filesystem
 * refuses.  The engine reads expert blobs through the same object (experts registered with
 * moeinf_register_expert_from_store: demand misses HIGH, speculative requests LOW and in the background). */
typedef struct moeinf_aio moeinf_aio;
int moeinf_aio_create(int threads, int64_t block_bytes, moeinf_aio** out);
int moeinf_aio_destroy(moeinf_aio* a);
int moeinf_aio_submit_read(moeinf_aio* a, const char* path, void* dst, int64_t nbytes, int64_t offset, int high_prio,
                           int try_direct, int64_t* request);
int moeinf_aio_promote(moeinf_aio* a, int64_t request);
int moeinf_aio_done(moeinf_aio* a, int64_t request, int32_t* done);
int moeinf_aio_wait(moeinf_aio* a, int64_t request); /* blocks; forgets the request; error = first failed block */
int moeinf_aio_stats(const moeinf_aio* a, int64_t out[5]); /* blocks_high, blocks_low, bytes, promoted, direct_fallbacks */

/* ---- expert-parallel exchange helpers (multi-GPU) ---------------------
 * Pack routed rows for an all-to-all and unpack the replies.  The collective itself (RCCL
 * all_to_all over xGMI) is issued by the host layer (torch.distributed) between these calls. */
/* Exchange rows have ep_row_elems = H + 16/sizeof(elem) elements: H activations followed by a 16-byte
 * tail whose first int32 is the row's expert id (-1 = padding row), so activations and ids cross the
 * fabric in ONE all-to-all per direction. */
int moeinf_ep_row_elems(const moeinf_engine* eng, int32_t* elems);
/* After a MOEINF_FWD_ROUTE_ONLY forward: write, for each destination rank r, the rows of x whose
 * expert lives on r ((e % ep_size) == r) into send_dev[r*cap_rows ...] ([ep_size*cap_rows, ep_row_elems],
 * cfg.dtype).  send_counts_dev[ep_size] (optional) receives the row counts.  cap_rows must be at least
 * tokens * min(K, ceil(E / ep_size)): a token sends a rank at most one row per expert that rank owns. */
int moeinf_ep_pack(moeinf_engine* eng, const void* x_dev, void* send_dev, int32_t* send_counts_dev, int cap_rows,
                   void* stream);
/* Sender side in ONE call: router (gate -> top-k -> dispatch index; a decode-sized DeepSeek forward carries its shared
 * expert's FFN inside these launches, i.e. under the exchange) + moeinf_ep_pack.  For decode-sized forwards the send rows
 * are written by the router's own launch: two launches before the all-to-all.  cap_rows >= tokens * min(K, experts per
 * rank).  Arguments as moeinf_moe_forward / moeinf_ep_pack. */
int moeinf_ep_route_pack(moeinf_engine* eng, int layer, const void* x_dev, int tokens, int batch_rows, const void* gate_w_dev,
                         void* send_dev, int32_t* send_counts_dev, int cap_rows, void* stream);
/* Run the expert FFN on rows received from all ranks: recv_dev [ep_size*cap_rows, ep_row_elems];
 * writes y_dev [ep_size*cap_rows, H] in the same row order (padding rows are left untouched: nobody reads them). */
int moeinf_ep_expert_ffn(moeinf_engine* eng, int layer, const void* recv_dev, void* y_dev, int cap_rows, void* stream);
/* Variable-split exchange for prefill-sized batches (tokens*K rows >> what a fixed per-peer capacity should carry):
 * send rows are COMPACT and sorted by destination rank, send_counts_dev[ep_size] holds the rows per destination; the
 * host layer exchanges the counts, then moves exactly the routed rows with split sizes.  After this call
 * moeinf_ep_combine takes cap_rows = 0 and ret_dev in the same compact order. */
int moeinf_ep_pack_compact(moeinf_engine* eng, const void* x_dev, void* send_dev, int32_t* send_counts_dev, void* stream);
/* moeinf_ep_expert_ffn over exactly nrows received rows (no padding rows), y_dev [nrows, H] in arrival order */
int moeinf_ep_expert_ffn_rows(moeinf_engine* eng, int layer, const void* recv_dev, void* y_dev, int nrows, void* stream);
/* Combine replies: ret_dev [ep_size*cap_rows, H] holds, in the order moeinf_ep_pack produced,
 * the expert outputs for this rank's routed rows; writes out_dev [tokens, H].  With a shared expert
 * (DeepSeek) registered, its FFN over x_dev runs here, on the token's home rank, and is added last. */
int moeinf_ep_combine(moeinf_engine* eng, const void* x_dev, const void* ret_dev, void* out_dev, int cap_rows,
                      void* stream);

/* ---- native transport of the exchange: a whole expert-parallel layer in ONE call ----------------------------------
 * RCCL (ncclSend/ncclRecv in one group = an all-to-all over xGMI) is called from inside the engine; the library is
 * bound at run time (the copy PyTorch-ROCm already loaded, else ROCm's).  Bootstrap like any NCCL program: rank 0 gets
 * a 128-byte unique id and hands it to the other ranks over whatever channel the host has (torch.distributed
 * broadcast, a file, MPI); every rank then calls moeinf_ep_comm_init (collective; rank/size = cfg.ep_rank/ep_size).
 * cap_tokens = the largest token count per forward through this path, THE SAME ON EVERY RANK (it sizes the per-peer row
 * slots of the fixed-capacity exchange: cap_tokens * min(K, ceil(E/ep_size))).  The reference has no collective here:
 * it copies rows between GPUs from one process (core/parallel/expert_dispatcher.cpp:284,405). */
int moeinf_ep_comm_available(int32_t* available); /* 1 if librccl could be bound in this process (no communicator is made) */
int moeinf_ep_comm_unique_id(void* id_out, int nbytes /* 128 */);
/* the LOCAL half of the bootstrap (validation, library binding, exchange buffers; no collective inside): call it on every
 * rank and agree on the outcome before any rank enters moeinf_ep_comm_init, so that a rank that fails here leaves nobody
 * blocked in ncclCommInitRank.  moeinf_ep_comm_init runs it itself when it was not called. */
int moeinf_ep_comm_prepare(moeinf_engine* eng, int cap_tokens);
int moeinf_ep_comm_init(moeinf_engine* eng, const void* unique_id, int nbytes, int cap_tokens); /* collective */
/* equal-split all-to-all of device buffers on `stream`: segment p (bytes_per_peer bytes) of send_dev goes to rank p */
int moeinf_ep_all_to_all(moeinf_engine* eng, const void* send_dev, void* recv_dev, int64_t bytes_per_peer, void* stream);
/* ---- direct peer-store exchange: expert parallelism with NO collective ----------------------------------------------
 * Replaces the reference's peer-access set-up (cudaDeviceEnablePeerAccess for every device pair,
 * core/prefetch/archer_prefetch_handle.cpp:37-61) and its implicit P2P row copies `tensor.to(device)`
 * (core/parallel/expert_dispatcher.cpp:284,405).  Every rank owns one exchange window in uncached device memory; the other
 * ranks map it (hipIpcOpenMemHandle between processes, the plain pointer inside one process).  The router's pack step and the
 * owner's FFN stage 2 store rows STRAIGHT into the destination rank's window over xGMI and publish an exchange number in its
 * flag words; the consumer kernels poll their own flags (bounded: MOEINF_EP_PEER_TIMEOUT_MS, default 10 000; on expiry the
 * device error flag reads 2; a flag AHEAD of the awaited exchange — a rank out of step after a failed call — reads 3).  The
 * flag is reported by moeinf_sync AND, without any sync, by a later moeinf_ep_moe_forward: the stream copies it to a pinned
 * word every MOEINF_EP_ERR_CHECK_EVERY (16) exchanges and every forward looks at that word first (MOEINF_ERR_STATE).
 * Unlike RCCL it also runs between processes that SHARE a GPU.  Which exchange form a forward takes (routed / batch-1
 * broadcast, polls inside the consumers / a wait kernel in front) is decided from ALL ranks' blobs, so MOEINF_EP_PEER_POLL and
 * MOEINF_EP_BCAST set on one rank only cannot split the group.
 * Bootstrap — every call is local and fails without blocking anyone; agree on the outcome between the steps:
 *   1. every rank: moeinf_ep_peer_export(eng, cap_tokens, blob)        allocates the window, writes a 192-byte blob
 *   2. exchange the blobs over any channel (torch.distributed all_gather, a file, MPI), concatenate them in rank order
 *   3. every rank: moeinf_ep_peer_attach(eng, blobs, ep_size * 192)    maps the peers, enables peer access
 *   4. every rank: moeinf_ep_peer_selftest(eng, stream, &ok)           tagged rows + flags both ways; ok = 0 on timeout
 * After step 3 moeinf_ep_moe_forward takes this transport.  Ranks must stop together: a peer may store into a window until
 * its owner's moeinf_destroy.  cap_tokens as for moeinf_ep_comm_init, the same on every rank. */
#define MOEINF_EP_PEER_BLOB_BYTES 192
int moeinf_ep_peer_export(moeinf_engine* eng, int cap_tokens, void* blob_out, int nbytes /* 192 */);
int moeinf_ep_peer_attach(moeinf_engine* eng, const void* blobs, int nbytes /* ep_size * 192 */);
int moeinf_ep_peer_selftest(moeinf_engine* eng, void* stream, int32_t* ok);
/* Undo steps 1-3 (local): unmap the peers, free the window and its staging buffers (kept if an RCCL communicator uses them).
 * For a group that agreed NOT to use this transport after a failed attach / self-test; every rank calls it or none (a peer
 * may store into a window until it is released).  The reference has no counterpart: its peer access is process-wide
 * (core/prefetch/archer_prefetch_handle.cpp:37-61). */
int moeinf_ep_peer_release(moeinf_engine* eng);
/* How long a consumer waits for a peer's flag before it gives up (error flag 2).  Set from MOEINF_EP_PEER_TIMEOUT_MS (10 000)
 * by moeinf_ep_peer_export; a host layer shortens it while a transport is on probation (bench.py: 3 000) so that a peer that
 * never publishes costs seconds, and restores it afterwards.  Takes effect with the next exchange; ms > 0. */
int moeinf_ep_peer_set_timeout_ms(moeinf_engine* eng, int ms);
/* ... and what it is now (a host layer that shortens it for a probation restores THIS, not the environment's default). */
int moeinf_ep_peer_get_timeout_ms(moeinf_engine* eng, int* ms);
/* out[0] = transport moeinf_ep_moe_forward will take (MOEINF_EP_TRANSPORT_*); out[1] = 1 if another rank shares this GPU
 * (then a one-wave wait kernel runs in front of the consumers instead of polls inside them); out[2] = 1: polls inside the
 * consumer kernels; out[3] = exchanges so far */
#define MOEINF_EP_TRANSPORT_NONE 0
#define MOEINF_EP_TRANSPORT_RCCL 1
#define MOEINF_EP_TRANSPORT_PEER_STORE 2
int moeinf_ep_transport(const moeinf_engine* eng, int32_t out[4]);
/* with both transports set up on one engine: which one moeinf_ep_moe_forward takes (default: the one set up last) */
int moeinf_ep_select_transport(moeinf_engine* eng, int kind);
/* on != 0: the caller guarantees that EVERY rank passes the same token count to every moeinf_ep_moe_forward (decode loops do).
 * With it, a ONE-token forward over the peer-store transport takes the broadcast form: the home rank sends its row and its E gate
 * logits to every rank and every owner's FFN stage 1 routes for itself (as the local batch-1 path does) — four launches per
 * layer instead of five, no router launch between the gate and the exchange.  All ranks must be in that form together, hence
 * the explicit promise.  Default off. */
int moeinf_ep_set_uniform_tokens(moeinf_engine* eng, int on);
/* moeinf_moe_forward for an expert-parallel engine, one host call per layer, everything enqueued on `stream`
 * (tokens <= cap_tokens).  Peer-store transport: router + pack into the peers' windows -> owner FFN -> combine (five
 * launches).  RCCL transport: moeinf_ep_route_pack -> all-to-all -> moeinf_ep_expert_ffn -> all-to-all -> moeinf_ep_combine. */
int moeinf_ep_moe_forward(moeinf_engine* eng, int layer, const void* x_dev, int tokens, int batch_rows, const void* gate_w_dev,
                          void* out_dev, void* stream);
/* per-phase HIP-event times of moeinf_ep_moe_forward calls made while moeinf_set_profiling was on (synchronises, resets) */
typedef struct moeinf_ep_profile {
  int64_t calls;
  double route_pack_ms, a2a_dispatch_ms, owner_ffn_ms, a2a_combine_ms, combine_ms;
} moeinf_ep_profile;
int moeinf_ep_get_profile(moeinf_engine* eng, moeinf_ep_profile* out);

#ifdef __cplusplus
}
#endif
#endif /* MOEINF_H_ */
