// kernels.h — launch wrappers of the gfx950 kernels (implemented in kernels.hip).
// Host code (engine.cpp, engine_ep.cpp) sees only these plain-C++ declarations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

namespace moeinf {

enum { DT_BF16 = 0, DT_F32 = 1, DT_F16 = 2 };  // = the reference's dtype ids (core/parallel/expert_module.h:20-23)
inline int dt_bytes(int dtype) { return dtype == DT_F32 ? 4 : 2; }
struct EpFuse;

// ---- direct peer-store exchange (expert parallelism without a collective; host side: ep_peer.h) ------------------
// Every rank owns one exchange WINDOW in uncached device memory, mapped into every other rank's address space
// (hipIpc* between processes, plain pointers inside one process):
//   [0, 2048)     recv flags: word p * 32 = the last exchange whose rows from rank p have landed in this window
//   [2048, 4096)  ret flags:  word p * 32 = the last exchange whose expert outputs from owner p have landed
//   recv_off      [ep_size][cap_rows] rows of H activations + 16-byte tail (segment p is written by rank p)
//   ret_off       [ep_size][cap_rows] rows of H expert outputs            (segment p is written by owner p)
// Producers store rows STRAIGHT into the consumer's window (system-scope write-through stores), drain them
// (s_waitcnt vmcnt(0)), and publish the exchange number in the consumer's flag word; consumers poll their own flags.
// One flag word per cache line; the number only grows, so nothing is ever reset.
constexpr int EP_MAX_PEERS = 16;
constexpr int EP_FLAG_WORDS = 32;  // 128 bytes between two flag words
constexpr int64_t EP_RET_FLAGS_OFF = 2048, EP_WINDOW_HDR = 4096;
struct EpPeers {
  uint64_t base[EP_MAX_PEERS];  // window of rank p as mapped in THIS process (base[rank] = this rank's own)
  int64_t recv_off, ret_off;    // byte offsets of the two regions (the same on every rank)
  int64_t recv_row_bytes, ret_row_bytes;  // (H + tail) * element size, H * element size
  int64_t timeout_ticks;        // bound of every poll, wall_clock64 ticks (100 MHz); on expiry *err = 2 and the kernel goes on
  int32_t* done;                // arrival counter of many-workgroup producers (this device, zero between launches)
  int32_t* err;
  uint32_t epoch;               // number of this exchange (1, 2, ...)
  int64_t bcast_off;            // broadcast region: [ep_size] x bcast_stride bytes (the E gate logits of rank p's token)
  int bcast_stride;
  int rank, size, cap_rows;
  int on;                       // 0: the classic form (send buffer + a collective); the rest of the struct is unused
  int poll;                     // 1: consumer kernels poll their flags themselves; 0: a one-wave wait kernel runs in front
};
struct EpWait {  // poll `n` flag words (stride EP_FLAG_WORDS) until each has reached `epoch`
  const uint32_t* flags;
  int n;
  uint32_t epoch;
  int64_t timeout_ticks;
  int32_t* err;
};
hipError_t launch_ep_wait(const EpWait& w, hipStream_t st);

// epilogues of the row-dot (weight-streaming) FFN kernel
enum {
  EPI_NONE = 0,       // out = Tr(acc)                      (down / w2 / wo projections)
  EPI_BIAS = 1,       // out = Tr(Tr(acc) + bias)           (NLLB fc2)
  EPI_RELU = 2,       // out = relu(Tr(acc))                (Switch wi)
  EPI_BIAS_RELU = 3,  // out = relu(Tr(Tr(acc) + bias))     (NLLB fc1)
  EPI_GATED_SILU = 4, // out = Tr(Tr(silu(Tr(acc0))) * Tr(acc1))   (Mixtral w1/w3, DeepSeek gate/up)
  EPI_GATED_GELU = 5  // out = Tr(Tr(gelu(Tr(acc0))) * Tr(acc1))   (gelu-gated Switch wi_0/wi_1, expert_module.cpp:54-59; erf form = torch::gelu's
                      // default).  Round 5: runs the row kernel at every size (the tuned grouped GEMMs are built for the SiLU gate only).
};

struct CombineArgs {
  const void* x;            // [T,H] (Switch/NLLB passthrough)
  const void* y;            // [rows,H] expert outputs, expert-sorted
  void* out;                // [T,H]
  const int32_t* topk_idx;  // [T,K]
  const float* topk_w;      // [T,K]
  const int32_t* pair_slot; // [T,K]
  const int32_t* pair_order;// [T,K]
  const float* router_prob; // [T] Switch
  const void* y_shared;           // non-null: DeepSeek shared-expert outputs, row t at y_shared[(row0 + t) * H]
  const int32_t* shared_offsets;  // row0 = shared_offsets ? shared_offsets[shared_E] : 0 (device value)
  int shared_E;
  int T, H, K;
  int kind;                 // MOEINF_ROUTER_* (selects the reference block's combine semantics)
  int dtype;
};

// One stage (x -> h, or h -> y) of the grouped expert FFN for every active expert of a layer.
struct FfnStage {
  const void* in;          // B-operand rows: x [tokens, ld_in] (stage 1) or h [rows, ld_in] (stage 2)
  int64_t ld_in;           // elements
  const int32_t* row_map;  // stage 1: expert-sorted row -> token id (fused gather); nullptr: identity
  void* out;               // [rows, ld_out]
  int64_t ld_out;
  const int32_t* out_map;  // expert-sorted row -> output row (scatter fused into the epilogue); nullptr: identity
  const uint64_t* wptr;    // [E+1] device base pointer of every expert blob of this layer (0 = absent);
                           // entry E is the shared expert (DeepSeek) or 0
  const int32_t* active;   // [<= E+1] ids of experts with tokens, ascending
  const int32_t* n_active; // device scalar (used when n_active_host < 0)
  int n_active_host;       // >= 0: number of entries of `active` to process (chunked dispatch)
  const int32_t* counts;   // [E+1] rows per expert
  const int32_t* offsets;  // [E+2] first expert-sorted row of each expert
  int32_t* miss_flag;      // set to 1 if an active expert has wptr == 0
  int E;                   // number of routed experts (index of the shared pseudo-expert)
  int K, R;                // routed experts: reduction length, output rows
  int K_sh, R_sh;          // shared expert
  int64_t off_a, off_b, off_bias;           // byte offsets inside a routed expert blob
  int64_t off_a_sh, off_b_sh;               // inside the shared blob
  int epi;
  int dtype;
  // decode-sized fused combine (stage 2 of a forward with <= 16 tokens, Mixtral/DeepSeek combine semantics): the
  // LAST block to finish a 16-column tile of y (over all active experts; arrival counter + agent-scope fences)
  // combines those 16 columns for every token, so the combine needs no launch of its own and stays deterministic
  int fuse_combine;
  int32_t* tile_done;      // [ceil(R/16)] arrival counters, zero between launches (the last block resets its own)
  CombineArgs comb;
  // batch-1 decode records (self-routing path): written by the meta block of ffn1_selfroute, read by ffn2_decode1
  uint64_t* dec_w;         // [8] blob pointer of the u-th active expert (ascending expert id)
  float* dec_cw;           // [8] the token's combine weight of that expert
  // ffn_gemm_ring2 with a split tail (set by its launcher): 1-D grid over units = (expert slot, 128-row block); units from
  // ring2_split on are dealt to TWO workgroups of four working waves each (64 rows), so a half-empty last round fills the chip
  int ring2_nblk;          // row blocks per expert (0: the plain 2-D grid)
  int ring2_split;
  // upper bound of the row index the stage reads from `in` (engine: max_tokens * (K + 1)); 0 = unknown.  ffn_gemm_ring2 keeps
  // element offsets into `in` in 32 bits: its launchers decline a stage whose rows_bound * ld_in does not fit (round-4 advice)
  int64_t rows_bound;
};

// ---- which form of ffn_gemm_ring2 a stage takes: pure host logic, shared by the launchers (ffn_gemm.hip) and the introspection
// export moeinf_ffn_ring2_form (engine.cpp), which tests/test_kernel_selection_cpu.py pins against DESIGN.md section 4.3.
// (Round 4: the launcher once asked for max_rows <= 192 where the sync-free path's estimate is 193 — the kernel silently never
// ran and six experiments measured its predecessor.)
struct Ring2Knobs {
  int enable_bits = 3;   // MOEINF_GEMM_RING2: bit 0 gated stage, bit 1 plain stage
  int min_k = 4096;      // MOEINF_RING_MIN_K
  int max_rows = 340;    // MOEINF_RING2_MAX_ROWS: above, a second pass over the weights begins -> the big-tile kernel
  int min_gated = 0;     // MOEINF_RING2_MIN_ROWS_GATED (0: where the hybrid kernel stops)
  int min_plain = 16;    // MOEINF_RING2_MIN_ROWS_PLAIN
  int tail = 1;          // MOEINF_RING2_TAIL: split a half-empty last round of the gated stage into half workgroups
  int hyb_rows = 0;      // MOEINF_GEMM_HYB_ROWS (0: 128 with <= 16 active experts, else 64)
  static int env_or(const char* n, int d) { const char* v = getenv(n); return v && *v ? atoi(v) : d; }
  static Ring2Knobs from_env() {
    Ring2Knobs k;
    k.enable_bits = env_or("MOEINF_GEMM_RING2", k.enable_bits); k.min_k = env_or("MOEINF_RING_MIN_K", k.min_k);
    k.max_rows = env_or("MOEINF_RING2_MAX_ROWS", k.max_rows); k.min_gated = env_or("MOEINF_RING2_MIN_ROWS_GATED", k.min_gated);
    k.min_plain = env_or("MOEINF_RING2_MIN_ROWS_PLAIN", k.min_plain); k.tail = env_or("MOEINF_RING2_TAIL", k.tail);
    k.hyb_rows = env_or("MOEINF_GEMM_HYB_ROWS", k.hyb_rows);
    return k;
  }
};
// rows per expert up to which the hybrid kernel runs (17 ..): 128 when at most 16 experts are active, else 64
inline int hyb_rows_for(int active, const Ring2Knobs& k) { return k.hyb_rows ? k.hyb_rows : (active <= 16 ? 128 : 64); }
struct Ring2Form {
  int ntb = 0;    // 0: not ring2; else token groups per pass: 8 / 12 / 16 (128 / 192 / 256 tokens)
  int tail = 0;   // 1: 1-D grid with a split tail (gated stage only)
  int nblk = 0;   // row blocks (128 rows) per expert
  int split = 0;  // first unit that is dealt to two half workgroups
  int blocks = 0; // workgroups launched
};
// elem_bytes: 2 (bf16 / fp16); f16: the hybrid kernel does not exist for fp16, the gated stage starts at 65 rows there;
// row_groups = ceil(max(R, R_sh) / 16); active = grid.y (upper bound of experts with rows); max_rows: see launch_ffn_stage
inline Ring2Form ring2_form(int elem_bytes, bool f16, int nmat, int K, int K_sh, int row_groups, int active, int max_rows, int num_cus,
                            const Ring2Knobs& k) {
  Ring2Form f;
  if (elem_bytes != 2 || !(k.enable_bits & (nmat == 2 ? 1 : 2))) return f;
  const bool k_ok = (K % 64) == 0 && K >= k.min_k && (K_sh == 0 || ((K_sh % 64) == 0 && K_sh >= k.min_k));
  const int min_rows = nmat == 2 ? (k.min_gated ? k.min_gated : (f16 ? 64 : hyb_rows_for(active, k))) : k.min_plain;
  if (!k_ok || max_rows <= min_rows || max_rows > k.max_rows) return f;
  f.ntb = max_rows <= 128 ? 8 : (max_rows <= 208 ? 12 : 16);
  f.nblk = (row_groups + 7) / 8;
  const int units = f.nblk * active, rem = num_cus > 0 ? units % num_cus : 0;
  f.blocks = units;
  if (nmat == 2 && k.tail && units > num_cus && rem > 0 && rem <= num_cus / 2) {
    f.tail = 1; f.split = units - rem; f.blocks = units + rem;
  }
  return f;
}

// max_rows_per_expert: upper bound of rows any one expert receives (selects the multi-token-tile variant)
hipError_t launch_ffn_stage(const FfnStage& s, int max_active, int max_rows_per_expert, hipStream_t st);
// row-major [R,K] -> MFMA A-operand tiles (see kernels.hip); dst needs tiled_bytes(R,K) bytes
hipError_t launch_retile(const void* src, void* dst, int R, int K, int dtype, hipStream_t st);
// all tensors of one staged blob in one launch: tensor t = (src + src_off[t]) row-major [R, K] -> (dst + dst_off[t]) tiled;
// K == 0: a vector of R 16-byte pieces, copied as is
struct RetileBlob {
  const void* src;
  void* dst;
  int n;
  int64_t src_off[4], dst_off[4];
  int R[4], K[4];
  int src_f8;  // pull form only: the source blob holds fp8 (e4m3fn) elements, the slot bf16 (1 source byte per destination element)
};
hipError_t launch_retile_blob(const RetileBlob& b, int dtype, hipStream_t st);
// the same from PINNED HOST memory (b.src = device-visible host pointer): the tier mover's pull form, `workgroups` x 256 threads
// ts: nullptr, or a 4 x u64 timing record in device memory {start tick of the copy's first launch (written when first != 0), max end tick, finished workgroups, -}
hipError_t launch_pull_retile(const RetileBlob& b, int dtype, int workgroups, hipStream_t st, unsigned long long* ts = nullptr, int first = 1);
inline int64_t tiled_bytes(int64_t R, int64_t K, int dtype) { const int64_t ept = dtype == DT_F32 ? 16 : 32; return ((R + 15) / 16) * ((K + ept - 1) / ept) * 1024; }

struct RouteArgs {
  const void* x;        // [T,H] dtype x_dtype
  const void* gate_w;   // [E,H] dtype gate_dtype
  float* logits;        // [T,E]
  int T, H, E, K;
  int x_dtype, gate_dtype;
  int kind;             // MOEINF_ROUTER_*
  int norm_topk_prob;
  float scale;
  int n_group, topk_group;
  int32_t* topk_idx;    // [T,K]
  float* topk_w;        // [T,K]
  int32_t* pair_valid;  // [T,K]
  int32_t* pair_order;  // [T,K] k-indices sorted by ascending expert id
  float* router_prob;   // [T] (Switch: max prob)
  int v3;               // kind DEEPSEEK only: 1 = DeepSeek-V3's gate (sigmoid scores, e_score_correction_bias, top-2-sum groups; modeling_deepseek_v3 MoEGate)
  const float* e_bias;  // ... its e_score_correction_bias [E] (fp32, device), or nullptr = zeros
  int no_renorm;        // kind MIXTRAL only: 1 = the top-k probabilities are NOT renormalised (Grok / Arctic, grok.py:38-45)
};
hipError_t launch_gate_logits(const RouteArgs& a, hipStream_t st);
hipError_t launch_route_topk(const RouteArgs& a, hipStream_t st);

struct IndexArgs {
  const int32_t* topk_idx;  // [T,K]; entries < 0 are never dispatched
  int idx_stride;           // distance between consecutive entries of topk_idx in int32 units (0/1: contiguous)
  int32_t* pair_valid;      // [T,K] in/out (Switch capacity clears entries); nullptr: all valid
  int T, K, E;
  int rows;                 // batch rows B (T = B*S); capacity applies per row
  int capacity;             // <=0: unlimited
  int shared;               // 1: append the shared pseudo-expert E with all T tokens
  int32_t* counts;          // [E+1]
  int32_t* offsets;         // [E+2]
  int32_t* active;          // [E+1]
  int32_t* n_active;        // scalar
  int32_t* pair_slot;       // [T,K]
  int32_t* slot_token;      // [T*K + T] expert-sorted row -> token id
  int32_t* slot_pair;       // [T*K + T] expert-sorted row -> pair id t*K+k (shared rows: -1)
  int slot_cap;             // mask_index: rows the slot_token/slot_pair buffers hold (<= 0: unchecked)
  int32_t* mirror;          // pinned HOST buffer {n_active, counts[E+1], active[E+1]} written by the kernel itself
                            // (counts pre-zeroed by the host; only active experts' counts are guaranteed written)
};
hipError_t launch_dispatch_index(const IndexArgs& a, hipStream_t st);
// the same index over many workgroups (3 launches) for long prefills; chunk_scratch: [ceil(T*K/1024) * E] ints.
// Not for a.capacity > 0 (Switch per-row capacity is a sequential pass).
hipError_t launch_dispatch_index_wide(const IndexArgs& a, int32_t* chunk_scratch, hipStream_t st);
// dispatch index from a dense router_mask[T,E] (element size 1, 4 or 8 bytes, non-zero = routed)
// keep: nullptr, or E bytes (device-visible): columns with keep[e] == 0 are treated as all-false
hipError_t launch_mask_index(const void* mask, int mask_elem_bytes, int T, int E, const IndexArgs& a, hipStream_t st, const uint8_t* keep = nullptr);
// fused route_topk + dispatch_index in one single-workgroup launch (use for T <= 64)
hipError_t launch_route_index(const RouteArgs& r, const IndexArgs& a, hipStream_t st, const EpFuse* pack = nullptr);
// Decode-sized DeepSeek forwards (bf16, T*K <= 64): the shared expert's FFN rides along with the router.
//   gate_shared1: gate logits + stage 1 of the shared expert (s = its stage-1 descriptor: in = x, row_map = nullptr,
//                 out = h_shared [T, R_sh]);
//   route_shared2: top-k + dispatch index (a.shared must be 0) + stage 2 of the shared expert (s: in = h_shared,
//                 out = y_shared [T, R_sh == H]).
hipError_t launch_gate_shared1(const RouteArgs& a, const FfnStage& s, hipStream_t st);
hipError_t launch_route_shared2(const RouteArgs& r, const IndexArgs& a, const FfnStage& s, hipStream_t st, const EpFuse* pack = nullptr);

// Batch-1 decode of the gated families (bf16, T == 1, K <= 8, T*K <= 64, every owned expert resident): FFN stage 1 that
// routes for itself from the gate logits (no top-k/index launch).  r/a as for launch_route_index (a.shared must be 0),
// s1 = the routed stage-1 descriptor, sh2 = the hidden shared expert's stage-2 descriptor or nullptr.
// Profiling: events armed here ride on the NEXT launch made through the KL macro (kdev.h: every FFN-stage launcher — launch_ffn_stage's
// kernels, launch_ffn1_selfroute, launch_ffn2_decode1, launch_moe_front1, launch_moe_layer1_switch) as hipExtLaunchKernel's start / stop events — the kernel's own begin and end on its dispatch packet.  An event
// RECORD in front of and behind a launch puts the command processor's barrier packets inside the interval (2.5-4 us per launch:
// the round-3..5 bench lines sat that far above rocprofv3's durations).  Thread-local; consumed by one launch.
void arm_kernel_timer(hipEvent_t start, hipEvent_t stop);
bool take_kernel_timer(hipEvent_t* start, hipEvent_t* stop);
inline void disarm_kernel_timer() { hipEvent_t a, b; (void)take_kernel_timer(&a, &b); }  // after a launch that may have failed before taking it
hipError_t launch_ffn1_selfroute(const RouteArgs& r, const IndexArgs& a, const FfnStage& s1, const FfnStage* sh2, hipStream_t st);
// The same for decode batches of 2..8 tokens (bf16 / fp16 gated families, T*K <= 64): the meta block routes every token and
// builds the index, every other workgroup routes the tokens for itself; stage 2 is the generic launch_ffn_stage (combine fused).
// max_active = min(E, T*K).
hipError_t launch_ffn1_selfroute_multi(const RouteArgs& r, const IndexArgs& a, const FfnStage& s1, const FfnStage* sh2, int max_active, hipStream_t st);
// ... and its stage 2 (s2.fuse_combine set, K = s2.comb.K active experts, one token): blob pointers and combine weights
// come from the records the self-routing launch left in s2.dec_w / s2.dec_cw
hipError_t launch_ffn2_decode1(const FfnStage& s2, hipStream_t st);
// A whole batch-1 decode layer (gated family, hidden shared expert) in ONE launch (layer_fused.hip): gate | shared stage 1 |
// meta | self-routing stage 1 | shared stage 2 | stage 2 + combine as workgroups of one grid; what used to be a kernel boundary
// is a counter that only grows.  ctr: LAYER1_CTRS words, LAYER1_CTR_STRIDE apart (one cache line each), zeroed once; launch =
// 1, 2, ... per counter set (every launch of a set must have the same shapes: the targets are launch * arrivals per launch).
constexpr int LAYER1_CTR_STRIDE = 1024, LAYER1_CTRS = 3 + 8;  // (one 4 KB page per counter: polls of different counters land on different memory channels)
struct LayerSync {
  uint32_t* ctr;
  uint32_t launch;
  int64_t timeout_ticks;  // bound of every wait, wall_clock64 ticks (100 MHz); on expiry *err = 4 and the workgroup goes on
  int32_t* err;
  int32_t* err_host;          // pinned, device-visible copy of the flag (nullptr: none): the host reads it on the forward path
  unsigned long long* trace;  // debugging (MOEINF_LAYER1_TRACE=<file>): [workgroup][4] wall-clock ticks (start, first wait over, second wait over, end); else nullptr
  int sleep;                  // s_sleep(2) repetitions between two polls
  int scalar_poll;            // 1: the counters are polled with scalar loads (they live in uncached memory); 0: agent-scope vector loads
  float* part;                // Switch form: [4][H] fp32 partial sums of the split stage-2 reduction
};
// the FRONT of a batch-1 layer of the gated families in one launch: gate | (shared stage 1) | meta | self-routing stage 1 |
// (shared stage 2); stage 2 + combine stay launch_ffn2_decode1.  sh1 / sh2: the hidden shared expert's stages or nullptr.
hipError_t launch_moe_front1(const RouteArgs& r, const IndexArgs& a, const FfnStage* sh1, const FfnStage* sh2, const FfnStage& s1, const LayerSync& sy, hipStream_t st);
// the Switch form (top-1, plain experts, no shared expert): E + 1 + F/16 + 4 * H/16 workgroups of eight waves, all resident at once;
// false: not handled (the caller runs the three launches)
bool launch_moe_layer1_switch(const RouteArgs& r, const IndexArgs& a, const FfnStage& s1, const FfnStage& s2, const LayerSync& sy, int num_cus, int wgs_per_cu, hipStream_t st);
// workgroups of that kernel one CU holds at a time (hipOccupancyMaxActiveBlocksPerMultiprocessor of the instantiation; 0: unknown)
int layer1_switch_wgs_per_cu(int x_dtype, int gate_dtype);

hipError_t launch_combine(const CombineArgs& a, hipStream_t st, const EpWait* wait = nullptr);  // wait: poll these flags first (peer-store exchange)
// out[i] = valid[i] ? idx[i] : -1
hipError_t launch_masked_idx(const int32_t* idx, const int32_t* valid, int32_t* out, int n, hipStream_t st);
// index arrays for "only the shared pseudo-expert E is active, with T rows" (expert-parallel path)
hipError_t launch_shared_only_index(const IndexArgs& a, hipStream_t st);

// residency-table update: table[idx[i]] = val[i], i < n (n <= 16), stream-ordered
struct PokeArgs {
  uint64_t* table;
  int n;
  int32_t idx[16];
  uint64_t val[16];
};
hipError_t launch_poke(const PokeArgs& a, hipStream_t st);

// expert-parallel helpers
// key[p] = valid pair ? topk_idx[p] % ep_size : -1   (destination rank of every (token,k) pair);
// also resets pair_pos[p] = -1 when pair_pos != nullptr
hipError_t launch_ep_dest_key(const int32_t* topk_idx, const int32_t* pair_valid, int32_t* key, int32_t* pair_pos,
                              int n_pairs, int ep_size, hipStream_t st);
struct EpPackArgs {
  const void* x;              // [T,H]
  void* send;                 // [ep_size*cap_rows, ld_send]: H activations + a 16-byte tail whose first int32 is
                              // the expert id of the row (-1 = padding) -> rows and ids travel in ONE all-to-all
  int64_t ld_send;            // elements per send row (H + 16/sizeof(elem))
  int32_t* pair_pos;          // [T*K] row of every pair inside `send`, -1 if not dispatched
  const int32_t* topk_idx;    // [T*K]
  const int32_t* counts;      // [ep_size] rows per destination (from dispatch_index on the dest keys)
  const int32_t* offsets;     // [ep_size+1]
  const int32_t* slot_pair;   // [T*K] destination-sorted slot -> pair id
  int K, H, ep_size, cap_rows, dtype;
};
hipError_t launch_ep_pack(const EpPackArgs& a, hipStream_t st, const EpPeers* peers = nullptr);
// the pack riding in the single-workgroup router launch of a decode-sized forward (launch_route_index /
// launch_route_shared2): on != 0 makes the workgroup that routed and indexed the tokens write the send rows as well
struct EpFuse {
  EpPackArgs a;
  const int32_t* pair_valid;
  int32_t* send_counts;  // optional [ep_size]
  int on;
  EpPeers peers;         // peers.on: the rows go straight into the destination ranks' windows (a.send unused)
};
// compact, destination-sorted send rows (variable-split exchange): row r = r-th pair in destination order
hipError_t launch_ep_pack_compact(const EpPackArgs& a, int n_pairs, hipStream_t st);
// n_pairs <= 64: dest keys + stable ranks + row copy in one launch (counts/offsets/slot_pair of `a` unused);
// send_counts (optional, [ep_size]) receives the rows per destination
hipError_t launch_ep_pack_small(const EpPackArgs& a, const int32_t* pair_valid, int n_pairs, int32_t* send_counts, hipStream_t st,
                                const EpPeers* peers = nullptr);

// Expert-parallel exchange, owner side, decode-sized (<= 64 received row slots, every owned expert resident): one FFN
// stage that INDEXES FOR ITSELF — every workgroup reads the expert ids in the received rows' tails, derives "its"
// expert (the blockIdx.y-th smallest id present) and that expert's rows, and streams the weights once over them: no
// dispatch-index launch between the all-to-all and the FFN.  stage 1: in = recv rows, out = h (expert-sorted rows);
// stage 2: in = h, out = the reply buffer, every row at its ARRIVAL position (no un-sort pass).
struct EpOwnArgs {
  const void* recv;     // [nrows, ld_recv]: H activations + 16-byte tail (first int32 = expert id, -1 = padding)
  int64_t ld_recv;      // elements
  int H;                // activation elements per received row
  int nrows;            // <= 64
  int ep_size, ep_rank;
  int stage;            // 1 or 2
  int max_active;       // grid.y
  int32_t* mirror;      // stage 1 only (optional): pinned routing mirror {n_active, counts[E+1], active[E+1]}
  // stage 1 leaves, per expert slot u, what stage 2 needs ("decode record", as dec_w / dec_cw of the local batch-1 path): stage 2
  // then starts with one round of loads from ordinary memory instead of re-deriving the index from the row tails
  struct Rec { uint64_t w; int32_t cnt, off, present, pad; int32_t rows[64]; };
  Rec* rec;             // [max_active] (nullptr: both stages index for themselves)
  int32_t* tile_done;   // [ceil(R/16)] zeroed counters: stage 2 of the peer-store exchange arrives per column tile first
  EpPeers peers;        // peers.on: recv = this rank's window; stage 1 polls the recv flags (peers.poll), stage 2 stores
                        // every output row straight into its home rank's window and the last workgroup publishes
};
hipError_t launch_ffn_ep_stage(const FfnStage& s, const EpOwnArgs& o, hipStream_t st);
// ---- batch-1 decode over the peer-store exchange, BROADCAST form (every rank brings ONE token; round 4) -------------------
// The routed form needs a router launch between the gate and the exchange (route_index + pack).  With one token per rank the
// home rank instead BROADCASTS (its token's row, its E gate logits) to every rank, and every owner's FFN stage 1 routes for
// itself — for all ep_size tokens — exactly as the local batch-1 path does for one (ffn1_selfroute, route_set_lean): the same
// logits through the same instructions give the same top-k on every rank, so the owners agree with the home rank's combine
// without a word of routing crossing the fabric.  Launches per layer: gate -> stage 1 (block 0: broadcast + home routing;
// others: poll, route ep_size tokens, stream) -> stage 2 (from stage 1's records, outputs stored home) -> combine.
struct EpBcastArgs {
  const void* x;          // [1, H] this rank's token
  int32_t* pair_pos;      // [K] -> ret row of every pair of the home token (owner * cap_rows + position)
  int32_t* mirror;        // pinned routing mirror of the OWNER side (optional)
  EpPeers peers;
};
// stage 1.  r / a: router arguments of the home token (r.logits = its gate logits, written by the gate launch); s1: routed
// stage-1 descriptor (in = this rank's recv region, ld_in = exchange row elements); sh2: hidden shared expert's stage 2 or
// nullptr; rec: stage-2 records.  with_bcast = 0: block 0's work was done by launch_ep_bcast (fallback paths).
hipError_t launch_ffn_epb_stage1(const RouteArgs& r, const FfnStage& s1, const FfnStage* sh2, const EpBcastArgs& b, EpOwnArgs::Rec* rec,
                                 int max_active, int with_bcast, hipStream_t st);
// the broadcast + the home token's routing as a launch of its own
hipError_t launch_ep_bcast(const RouteArgs& r, const EpBcastArgs& b, hipStream_t st);
// owner side, slow path (an owned expert is not resident: the host must see the routing): wait for the broadcasts, route the
// ep_size tokens and write what the ROUTED form would have delivered — rows with expert-id tails — into a local staging buffer
// `recv_like` [ep_size*cap_rows, ld] so that the generic owner path can take over
hipError_t launch_ep_bcast_unpack(const RouteArgs& r, const EpBcastArgs& b, void* recv_like, int64_t ld, int dtype, hipStream_t st);

// peer-store exchange, owner side, generic path (more rows than the self-indexing kernel takes): copy the valid rows of
// y [ep_size*cap_rows, H] (valid = the received row's tail >= 0) into their home ranks' windows and publish
hipError_t launch_ep_push(const void* y, const void* recv, int64_t ld_recv, int H, int dtype, const EpPeers& peers, hipStream_t st);
// transport self-test: rank r writes a tagged pattern into segment r of every peer's two regions and publishes `peers.epoch`
// on both flag sets; check: after both flag sets arrived, every segment p of this rank's regions must hold rank p's tag
hipError_t launch_ep_selftest_send(const EpPeers& peers, int words, hipStream_t st);
hipError_t launch_ep_selftest_check(const EpPeers& peers, int words, int32_t* ok_dev, hipStream_t st);
// [T,K] routing of a caller that kept its own router -> the engine's pair arrays (moeinf_combine): idx < 0 = dropped pair
hipError_t launch_prep_pairs(const int32_t* idx_in, const float* w_in, int T, int K, int32_t* topk_idx, float* topk_w,
                             int32_t* pair_valid, int32_t* pair_order, hipStream_t st);

}  // namespace moeinf
