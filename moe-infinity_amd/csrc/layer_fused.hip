// layer_fused.hip — counters instead of kernel boundaries, where that pays (round 5):
//   * moe_front1_kernel: the FRONT of a batch-1 decode layer of the gated families in one launch — gate | (shared expert stage 1)
//     | meta (router, index, records, mirror) | self-routing stage 1 | (shared expert stage 2); DeepSeek-V2-Lite 0.958 -> 0.928
//     ms/token (two launches per layer instead of three);
//   * moe_layer1_switch_kernel: a whole Switch layer (top-1, plain experts) in ONE launch, 0.285 -> 0.204 ms/token.
// Role = workgroup id; a workgroup only waits for smaller ids and the dispatcher hands workgroups out in id order, so no wait can
// be circular.  Every wait is bounded by the wall clock (MOEINF_LAYER1_TIMEOUT_MS): on expiry the device error flag reads 4 and the
// workgroup goes on — a wrong result that the next forward / sync point reports, never a hung GPU.  The counters only grow: launch
// number n (per engine) waits for n * (arrivals per launch), compared with a signed difference.  They live in UNCACHED memory, one
// 4 KB page each, and are polled on the SCALAR path (s_load glc): a vector load waits behind every weight tile its CU has requested.
// Data that crosses workgroups inside the launch is written with write-through stores that are drained before the arrival, and read
// with agent-scope loads after the wait.
//
// What is NOT here any more (round 6): the whole DeepSeek layer as one persistent launch (moe_layer1_kernel, its item table and
// the stage-2 pre-load helpers).  Built and measured in round 5: parity-green and slower than the two launches it would replace
// (1.09-1.20 vs 0.928 ms/token; profiles/r05_layer1_vs_three_launches_*.txt, timelines profiles/r05_layer1_timeline_*.txt; DESIGN.md
// section 4.5.1 keeps the analysis) — the stage-1 -> stage-2 hand-over is a chip-wide barrier under full load, and a kernel
// boundary is as cheap.  Deleted rather than shipped as 350 lines of opt-in code.
//
// Replaces in the reference: one layer's Python router + dispatch_local + per-expert ATen GEMMs + combine loop
// (moe_infinity/models/deepseek.py:55-136, switch_transformers.py:74-113, core/parallel/expert_module.cpp:24-36,193-204).
#include "kdev.h"

#include <algorithm>
#include <vector>

namespace moeinf {

enum { LC_GATE = 0, LC_META = 1, LC_SH1 = 2, LC_H0 = 3 };  // words of LayerSync::ctr, LAYER1_CTR_STRIDE apart

__device__ __forceinline__ void layer_wait(const LayerSync& sy, const int which, const uint32_t per_launch) {
  if (threadIdx.x == 0) {
    const uint32_t* c = sy.ctr + which * LAYER1_CTR_STRIDE;
    const uint32_t target = sy.launch * per_launch;
    const long long t0 = wall_clock64();
    // The counters live in UNCACHED memory and are polled on the SCALAR path (sy.scalar_poll): a vector load waits its turn
    // behind every weight tile this CU has requested (tens of KB in flight per workgroup, microseconds), the scalar unit has
    // its own way to memory.  glc: not from the scalar cache.
    for (;;) {
      uint32_t v;
      if (sy.scalar_poll) asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(c) : "memory");
      else v = ld_coherent(c);
      if ((int32_t)(v - target) >= 0) break;
      if (wall_clock64() - t0 > sy.timeout_ticks) { atomicExch(sy.err, 4); if (sy.err_host) __hip_atomic_store(sy.err_host, 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
      for (int i = 0; i < sy.sleep; ++i) __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  asm volatile("" ::: "memory");  // later loads stay behind the wait
}
__device__ __forceinline__ void layer_trace(const LayerSync& sy, const int slot) {
  if (sy.trace && threadIdx.x == 0) sy.trace[slot] = (unsigned long long)wall_clock64();
}
// A counter that MANY items arrive at, spread over EIGHT words of its page, 128 B apart (atomics on one address serialise at
// ~20 ns each: 192 arrivals on one word were visible 3.6 us after the last one was issued; eight words polled one after the other
// cost eight uncached round trips, 3.1 us — profiles/r05_layer1_switch_timeline_*.txt): item i arrives at word i % 8, the waiter
// fetches all eight in ONE batch of scalar loads and needs every word at its share of `total`.
constexpr int LAYER1_SPREAD = 8, LAYER1_SPREAD_STRIDE = 32;  // words
__device__ __forceinline__ uint32_t* layer_spread_word(const LayerSync& sy, const int which, const int i) {
  return sy.ctr + which * LAYER1_CTR_STRIDE + (i % LAYER1_SPREAD) * LAYER1_SPREAD_STRIDE;
}
__device__ __forceinline__ void layer_wait_spread(const LayerSync& sy, const int which, const int total) {
  if (threadIdx.x == 0) {
    const uint32_t* c = sy.ctr + which * LAYER1_CTR_STRIDE;
    const long long t0 = wall_clock64();
    for (;;) {
      uint32_t v[LAYER1_SPREAD];
      if (sy.scalar_poll) {
        asm volatile("s_load_dword %0, %8, 0x0 glc\n\ts_load_dword %1, %8, 0x80 glc\n\ts_load_dword %2, %8, 0x100 glc\n\ts_load_dword %3, %8, 0x180 glc\n\t"
                     "s_load_dword %4, %8, 0x200 glc\n\ts_load_dword %5, %8, 0x280 glc\n\ts_load_dword %6, %8, 0x300 glc\n\ts_load_dword %7, %8, 0x380 glc\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(v[0]), "=&s"(v[1]), "=&s"(v[2]), "=&s"(v[3]), "=&s"(v[4]), "=&s"(v[5]), "=&s"(v[6]), "=&s"(v[7]) : "s"(c) : "memory");
      } else {
#pragma unroll
        for (int j = 0; j < LAYER1_SPREAD; ++j) v[j] = ld_coherent(c + j * LAYER1_SPREAD_STRIDE);
      }
      bool ok = true;
#pragma unroll
      for (int j = 0; j < LAYER1_SPREAD; ++j) ok = ok && (int32_t)(v[j] - sy.launch * (uint32_t)((total + LAYER1_SPREAD - 1 - j) / LAYER1_SPREAD)) >= 0;
      if (ok) break;
      if (wall_clock64() - t0 > sy.timeout_ticks) { atomicExch(sy.err, 4); if (sy.err_host) __hip_atomic_store(sy.err_host, 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
      for (int i = 0; i < sy.sleep; ++i) __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  asm volatile("" ::: "memory");
}
// every thread's write-through stores have been acknowledged -> one arrival
__device__ __forceinline__ void layer_arrive(const LayerSync& sy, const int which) {
  wait_stores_acked();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(sy.ctr + which * LAYER1_CTR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The blob pointer of the u-th chosen expert (ascending id) of the one token, derived by THIS workgroup from the gate logits
// (wave 0: route_set_lean, the same arithmetic as the generic router on this path; lane j holds wptr[j], the chosen one is a
// v_readlane away).  Every workgroup derives the same set from the same logits with the same instructions.
__device__ __forceinline__ const char* layer_selfroute(const RouteArgs& r, const FfnStage& s, const int u, unsigned long long* sh_w, int* sh_ok,
                                                        const int round_p_dtype = DT_F32) {
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 64) {
    uint64_t wp = 0;
    if (lane < r.E) wp = s.wptr[lane];
    uint64_t chosen = route_set_lean<true>(r.logits, r.E, r.K, lane, round_p_dtype);
    for (int i = 0; i < u; ++i) chosen &= chosen - 1;  // drop the u smallest ids
    const int e = chosen ? (int)__builtin_ctzll(chosen) : -1;
    uint64_t wsel = 0;
    if (e >= 0) {
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wp, e);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(wp >> 32), e);
      wsel = ((uint64_t)hi << 32) | lo;
    }
    if (lane == 0) { *sh_w = wsel; *sh_ok = e >= 0; }
  }
  __syncthreads();
  return reinterpret_cast<const char*>(*sh_w);
}

// ------------------------------------------------------------------------------------------------
// The FRONT of a batch-1 decode layer of the gated families in one launch (round 5): gate | (shared stage 1) | meta |
// self-routing stage 1 | (shared stage 2) as workgroups of ONE grid — the gate launch and its boundary leave the layer
// (Mixtral: two launches per layer; DeepSeek: two instead of three).  This is the part of the one-launch layer above that the
// timelines showed to work: the GATE hand-over happens while nothing streams yet (flag seen 1 us after the last gate workgroup),
// stage 1 starts 3 us into the launch instead of 8 us into the layer, and the shared expert's two stages overlap it as before.
// What does NOT go into this launch is the stage-1 -> stage-2 hand-over (a chip-wide barrier under full load: a kernel boundary
// is as cheap, section 4.5 of DESIGN.md): stage 2 + combine stay the launch they were (ffn2_decode1[_pair]).
// Role = workgroup id (gate first): a workgroup only waits for smaller ids, the dispatcher hands them out in id order.
// Item bodies = ffn1_selfroute_kernel's / gate_shared1_kernel's (ffn_rows_item<T, 2, 4, U, 1>, gate_body): bit-identical rows.
template <typename T, typename GW, int U>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(96))) void moe_front1_kernel(RouteArgs r, IndexArgs a, FfnStage sh1, FfnStage sh2, FfnStage s1, LayerSync sy,
                                                                                              int round_logits, int n_sh1, int n_sh2) {
  constexpr int NW = 4;
  __shared__ float red[NW][2][256];
  __shared__ double redg[4][1];
  __shared__ unsigned long long sh_w;
  __shared__ int sh_flag;
  static_assert(sizeof(float) * NW * 2 * 256 >= sizeof(int) * (2 * IDX_MAXE + 1), "index scratch aliases the reduction buffer");
  const int lane = threadIdx.x & 63;
  const int K = r.K, E = r.E;
  const int n_rg = (s1.R + 15) / 16;
  int b = blockIdx.x;
  const int tslot = (int)blockIdx.x * 4;
  layer_trace(sy, tslot + 0);
  if (b < E) {  // ---- gate
    gate_body<T, GW, 1, true>(reinterpret_cast<const T*>(r.x), reinterpret_cast<const GW*>(r.gate_w), r.logits, 1, r.H, E, round_logits, redg, b, 0);  // (one token: one reduction, not gate_logits_kernel's four)
    layer_arrive(sy, LC_GATE);
    layer_trace(sy, tslot + 3);
    return;
  }
  b -= E;
  if (b < n_sh1) {  // ---- shared expert, stage 1 (h_shared is read by THIS launch's shared stage 2: write-through + counter)
    const char* W = reinterpret_cast<const char*>(sh1.wptr[sh1.E]);
    ffn_rows_item<T, 2, NW, U, 1, false, true>(sh1, b, W, true, 1, 0, red);
    wait_stores_acked();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(layer_spread_word(sy, LC_SH1, b), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    layer_trace(sy, tslot + 3);
    return;
  }
  b -= n_sh1;
  if (b == 0) {  // ---- meta (one wave): the generic router, the one-wave index, the decode records, the pinned mirror
    if (threadIdx.x >= 64) return;
    layer_wait(sy, LC_GATE, E);
    layer_trace(sy, tslot + 1);
    int* scratch = reinterpret_cast<int*>(&red[0][0][0]);
    Routed o;
    route_core<true>(r, 0, lane, o);
    int my_sel, rank;
    float my_w;
    route_store(r, 0, lane, o, &my_sel, &my_w, &rank);
    if (lane < K && s1.dec_w) {  // read by the NEXT launch (stage 2): plain stores
      s1.dec_w[rank] = my_sel >= 0 ? s1.wptr[my_sel] : 0ull;
      s1.dec_cw[rank] = my_w;
    }
    __threadfence_block();
    index_small(a, scratch, scratch + IDX_MAXE);
    layer_trace(sy, tslot + 3);
    return;
  }
  b -= 1;
  if (b < K * n_rg) {  // ---- routed stage 1 (h is read by the next launch: plain stores)
    const int u = b / n_rg, rg = b - u * n_rg;
    layer_wait(sy, LC_GATE, E);
    layer_trace(sy, tslot + 1);
    const char* W = layer_selfroute(r, s1, u, &sh_w, &sh_flag);
    if (!sh_flag) return;
    if (W == nullptr) {  // never on the sync-free path
      if (threadIdx.x == 0 && rg == 0) atomicExch(s1.miss_flag, 1);
      return;
    }
    ffn_rows_item<T, 2, NW, U, 1>(s1, rg, W, false, 1, u, red, 0);
    layer_trace(sy, tslot + 3);
    return;
  }
  b -= K * n_rg;
  {  // ---- shared expert, stage 2 (y_shared is read by the next launch: plain stores; h_shared comes from this one)
    const char* W = reinterpret_cast<const char*>(sh2.wptr[sh2.E]);
    layer_wait_spread(sy, LC_SH1, n_sh1);
    layer_trace(sy, tslot + 2);
    ffn_rows_item<T, 1, NW, U, 1, true, false>(sh2, b, W, true, 1, 0, reinterpret_cast<float (*)[1][256]>(&red[0][0][0]));
    layer_trace(sy, tslot + 3);
    (void)n_sh2;
  }
}

// bf16 / fp16 model with the gate in the model dtype or fp32; sh1 / sh2 = the hidden shared expert's stages or nullptr
hipError_t launch_moe_front1(const RouteArgs& r, const IndexArgs& a, const FfnStage* sh1, const FfnStage* sh2, const FfnStage& s1, const LayerSync& sy, hipStream_t st) {
  const int n_rg = (s1.R + 15) / 16;
  const int n_sh1 = sh1 ? (sh1->R_sh + 15) / 16 : 0, n_sh2 = sh2 ? (sh2->R_sh + 15) / 16 : 0;
  const dim3 grid(r.E + n_sh1 + 1 + r.K * n_rg + n_sh2);
  // (as launch_ffn1_selfroute: a multi-round grid streams best with FOUR workgroups per CU, capped through dynamic LDS, and
  // four tiles per wave, matrix and batch; a grid that is resident all at once takes eight)
  static const int lds_env = env_int("MOEINF_SR_LDS_KB", -1);
  const size_t dyn = (size_t)(lds_env >= 0 ? lds_env : (grid.x > 4 * 256 ? 30 : 0)) * 1024;
  static const int sr_u_env = env_int("MOEINF_SR_U", 0);
  const int sr_u = sr_u_env ? sr_u_env : (grid.x > 4 * 256 ? 4 : 8);
  const int rl = r.kind != 0 ? 0 : (r.x_dtype == DT_BF16 ? 1 : (r.x_dtype == DT_F16 ? 2 : 0));
  const FfnStage& a1 = sh1 ? *sh1 : s1;
  const FfnStage& a2 = sh2 ? *sh2 : s1;
#define F1(TT, GW, UU) KL((moe_front1_kernel<TT, GW, UU>), grid, dim3(256), dyn, st, r, a, a1, a2, s1, sy, rl, n_sh1, n_sh2)
#define F1U(TT, GW) do { if (sr_u == 8) F1(TT, GW, 8); else F1(TT, GW, 4); } while (0)
  if (s1.dtype == DT_BF16) { if (r.gate_dtype == DT_BF16) F1U(uint16_t, uint16_t); else if (r.gate_dtype == DT_F32) F1U(uint16_t, float); else return hipErrorInvalidValue; }
  else if (s1.dtype == DT_F16) { if (r.gate_dtype == DT_F16) F1U(half_t, half_t); else if (r.gate_dtype == DT_F32) F1U(half_t, float); else return hipErrorInvalidValue; }
  else return hipErrorInvalidValue;
#undef F1U
#undef F1
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The same idea for Switch (top-1, plain ReLU experts, no shared expert; Switch-base-8: 18.9 MB per layer in three launches of
// 3-10 us = pure fixed cost) — and here it PAYS (0.32 -> 0.27 ms/token in its first form): with hardly any traffic in flight a
// flag costs ~1 us.  ONE launch of E + 1 + F/16 + KS * H/16 workgroups of eight waves, every one resident at once (the launcher
// checks: two per CU); role = workgroup id, so a workgroup only waits for smaller ids:
//   gate (E) -> GATE | meta (1): waits GATE -> META | stage 1 (F/16 row groups): waits GATE, routes for itself -> H[0]
//   stage 2 (H/16 column tiles x KS quarters of the reduction): waits GATE, routes for itself, requests ALL its weight tiles, THEN
//   waits for H[0], multiplies, hands its fp32 partial sums to the tile's last arriver, which adds the KS partials in a fixed
//   order, rounds once and writes y and out = Tr(router_prob * y) (top-1: nothing else to combine).
// Why the reduction is split (KS = 4): one CU pulls ~26 GB/s whatever it has in flight — a column tile's 196 KB (Switch-base-8)
// took 7.6 us on one CU (first form: launch span 16.6 us, profiles/r05_layer1_switch_timeline_*.txt); 49 KB take 2.
// Stage 1 is ffn_rows_item<T, 1, 8, 6, 1> (the three launches: sixteen waves); stage 2 adds KS partial sums where the three
// launches add one chain: the same products in another fp32 summation order, ONE rounding each — the 1-ulp row bar and the
// fp32-exact arm hold as they do for any of the kernels (tests/test_gpu_fullsize.py::test_switch_base_8_layer_fp32).
template <typename T, typename GW, int P2, int KS>
__global__ __launch_bounds__(512) void moe_layer1_switch_kernel(RouteArgs r, IndexArgs a, FfnStage s1, FfnStage s2, LayerSync sy) {
  constexpr int NW = 8;
  __shared__ float red[NW][1][256];
  __shared__ double redg[4][1];
  __shared__ unsigned long long sh_w;
  __shared__ int sh_flag;
  static_assert(sizeof(float) * NW * 256 >= sizeof(int) * (2 * IDX_MAXE + 1), "index scratch aliases the reduction buffer");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int E = r.E;
  const int n_rg = (s1.R + 15) / 16;
  int b = blockIdx.x;
  const int tslot = (int)blockIdx.x * 4;
  layer_trace(sy, tslot + 0);
  if (b < E) {  // ---- gate: one logit per workgroup, four waves (the others leave: a finished wave does not count at a barrier)
    if (tid >= 256) return;
    gate_body<T, GW, 1, true>(reinterpret_cast<const T*>(r.x), reinterpret_cast<const GW*>(r.gate_w), r.logits, 1, r.H, E, 0, redg, b, 0);
    layer_arrive(sy, LC_GATE);
    layer_trace(sy, tslot + 3);
    return;
  }
  b -= E;
  if (b == 0) {  // ---- meta: the generic Switch router (top-1 on probabilities cast to the model dtype, router_prob), index, mirror
    if (tid >= 64) return;
    layer_wait(sy, LC_GATE, E);
    layer_trace(sy, tslot + 1);
    int* scratch = reinterpret_cast<int*>(&red[0][0][0]);
    Routed o;
    route_core<true>(r, 0, lane, o);
    int my_sel, rank;
    float my_w;
    route_store(r, 0, lane, o, &my_sel, &my_w, &rank);
    if (lane == 0) {  // stage 2 of THIS launch reads the combine factor (= router_prob, route_core kind 2: w[0] = val[0])
      st_coherent(&s2.dec_w[0], my_sel >= 0 ? (uint64_t)s2.wptr[my_sel] : (uint64_t)0);
      st_coherent(&s2.dec_cw[0], my_w);
    }
    wait_stores_acked();
    if (lane == 0) __hip_atomic_fetch_add(sy.ctr + LC_META * LAYER1_CTR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_block();
    index_small(a, scratch, scratch + IDX_MAXE);
    layer_trace(sy, tslot + 3);
    return;
  }
  b -= 1;
  if (b < n_rg) {  // ---- stage 1: 16 rows of the chosen expert's wi, ReLU
    layer_wait(sy, LC_GATE, E);
    layer_trace(sy, tslot + 1);
    const char* W = layer_selfroute(r, s1, 0, &sh_w, &sh_flag, r.x_dtype);
    if (sh_flag && W == nullptr && tid == 0 && b == 0) atomicExch(s1.miss_flag, 1);
    ffn_rows_item<T, 1, NW, 6, 1, false, true>(s1, b, W, false, (sh_flag && W) ? 1 : 0, 0, red, 0);
    layer_trace(sy, tslot + 2);
    wait_stores_acked();  // (= layer_arrive, at one of the eight words that share the 192 arrivals)
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(layer_spread_word(sy, LC_H0, b), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    layer_trace(sy, tslot + 3);
    return;
  }
  b -= n_rg;
  {  // ---- stage 2: quarter q of the reduction of 16 output columns; the tile's last arriver finishes it
    constexpr int EPV = DT<T>::EPV, EPT = 4 * EPV;
    const int col = b / KS, q = b - col * KS;
    layer_wait(sy, LC_GATE, E);
    layer_trace(sy, tslot + 1);
    const char* W = layer_selfroute(r, s2, 0, &sh_w, &sh_flag, r.x_dtype);
    const bool live = sh_flag && W;
    if (sh_flag && W == nullptr && tid == 0 && b == 0) atomicExch(s2.miss_flag, 1);
    const int KB = s2.K / EPT, KBq = KB / KS;
    const char* a0 = W + s2.off_a + (size_t)col * KB * 1024 + (size_t)q * KBq * 1024 + lane * 16;
    u32x4 wv[P2];
#pragma unroll
    for (int i = 0; i < P2; ++i)
      if (live && wave + i * NW < KBq) wv[i] = ld16_nt(a0 + (size_t)(wave + i * NW) * 1024);
    layer_wait_spread(sy, LC_H0, n_rg);
    layer_trace(sy, tslot + 2);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (live) {
      const T* xr = reinterpret_cast<const T*>(s2.in) + (size_t)q * KBq * EPT + (lane >> 4) * EPV;  // row 0 of h: the one token's expert row
      u32x4 xv[P2];
#pragma unroll
      for (int i = 0; i < P2; ++i)
        if (wave + i * NW < KBq) xv[i] = ld16_coherent(xr + (size_t)(wave + i * NW) * EPT);
#pragma unroll
      for (int i = 0; i < P2; ++i)
        if (wave + i * NW < KBq) mma16<T>(acc, wv[i], xv[i]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wave][0][lane * 4 + j] = acc[j];
    __syncthreads();
    // the 16 sums of this quarter (token column 0 of the tile): lanes with (l & 15) == 0, as in ffn_rows_item
    const int l = tid >> 2, jj = tid & 3;
    const int orow = col * 16 + (l >> 4) * 4 + jj;
    const bool mine = tid < 256 && (l & 15) == 0 && orow < s2.R;
    if (mine) {
      float s0 = 0.f;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) s0 += red[ww][0][tid];
      st_coherent(&sy.part[(size_t)q * s2.R + orow], s0);
    }
    wait_stores_acked();
    __syncthreads();
    if (tid == 0) sh_flag = __hip_atomic_fetch_add(&s2.tile_done[col], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == KS - 1;
    __syncthreads();
    if (sh_flag) {  // last arriver of the column tile: the KS partial sums in ascending q, ONE rounding, then the block's combine
      layer_wait(sy, LC_META, 1);
      if (mine && live) {
        float v = 0.f;
#pragma unroll
        for (int qq = 0; qq < KS; ++qq) v += ld_coherent(&sy.part[(size_t)qq * s2.R + orow]);
        v = DT<T>::round(v);
        DT<T>::store(reinterpret_cast<T*>(s2.out) + orow, v);  // y row 0 (expert-sorted row of the one token)
        DT<T>::store(reinterpret_cast<T*>(s2.comb.out) + orow, DT<T>::round(ld_coherent(&s2.dec_cw[0]) * v));
      }
      if (tid == 0) s2.tile_done[col] = 0;
    }
    layer_trace(sy, tslot + 3);
  }
}

// Switch: x and experts in one dtype (fp32 for Switch-base, bf16), gate in the model dtype or fp32.  false: not handled (more
// workgroups than the chip holds at once — they must all be resident —, a reduction the split does not cover, an odd dtype mix).
int layer1_switch_wgs_per_cu(int x_dtype, int gate_dtype) {
  int n = 0;
  hipError_t e = hipErrorInvalidValue;
  if (x_dtype == DT_F32) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, moe_layer1_switch_kernel<float, float, 6, 4>, 512, 0);
  else if (gate_dtype == DT_BF16) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, moe_layer1_switch_kernel<uint16_t, uint16_t, 6, 4>, 512, 0);
  else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, moe_layer1_switch_kernel<uint16_t, float, 6, 4>, 512, 0);
  if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

bool launch_moe_layer1_switch(const RouteArgs& r, const IndexArgs& a, const FfnStage& s1, const FfnStage& s2, const LayerSync& sy, int num_cus, int wgs_per_cu, hipStream_t st) {
  constexpr int KS = 4, P2 = 6, NW = 8;
  const int n_rg = (s1.R + 15) / 16, n_col = (s2.R + 15) / 16;
  const dim3 grid(r.E + 1 + n_rg + KS * n_col);
  const int ept = s2.dtype == DT_F32 ? 16 : 32;
  // every workgroup resident at once: what the chip HOLDS is asked (occupancy query), not assumed; at most two per CU even if more fit
  const int per_cu = wgs_per_cu > 0 ? (wgs_per_cu < 2 ? wgs_per_cu : 2) : 0;
  if (per_cu == 0 || (int)grid.x > per_cu * num_cus || (s2.K % (ept * KS)) != 0 || (s1.K % ept) != 0 || s2.K / ept / KS > NW * P2 || r.K != 1 || s2.dtype == DT_F16 || !sy.part) return false;
  if (s2.dtype == DT_F32) {
    if (r.gate_dtype != DT_F32) return false;
    KL((moe_layer1_switch_kernel<float, float, P2, KS>), grid, dim3(512), 0, st, r, a, s1, s2, sy);
  } else {
    if (r.gate_dtype == DT_BF16) KL((moe_layer1_switch_kernel<uint16_t, uint16_t, P2, KS>), grid, dim3(512), 0, st, r, a, s1, s2, sy);
    else if (r.gate_dtype == DT_F32) KL((moe_layer1_switch_kernel<uint16_t, float, P2, KS>), grid, dim3(512), 0, st, r, a, s1, s2, sy);
    else return false;
  }
  return hipGetLastError() == hipSuccess;
}

}  // namespace moeinf
