// ffn_ring2_kernel.h — ffn_gemm_ring2: the register-ring form of the grouped expert-FFN GEMM (device code + its launcher template).
// Included by ffn_gemm_ring2.hip (bf16 instantiations) and ffn_gemm_ring2_f16.hip (fp16): two translation units, because every tile
// form is instantiated per number of token-group pairs present and one unit with both types was the slowest compile of the library.
#pragma once
#include "kdev.h"

namespace moeinf {

__device__ __forceinline__ void ring_load(u32x4& dst, const char* p) {
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
}

// ------------------------------------------------------------------------------------------------
// ffn_gemm_ring2 (round 4): BOTH stages (NMAT = 2: gated, NMAT = 1: plain with the bias / ReLU epilogues) for experts with
// ~33..340 rows and long reductions (K >= 4096, K % 64 == 0), 2-byte types.  Weight-streaming GEMM at ~128 rows per expert:
// 128 flop per weight byte, i.e. bound by the HBM stream as long as the multiply phase hides behind it.
//   * block = 8 waves, ONE block per CU; wave w owns the same 16 rows of the stage's matrix (gated: of BOTH matrices) against
//     NTB token groups (128 / 192 / 256 tokens per pass): the accumulators never leave the registers;
//   * weights go HBM -> VGPRs directly (the tiled layout IS the MFMA A fragment) through a ring of D register stages (a stage =
//     2 k-tiles = 2 or 4 one-KiB tiles per wave); inline-asm loads, so the compiler's vmcnt bookkeeping cannot drain the ring;
//   * activations go L2 -> LDS by global_load_lds in full 128-byte lines (8 rows x 128 B per instruction, source-side XOR
//     swizzle, conflict-free ds_read_b128) through a ring of 3 LDS stages;
//   * ONE raw s_barrier per stage and a COUNTED s_waitcnt: every wave issues the same VM ops in the same order.
// It replaces round 2's ffn_gemm_ring (gated stage; same data path) and, for these shapes, ffn_gemm_lds (plain stage).  What
// the counters said about those two at 512 Mixtral tokens (profiles/r04_pmc_prefill512_ring_lds_before_ring2.json): fabric
// traffic 1.05x / 1.25x the algorithmic bytes, no LDS bank conflicts, MFMA pipe 31 % / 22 % busy, streams at 4.1 / 3.6 TB/s.
// The ISA showed why.  ffn_gemm_ring at 254 registers had eight left for activation fragments: a wave ran
// `2 x ds_read_b128 -> s_waitcnt lgkmcnt(0) -> 4 MFMAs` twelve times per k-tile — an exposed LDS round trip in front of every
// 64 cycles of matrix work — and all eight VMEM instructions of a stage went out back to back right behind the barrier, from
// all eight waves at once.  ffn_gemm_lds has ONE stage in flight per block (two buffers, `vmcnt(0)` + __syncthreads per
// k-step): 224 stages x one memory latency = the 262 us it took.  Here, per stage:
//   wait (counted vmcnt) -> s_barrier -> fragments of chunk 0 -> for every chunk: [fragments of the NEXT chunk ->
//   one or two of the stage's VMEM instructions -> the chunk's 8 MFMAs];
//   * two fragment sets (2 x CW x 4 registers): an LDS read is always one chunk of MFMAs ahead of its use.  The reads are
//     inline asm with counted lgkmcnt waits of their own — with compiler-scheduled reads every LDS-DMA between a read and its
//     use turned the wait into lgkmcnt(0), i.e. the prefetched chunk was waited for as well;
//   * the stage's VMEM instructions (activation DMA pieces of stage S+2, weight tiles of stage S+D-1) are spread over the
//     chunks — same order in every wave, so the counted wait still holds — and overlap the partner wave's MFMAs;
//   * the number of token-group PAIRS present is a compile-time constant of the pass (switch over instantiations of the whole
//     k-loop): no branch inside a stage, absent pairs are neither fetched nor multiplied (a 140-row expert: 10 groups of 16,
//     not 12 or 16);
//   * registers: accumulators 16 x pairs (gated) / 8 x pairs (plain) + ring D x WL x 4 + fragments 32 / 64.
// The plain stage: a wave owns 16 rows of ONE matrix, a stage is still two k-tiles (2 KiB per wave); 128 rows per workgroup
// = 256 workgroups for Mixtral's down projection, one per CU.
// Mixtral-8x7B, 512 tokens, us per layer: gate/up 441 -> 374 (4.3 -> 5.1 TB/s), down 261 -> 202 (3.7 -> 4.8 TB/s).
// ------------------------------------------------------------------------------------------------
template <int LO, int HI, typename F>
__device__ __forceinline__ void dispatch_np(int np, F&& f) {
  if constexpr (LO >= HI) {
    f(std::integral_constant<int, HI>{});
  } else {
    if (np <= LO) f(std::integral_constant<int, LO>{});
    else dispatch_np<LO + 1, HI>(np, f);
  }
}

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
template <int OFF>
__device__ __forceinline__ void lds_read16(u32x4& dst, uint32_t lds_addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "n"(OFF) : "memory");
}
// s_waitcnt lgkmcnt(N) that carries the first W fragment registers: no MFMA reading them can be scheduled above it
template <int W, int N, int CW>
__device__ __forceinline__ void frag_wait(u32x4 (&f)[CW]) {
  static_assert(W == 2 || W == 4 || W == 6 || W == 8, "chunk widths are whole pairs");
  if constexpr (W == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f[0]), "+v"(f[1]) : "n"(N) : "memory");
  else if constexpr (W == 4) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(N) : "memory");
  else if constexpr (W == 6) asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]) : "n"(N) : "memory");
  else asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "n"(N) : "memory");
}

template <int WL, int N>
__device__ __forceinline__ void ring2_wait(u32x4 (&w)[WL]) {
  static_assert(WL == 2 || WL == 4, "2 or 4 weight tiles per stage");
  if constexpr (WL == 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "n"(N) : "memory");
  else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w[0]), "+v"(w[1]) : "n"(N) : "memory");
}

// (Round 4 built an EARLY form — four-deep LDS ring, stage barrier one stage early, the first fragments of stage S+1 read at the
// end of step S.  Measured in round 5 (profiles/r05_ffn_sweep_ring2_early_vs_shipped.txt): parity green, 512 tokens 363.5-386.8 us
// vs 359.8-384.4 us for this form, 384 / 768 tokens within 1 % — no gain for 24 KiB more LDS, so it was removed.)
template <typename T, int NMAT, int NTB, int D, bool TAIL = false>
__global__ __launch_bounds__(512) void ffn_gemm_ring2_kernel(FfnStage s) {
  static_assert(sizeof(T) == 2, "bf16 / fp16");
  static_assert(D >= 3 && D <= 6, "register ring of 3..6 stages");
  static_assert(NTB % 4 == 0, "whole activation DMA pieces per wave");
  constexpr int NWV = 8, KT = 2, EPT = 32, EPV = 8;
  constexpr int WL = KT * NMAT;             // weight tiles (1 KiB) per wave and stage
  constexpr int XSTAGE = KT * NTB * 1024;   // activation bytes per stage
  constexpr int NX = 3;                       // LDS ring
  constexpr int CW = 8 / NMAT;              // token groups per chunk: 8 MFMAs between two fragment batches
  static_assert(NX * XSTAGE <= 160 * 1024, "the activation ring must fit the 160 KB of LDS of a gfx950 CU (this kernel is built for gfx950 only)");
  __shared__ __attribute__((aligned(16))) char smem[NX * XSTAGE];

  // TAIL: 1-D grid over units (expert slot, row block); units from ring2_split on are shared by two workgroups (half 0 / 1)
  int u = blockIdx.y, bx = blockIdx.x, half = -1;
  if constexpr (TAIL) {
    const int lin = blockIdx.x;
    int unit = lin;
    if (lin >= s.ring2_split) { const int h = lin - s.ring2_split; unit = s.ring2_split + (h >> 1); half = h & 1; }
    u = unit / s.ring2_nblk; bx = unit - u * s.ring2_nblk;
  }
  if (u >= (s.n_active_host >= 0 ? s.n_active_host : *s.n_active)) return;
  const int e = s.active[u];
  const bool sh = (e == s.E);
  const int K = sh ? s.K_sh : s.K;
  const int R = sh ? s.R_sh : s.R;
  const int nrg_total = (R + 15) / 16;
  if (bx * NWV >= nrg_total) return;
  const int cnt = s.counts[e];
  const int off = s.offsets[e];
  const char* W = reinterpret_cast<const char*>(s.wptr[e]);
  if (W == nullptr) {
    if (threadIdx.x == 0 && bx == 0) atomicExch(s.miss_flag, 1);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int KB = K / EPT;
  const int KS = KB / KT;
  const size_t rg_stride = (size_t)KB * 1024;
  // this wave's weight-tile stream(s) (a row group past the end re-reads the last one; its results are dropped)
  // (a half workgroup: waves 0-3 own row groups half*4 .. half*4+3 of the block, waves 4-7 only keep the barriers company)
  const int rg_want = bx * NWV + (half > 0 ? 4 : 0) + wave;
  const int rg = min(rg_want, nrg_total - 1);
  const bool rg_live = rg_want < nrg_total;
  const char* ap[NMAT];
  ap[0] = W + (sh ? s.off_a_sh : s.off_a) + (size_t)rg * rg_stride + lane * 16;
  if (NMAT == 2) ap[NMAT - 1] = W + (sh ? s.off_b_sh : s.off_b) + (size_t)rg * rg_stride + lane * 16;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // fragment read of (token n of group g, k-tile kk): piece 2g + n/8, byte (n%8)*128 + (((kk*4 + q) ^ (n%8)) << 4)
  const int rr = n & 7;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
  uint32_t frag_off[KT];
#pragma unroll
  for (int kk = 0; kk < KT; ++kk) frag_off[kk] = (n >> 3) * 1024 + rr * 128 + ((((kk & 1) * 4 + q) ^ rr) << 4);

  for (int tile0 = 0; tile0 * 16 < cnt; tile0 += NTB) {
    const int ntl = min(NTB, (cnt - tile0 * 16 + 15) / 16);  // token groups present in this pass (block-uniform)
    auto pass = [&](auto npc, auto nwc) {
      constexpr int NG = decltype(npc)::value * 2;  // token groups multiplied in this pass
      constexpr int NWVE = decltype(nwc)::value;     // waves that work (4 in a half workgroup)
      if constexpr (NWVE < NWV) {
        if (wave >= NWVE) {  // same barriers as the working waves, nothing else
          for (int ks = 0; ks < KS; ++ks) __builtin_amdgcn_s_barrier();
          return;
        }
      }
      constexpr int NCH = (NG + CW - 1) / CW;       // chunks per k-tile
      constexpr int NSLOT = KT * NCH;
      constexpr int XPWP = (2 * NG + NWVE - 1) / NWVE;  // activation DMA pieces per wave and stage that hold rows of this pass
      // element offset of this lane's 16 bytes in each of its pieces (8 rows x 128 B; piece id = wave + NWVE * i)
      uint32_t xoff[XPWP];
#pragma unroll
      for (int i = 0; i < XPWP; ++i) {
        const int pg = wave + NWVE * i;
        const int trow = tile0 * 16 + pg * 8 + (lane >> 3);
        const int srow = off + min(trow, cnt - 1);
        const int64_t xrow = s.row_map ? (int64_t)s.row_map[srow] : (int64_t)srow;
        xoff[i] = (uint32_t)(xrow * s.ld_in + (((lane & 7) ^ (lane >> 3)) * EPV));
      }
      constexpr int OPS = XPWP + WL;                  // VMEM instructions per wave and stage
      constexpr int NWAIT = (D == 3 ? WL : 2 * WL) + XPWP;  // what may stay in flight when stage S is consumed (see below)
      f32x4 acc[NG][NMAT];
#pragma unroll
      for (int b = 0; b < NG; ++b)
#pragma unroll
        for (int m = 0; m < NMAT; ++m) acc[b][m] = f32x4{0.f, 0.f, 0.f, 0.f};
      u32x4 wr[D][WL];  // [ring stage][k-tile * NMAT + matrix]
      auto issue_w1 = [&](int ks, u32x4 (&dst)[WL], int t) {  // past the end: re-read the last stage (never multiplied)
        const int kk = t / NMAT, m = t % NMAT;
        ring_load(dst[t], ap[m] + (size_t)(min(ks, KS - 1) * KT + kk) * 1024);
      };
      auto issue_x1 = [&](int ks, int i) {
        __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const T*>(s.in) + ((size_t)xoff[i] + (size_t)min(ks, KS - 1) * KT * EPT)),
                                         (lptr_t)(smem + (ks % NX) * XSTAGE + (wave + NWVE * i) * 1024), 16, 0, 0);
      };
      // Issue order of every wave: prologue W0 X0 W1 X1 W2 .. W(D-2); step S issues X(S+2) then W(S+D-1), spread over its
      // chunks.  When stage S is consumed, what was issued after X(S) (D > 3; after W(S) for D == 3) may still be in flight:
      // D > 3: W(S+D-3) X(S+1) W(S+D-2) = 2 WL + XPWP;  D == 3: X(S+1) W(S+1) = WL + XPWP.
#pragma unroll
      for (int t = 0; t < WL; ++t) issue_w1(0, wr[0], t);
#pragma unroll
      for (int i = 0; i < XPWP; ++i) issue_x1(0, i);
#pragma unroll
      for (int t = 0; t < WL; ++t) issue_w1(1, wr[1], t);
#pragma unroll
      for (int i = 0; i < XPWP; ++i) issue_x1(1, i);
#pragma unroll
      for (int d = 2; d <= D - 2; ++d)
#pragma unroll
        for (int t = 0; t < WL; ++t) issue_w1(d, wr[d], t);
      auto step = [&](int S, u32x4 (&wc)[WL], u32x4 (&wn)[WL]) {
        ring2_wait<WL, NWAIT>(wc);      // this wave's W(S) and X(S) landed
        __builtin_amdgcn_s_barrier();   // everybody's X(S) landed; the LDS buffer the step refills and ring slot (S-1)%D are free
        const uint32_t sbase = lds0 + (S % NX) * XSTAGE;
        uint32_t fa[KT];
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) fa[kk] = sbase + frag_off[kk];
        u32x4 fb[2][CW];
        // the fragment reads are inline asm with counted lgkmcnt waits of their own: with compiler-scheduled reads every
        // LDS-DMA between a read and its use turns the compiler's wait into lgkmcnt(0) — the prefetched chunk would be waited for
        auto read_slot = [&](auto jc, u32x4 (&f)[CW]) {
          constexpr int j = decltype(jc)::value, kk = j / NCH, c = j % NCH;
          static_for<CW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (c * CW + i < NG) lds_read16<(c * CW + i) * 2048>(f[i], fa[kk]);
          });
        };
        read_slot(std::integral_constant<int, 0>{}, fb[0]);
        static_for<NSLOT>([&](auto jc) {
          constexpr int j = decltype(jc)::value, kk = j / NCH, c = j % NCH;
          constexpr int width = (NG - c * CW) < CW ? (NG - c * CW) : CW;
          constexpr int cn = (j + 1) % NCH;
          constexpr int width_next = (j + 1 < NSLOT) ? ((NG - cn * CW) < CW ? (NG - cn * CW) : CW) : 0;
          if constexpr (j + 1 < NSLOT) read_slot(std::integral_constant<int, j + 1>{}, fb[(j + 1) & 1]);
          static_for<(j + 1) * OPS / NSLOT - j * OPS / NSLOT>([&](auto oc) {
            constexpr int o = j * OPS / NSLOT + decltype(oc)::value;
            if constexpr (o < XPWP) issue_x1(S + 2, o);
            else issue_w1(S + D - 1, wn, o - XPWP);
          });
          frag_wait<width, width_next>(fb[j & 1]);  // this chunk's fragments landed; the next chunk's stay in flight
          __builtin_amdgcn_sched_barrier(0);          // the reads and loads above stay above the MFMAs below
          static_for<width>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
#pragma unroll
            for (int m = 0; m < NMAT; ++m) mma16<T>(acc[c * CW + i][m], wc[kk * NMAT + m], fb[j & 1][i]);
          });
          __builtin_amdgcn_sched_barrier(0);
        });
      };
      // unrolled by D: the register ring is indexed statically; past the end the issues are clamped re-reads
      for (int ks = 0; ks < KS; ks += D) {
#pragma unroll
        for (int d = 0; d < D; ++d)
          if (ks + d < KS) step(ks + d, wr[d], wr[(d + D - 1) % D]);
      }
      // the clamped tail issues: the drain names every ring register, so none of them can be handed to another value while a
      // load that nobody reads is still on its way into it
#pragma unroll
      for (int d = 0; d < D; ++d) ring2_wait<WL, 0>(wr[d]);
      // epilogue straight from the accumulators: lane holds 4 consecutive rows of one token
      epi_switch<NMAT>(s.epi, [&](auto epic) {
        constexpr int EPI = decltype(epic)::value;
        const T* bias = reinterpret_cast<const T*>(W + s.off_bias);
        const bool aligned = (s.ld_out & 3) == 0;
#pragma unroll
        for (int b = 0; b < NG; ++b) {
          const int tok = (tile0 + b) * 16 + n;
          if (tok < cnt && rg_live) {
            const int srow = s.out_map ? s.out_map[off + tok] : off + tok;
            epi_quad<T, EPI>(acc[b][0], acc[b][NMAT - 1], bias, rg * 16 + q * 4, R, aligned, reinterpret_cast<T*>(s.out) + (size_t)srow * s.ld_out);
          }
        }
      });
    };
    // the pass body is instantiated per number of token-group pairs present (block-uniform switch)
    if (TAIL && half >= 0) dispatch_np<2, NTB / 2>((ntl + 1) >> 1, [&](auto npc) { pass(npc, std::integral_constant<int, 4>{}); });
    else dispatch_np<2, NTB / 2>((ntl + 1) >> 1, [&](auto npc) { pass(npc, std::integral_constant<int, NWV>{}); });
    __syncthreads();  // the next pass re-uses the LDS ring from stage 0
  }
}

// launch the form ring2_form (kernels.h) chose: 128 / 192 / 256 tokens per pass; the gated stage with a split tail when the last
// round of workgroups would fill at most half of the CUs (Mixtral's gate-up: 112 row blocks x 8 experts = 896 workgroups = 3.5
// rounds on 256 CUs -> the last 128 units go out as 256 half workgroups of four working waves, 64 rows each)
static int ring2_num_cus() {
  static const int ncu = [] { int d = 0, n = 256; if (hipGetDevice(&d) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n > 0 ? n : 256; }();
  return ncu;
}
template <typename T, int NMAT>
static void launch_ring2(const FfnStage& s0, dim3 grid, const Ring2Form& f, hipStream_t st) {
  const dim3 g2((unsigned)f.nblk, grid.y);
  FfnStage s = s0;
  if constexpr (NMAT == 2) {
    if (f.tail) {
      s.ring2_nblk = f.nblk; s.ring2_split = f.split;
      const dim3 g1((unsigned)f.blocks);
      if (f.ntb == 8) KL((ffn_gemm_ring2_kernel<T, 2, 8, 4, true>), g1, dim3(512), 0, st, s);
      else if (f.ntb == 12) KL((ffn_gemm_ring2_kernel<T, 2, 12, 3, true>), g1, dim3(512), 0, st, s);
      else KL((ffn_gemm_ring2_kernel<T, 2, 16, 3, true>), g1, dim3(512), 0, st, s);
      return;
    }
    if (f.ntb == 8) KL((ffn_gemm_ring2_kernel<T, 2, 8, 4>), g2, dim3(512), 0, st, s);
    else if (f.ntb == 12) KL((ffn_gemm_ring2_kernel<T, 2, 12, 3>), g2, dim3(512), 0, st, s);  // (D = 4: 386 vs 380 us and 16 B of scratch)
    else KL((ffn_gemm_ring2_kernel<T, 2, 16, 3>), g2, dim3(512), 0, st, s);
  } else {
    if (f.ntb == 8) KL((ffn_gemm_ring2_kernel<T, 1, 8, 4>), g2, dim3(512), 0, st, s);
    else if (f.ntb == 12) KL((ffn_gemm_ring2_kernel<T, 1, 12, 4>), g2, dim3(512), 0, st, s);  // (D = 6: 199 vs 201 us)
    else KL((ffn_gemm_ring2_kernel<T, 1, 16, 4>), g2, dim3(512), 0, st, s);
  }
}

}  // namespace moeinf
