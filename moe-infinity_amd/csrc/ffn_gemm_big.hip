// ffn_gemm_big.hip — grouped expert GEMM for the COMPUTE-bound regime (more than 128 rows per expert: long prefills).
//
// What limits ffn_gemm_lds / ffn_gemm_ring there (0.63-0.76 PFLOP/s per layer at 4096 tokens, 23-31 % MFMA-busy by PMC):
// 256 tokens against 64-128 weight rows per block and the 16x16x32 MFMA — every B fragment read from LDS feeds two
// MFMAs, a barrier every 32 MFMAs per wave.  Here
//   * block tile 256 x 256: 256 weight rows (gated stage: 128 rows of BOTH matrices, so SiLU*mul stays in registers;
//     plain stage: 256 rows) x 256 tokens, BK = 64 per stage, 8 waves as 2 (rows) x 4 (tokens);
//   * v_mfma_f32_32x32x16_bf16: a wave owns 128 weight rows x 64 tokens = 8 accumulator tiles of 32x32 (128 VGPRs);
//     per 16-deep k-step 4 A + 2 B fragment reads feed 8 MFMAs of 32 cycles each — a third of the LDS bytes per flop of
//     the 16x16x32 kernels, 32 MFMAs (1024 matrix-pipe cycles) per wave between barriers;
//   * both operands through LDS by the asynchronous global->LDS DMA: weights in a 3-deep ring two stages ahead,
//     activations in two buffers one stage ahead (RING3 below), the DMA instructions spread between the MFMAs;
//   * ping-pong schedule (MODE 2, the default): the two waves of a SIMD alternate LOAD slots (fragment reads of one
//     16-deep k-step) and COMPUTE slots (its 8 MFMAs + 2 DMAs), one s_barrier per slot, the second row half one slot
//     behind — on every SIMD one wave's MFMAs run beside the other's LDS traffic (MFMA-busy 46 -> 61 % by PMC);
//   * the output tile leaves through the idle LDS as whole rows (16-byte stores), epilogue kind compile-time;
//   * weight tiles are already MFMA fragments in HBM (the tiled slot layout): a 32-row A fragment is two vertically
//     adjacent 1-KiB tiles, read from the DMA image at tile[(row>>4)] + ((q*16 + (row&15)) * 16) — 256 contiguous bytes
//     per 16 lanes, conflict-free for ds_read_b128;
//   * activations in full 128-byte lines (8 token rows x 128 B per DMA), XOR-swizzled on the SOURCE side with
//     f(piece, row) = (row>>1 & 3) | (piece & 1) << 2 so that the 32-token fragments of the 32x32 MFMA read
//     conflict-free (the (chunk ^ row) swizzle of the 16-token kernels is 2-way conflicting for 32 tokens);
//   * an expert's 256-token passes run as separate workgroups, placed on ONE XCD and dispatched together (1-D grid,
//     id -> (slab, pass) below): the weight slab comes from HBM once, the other passes read it from that XCD's L2; short
//     matrices still fill the chip.
// Replaces the reference's per-expert torch::matmul triple (core/parallel/expert_module.cpp:171-175) for prefill-sized
// batches.  bf16 only; K % 64 == 0.
#include "kdev.h"

#include <type_traits>

namespace moeinf {

typedef f32x16_ f32x16;

// the fragment reads of k-step J of a stage (ping-pong loop below) as inline asm, into register set J & 1
template <int NMAT, int RT, int RGB, int J>
__device__ __forceinline__ void big_read_frags(u32x4 (&af)[2][NMAT][RT], u32x4 (&bf)[2][2], const uint32_t (&a_addr)[NMAT][RT], const uint32_t (&b_base)[2]) {
  constexpr int kk = J >> 1, ks2 = J & 1;
#pragma unroll
  for (int m = 0; m < NMAT; ++m)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[J & 1][m][rt]) : "v"(a_addr[m][rt]), "n"(kk * NMAT * RGB * 1024 + ks2 * 512));
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)  // chunk (2J + kg) ^ f: the (kg ^ f) part is in b_base, 2J flips bits 5-6 of the byte address
    asm volatile("ds_read_b128 %0, %1" : "=v"(bf[J & 1][tt]) : "v"(b_base[tt] ^ (uint32_t)(J << 5)));
}

// RING3: the weight image gets a THREE-deep LDS ring (3 x 32 KiB) and is fetched two stages ahead, the activation image
// keeps two buffers (2 x 32 KiB) one stage ahead — 160 KiB, all of a CU's LDS; counted s_waitcnt vmcnt(4) + raw
// s_barrier, so the four weight DMAs of stage s+1 stay in flight across the barrier that opens stage s (weights come
// from HBM/MALL and need the longer lead; activations mostly hit in L2).  Past the end the issues are clamped re-reads
// into slots that are already consumed, so the count never varies.
template <typename T, int NMAT, bool RING3, int MODE>  // T: uint16_t = bf16, half_t = fp16 (the same tiles, the f16 matrix instruction)
__global__ __launch_bounds__(512) void ffn_gemm_big_kernel(FfnStage s, int nx, int ny, int nz, int xcd_map, int tail_max, int chunk, int move_short, int num_cus) {
  constexpr int EPT = 32, EPV = 8;
  constexpr int RGB = 16 / NMAT;   // row groups (16 rows) of EACH matrix per block
  constexpr int RT = 4 / NMAT;     // 32-row tiles of each matrix per wave
  constexpr int KK = 2;            // k-tiles (32 k) per stage
  constexpr int A_TILES = KK * NMAT * RGB;  // 32
  constexpr int B_PIECES = 32;              // 256 tokens in pieces of 8 rows x 128 B
  constexpr int ABYTES = A_TILES * 1024, BBYTES = B_PIECES * 1024;  // 32 KiB each per stage
  constexpr int NA = RING3 ? 3 : 2;
  constexpr int BOFF = NA * ABYTES;  // the activation buffers start behind the weight ring
  // the ONLY __shared__ object (a second one de-pipelines the DMA); always all 160 KiB: the short-pass ring (3 x 48 KiB) and the
  // staged output tile (256 tokens x 528 B for the plain stage) need more than the two-buffer form's 128 KiB
  __shared__ __attribute__((aligned(16))) char smem[3 * ABYTES + 2 * BBYTES];

  // 1-D grid, XCD-aware: workgroup id -> XCD id % 8 (observed dispatch rule; a wrong guess costs speed, never
  // correctness).  The token passes of ONE weight slab (row block bx of expert slot u, g = u * nx + bx) get ids 8 apart —
  // the same XCD, dispatched together — so the slab streams from HBM once and the other passes hit it in that XCD's L2.
  int pass0, g;
  if (xcd_map) {
    // Slabs are dealt to the XCDs (= id % 8) in CHUNKS of `chunk` (4) consecutive slabs in (expert, row block) order, so
    // the ~32 workgroups an XCD runs at a time are a few slabs x (nz - 1) passes of ONE expert: they stream those weight slabs
    // and nz - 1 activation tiles between them through that XCD's L2.  Dealt one by one, the down projection (16 slabs
    // per expert) shared each activation tile between only two workgroups of an XCD; one contiguous run per XCD puts
    // whole experts — and the DeepSeek shared expert's 16 passes — on one XCD (measured: gate/up 586 -> 1 022 us).
    // ... and the LAST pass of every slab (the ragged one: 1..256 tokens, or none) gets the highest ids: the full passes
    // fill whole rounds of the chip first, the short ones pack the final partial round (longest jobs first).
    const int per = chunk * (((nx * ny + chunk - 1) / chunk + 7) / 8);  // slab positions per XCD
    const int nfull = per * 8 * (nz - 1);
    int xcd, loc;
    if ((int)blockIdx.x < nfull) {
      const int tq = blockIdx.x >> 3;
      xcd = blockIdx.x & 7; pass0 = tq % (nz - 1); loc = tq / (nz - 1);
    } else if ((int)blockIdx.x < nfull + per * 8) {
      const int idb = (int)blockIdx.x - nfull;
      xcd = idb & 7; pass0 = nz - 1; loc = idb >> 3;
    } else {
      // SHORT region (round 6; move_short >= 1): the very end of the grid, one workgroup per slab, for a slab's SHORT last pass
      // when it is moved behind the full passes (below)
      const int idb = (int)blockIdx.x - nfull - per * 8;
      xcd = idb & 7; pass0 = -1; loc = idb >> 3;  // (the pass index comes from the device-side count)
    }
    g = ((loc / chunk) * 8 + xcd) * chunk + loc % chunk;
  } else {  // MOEINF_GEMM_BIG_XCD=0 (A/B): passes adjacent in id, i.e. spread over the XCDs
    pass0 = blockIdx.x % nz; g = blockIdx.x / nz;
  }
  if (g >= nx * ny) return;
  const int u = g / nx, bx = g - u * nx;
  if (u >= (s.n_active_host >= 0 ? s.n_active_host : *s.n_active)) return;
  const int e = s.active[u];
  const bool sh = (e == s.E);
  const int K = sh ? s.K_sh : s.K;
  const int R = sh ? s.R_sh : s.R;
  const int nrg_total = (R + 15) / 16;
  const int rg0 = bx * RGB;
  if (rg0 >= nrg_total) return;
  const int cnt = s.counts[e];
  const int off = s.offsets[e];
  const char* W = reinterpret_cast<const char*>(s.wptr[e]);
  if (W == nullptr) {
    if (threadIdx.x == 0 && bx == 0 && pass0 == 0) atomicExch(s.miss_flag, 1);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int KB = K / EPT;  // K % 64 == 0 (checked by the launcher)
  const int KS = KB / KK;
  // SHORT LAST PASSES BEHIND THE FULL ONES, WHEN THAT PAYS (round 6).  A short last pass (up to tail_max tokens) IN ITS SLOT runs
  // beside its slab's full passes on the same XCD and rides their weight stream through L2 — cheap.  But when the full passes are
  // two or more EXACT rounds of the chip (4 096 Mixtral tokens: 512 = 2 x 256), the short passes scattered between them open a
  // third round for everybody; then — and only then — they are run from the SHORT region at the end of the grid (longest jobs
  // first): down projection 1 004 -> 960 us at 4 096 tokens, 780 -> 753 at 3 840, 1 015 -> 992 at 4 224.  Unconditionally
  // moved they lose 5-60 % at 1 536 / 2 048 / 3 072 tokens (alone at the end a short pass streams its slab again, latency-bound),
  // and SPLITTING their reduction over idle CUs (fp32 partials, last arriver adds) never beat moving them: both measured,
  // profiles/r06_big_gemm_short_passes_moved_and_split_ab.txt.  Every workgroup concerned derives the same decision from the
  // device-side counts.
  const bool can_move = NMAT == 1 && move_short >= 1 && xcd_map && !sh;
  const int pe_own = (cnt + 255) >> 8;
  const bool short_last = can_move && pe_own >= 1 && pe_own <= nz && cnt - (pe_own - 1) * 256 <= tail_max;
  if (pass0 < 0 || (short_last && pass0 == pe_own - 1)) {
    if (!short_last) return;  // (SHORT region: nothing to run for this slab)
    bool moved = move_short >= 2;  // (2: always, A/B)
    if (!moved) {
      const int nact = s.n_active_host >= 0 ? s.n_active_host : *s.n_active;
      int n_full = 0;
      for (int u0 = 0; u0 < nact; u0 += 64) {  // every wave computes it (64 experts per step): no LDS, no barrier
        const int uu = u0 + lane;
        int f = 0;
        if (uu < nact) {
          const int ee = s.active[uu];
          const int c = s.counts[ee];
          const int pe = (c + 255) >> 8;
          const int nxe = (((ee == s.E ? s.R_sh : s.R) + 15) / 16 + RGB - 1) / RGB;
          const bool shrt = ee != s.E && pe >= 1 && pe <= nz && c - (pe - 1) * 256 <= tail_max;
          f = nxe * (pe - (shrt ? 1 : 0));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) f += __shfl_xor(f, o);
        n_full += f;
      }
      moved = n_full >= 2 * num_cus && n_full % num_cus == 0;
    }
    if ((pass0 < 0) != moved) return;  // the slot's workgroup runs it in place, or the SHORT region's does — never both
    pass0 = pe_own - 1;
  }

  const size_t rg_stride = (size_t)KB * 1024;
  const char* wbase[NMAT];
  wbase[0] = W + (sh ? s.off_a_sh : s.off_a);
  if (NMAT == 2) wbase[NMAT - 1] = W + (sh ? s.off_b_sh : s.off_b);
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  // this wave's DMA work per stage: A tiles t = wave + 8i, B pieces pc = wave + 8i (i < 4)
  const char* asrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = wave + 8 * i;
    const int rg_l = t % RGB, m = (t / RGB) % NMAT, kk = t / (RGB * NMAT);
    asrc[i] = wbase[m] + (size_t)min(rg0 + rg_l, nrg_total - 1) * rg_stride + (size_t)kk * 1024 + lane * 16;  // + ks * KK * 1024 per stage
  }
  // fragment read offsets inside a stage (bytes)
  const int row32 = lane & 31, kg = lane >> 5;
  const int wrg0 = wm * (RGB / 2);
  int a_off[NMAT][RT];  // + (kk * NMAT * RGB) * 1024 + (ks2 * 2 * 16) * 16 per k16 step
#pragma unroll
  for (int m = 0; m < NMAT; ++m)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      a_off[m][rt] = (m * RGB + wrg0 + rt * 2 + (row32 >> 4)) * 1024 + (kg * 16 + (row32 & 15)) * 16;
  int b_off[2], b_f[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int p_local = wn * 8 + tt * 4 + (row32 >> 3), r8 = row32 & 7;
    b_off[tt] = p_local * 1024 + r8 * 128;
    b_f[tt] = ((r8 >> 1) & 3) | ((p_local & 1) << 2);
  }

  for (int tile0 = pass0 * 16; tile0 * 16 < cnt; tile0 += nz * 16) {
    // activation rows this wave DMA-loads: 8-row pieces pc = wave + 8i, source chunk swizzled
    const T* xrp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pc = wave + 8 * i, r8 = lane >> 3;
      const int trow = tile0 * 16 + pc * 8 + r8;
      const int srow = off + min(trow, cnt - 1);
      const int64_t xrow = s.row_map ? (int64_t)s.row_map[srow] : (int64_t)srow;
      const int f = ((r8 >> 1) & 3) | ((pc & 1) << 2);
      xrp[i] = reinterpret_cast<const T*>(s.in) + xrow * s.ld_in + (((lane & 7) ^ f) * EPV);
    }
    f32x16 acc[NMAT][RT][2];
#pragma unroll
    for (int m = 0; m < NMAT; ++m)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[m][rt][tt][i] = 0.f;

    // DMA piece i of a stage: i < 4 activation pieces (8 token rows x 128 B each lane-group), i >= 4 weight tiles
    auto issue_a1 = [&](int ks, int slot, int i) {
      __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + (size_t)ks * KK * 1024), (lptr_t)(smem + slot * ABYTES + (wave + 8 * i) * 1024), 16, 0, 0);
    };
    auto issue_b1 = [&](int ks, int slot, int i) {
      __builtin_amdgcn_global_load_lds((gptr_t)(xrp[i] + (size_t)ks * KK * EPT), (lptr_t)(smem + BOFF + slot * BBYTES + (wave + 8 * i) * 1024), 16, 0, 0);
    };
    auto issue_a = [&](int ks, int slot) {
#pragma unroll
      for (int i = 0; i < 4; ++i) issue_a1(ks, slot, i);
    };
    auto issue_b = [&](int ks, int slot) {
#pragma unroll
      for (int i = 0; i < 4; ++i) issue_b1(ks, slot, i);
    };

    // SHORT LAST PASS.  An expert's rows are cut into 256-token passes; the last one holds anything from 1 to 256 tokens and
    // as a full tile costs a full pass (at 4 096 Mixtral tokens half the experts have ~30 tokens in a fifth pass: 12 % of
    // the gated stage's MFMA work, and the down projection's 576 workgroups are 2.25 rounds of 256 CUs because of them).
    // Up to tail_max (128) tokens run this variant instead: all eight waves along the ROW dimension (wave w: 32 rows x up
    // to four 32-token column tiles; gated stage: an A fragment made of row group w of BOTH matrices — gate rows in its
    // first 16 rows, up rows in the last 16 land in the same lane's registers i and i + 8), a quarter or half of the
    // MFMAs, half the activation bytes, a three-deep 48-KiB stage ring fetched two stages ahead (the pass is bound by
    // the weight stream, not by the matrix pipe).  Accumulators: the first four tiles of acc.
    const int ntok = min(256, cnt - tile0 * 16);
    const bool tail = ntok <= tail_max;  // block-uniform
    const int nct = (ntok + 31) >> 5;    // 32-token column tiles in a tail pass
    constexpr int TSTAGE = ABYTES + 16 * 1024;
    if (tail) {
      const int a_off_t = (NMAT == 2 ? ((row32 >> 4) * RGB + wave) : (2 * wave + (row32 >> 4))) * 1024 + (kg * 16 + (row32 & 15)) * 16;
      const int b_off_t = (row32 >> 3) * 1024 + (row32 & 7) * 128;  // + column tile * 4096
      const int b_f_t = (((row32 & 7) >> 1) & 3) | (((row32 >> 3) & 1) << 2);
      auto issue_t = [&](int ks, int slot) {
        char* base = smem + slot * TSTAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + (size_t)ks * KK * 1024), (lptr_t)(base + (wave + 8 * i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
          __builtin_amdgcn_global_load_lds((gptr_t)(xrp[i] + (size_t)ks * KK * EPT), (lptr_t)(base + ABYTES + (wave + 8 * i) * 1024), 16, 0, 0);
      };
      issue_t(0, 0);
      issue_t(min(1, KS - 1), 1);
      for (int ks = 0; ks < KS; ++ks) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // stage ks landed (this wave's share); stage ks+1 may be in flight
        __builtin_amdgcn_s_barrier();                      // ... everybody's; stage ks-1 is consumed
        issue_t(min(ks + 2, KS - 1), (ks + 2) % 3);        // past the end: clamped re-reads into a consumed slot, the count stays fixed
        const char* ab = smem + (ks % 3) * TSTAGE;
        const char* bb = ab + ABYTES + b_off_t;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32x4 af = *reinterpret_cast<const u32x4*>(ab + a_off_t + (j >> 1) * NMAT * RGB * 1024 + (j & 1) * 512);
          const int ch = ((j * 2 + kg) ^ b_f_t) << 4;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (c < nct) {
              const u32x4 bfr = *reinterpret_cast<const u32x4*>(bb + c * 4096 + ch);
              acc[0][c >> 1][c & 1] = mma32<T>(af, bfr, acc[0][c >> 1][c & 1]);
            }
          }
        }
      }
    } else
    if constexpr (MODE >= 2) {
      // Ping-pong: the two waves of a SIMD (wave w and w + 4, i.e. the row halves wm = 0 / 1) alternate between a LOAD
      // slot (fragment reads of one 16-deep k-step + two DMAs, retired before the slot ends) and a COMPUTE slot (its
      // eight MFMAs), one s_barrier per slot; the wm = 1 half runs one slot behind, so on every SIMD one wave's MFMAs run
      // beside the other's LDS / DMA issue instead of both doing the same thing at the same time.
      //   slot:   0   1   2   3   4   5   6   7  | 8
      //   wm=0:   L0  C0  L1  C1  L2  C2  L3  C3 | L0'        (Lj: k-step j of the stage, Cj: its MFMAs)
      //   wm=1:   C3p L0  C0  L1  C1  L2  C2  L3 | C3  L0' ...
      // A wave's eight DMAs per stage (d0..d3: its share of stage s+1's activations, d4..d7: of stage s+2's weights) are
      // issued two per COMPUTE slot.  Hazards: stage s+1 is first read in slot 8 (wm = 0), so every wave retires its share
      // of it before the barrier that ends slot 7 — wm = 0 after C3 (all eight issued: vmcnt(4)), wm = 1 after L3 (d0..d5
      // issued, its C3 is slot 8: vmcnt(2); the weights of stage s+1 are older and retire first).  The buffers the DMAs of
      // stage s overwrite were last read in stage s-1's L3 of wm = 1 (slot 7), retired by that slot's lgkmcnt(0), two
      // barriers before the first DMA issue of stage s (slot 1).
      static_assert(MODE < 2 || RING3, "ping-pong uses the 3-deep weight ring");
      issue_a(0, 0); issue_b(0, 0); issue_a(min(1, KS - 1), 1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (wm == 1) __builtin_amdgcn_s_barrier();  // one slot behind
      for (int ks = 0; ks < KS; ++ks) {
        const char* abase = smem + (ks % 3) * ABYTES;
        const char* bbase = smem + BOFF + (ks & 1) * BBYTES;
        auto issue_next = [&](int d) {
          if (d < 4) issue_b1(min(ks + 1, KS - 1), (ks + 1) & 1, d);
          else issue_a1(min(ks + 2, KS - 1), (ks + 2) % 3, d - 4);
        };
        u32x4 af[2][NMAT][RT], bf[2][2];
        // fragment reads as inline asm: the compiler's own wait insertion puts lgkmcnt(0) in front of every MFMA group that
        // follows LDS reads issued after the ones it needs (it does not count reads past a pending LDS DMA), which would
        // make each COMPUTE slot wait for the NEXT step's fragments; the counted waits below are the only ones
        const uint32_t a_lds = (uint32_t)(uintptr_t)(lptr_t)abase, b_lds = (uint32_t)(uintptr_t)(lptr_t)bbase;
        uint32_t a_addr[NMAT][RT], b_base[2];
#pragma unroll
        for (int m = 0; m < NMAT; ++m)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) a_addr[m][rt] = a_lds + a_off[m][rt];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) b_base[tt] = (b_lds + b_off[tt]) | (uint32_t)(((kg ^ b_f[tt]) & 7) << 4);  // 128-byte aligned base | chunk
        auto slot_pair = [&](auto jc) {
          constexpr int j = decltype(jc)::value;
          // LOAD slot j: the fragments of k-step j+1 are requested one slot ahead (they land under the MFMAs of step j), so
          // only L0 waits out an LDS round trip; L3 requests nothing — the next stage is not known to have landed yet
          big_read_frags<NMAT, RT, RGB, j>(af, bf, a_addr, b_base);
          if constexpr (MODE == 3) {
            // MODE 3 (round 6): the stage's eight DMAs ride in the LOAD slots, two behind each k-step's fragment reads (their
            // issue — address processing for 64 lanes each — then overlaps the reads' LDS round trip and the partner wave's bare
            // MFMAs, instead of stalling this wave between its own MFMAs while the matrix pipe drains).  Same buffers, same
            // hazards: a LOAD slot of stage s starts after the barrier that ended the last read of stage s-1.
            __builtin_amdgcn_sched_barrier(0);
            issue_next(j * 2);
            issue_next(j * 2 + 1);
          }
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          // wm = 1 has issued d0..d5 of this stage by now (its C3 is slot 8): all but d4, d5 — i.e. its share of stage s+1's
          // activations and, older, of stage s+1's weights — must have landed before the barrier that ends slot 7
          // (MODE 3: d0..d7 are issued by now: all but the newest four)
          if (j == 3 && wm == 1) { if constexpr (MODE == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < NMAT; ++m)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
              for (int tt = 0; tt < 2; ++tt) {
                acc[m][rt][tt] = mma32<T>(af[j & 1][m][rt], bf[j & 1][tt], acc[m][rt][tt]);
                // the stage's eight DMAs ride in the COMPUTE slots, one behind every fourth MFMA (32 matrix-pipe cycles
                // each cover the issue): in the LOAD slot they would lengthen the slot the matrix pipe waits for
                const int idx = (m * RT + rt) * 2 + tt;
                if (MODE == 2 && (idx & 3) == 3) {
                  __builtin_amdgcn_sched_barrier(0);
                  issue_next(j * 2 + (idx >> 2));
                  __builtin_amdgcn_sched_barrier(0);
                }
              }
          if (j == 3 && wm == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
        };
        slot_pair(std::integral_constant<int, 0>{});
        slot_pair(std::integral_constant<int, 1>{});
        slot_pair(std::integral_constant<int, 2>{});
        slot_pair(std::integral_constant<int, 3>{});
      }
      if (wm == 0) __builtin_amdgcn_s_barrier();  // re-align the halves
    } else {
    if constexpr (RING3) {
      issue_a(0, 0); issue_b(0, 0); issue_a(min(1, KS - 1), 1);
    } else {
      issue_a(0, 0); issue_b(0, 0);
    }
    for (int ks = 0; ks < KS; ++ks) {
      const char *abase, *bbase;
      // the eight DMAs a wave issues per stage, in issue order d = 0..7: activations of stage ks+1 first, then the weights
      // (RING3: of stage ks+2 — the newest four, the ones vmcnt(4) leaves in flight)
      auto issue_next = [&](int d) {
        if constexpr (RING3) {
          if (d < 4) issue_b1(min(ks + 1, KS - 1), (ks + 1) & 1, d);
          else issue_a1(min(ks + 2, KS - 1), (ks + 2) % 3, d - 4);
        } else {
          if (ks + 1 < KS) {
            if (d < 4) issue_b1(ks + 1, (ks + 1) & 1, d);
            else issue_a1(ks + 1, (ks + 1) & 1, d - 4);
          }
        }
      };
      if constexpr (RING3) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // all but the newest four DMAs (the weights of stage ks+1) have landed
        __builtin_amdgcn_s_barrier();                      // ... everybody's have, and stage ks-1 is fully consumed
        abase = smem + (ks % 3) * ABYTES;
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA of stage ks has landed
        __syncthreads();                                   // ... everybody's has, and stage ks-1 is fully consumed
        abase = smem + (ks & 1) * ABYTES;
      }
      bbase = smem + BOFF + (ks & 1) * BBYTES;
      // four 16-deep k-steps per stage (k-tile kk = j >> 1, half ks2 = j & 1).  The fragments of step j+1 are read into
      // a second register set BEFORE the eight MFMAs of step j issue (the compiler left to itself reads each pair of
      // fragments just in time behind an lgkmcnt(0): every MFMA pair then waits out an LDS round trip)
      u32x4 af[2][NMAT][RT], bf[2][2];
      auto read_frags = [&](int j, u32x4 (&fa)[NMAT][RT], u32x4 (&fb)[2]) {
        const int kk = j >> 1, ks2 = j & 1;
#pragma unroll
        for (int m = 0; m < NMAT; ++m)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            fa[m][rt] = *reinterpret_cast<const u32x4*>(abase + a_off[m][rt] + kk * NMAT * RGB * 1024 + ks2 * 512);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) fb[tt] = *reinterpret_cast<const u32x4*>(bbase + b_off[tt] + (((j * 2 + kg) ^ b_f[tt]) << 4));
      };
      read_frags(0, af[0], bf[0]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j + 1 < 4) read_frags(j + 1, af[(j + 1) & 1], bf[(j + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);  // the reads above stay above the MFMAs below
#pragma unroll
        for (int m = 0; m < NMAT; ++m)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
              acc[m][rt][tt] = mma32<T>(af[j & 1][m][rt], bf[j & 1][tt], acc[m][rt][tt]);
              {
                // one DMA behind every fourth MFMA: all eight waves issuing their eight DMAs at the top of the stage queue
                // up at the CU's one address unit (64 x 1 KiB at 64 B/clk ~ 1 000 cycles) with nothing in the matrix pipe —
                // measured +450 us of 1 530 for the gated stage at 4 096 tokens; spread out, a wave that waits at the
                // address unit has four MFMAs (128 cycles) queued and its SIMD partner fills the rest
                const int idx = (m * RT + rt) * 2 + tt;
                if ((idx & 3) == 3) {
                  __builtin_amdgcn_sched_barrier(0);
                  issue_next(j * 2 + (idx >> 2));
                  __builtin_amdgcn_sched_barrier(0);
                }
              }
            }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    }
    if (RING3 || tail) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped issues past the last stage
    // epilogue: the output tile goes through the (now idle) LDS so that the global stores are whole rows.  Straight from
    // the accumulators a lane owns 4 consecutive rows of ONE token — every store instruction touches 32 token rows with
    // 16 bytes each, 4 096 partial-line writes per workgroup: 17 us of a 86-us workgroup (ablation: 1 532 -> 1 254 us for
    // the gated stage at 4 096 tokens without it).  Staged as [token][row] with a 16-byte pad per token, then each wave
    // stores 32 tokens, 1 KiB (4 or 2 full token rows of this block's columns) per instruction.
    __syncthreads();  // every wave is done with the fragment images
    constexpr int OROWS = 256 / NMAT;     // output rows of this block (columns of `out`)
    constexpr int OSTR = OROWS * 2 + 16;  // bytes per token in the staged tile
    auto epilogue_tile = [&](auto rtc, auto epic) {  // rt and the epilogue kind as compile-time constants: the accumulator
      constexpr int rt = decltype(rtc)::value;        // arrays must never be indexed dynamically, and a run-time switch on
      constexpr int EPI = decltype(epic)::value;      // s.epi per element costs more than the arithmetic
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int tokl = wn * 64 + tt * 32 + row32;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          // lane holds, for token (lane & 31), rows 8*g4 + 4*kg + (0..3) of each 32-row tile
          const int rowl = (wrg0 + rt * 2) * 16 + 8 * g4 + 4 * kg;
          float v[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            float a0 = acc[0][rt][tt][g4 * 4 + jj];
            if constexpr (EPI == EPI_GATED_SILU) {
              a0 = DT<T>::round(a0);
              const float bb = DT<T>::round(acc[NMAT - 1][rt][tt][g4 * 4 + jj]);
              const float sl = DT<T>::round(a0 / (1.0f + expf(-a0)));
              a0 = sl * bb;
            }
            v[jj] = a0;  // rounded by the store; bias / ReLU of the plain stages are applied by the row-store loop below
          }
          DT<T>::store4(reinterpret_cast<T*>(smem + tokl * OSTR + rowl * 2), v);
        }
      }
    };
    auto epilogue_all = [&](auto epic) {
      epilogue_tile(std::integral_constant<int, 0>{}, epic);
      epilogue_tile(std::integral_constant<int, 1>{}, epic);
      if constexpr (RT == 4) {
        epilogue_tile(std::integral_constant<int, 2>{}, epic);
        epilogue_tile(std::integral_constant<int, 3>{}, epic);
      }
    };
    {
      if (tail) {
        // wave w's tile: gated — rows 16w..16w+15 of the block's 128 (registers i < 8 gate, i + 8 up); plain — rows 32w..32w+31
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (c < nct) {
            const f32x16& A = acc[0][c >> 1][c & 1];
            const int tokl = c * 32 + row32;
#pragma unroll
            for (int g4 = 0; g4 < (NMAT == 2 ? 2 : 4); ++g4) {
              const int rowl = wave * (NMAT == 2 ? 16 : 32) + 8 * g4 + 4 * kg;
              float v[4];
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) {
                float a0 = A[g4 * 4 + jj];
                if constexpr (NMAT == 2) {
                  a0 = DT<T>::round(a0);
                  const float bb2 = DT<T>::round(A[8 + g4 * 4 + jj]);
                  const float sl = DT<T>::round(a0 / (1.0f + expf(-a0)));
                  a0 = sl * bb2;
                }
                v[jj] = a0;
              }
              DT<T>::store4(reinterpret_cast<T*>(smem + tokl * OSTR + rowl * 2), v);
            }
          }
        }
      } else if constexpr (NMAT == 2) {
        epilogue_all(std::integral_constant<int, EPI_GATED_SILU>{});  // the launcher admits two matrices for the gated stage only
      } else {
        epilogue_all(std::integral_constant<int, EPI_NONE>{});
      }
      __syncthreads();
      constexpr int CPR = OROWS * 2 / 16;  // 16-byte chunks per token: 16 (gated) / 32
      constexpr int TPI = 64 / CPR;        // tokens per wave instruction
      const int c = lane % CPR, tl = lane / CPR;
      const int orow0 = rg0 * 16 + c * 8;
      const bool has_bias = NMAT == 1 && (s.epi == EPI_BIAS || s.epi == EPI_BIAS_RELU);
      const bool has_relu = NMAT == 1 && (s.epi == EPI_RELU || s.epi == EPI_BIAS_RELU);
      float bias8[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj)
        bias8[jj] = has_bias ? DT<T>::load(reinterpret_cast<const T*>(W + s.off_bias) + min(orow0 + jj, R - 1)) : 0.f;
#pragma unroll 4
      for (int it = 0; it < 32 / TPI; ++it) {
        const int tokl = wave * 32 + it * TPI + tl;
        const int tok = tile0 * 16 + tokl;
        if (tok < cnt && orow0 < R) {
          const int srow = s.out_map ? s.out_map[off + tok] : off + tok;
          T* op = reinterpret_cast<T*>(s.out) + (size_t)srow * s.ld_out + orow0;
          u32x4 v = *reinterpret_cast<const u32x4*>(smem + tokl * OSTR + c * 16);
          if (has_bias || has_relu) {  // out = relu(Tr(Tr(acc) + bias)): the staged values are Tr(acc)
            uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int h2 = 0; h2 < 4; ++h2) {
              float lo, hi;
              DT<T>::unpack2(w4[h2], lo, hi);
              if (has_bias) { lo += bias8[h2 * 2]; hi += bias8[h2 * 2 + 1]; }
              if (has_relu) { lo = fmaxf(DT<T>::round(lo), 0.f); hi = fmaxf(DT<T>::round(hi), 0.f); }
              w4[h2] = DT<T>::pack2(lo, hi);
            }
            v = u32x4{w4[0], w4[1], w4[2], w4[3]};
          }
          if (orow0 + 8 <= R) {
            *reinterpret_cast<u32x4*>(op) = v;
          } else {
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
              if (orow0 + jj < R) op[jj] = DT<T>::from_bits((uint16_t)(w4[jj >> 1] >> ((jj & 1) * 16)));
          }
        }
      }
    }
    __syncthreads();  // the next pass re-uses buffer 0
  }
}

// max_rows: (an estimate of) the rows of the busiest expert; the kernel's pass loop covers more
bool launch_ffn_gemm_big(const FfnStage& s, int nmat, dim3 grid, int max_rows, hipStream_t st) {
  if (s.dtype == DT_F32 || (s.K % 64) != 0 || (s.K_sh % 64) != 0 || (s.ld_out % 8) != 0) return false;  // 16-byte row stores
  if ((nmat == 2) != (s.epi == EPI_GATED_SILU)) return false;
  const int rmax = s.R > s.R_sh ? s.R : s.R_sh;
  const int passes = max_rows <= 256 ? 1 : (max_rows + 255) / 256;
  const int nx = (rmax + 255 / nmat) / (256 / nmat), ny = (int)grid.y, nz = passes > 8 ? 8 : passes;
  static const int chunk_env = env_int("MOEINF_GEMM_BIG_CHUNK", 4);
  const int chunk = chunk_env < 1 ? 1 : chunk_env;
  // short last passes of the plain stage with a long reduction run from the END of the grid when the full passes are two or more
  // exact rounds (kernel: SHORT region).  MOEINF_GEMM_BIG_MOVE: 0 = never (the round-5 form), 1 = by that rule (default), 2 = always
  static const int move_env = env_int("MOEINF_GEMM_BIG_MOVE", 1);
  static int num_cus = 0;
  if (!num_cus) { int dev = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || num_cus <= 0) num_cus = 256; }
  const int move_short = (nmat == 1 && s.K >= 4096) ? std::max(0, std::min(2, move_env)) : 0;
  const int per8 = chunk * (((nx * ny + chunk - 1) / chunk + 7) / 8) * 8;
  const dim3 g((unsigned)(per8 * (nz + (move_short ? 1 : 0))));
  static const int xcd_map = env_int("MOEINF_GEMM_BIG_XCD", 1);
  static const int ring3 = env_int("MOEINF_GEMM_BIG_RING3", 1);
  static const int tail_max = env_int("MOEINF_GEMM_BIG_TAIL", 128);  // tokens up to which a last pass runs the short-pass variant (0: never)
  static const int mode = env_int("MOEINF_GEMM_BIG_MODE", 2);  // 2: ping-pong (the two waves of a SIMD alternate load / compute slots), 1: both in step
#define BIGGO(NM, R3, MD) KL((ffn_gemm_big_kernel<uint16_t, NM, R3, MD>), g, dim3(512), 0, st, s, nx, ny, nz, xcd_map, tail_max, chunk, move_short, num_cus)
  if (s.dtype == DT_F16) {  // fp16 experts: the default schedule only (three-deep weight ring, ping-pong)
    if (nmat == 2) KL((ffn_gemm_big_kernel<half_t, 2, true, 2>), g, dim3(512), 0, st, s, nx, ny, nz, xcd_map, tail_max, chunk, move_short, num_cus);
    else KL((ffn_gemm_big_kernel<half_t, 1, true, 2>), g, dim3(512), 0, st, s, nx, ny, nz, xcd_map, tail_max, chunk, move_short, num_cus);
    return true;
  }
  if (ring3) {
    if (mode == 3) { if (nmat == 2) BIGGO(2, true, 3); else BIGGO(1, true, 3); }
    else if (mode == 2) { if (nmat == 2) BIGGO(2, true, 2); else BIGGO(1, true, 2); }
    else { if (nmat == 2) BIGGO(2, true, 1); else BIGGO(1, true, 1); }
  } else {
    if (nmat == 2) BIGGO(2, false, 1); else BIGGO(1, false, 1);
  }
#undef BIGGO
  return true;
}

}  // namespace moeinf
