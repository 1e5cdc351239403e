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
//   * both operands through LDS by the asynchronous global->LDS DMA, two 64-KiB stages: the DMA of stage s+1 is
//     issued right after the barrier that opens stage s and has a whole stage of MFMAs (~1 us) to land;
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

typedef __attribute__((ext_vector_type(16))) float f32x16;

// RING3: the weight image gets a THREE-deep LDS ring (3 x 32 KiB) and is fetched two stages ahead, the activation image
// keeps two buffers (2 x 32 KiB) one stage ahead — 160 KiB, all of a CU's LDS; counted s_waitcnt vmcnt(4) + raw
// s_barrier, so the four weight DMAs of stage s+1 stay in flight across the barrier that opens stage s (weights come
// from HBM/MALL and need the longer lead; activations mostly hit in L2).  Past the end the issues are clamped re-reads
// into slots that are already consumed, so the count never varies.
template <int NMAT, bool RING3, int ABL = 0>
__global__ __launch_bounds__(512) void ffn_gemm_big_kernel(FfnStage s, int nx, int ny, int nz, int xcd_map) {
  typedef uint16_t T;
  constexpr int EPT = 32, EPV = 8;
  constexpr int RGB = 16 / NMAT;   // row groups (16 rows) of EACH matrix per block
  constexpr int RT = 4 / NMAT;     // 32-row tiles of each matrix per wave
  constexpr int KK = 2;            // k-tiles (32 k) per stage
  constexpr int A_TILES = KK * NMAT * RGB;  // 32
  constexpr int B_PIECES = 32;              // 256 tokens in pieces of 8 rows x 128 B
  constexpr int ABYTES = A_TILES * 1024, BBYTES = B_PIECES * 1024;  // 32 KiB each per stage
  constexpr int NA = RING3 ? 3 : 2;
  constexpr int BOFF = NA * ABYTES;  // the activation buffers start behind the weight ring
  __shared__ __attribute__((aligned(16))) char smem[NA * ABYTES + 2 * BBYTES];  // the ONLY __shared__ object (a second one de-pipelines the DMA)

  // 1-D grid, XCD-aware: workgroup id -> XCD id % 8 (observed dispatch rule; a wrong guess costs speed, never
  // correctness).  The nz token passes of ONE weight slab (row block bx of expert slot u) get ids 8 apart — the same XCD,
  // dispatched together — so the slab streams from HBM once and the other passes hit it in that XCD's L2:
  //   id = ((g / 8) * nz + pass) * 8 + g % 8,   g = u * nx + bx
  int pass0, g;
  if (xcd_map) {
    const int xcd = blockIdx.x & 7, tq = blockIdx.x >> 3;
    pass0 = tq % nz; g = (tq / nz) * 8 + xcd;
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

    auto issue_a = [&](int ks, int slot) {
      char* base = smem + slot * ABYTES;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + (size_t)ks * KK * 1024), (lptr_t)(base + (wave + 8 * i) * 1024), 16, 0, 0);
    };
    auto issue_b = [&](int ks, int slot) {
      char* base = smem + BOFF + slot * BBYTES;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(xrp[i] + (size_t)ks * KK * EPT), (lptr_t)(base + (wave + 8 * i) * 1024), 16, 0, 0);
    };

    if constexpr (RING3) {
      issue_a(0, 0); issue_b(0, 0); issue_a(min(1, KS - 1), 1);
    } else {
      issue_a(0, 0); issue_b(0, 0);
    }
    for (int ks = 0; ks < KS; ++ks) {
      const char *abase, *bbase;
      if constexpr (RING3) {
        if constexpr (ABL == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (ABL != 2 && ABL != 4 && ABL < 8) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // all but the newest four DMAs (the weights of stage ks+1) have landed
        if constexpr (ABL != 9) __builtin_amdgcn_s_barrier();                      // ... everybody's have, and stage ks-1 is fully consumed
        if constexpr (ABL != 2 && ABL != 4 && ABL < 8) {
        if constexpr (ABL != 6) issue_b(min(ks + 1, KS - 1), (ks + 1) & 1);
        if constexpr (ABL != 7) issue_a(min(ks + 2, KS - 1), (ks + 2) % 3);
        }
        abase = smem + (ks % 3) * ABYTES;
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA of stage ks has landed
        __syncthreads();                                   // ... everybody's has, and stage ks-1 is fully consumed
        if (ks + 1 < KS) { issue_a(ks + 1, (ks + 1) & 1); issue_b(ks + 1, (ks + 1) & 1); }
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
      if (!(ABL >= 3) || ks == 0) read_frags(0, af[0], bf[0]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j + 1 < 4 && (!(ABL >= 3) || ks == 0)) read_frags(j + 1, af[(j + 1) & 1], bf[(j + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);  // the reads above stay above the MFMAs below
#pragma unroll
        for (int m = 0; m < NMAT; ++m)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
              if constexpr (ABL == 1 || ABL == 5) acc[m][rt][tt][0] += __builtin_bit_cast(float, af[j & 1][m][rt].x ^ bf[j & 1][tt].x);
              else
              acc[m][rt][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[j & 1][m][rt]), __builtin_bit_cast(bf16x8, bf[j & 1][tt]),
                                                                       acc[m][rt][tt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (RING3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped tail issues
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
    if constexpr (ABL >= 8) {
      float sacc = 0.f;
#pragma unroll
      for (int m = 0; m < NMAT; ++m)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc += acc[m][rt][tt][i];
      if (sacc == 12345.678f) reinterpret_cast<float*>(s.out)[0] = sacc;
    } else {
      if constexpr (NMAT == 2) epilogue_all(std::integral_constant<int, EPI_GATED_SILU>{});  // the launcher admits two matrices for the gated stage only
      else epilogue_all(std::integral_constant<int, EPI_NONE>{});
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
              float lo = __uint_as_float(w4[h2] << 16), hi = __uint_as_float(w4[h2] & 0xffff0000u);
              if (has_bias) { lo += bias8[h2 * 2]; hi += bias8[h2 * 2 + 1]; }
              if (has_relu) { lo = fmaxf(DT<T>::round(lo), 0.f); hi = fmaxf(DT<T>::round(hi), 0.f); }
              w4[h2] = f2bf2(lo, hi);
            }
            v = u32x4{w4[0], w4[1], w4[2], w4[3]};
          }
          if (orow0 + 8 <= R) {
            *reinterpret_cast<u32x4*>(op) = v;
          } else {
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
              if (orow0 + jj < R) op[jj] = (T)(w4[jj >> 1] >> ((jj & 1) * 16));
          }
        }
      }
    }
    __syncthreads();  // the next pass re-uses buffer 0
  }
}

// max_rows: (an estimate of) the rows of the busiest expert; the kernel's pass loop covers more
bool launch_ffn_gemm_big(const FfnStage& s, int nmat, dim3 grid, int max_rows, hipStream_t st) {
  if (s.dtype != DT_BF16 || (s.K % 64) != 0 || (s.K_sh % 64) != 0 || (s.ld_out % 8) != 0) return false;  // 16-byte row stores
  if ((nmat == 2) != (s.epi == EPI_GATED_SILU)) return false;
  const int rmax = s.R > s.R_sh ? s.R : s.R_sh;
  const int passes = max_rows <= 256 ? 1 : (max_rows + 255) / 256;
  const int nx = (rmax + 255 / nmat) / (256 / nmat), ny = (int)grid.y, nz = passes > 8 ? 8 : passes;
  const dim3 g((unsigned)(((nx * ny + 7) / 8) * nz * 8));
  static const int xcd_map = env_int("MOEINF_GEMM_BIG_XCD", 1);
  static const int ring3 = env_int("MOEINF_GEMM_BIG_RING3", 1);
  static const int abl = env_int("MOEINF_GEMM_BIG_ABL", 0);
#define ABLGO(A) if (abl == A) { if (nmat == 2) hipLaunchKernelGGL((ffn_gemm_big_kernel<2, true, A>), g, dim3(512), 0, st, s, nx, ny, nz, xcd_map); else hipLaunchKernelGGL((ffn_gemm_big_kernel<1, true, A>), g, dim3(512), 0, st, s, nx, ny, nz, xcd_map); return true; }
  ABLGO(1) ABLGO(2) ABLGO(3) ABLGO(4) ABLGO(5) ABLGO(6) ABLGO(7) ABLGO(8) ABLGO(9)
  if (ring3) {
    if (nmat == 2) hipLaunchKernelGGL((ffn_gemm_big_kernel<2, true>), g, dim3(512), 0, st, s, nx, ny, nz, xcd_map);
    else hipLaunchKernelGGL((ffn_gemm_big_kernel<1, true>), g, dim3(512), 0, st, s, nx, ny, nz, xcd_map);
  } else {
    if (nmat == 2) hipLaunchKernelGGL((ffn_gemm_big_kernel<2, false>), g, dim3(512), 0, st, s, nx, ny, nz, xcd_map);
    else hipLaunchKernelGGL((ffn_gemm_big_kernel<1, false>), g, dim3(512), 0, st, s, nx, ny, nz, xcd_map);
  }
  return true;
}

}  // namespace moeinf
