// ffn_gemm_kernels.h — the grouped-GEMM forms of the expert FFN stage for experts with MANY rows (prefill, large batches):
// ffn_gemm (register-tiled), ffn_gemm_lds (both operands through LDS), ffn_gemm_hyb (activations through LDS, weights
// straight to registers); ffn_gemm_ring2 (register ring of weight tiles, software-pipelined; long reductions) lives in
// ffn_gemm_ring2.hip.  Selected by launch_ffn_gemm, which
// launch_ffn_stage (kernels.hip) calls for more than 16 rows per expert.
// Included by ffn_gemm.hip (bf16 and fp32 instantiations) and ffn_gemm_f16.hip (fp16, round 5): two translation units, so that
// the dtypes compile in parallel.
#pragma once
#include "kdev.h"

#include <type_traits>

namespace moeinf {

// ffn_gemm_ring2.hip: the register-ring kernel, launched when ring2_form (kernels.h) picks one of its forms; false: not handled
bool launch_ffn_gemm_ring2_bf16(const FfnStage& s, int nmat, dim3 grid, int max_rows, hipStream_t st);

// ------------------------------------------------------------------------------------------------
// ffn_gemm: the same stage for experts with MANY tokens (prefill, large batches) — a register-tiled
// grouped GEMM on MFMA.  A block owns RG row groups (16*RG weight rows, for the gated stage of BOTH
// matrices) and walks the expert's tokens 64 at a time; per k-tile a wave issues RG*NMAT weight-tile
// loads (contiguous 1 KiB each, the tiled layout IS the MFMA A fragment) + 4 activation-fragment loads
// and RG*NMAT*4 MFMAs — 16 MFMAs per 8 loads, against 8 per 6 in ffn_rows' 64-token variant — and the
// next k-tile's fragments are loaded into a second register set BEFORE the current MFMAs issue, so the
// L2 latency hides behind the matrix pipe even at 2-3 waves per SIMD.  K is split over the block's
// waves (no operand is loaded twice inside a block); partial tiles meet in LDS for the epilogue.
// ------------------------------------------------------------------------------------------------
template <typename T, int NMAT, int RG, int NT, int NW>
__global__ __launch_bounds__(NW * 64) void ffn_gemm_kernel(FfnStage s) {
  constexpr int EPV = DT<T>::EPV;
  constexpr int EPT = 4 * EPV;
  // NT = token groups (16 tokens each) per pass over the weights.  At t_e < ridge (~300 tokens) the stage
  // is still bound by HBM weight traffic, so the launcher picks NT to cover an expert's tokens in as few
  // passes as possible (weights stream from HBM once per pass; activations are re-read from L2).
  __shared__ float red[NW][NMAT][256];

  const int u = blockIdx.y;
  if (u >= (s.n_active_host >= 0 ? s.n_active_host : *s.n_active)) return;
  const int e = s.active[u];
  const bool sh = (e == s.E);
  const int K = sh ? s.K_sh : s.K;
  const int R = sh ? s.R_sh : s.R;
  const int rg0 = blockIdx.x * RG;  // first row group of this block
  if (rg0 * 16 >= R) return;
  const int cnt = s.counts[e];
  const int off = s.offsets[e];
  const char* W = reinterpret_cast<const char*>(s.wptr[e]);
  if (W == nullptr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicExch(s.miss_flag, 1);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int KB = (K + EPT - 1) / EPT, KBfull = K / EPT;
  const int nrg = min(RG, (R + 15) / 16 - rg0);  // live row groups (block-uniform)
  const char* a0 = W + (sh ? s.off_a_sh : s.off_a) + (size_t)rg0 * KB * 1024 + lane * 16;
  const char* a1 = NMAT == 2 ? W + (sh ? s.off_b_sh : s.off_b) + (size_t)rg0 * KB * 1024 + lane * 16 : nullptr;
  const size_t rg_stride = (size_t)KB * 1024;
  const int kq = q * EPV;
  const u32x4 z = {0u, 0u, 0u, 0u};

  for (int tile0 = 0; tile0 * 16 < cnt; tile0 += NT) {
    const int ntl = min(NT, (cnt - tile0 * 16 + 15) / 16);
    const T* xr[NT];
    f32x4 acc[RG][NT][NMAT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      const int srow = off + min((tile0 + tt) * 16 + n, cnt - 1);
      const int64_t xrow = s.row_map ? (int64_t)s.row_map[srow] : (int64_t)srow;
      xr[tt] = reinterpret_cast<const T*>(s.in) + xrow * s.ld_in + kq;
#pragma unroll
      for (int rg = 0; rg < RG; ++rg)
#pragma unroll
        for (int m = 0; m < NMAT; ++m) acc[rg][tt][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    u32x4 ca[RG][NMAT], cx[NT], na[RG][NMAT], nx[NT];
    auto load_frags = [&](u32x4 (&fa)[RG][NMAT], u32x4 (&fx)[NT], int kb, bool guard_x) {
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) {
        if (rg < nrg) {
          fa[rg][0] = ld16(a0 + rg * rg_stride + (size_t)kb * 1024);
          if (NMAT == 2) fa[rg][NMAT - 1] = ld16(a1 + rg * rg_stride + (size_t)kb * 1024);
        }
      }
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
        if (tt < ntl) fx[tt] = (!guard_x || kb * EPT + kq < K) ? ld16(xr[tt] + (size_t)kb * EPT) : z;
    };
    auto mma_frags = [&](const u32x4 (&fa)[RG][NMAT], const u32x4 (&fx)[NT]) {
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) {
        if (rg < nrg) {
#pragma unroll
          for (int tt = 0; tt < NT; ++tt) {
            if (tt < ntl) {
              mma16<T>(acc[rg][tt][0], fa[rg][0], fx[tt]);
              if (NMAT == 2) mma16<T>(acc[rg][tt][NMAT - 1], fa[rg][NMAT - 1], fx[tt]);
            }
          }
        }
      }
    };
    // k-tiles wave, wave+NW, ... (the zero-padded last tile, if any, is just one more tile with a guarded x read)
    int kb = wave;
    if (kb < KB) load_frags(ca, cx, kb, kb >= KBfull);
    for (; kb < KB; kb += NW) {
      const int nk = kb + NW;
      if (nk < KB) load_frags(na, nx, nk, nk >= KBfull);
      mma_frags(ca, cx);
      if (nk < KB) {
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
          for (int m = 0; m < NMAT; ++m) ca[rg][m] = na[rg][m];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) cx[tt] = nx[tt];
      }
    }
    // reduction over the K split + epilogue, one 16x16 tile at a time
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      if (rg >= nrg) break;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        if (tt >= ntl) break;
        const int tile = tile0 + tt;
        const int r0 = (rg0 + rg) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          red[wave][0][lane * 4 + j] = acc[rg][tt][0][j];
          if (NMAT == 2) red[wave][NMAT - 1][lane * 4 + j] = acc[rg][tt][NMAT - 1][j];
        }
        __syncthreads();
        for (int i = tid; i < 256; i += NW * 64) {
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int ww = 0; ww < NW; ++ww) {
            s0 += red[ww][0][i];
            if (NMAT == 2) s1 += red[ww][1][i];
          }
          const int l = i >> 2, j = i & 3;
          const int tn = l & 15;
          const int orow = r0 + (l >> 4) * 4 + j;
          if (tile * 16 + tn < cnt && orow < R) {
            float v = DT<T>::round(s0);
            if (s.epi == EPI_GATED_SILU) {
              const float b = DT<T>::round(s1);
              const float sl = DT<T>::round(v / (1.0f + expf(-v)));
              v = DT<T>::round(sl * b);
            } else {
              if (s.epi == EPI_BIAS || s.epi == EPI_BIAS_RELU)
                v = DT<T>::round(v + DT<T>::load(reinterpret_cast<const T*>(W + s.off_bias) + orow));
              if (s.epi == EPI_RELU || s.epi == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
            }
            const int srow = off + tile * 16 + tn;
            DT<T>::store(reinterpret_cast<T*>(s.out) + (size_t)(s.out_map ? s.out_map[srow] : srow) * s.ld_out + orow, v);
          }
        }
        __syncthreads();
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// ffn_gemm_lds: grouped GEMM for experts with many tokens, operands staged through LDS by the
// asynchronous global->LDS DMA (global_load_lds, 16 B per lane) in a two-buffer ring.
//   block = 4 waves as 2 (row halves) x 2 (token halves); block tile = RGB row groups x 8 token groups
//   (gated: 64 rows of BOTH matrices x 128 tokens; plain: 128 rows x 128 tokens); every wave owns 16
//   accumulator tiles; a stage = 2 k-tiles = 32 one-KiB tiles.
//   Both operand images in LDS are in MFMA FRAGMENT ORDER (bytes [16*lane, +16) of a 1-KiB tile belong
//   to lane `lane`): the weight tiles already are (tiled HBM layout, a contiguous 1-KiB DMA), and an
//   activation tile becomes one DMA whose per-lane SOURCE address is x[token lane%16][k + (lane/16)*8]
//   — the DMA writes base + lane*16, which is exactly the fragment slot.  Fragment reads are therefore
//   linear ds_read_b128 at lane*16: conflict-free, no swizzle, no transpose.
//   Loop: barrier (stage s landed, stage s-1 fully consumed) -> issue DMA of stage s+1 -> 32 MFMAs per
//   wave on stage s.  Requires K % (k-tile) == 0 (no zero-fill path for the activations).
// ------------------------------------------------------------------------------------------------
// NWV waves per block in a 2 x (NWV/2) grid: 4 waves cover 128 tokens per pass over the weights, 8 waves 256
// (experts with more than 128 rows would otherwise stream their weights from HBM twice).
// XL (needs K % (2 k-tiles) == 0): the activation image of a stage is filled in FULL 128-byte lines — one DMA =
// 8 token rows x 128 B (both k-tiles of the stage) instead of 16 rows x 64 B: half the cache lines per
// instruction on the texture-addresser path, which is what bounds this kernel at 128-256 tokens per expert.  The
// DMA writes LDS linearly (base + lane*16), so the bank swizzle is applied to the SOURCE: lane (r = lane/8,
// c = lane%8) fetches 16-byte chunk (c ^ r) of row r; a fragment read of (token n, chunk ch) then goes to piece
// n/8, byte r*128 + ((ch ^ r) << 4), r = n%8 — conflict-free for ds_read_b128.
// (A 3-buffer variant — stage ks+2 issued while stage ks is multiplied, counted s_waitcnt + raw s_barrier so that one
// stage stays in flight across the barrier — was built and measured: Mixtral's down projection 242-272 -> 346-368 us
// at 512 tokens, 808 -> 970-1005 us at 2048; DeepSeek +-10 % either way.  Not kept.)
template <typename T, int NMAT, int RGB, int NWV, bool XL>
__global__ __launch_bounds__(NWV * 64) void ffn_gemm_lds_kernel(FfnStage s) {
  constexpr int EPV = DT<T>::EPV;
  constexpr int EPT = 4 * EPV;
  constexpr int RGW = RGB / 2;
  constexpr int WC = NWV / 2;          // wave columns
  constexpr int NTW = 4, NTB = WC * NTW;
  constexpr int XPW = XL ? 2 * NTB / NWV : NTB / NWV;  // activation DMA pieces per wave and k-tile pair
  constexpr int KK = 2;
  constexpr int A_TILES = KK * NMAT * RGB;
  constexpr int B_TILES = KK * NTB;
  constexpr int STAGE = (A_TILES + B_TILES) * 1024;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int u = blockIdx.y, bx = blockIdx.x;
  if (u >= (s.n_active_host >= 0 ? s.n_active_host : *s.n_active)) return;
  const int e = s.active[u];
  const bool sh = (e == s.E);
  const int K = sh ? s.K_sh : s.K;
  const int R = sh ? s.R_sh : s.R;
  const int rg0 = bx * RGB;
  const int nrg_total = (R + 15) / 16;
  if (rg0 >= nrg_total) return;
  const int cnt = s.counts[e];
  const int off = s.offsets[e];
  const char* W = reinterpret_cast<const char*>(s.wptr[e]);
  if (W == nullptr) {
    if (threadIdx.x == 0 && bx == 0) atomicExch(s.miss_flag, 1);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const int n = lane & 15, q = lane >> 4;
  const int KB = K / EPT;  // K % EPT == 0 (checked by the launcher)
  const int KS = (KB + KK - 1) / KK;
  const size_t rg_stride = (size_t)KB * 1024;
  const char* am[NMAT];
  am[0] = W + (sh ? s.off_a_sh : s.off_a) + (size_t)rg0 * rg_stride + lane * 16;
  if (NMAT == 2) am[NMAT - 1] = W + (sh ? s.off_b_sh : s.off_b) + (size_t)rg0 * rg_stride + lane * 16;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  for (int tile0 = 0; tile0 * 16 < cnt; tile0 += NTB) {
    const int ntl = min(NTB, (cnt - tile0 * 16 + 15) / 16);
    // activation rows this wave DMA-loads: token groups `wave`, `wave + NWV` (16 rows x 64 B each), or with XL the
    // 8-row pieces `wave + NWV*i` (8 rows x 128 B, source chunk swizzled)
    const T* xrp[XPW];
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
      const int trow = XL ? (tile0 * 16 + (wave + NWV * i) * 8 + (lane >> 3)) : ((tile0 + wave + NWV * i) * 16 + n);
      const int srow = off + min(trow, cnt - 1);
      const int64_t xrow = s.row_map ? (int64_t)s.row_map[srow] : (int64_t)srow;
      xrp[i] = reinterpret_cast<const T*>(s.in) + xrow * s.ld_in + (XL ? (((lane & 7) ^ (lane >> 3)) * EPV) : q * EPV);
    }
    f32x4 acc[RGW][NTW][NMAT];
#pragma unroll
    for (int a = 0; a < RGW; ++a)
#pragma unroll
      for (int b = 0; b < NTW; ++b)
#pragma unroll
        for (int m = 0; m < NMAT; ++m) acc[a][b][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto issue = [&](int ks, int buf) {
      char* base = smem + buf * STAGE;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const int kb = ks * KK + kk;
        if (kb < KB) {
#pragma unroll
          for (int i = 0; i < (RGB + NWV - 1) / NWV; ++i) {
            const int rg_l = wave + NWV * i;
            if (rg_l < RGB && rg0 + rg_l < nrg_total) {
#pragma unroll
              for (int m = 0; m < NMAT; ++m)
                __builtin_amdgcn_global_load_lds((gptr_t)(am[m] + rg_l * rg_stride + (size_t)kb * 1024),
                                                 (lptr_t)(base + ((kk * NMAT + m) * RGB + rg_l) * 1024), 16, 0, 0);
            }
          }
          if constexpr (!XL) {
#pragma unroll
            for (int i = 0; i < XPW; ++i) {
              const int tg_l = wave + NWV * i;
              if (tg_l < ntl)
                __builtin_amdgcn_global_load_lds((gptr_t)(xrp[i] + (size_t)kb * EPT),
                                                 (lptr_t)(base + (A_TILES + kk * NTB + tg_l) * 1024), 16, 0, 0);
            }
          }
        }
      }
      if constexpr (XL) {
#pragma unroll
        for (int i = 0; i < XPW; ++i) {
          const int pc = wave + NWV * i;  // 8-row piece; token group pc/2
          if (pc < 2 * ntl)
            __builtin_amdgcn_global_load_lds((gptr_t)(xrp[i] + (size_t)ks * KK * EPT), (lptr_t)(base + (A_TILES + pc) * 1024), 16, 0, 0);
        }
      }
    };

    issue(0, 0);
    for (int ks = 0; ks < KS; ++ks) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA of stage ks has landed
      __syncthreads();                                   // ... everybody's has, and stage ks-1 is fully consumed
      if (ks + 1 < KS) issue(ks + 1, (ks + 1) & 1);
      const char* base = smem + (ks & 1) * STAGE + lane * 16;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        if (ks * KK + kk < KB) {
          u32x4 af[RGW][NMAT], bf[NTW];
#pragma unroll
          for (int a = 0; a < RGW; ++a) {
            const int rg_l = wr * RGW + a;
#pragma unroll
            for (int m = 0; m < NMAT; ++m) af[a][m] = *reinterpret_cast<const u32x4*>(base + ((kk * NMAT + m) * RGB + rg_l) * 1024);
          }
#pragma unroll
          for (int b = 0; b < NTW; ++b) {
            if constexpr (XL) {
              const int r = n & 7, ch = kk * 4 + q;
              bf[b] = *reinterpret_cast<const u32x4*>(smem + (ks & 1) * STAGE + (A_TILES + (wc * NTW + b) * 2 + (n >> 3)) * 1024 + r * 128 + ((ch ^ r) << 4));
            } else {
              bf[b] = *reinterpret_cast<const u32x4*>(base + (A_TILES + kk * NTB + wc * NTW + b) * 1024);
            }
          }
#pragma unroll
          for (int a = 0; a < RGW; ++a) {
            if (rg0 + wr * RGW + a < nrg_total) {
#pragma unroll
              for (int b = 0; b < NTW; ++b) {
                if (wc * NTW + b < ntl) {
                  mma16<T>(acc[a][b][0], af[a][0], bf[b]);
                  if (NMAT == 2) mma16<T>(acc[a][b][NMAT - 1], af[a][NMAT - 1], bf[b]);
                }
              }
            }
          }
        }
      }
    }
    // epilogue straight from the accumulators (no K split): lane holds 4 consecutive rows of one token
    epi_switch<NMAT>(s.epi, [&](auto epic) {
      constexpr int EPI = decltype(epic)::value;
      const T* bias = reinterpret_cast<const T*>(W + s.off_bias);
      const bool aligned = (s.ld_out & 3) == 0;
#pragma unroll
      for (int b = 0; b < NTW; ++b) {
        const int tok = (tile0 + wc * NTW + b) * 16 + n;
        if (tok < cnt) {
          T* orow_p = reinterpret_cast<T*>(s.out) + (size_t)(s.out_map ? s.out_map[off + tok] : off + tok) * s.ld_out;
#pragma unroll
          for (int a = 0; a < RGW; ++a)
            if (rg0 + wr * RGW + a < nrg_total)
              epi_quad<T, EPI>(acc[a][b][0], acc[a][b][NMAT - 1], bias, (rg0 + wr * RGW + a) * 16 + q * 4, R, aligned, orow_p);
        }
      }
    });
    __syncthreads();  // the next pass re-uses buffer 0
  }
}

// ------------------------------------------------------------------------------------------------
// ffn_gemm_hyb: grouped GEMM for experts with up to a few hundred tokens, where the stage is still bound
// by streaming the weights from HBM (ridge: ~300 tokens per expert).  What limits ffn_gemm_lds there is
// BYTES IN FLIGHT: a CU has to keep latency x bandwidth (~2 us x 25 B/ns) of weight bytes outstanding, and
// with both operands staged in LDS the 160 KiB cap that at 2 blocks x one 16-KiB weight stage.
// Here only the ACTIVATIONS go through LDS (they are shared by all waves of the block); every wave owns
// private weight rows and streams its tiles straight into registers, like the decode kernel (the tiled HBM
// layout is the MFMA A fragment).  LDS per block drops to 2 x KK x 8 KiB, so 3-4 blocks fit a CU and the
// weight bytes in flight no longer depend on LDS.
//   block = 4 waves; wave w owns RW row groups (16 rows each) of NMAT matrices (RW*NMAT = 2) against 8 token
//   groups (128 tokens): 16 accumulator tiles.  Stage = KK k-tiles: A fragments of stage s+1 are loaded into a
//   second register set and B tiles of stage s+1 are DMA'd into the other LDS buffer while stage s computes.
// ------------------------------------------------------------------------------------------------
//   XL: activation image in full 128-byte lines with the source-side swizzle of ffn_gemm_lds (needs KK even and
//   an even number of k-tiles).
template <typename T, int NMAT, int RW, int KK, bool XL>
__global__ __launch_bounds__(256) void ffn_gemm_hyb_kernel(FfnStage s) {
  static_assert(!XL || KK % 2 == 0, "full-line staging moves k-tiles in pairs");
  constexpr int EPV = DT<T>::EPV;
  constexpr int EPT = 4 * EPV;
  constexpr int NTB = 8;
  constexpr int RGB = 4 * RW;            // row groups per block
  constexpr int STAGE = KK * NTB * 1024;  // activation bytes per stage
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int u = blockIdx.y;
  if (u >= (s.n_active_host >= 0 ? s.n_active_host : *s.n_active)) return;
  const int e = s.active[u];
  const bool sh = (e == s.E);
  const int K = sh ? s.K_sh : s.K;
  const int R = sh ? s.R_sh : s.R;
  const int nrg_total = (R + 15) / 16;
  if ((int)blockIdx.x * RGB >= nrg_total) return;
  const int cnt = s.counts[e];
  const int off = s.offsets[e];
  const char* W = reinterpret_cast<const char*>(s.wptr[e]);
  if (W == nullptr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicExch(s.miss_flag, 1);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int KB = K / EPT;  // K % EPT == 0 (checked by the launcher)
  const int KS = (KB + KK - 1) / KK;
  const size_t rg_stride = (size_t)KB * 1024;
  const int rgw0 = blockIdx.x * RGB + wave * RW;  // first row group of this wave
  // row groups past the end (R not a multiple of the block's rows) re-read the last one; their results are dropped
  const char* ap[RW][NMAT];
#pragma unroll
  for (int a = 0; a < RW; ++a) {
    const int rg = min(rgw0 + a, nrg_total - 1);
    ap[a][0] = W + (sh ? s.off_a_sh : s.off_a) + (size_t)rg * rg_stride + lane * 16;
    if (NMAT == 2) ap[a][NMAT - 1] = W + (sh ? s.off_b_sh : s.off_b) + (size_t)rg * rg_stride + lane * 16;
  }
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  for (int tile0 = 0; tile0 * 16 < cnt; tile0 += NTB) {
    const int ntl = min(NTB, (cnt - tile0 * 16 + 15) / 16);
    constexpr int XPW = XL ? 4 : 2;
    const T* xrp[XPW];  // activation rows this wave DMA-loads: token groups `wave`, `wave + 4` / 8-row pieces `wave + 4i`
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
      const int trow = XL ? (tile0 * 16 + (wave + 4 * i) * 8 + (lane >> 3)) : ((tile0 + wave + 4 * i) * 16 + n);
      const int srow = off + min(trow, cnt - 1);
      const int64_t xrow = s.row_map ? (int64_t)s.row_map[srow] : (int64_t)srow;
      xrp[i] = reinterpret_cast<const T*>(s.in) + xrow * s.ld_in + (XL ? (((lane & 7) ^ (lane >> 3)) * EPV) : q * EPV);
    }
    f32x4 acc[RW][NTB][NMAT];
#pragma unroll
    for (int a = 0; a < RW; ++a)
#pragma unroll
      for (int b = 0; b < NTB; ++b)
#pragma unroll
        for (int m = 0; m < NMAT; ++m) acc[a][b][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 af[2][KK][RW][NMAT];  // two register sets of weight fragments (current / next stage)
    auto issue = [&](int ks, int buf, u32x4 (&dst)[KK][RW][NMAT]) {
      char* base = smem + buf * STAGE;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const int kb = min(ks * KK + kk, KB - 1);  // a short last stage re-reads tile KB-1 (never multiplied)
#pragma unroll
        for (int a = 0; a < RW; ++a)
#pragma unroll
          for (int m = 0; m < NMAT; ++m) dst[kk][a][m] = ld16_nt(ap[a][m] + (size_t)kb * 1024);
        if constexpr (!XL) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int tg_l = wave + 4 * i;
            if (tg_l < ntl)
              __builtin_amdgcn_global_load_lds((gptr_t)(xrp[i] + (size_t)kb * EPT), (lptr_t)(base + (kk * NTB + tg_l) * 1024), 16, 0, 0);
          }
        }
      }
      if constexpr (XL) {
#pragma unroll
        for (int j = 0; j < KK / 2; ++j) {
          const int pr = min(ks * (KK / 2) + j, KB / 2 - 1);  // k-tile pair (a short last stage re-reads the last pair)
#pragma unroll
          for (int i = 0; i < XPW; ++i) {
            const int pc = wave + 4 * i;
            if (pc < 2 * ntl)
              __builtin_amdgcn_global_load_lds((gptr_t)(xrp[i] + (size_t)pr * 2 * EPT), (lptr_t)(base + (j * 2 * NTB + pc) * 1024), 16, 0, 0);
          }
        }
      }
    };
    auto compute = [&](int ks, int buf, const u32x4 (&cur)[KK][RW][NMAT]) {
      const char* base = smem + buf * STAGE + lane * 16;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        if (ks * KK + kk < KB) {
#pragma unroll
          for (int b = 0; b < NTB; ++b) {
            if (b < ntl) {
              const int r = n & 7, ch = (kk & 1) * 4 + q;
              const u32x4 bf = XL ? *reinterpret_cast<const u32x4*>(smem + buf * STAGE + ((kk >> 1) * 2 * NTB + b * 2 + (n >> 3)) * 1024 + r * 128 + ((ch ^ r) << 4))
                                  : *reinterpret_cast<const u32x4*>(base + (kk * NTB + b) * 1024);
#pragma unroll
              for (int a = 0; a < RW; ++a) {
                mma16<T>(acc[a][b][0], cur[kk][a][0], bf);
                if (NMAT == 2) mma16<T>(acc[a][b][NMAT - 1], cur[kk][a][NMAT - 1], bf);
              }
            }
          }
        }
      }
    };

    issue(0, 0, af[0]);
    for (int ks = 0; ks < KS; ks += 2) {  // unrolled by two so both register sets are indexed statically
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stage ks: this wave's fragments and activation DMA landed
      __syncthreads();                                   // ... everybody's DMA has, and stage ks-1 is fully consumed
      if (ks + 1 < KS) issue(ks + 1, 1, af[1]);
      compute(ks, 0, af[0]);
      if (ks + 1 < KS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ks + 2 < KS) issue(ks + 2, 0, af[0]);
        compute(ks + 1, 1, af[1]);
      }
    }
    // epilogue straight from the accumulators (no K split): lane holds 4 consecutive rows of one token
    epi_switch<NMAT>(s.epi, [&](auto epic) {
      constexpr int EPI = decltype(epic)::value;
      const T* bias = reinterpret_cast<const T*>(W + s.off_bias);
      const bool aligned = (s.ld_out & 3) == 0;
#pragma unroll
      for (int b = 0; b < NTB; ++b) {
        const int tok = (tile0 + b) * 16 + n;
        if (tok < cnt) {
          T* orow_p = reinterpret_cast<T*>(s.out) + (size_t)(s.out_map ? s.out_map[off + tok] : off + tok) * s.ld_out;
#pragma unroll
          for (int a = 0; a < RW; ++a)
            if (rgw0 + a < nrg_total)
              epi_quad<T, EPI>(acc[a][b][0], acc[a][b][NMAT - 1], bias, (rgw0 + a) * 16 + q * 4, R, aligned, orow_p);
        }
      }
    });
    __syncthreads();  // the next pass re-uses LDS buffer 0
  }
}

template <typename T, int NMAT>
bool launch_ffn_gemm(const FfnStage& s, dim3 grid, int max_rows, hipStream_t st) {
  if (NMAT == 2 && s.epi != EPI_GATED_SILU) return false;  // (the gelu gate: the row kernel, kernels.hip)
  static const int use_gemm = env_int("MOEINF_FFN_GEMM", 2);
  static const int force_nt = env_int("MOEINF_FFN_GEMM_NT", 0);
  // long prefills: the 256 x 256 / 32x32x16-MFMA kernel (ffn_gemm_big.hip).  Measured (profiles/r03_ffn_sweep_prefill_big_*.txt):
  // it beats ffn_gemm_lds from 257 rows per expert on (Mixtral down projection at 2048 tokens 846 -> 730 us, DeepSeek-V2-Lite
  // at 4096 tokens 2.54 -> 1.96 ms per layer).  The register-ring kernel (gated stage, K >= 4096) held out to ~640 rows
  // against the first ping-pong version; with the short-last-pass variant the big kernel wins from 257 rows on (Mixtral
  // gate/up: 768 tokens 503 vs 535 us, 1024 tokens 647 vs 748, 1536 tokens 845 vs 1 031, 2048 tokens 1 010 vs 1 250), so
  // both stages switch at the same row count now; below it (512 tokens: 436 vs 450 gate/up but 324 vs 265 down) ring / lds stay
  static const int big_env = env_int("MOEINF_GEMM_BIG", 1);
  static const int big_rows = env_int("MOEINF_GEMM_BIG_ROWS", 256);
  const int ept = sizeof(T) == 2 ? 32 : 16;
  const bool k_ok = (s.K % ept) == 0 && (s.K_sh % ept) == 0;
  // 17-64 rows per expert (e.g. NLLB's 128 experts at a 2048-token batch): too many for the decode kernel, too few to
  // amortise staging the weights in LDS -> the hybrid kernel (measured -15 % on that shape, sweep in profiles/)
  // hybrid kernel (weights -> registers) up to 64 rows per expert; up to 128 when few experts are active (<= 16: big
  // matrices, few workgroups — Mixtral at 192 / 256 / 320 tokens: down projection 213 -> 174, 227 -> 208, 232 -> 226 us; with
  // NLLB's 128 experts at 4096 tokens the same switch costs +11 %)
  static const Ring2Knobs knobs0 = Ring2Knobs::from_env();
  const int hyb_rows = hyb_rows_for((int)grid.y, knobs0);
  if constexpr (sizeof(T) == 2) {
    // long reductions (K >= 4096: Mixtral's two stages, NLLB's second), 17 (plain) / hyb_rows+1 (gated) .. 340 rows per expert:
    // the software-pipelined register ring.  Measured against what ran there before (profiles/r04_ffn_sweep_ring2_*.txt,
    // Mixtral-8x7B, us per layer, gate-up / down):
    //   tokens   96       224       336       384       512       640       768       896
    //   before   320/162  337/196   365/233   404/252   441/261   494/274   492/342   510/347   (hybrid | ring + lds | big)
    //   ring2    (hyb)/147 (hyb)/157 (hyb)/173 347/178   374/202   401/233   438/256   487/287
    // gated stage below 129 rows: the hybrid kernel is 1-2 % ahead and stays.  Above ~256 rows per expert (the row estimate of
    // the sync-free path is 1.5 x the mean + 1 = 337 at 896 tokens, 385 at 1 024) a second pass over the weights begins and the
    // big-tile kernel takes over.  An expert with more rows than a pass holds takes another pass; correctness never depends on
    // the estimate.
    if constexpr (std::is_same<T, uint16_t>::value) {
      if (use_gemm == 2 && launch_ffn_gemm_ring2_bf16(s, NMAT, grid, max_rows, st)) return true;
    }
  }
  if (use_gemm == 2 && big_env && sizeof(T) == 2 && max_rows > big_rows && launch_ffn_gemm_big(s, NMAT, grid, max_rows, st)) return true;
  if ((use_gemm == 3 || (use_gemm == 2 && max_rows <= hyb_rows)) && k_ok) {  // weights -> registers, activations -> LDS
    static const int kk = env_int("MOEINF_GEMM_HYB_KK", 4);
    static const int hxl_env = env_int("MOEINF_GEMM_XL", 1);
    const bool hxl = hxl_env && (s.K % (2 * ept)) == 0 && (s.K_sh % (2 * ept)) == 0;
#define HYB(NM, RWV, KKV, XLV) KL((ffn_gemm_hyb_kernel<T, NM, RWV, KKV, XLV>), dim3((grid.x + 4 * RWV - 1) / (4 * RWV), grid.y), dim3(256), 0, st, s)
    if constexpr (NMAT == 2) {
      if (kk == 2) { if (hxl) HYB(2, 1, 2, true); else HYB(2, 1, 2, false); }
      else { if (hxl) HYB(2, 1, 4, true); else HYB(2, 1, 4, false); }
    } else {
      if (kk == 2) { if (hxl) HYB(1, 2, 2, true); else HYB(1, 2, 2, false); }
      else { if (hxl) HYB(1, 2, 4, true); else HYB(1, 2, 4, false); }
    }
#undef HYB
  } else if (use_gemm == 2 && k_ok) {  // LDS-staged grouped GEMM
    static const int rgb_plain = env_int("MOEINF_FFN_GEMM_RGB", 0);
    static const int wide_env = env_int("MOEINF_GEMM_WIDE", -1);
    const bool wide = wide_env >= 0 ? wide_env != 0 : max_rows > 128;  // 8 waves: 256 tokens per pass over the weights
    static const int xl_env = env_int("MOEINF_GEMM_XL", 1);
    const bool xl = xl_env && (s.K % (2 * ept)) == 0 && (s.K_sh % (2 * ept)) == 0;  // full-line activation staging
    auto go = [&](auto kern, int rgb, int nwv) {
      KL(kern, dim3((grid.x + rgb - 1) / rgb, grid.y), dim3(nwv * 64), 0, st, s);
    };
#define GO(NM, RG, NW) do { if (xl) go(ffn_gemm_lds_kernel<T, NM, RG, NW, true>, RG, NW); else go(ffn_gemm_lds_kernel<T, NM, RG, NW, false>, RG, NW); } while (0)
    if constexpr (NMAT == 2) {
      static const int rgb_gated = env_int("MOEINF_FFN_GEMM_RGB2", 4);
      if (rgb_gated == 8) { if (wide) GO(2, 8, 8); else GO(2, 8, 4); }
      else { if (wide) GO(2, 4, 8); else GO(2, 4, 4); }
    } else {
      // 128-row blocks need >= 2 blocks per CU to hide the DMA latency; fall back to 64-row blocks otherwise
      const bool big = rgb_plain ? rgb_plain == 8 : (((grid.x + 7) / 8) * grid.y >= 512 && s.K >= 4096);
      if (big) { if (wide) GO(1, 8, 8); else GO(1, 8, 4); }
      else     { if (wide) GO(1, 4, 8); else GO(1, 4, 4); }
    }
#undef GO
  } else if (use_gemm) {
    const int nt = force_nt ? force_nt : 4;  // measured: (RG,NT)=(2,4)/(4,4) beats (1,8)/(2,8) at t_e ~128 (profiles/r01_ffn_sweep_prefill_gemm.txt)
    if constexpr (NMAT == 2) {  // gated: 2 matrices -> (RG, NT) = (2,4) or (1,8)
      if (nt <= 4) KL((ffn_gemm_kernel<T, 2, 2, 4, 4>), dim3((grid.x + 1) / 2, grid.y), dim3(256), 0, st, s);
      else KL((ffn_gemm_kernel<T, 2, 1, 8, 4>), grid, dim3(256), 0, st, s);
    } else {                    // plain: (4,4) or (2,8)
      if (nt <= 4) KL((ffn_gemm_kernel<T, 1, 4, 4, 4>), dim3((grid.x + 3) / 4, grid.y), dim3(256), 0, st, s);
      else KL((ffn_gemm_kernel<T, 1, 2, 8, 4>), dim3((grid.x + 1) / 2, grid.y), dim3(256), 0, st, s);
    }
  } else {
    return false;
  }
  return true;
}

}  // namespace moeinf
