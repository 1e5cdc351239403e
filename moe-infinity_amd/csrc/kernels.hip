// kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the expert-offload hot path.
//
//   gate_logits      router GEMV/GEMM, fp64 accumulate (bit-stable routing)        HBM/latency bound
//   route_topk       softmax + wave-level top-k + renorm, one wave per token       latency bound
//   dispatch_index   per-expert counts (ballot/popc), prefix sums, stable permutation
//   ffn_rows         grouped expert FFN as a weight-streaming row-dot kernel: MFMA 16x16 tiles with
//                    the WEIGHT rows on the M side and <=16 routed tokens on the N side, so one
//                    pass over an expert's weights serves every token routed to it; fused
//                    gather (stage 1), fused SiLU*mul / ReLU / bias epilogues      HBM bound
//   combine          deterministic weighted gather (ascending expert id), reference rounding points
//
// What these replace in the reference is a SEQUENCE OF ATen OPS, not kernels (the reference has no
// device code): SURVEY.md section 2.3 K1..K10.  Rounding points of the reference's dtype-typed ATen ops
// are reproduced (Tr() below) so results match its CPU path to accumulation-order noise.
#include "kdev.h"

#include <algorithm>

#include <type_traits>

#include <string.h>

namespace moeinf {

// ------------------------------------------------------------------------------------------------
// Weight layout in an HBM slot ("tiled"): every [R,K] matrix is stored as MFMA A-operand tiles.
// Tile (rg, kb) covers rows [16rg,16rg+16) x 64 bytes of k (32 bf16 / 16 fp32) and occupies 1 KiB
// laid out in LANE ORDER: bytes [16*lane, 16*lane+16) = W[16rg + (lane&15)][kb*EPT + (lane>>4)*EPV ...].
// Tiles of one row group are consecutive (kb fastest), so a wave that streams a row group issues
// plain contiguous 1-KiB loads — the access pattern that measured 6.8-7.2 TB/s on this chip
// (tools/stream_patterns.hip, pattern E) against 6.0-6.1 TB/s for 16 strided 64-byte row segments.
// The host arena keeps the reference's row-major blob; retile_kernel converts after each H2D copy.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void retile_kernel(const T* __restrict__ src, char* __restrict__ dst, int R, int K) {
  constexpr int EPV = DT<T>::EPV;
  constexpr int EPT = 4 * EPV;  // k elements per tile
  const int KB = (K + EPT - 1) / EPT;
  const int lane = threadIdx.x & 63;
  const int kb = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int rg = blockIdx.y;
  if (kb >= KB) return;
  const int row = rg * 16 + (lane & 15);
  const int k = kb * EPT + (lane >> 4) * EPV;
  u32x4 v = {0u, 0u, 0u, 0u};
  if (row < R && k < K) v = ld16(src + (size_t)row * K + k);  // K % EPV == 0
  *reinterpret_cast<u32x4*>(dst + ((size_t)rg * KB + kb) * 1024 + lane * 16) = v;
}
// Every tensor of ONE staged expert blob in one launch (small experts travel host -> HBM as a whole blob, one
// hipMemcpyAsync; engine.cpp issue_copy): blockIdx.z = tensor, a matrix is re-tiled as above, a bias vector (K == 0) is
// copied as 16-byte pieces.
template <typename T>
__global__ __launch_bounds__(256) void retile_blob_kernel(RetileBlob b) {
  constexpr int EPV = DT<T>::EPV;
  constexpr int EPT = 4 * EPV;
  const int t = blockIdx.z;
  const char* src = reinterpret_cast<const char*>(b.src) + b.src_off[t];
  char* dst = reinterpret_cast<char*>(b.dst) + b.dst_off[t];
  const int R = b.R[t], K = b.K[t];
  if (K == 0) {  // not a matrix: R = bytes / 16 (sizes are multiples of 16: checked by the launcher)
    const int64_t i = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
    if (i < R) reinterpret_cast<u32x4*>(dst)[i] = reinterpret_cast<const u32x4*>(src)[i];
    return;
  }
  const int KB = (K + EPT - 1) / EPT;
  const int lane = threadIdx.x & 63;
  const int kb = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int rg = blockIdx.y;
  if (kb >= KB || rg * 16 >= R) return;
  const int row = rg * 16 + (lane & 15);
  const int k = kb * EPT + (lane >> 4) * EPV;
  u32x4 v = {0u, 0u, 0u, 0u};
  if (row < R && k < K) v = ld16(reinterpret_cast<const T*>(src) + (size_t)row * K + k);
  *reinterpret_cast<u32x4*>(dst + ((size_t)rg * KB + kb) * 1024 + lane * 16) = v;
}
hipError_t launch_retile_blob(const RetileBlob& b, int dtype, hipStream_t st) {
  const int ept = dtype == DT_F32 ? 16 : 32;
  unsigned gx = 1, gy = 1;
  for (int t = 0; t < b.n; ++t) {
    if (b.K[t] > 0) { gx = std::max(gx, (unsigned)(((b.K[t] + ept - 1) / ept + 3) / 4)); gy = std::max(gy, (unsigned)((b.R[t] + 15) / 16)); }
  }
  for (int t = 0; t < b.n; ++t)
    if (b.K[t] == 0) {  // a vector's pieces are spread over the (gx, gy) plane the matrices set
      const int64_t blocks = ((int64_t)b.R[t] + 255) / 256;
      while ((int64_t)gx * gy < blocks) gy += 1;
    }
  const dim3 grid(gx, gy, b.n);
  if (dtype != DT_F32) hipLaunchKernelGGL(retile_blob_kernel<uint16_t>, grid, dim3(256), 0, st, b);
  else hipLaunchKernelGGL(retile_blob_kernel<float>, grid, dim3(256), 0, st, b);
  return hipGetLastError();
}
// PULL form of the tier mover (round 6): the GPU fetches an expert blob from PINNED HOST memory itself and writes it straight into
// the tiled slot — no SDMA copy, no staging buffer, no re-tile launch.  Measured (tools/pcie_read.hip,
// profiles/r06_pcie_pull_kernel_vs_hipmemcpy.txt): 8-16 workgroups with four 16-byte loads per lane in flight pull 56.5 GB/s
// from a hipHostMalloc arena — the link's rate — while a hipMemcpyAsync that has an event recorded in front of and behind it
// (every expert copy of the engine had: staging-buffer hand-over, ready events, timers) pays ~60 us of SDMA <-> queue hand-overs
// on top of its bytes (16.5 MiB DeepSeek experts: 46-52 GB/s; 112 MiB Mixtral pieces: 54.6).  A kernel on the copy stream orders
// against events inside ONE hardware queue.
// Work unit = 16 rows x 1 KiB (one row group, sixteen k-tiles): every wave reads four ROWS, 1 KiB CONTIGUOUS each (a first form
// that read the 64-byte row pieces of a tile directly pulled 21-35 GB/s: sixteen 64-byte requests per wave-load to sixteen
// different pages; contiguous KiBs reach the link's 56), the rows cross through LDS, and the workgroup writes sixteen 1-KiB tiles.
// The loads of unit i+1 are in flight while unit i leaves LDS.  A bias vector is copied in 16-KiB units.
template <typename T, bool F8>  // F8: the host blob holds fp8 (e4m3fn) elements — half the bytes on the link —, up-cast to bf16 (T) on the way in
__global__ __launch_bounds__(256) void pull_retile_kernel(RetileBlob b, unsigned long long* ts, int first) {
  // link-busy timing WITHOUT queue packets (an event record between two kernels of a stream costs ~10 us, and the engine's timers
  // were two per copy: -5.5 % ms/token on the DeepSeek offload leg): ts[0] = start tick of the copy's first launch, ts[1] = max
  // end tick over its workgroups, ts[2] = workgroups that have finished (all three only ever grow: no re-initialisation)
  if (ts && first && blockIdx.x == 0 && threadIdx.x == 0) ts[0] = (unsigned long long)wall_clock64();
  constexpr int EPV = DT<T>::EPV;          // elements per 16 bytes of the SLOT's dtype
  constexpr int EPT = 4 * EPV;             // k elements per tile (64 bytes of a row)
  constexpr int W = F8 ? 2 : 1;            // 16-byte pieces of the slot a lane produces per load (a lane always LOADS 16 source bytes)
  constexpr int CK = 64 * EPV * W;         // k elements per unit (1 KiB of a SOURCE row)
  constexpr int LROW = 1024 * W + 16;      // LDS bytes per row (padded: the tile reads walk sixteen rows)
  __shared__ __attribute__((aligned(16))) char lds[16 * LROW];
  __shared__ uint16_t lut[F8 ? 256 : 1];  // e4m3fn byte -> bf16 bits (exact: three mantissa bits)
  if (F8) {
    const uint32_t v8 = threadIdx.x, sg = v8 >> 7, ex = (v8 >> 3) & 15, m = v8 & 7;
    float f = (ex == 15 && m == 7) ? __builtin_nanf("") : (ex == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + (float)m * 0.125f, (int)ex - 7));
    if (sg) f = -f;
    lut[v8 & (F8 ? 255 : 0)] = (uint16_t)(__float_as_uint(f) >> 16);
    __syncthreads();
  }
  // 16 RAW source bytes -> W pieces of 16 slot bytes (fp8: the up-cast; else the bytes themselves)
  auto widen = [&](const u32x4 q, u32x4 (&o)[W]) {
    if constexpr (F8) {
      const uint32_t qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        uint32_t r4[4];
#pragma unroll
        for (int h2 = 0; h2 < 4; ++h2) {
          const uint32_t ww = qq[w * 2 + (h2 >> 1)], sh = (h2 & 1) * 16;
          r4[h2] = (uint32_t)lut[(ww >> sh) & 255] | ((uint32_t)lut[(ww >> (sh + 8)) & 255] << 16);
        }
        o[w] = u32x4{r4[0], r4[1], r4[2], r4[3]};
      }
    } else {
      o[0] = q;
    }
  };
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int nunit[4], kcs[4], total = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    kcs[t] = (t < b.n && b.K[t] > 0) ? (b.K[t] + CK - 1) / CK : 1;
    nunit[t] = t < b.n ? (b.K[t] > 0 ? ((b.R[t] + 15) / 16) * kcs[t] : (b.R[t] + 1024 * W - 1) / (1024 * W)) : 0;  // vector: R = 16-byte pieces, 1024 W per unit
    total += nunit[t];
  }
  // The four loads of a unit are issued UNCONDITIONALLY from clamped (always valid) addresses and the lanes that are out of range
  // are zeroed when the data is USED: a register that is zeroed before a predicated load makes the compiler wait for every load
  // in flight first (vmcnt(0) in front of each load: one request per lane in flight instead of four — the link at 42 GB/s with 16
  // workgroups instead of 56).
  u32x4 raw[4];
  uint32_t okmask = 0;  // bit j: this lane's j-th load of the unit in flight is in range
  constexpr int SRC_EPL = F8 ? 16 : EPV;  // source elements per lane-load (16 bytes)
  auto decode = [&](int i, int& t, int& rg, int& kc) {
    t = 0;
    while (t < 3 && i >= nunit[t]) { i -= nunit[t]; ++t; }
    rg = i / kcs[t]; kc = i - rg * kcs[t];
  };
  auto load = [&](int i) {
    int t, rg, kc;
    decode(i, t, rg, kc);
    const char* src = reinterpret_cast<const char*>(b.src) + b.src_off[t];
    constexpr int SRC_ES = F8 ? 1 : (int)sizeof(T);
    okmask = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      size_t elem;
      bool ok;
      if (b.K[t] > 0) {
        const int row = rg * 16 + wave * 4 + j, k = kc * CK + lane * SRC_EPL;  // (K % SRC_EPL == 0: checked by the engine)
        ok = row < b.R[t] && k < b.K[t];
        elem = (size_t)min(row, b.R[t] - 1) * b.K[t] + min(k, b.K[t] - SRC_EPL);
      } else {
        const int piece = (rg * 1024 + (wave * 4 + j) * 64 + lane) * W;  // (kcs = 1: rg = the unit); W consecutive 16-byte pieces of the slot per lane
        ok = piece < b.R[t];
        elem = (size_t)min(piece, b.R[t] - W) * EPV;  // (R % W == 0 and R >= W for fp8 blobs)
      }
      raw[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + elem * SRC_ES));
      okmask |= (ok ? 1u : 0u) << j;
    }
  };
  int i = blockIdx.x;
  if (i < total) load(i);
  for (; i < total; i += gridDim.x) {
    int t, rg, kc;
    decode(i, t, rg, kc);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32x4 o[W];
      widen(raw[j], o);
#pragma unroll
      for (int w = 0; w < W; ++w) {
        if (!((okmask >> j) & 1u)) o[w] = u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(lds + (wave * 4 + j) * LROW + (lane * W + w) * 16) = o[w];
      }
    }
    __syncthreads();
    if (i + (int)gridDim.x < total) load(i + gridDim.x);  // in flight while this unit leaves through LDS
    char* dst = reinterpret_cast<char*>(b.dst) + b.dst_off[t];
    if (b.K[t] > 0) {
      const int KB = (b.K[t] + EPT - 1) / EPT;
#pragma unroll
      for (int j = 0; j < 4 * W; ++j) {
        const int kbl = wave * 4 * W + j, kb = kc * 16 * W + kbl;  // tile of this unit: row lane & 15, 16-byte piece lane >> 4
        if (kb < KB) {
          const u32x4 w4 = *reinterpret_cast<const u32x4*>(lds + (lane & 15) * LROW + kbl * 64 + (lane >> 4) * 16);
          *reinterpret_cast<u32x4*>(dst + ((size_t)rg * KB + kb) * 1024 + lane * 16) = w4;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int w = 0; w < W; ++w) {
          const int piece = (rg * 1024 + (wave * 4 + j) * 64 + lane) * W + w;
          if (piece < b.R[t]) *reinterpret_cast<u32x4*>(dst + (size_t)piece * 16) = *reinterpret_cast<const u32x4*>(lds + (wave * 4 + j) * LROW + (lane * W + w) * 16);
        }
    }
    __syncthreads();
  }
  if (ts && threadIdx.x == 0) {
    atomicMax(&ts[1], (unsigned long long)wall_clock64());
    __threadfence();
    atomicAdd(&ts[2], 1ull);
  }
}
hipError_t launch_pull_retile(const RetileBlob& b, int dtype, int workgroups, hipStream_t st, unsigned long long* ts, int first) {
  const dim3 grid(workgroups < 1 ? 1 : workgroups);
  if (b.src_f8) {
    if (dtype != DT_BF16) return hipErrorInvalidValue;  // fp8 host blobs are up-cast to bf16 slots only
    hipLaunchKernelGGL((pull_retile_kernel<uint16_t, true>), grid, dim3(256), 0, st, b, ts, first);
  } else if (dtype != DT_F32) hipLaunchKernelGGL((pull_retile_kernel<uint16_t, false>), grid, dim3(256), 0, st, b, ts, first);
  else hipLaunchKernelGGL((pull_retile_kernel<float, false>), grid, dim3(256), 0, st, b, ts, first);
  return hipGetLastError();
}
hipError_t launch_retile(const void* src, void* dst, int R, int K, int dtype, hipStream_t st) {
  const int ept = dtype == DT_F32 ? 16 : 32;
  dim3 grid(((K + ept - 1) / ept + 3) / 4, (R + 15) / 16);
  if (dtype != DT_F32) hipLaunchKernelGGL(retile_kernel<uint16_t>, grid, dim3(256), 0, st, (const uint16_t*)src, (char*)dst, R, K);
  else hipLaunchKernelGGL(retile_kernel<float>, grid, dim3(256), 0, st, (const float*)src, (char*)dst, R, K);
  return hipGetLastError();
}

template <typename T, int NMAT, int NW, int U, int NT>
__global__ __launch_bounds__(NW * 64) void ffn_rows_kernel(FfnStage s) {
  __shared__ float red[NW][NMAT][256];

  // Prologue loads in two dependent rounds instead of three: active[u] is fetched together with n_active (entries
  // past n_active hold stale but valid expert ids — the array is zero-initialised and only ever written with ids).
  const int u = blockIdx.y;
  const int e = s.active[u];
  const int nact_dev = *s.n_active;
  asm volatile("" ::"s"(e), "s"(nact_dev));  // keep both loads ahead of the exit branch (the compiler would sink active[u] below it)
  const int nact = s.n_active_host >= 0 ? s.n_active_host : nact_dev;
  if (u >= nact) return;
  const bool sh = (e == s.E);
  const int R = sh ? s.R_sh : s.R;
  const int r0 = blockIdx.x * 16;
  if (r0 >= R) return;
  const int off = s.offsets[e];
  const int cnt_e = s.counts[e];  // loaded alongside wptr[e], not after it: one dependent round trip less per block
  const char* W = reinterpret_cast<const char*>(s.wptr[e]);
  // an absent expert (never on the sync-free path) computes nothing but still reports its arrival below
  if (W == nullptr && threadIdx.x == 0 && blockIdx.x == 0) atomicExch(s.miss_flag, 1);
  const int cnt = W ? cnt_e : 0;
  const int tid = threadIdx.x;
  ffn_rows_item<T, NMAT, NW, U, NT>(s, blockIdx.x, W, sh, cnt, off, red);
  if constexpr (NMAT == 1 && NT == 1) {
    if (s.fuse_combine) {
      // this block's y columns [r0, r0+16) are written; the last of the layer's `nact` blocks to arrive for
      // the column tile combines it for every token (fixed ascending-expert order -> deterministic)
      __shared__ int is_last;
      wait_stores_acked();  // this thread's (write-through) y stores have reached device-coherent memory
      __syncthreads();
      if (tid == 0) is_last = (__hip_atomic_fetch_add(&s.tile_done[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nact - 1);
      __syncthreads();
      if (is_last) {
        for (int i = tid; i < s.comb.T * 4; i += NW * 64) combine_cols<T, true>(s.comb, i >> 2, r0 + (i & 3) * 4);
        if (tid == 0) s.tile_done[blockIdx.x] = 0;
      }
    }
  }
}


// grouped GEMM variants for experts with many rows (ffn_gemm.hip); false: not handled (MOEINF_FFN_GEMM=0)
template <typename T, int NMAT>
bool launch_ffn_gemm(const FfnStage& s, dim3 grid, int max_rows, hipStream_t st);
bool launch_ffn_gemm_big(const FfnStage& s, int nmat, dim3 grid, int max_rows, hipStream_t st);  // ffn_gemm_big.hip (bf16, fp16)
bool launch_ffn_gemm_ring2_f16(const FfnStage& s, int nmat, dim3 grid, int max_rows, hipStream_t st);  // ffn_gemm.hip

template <typename T, int NMAT>
static void launch_ffn_t(const FfnStage& s, dim3 grid, int nw, int u, bool many_tokens, int max_rows, hipStream_t st) {
#define LAUNCH(NWV, UU, NTT) KL((ffn_rows_kernel<T, NMAT, NWV, UU, NTT>), grid, dim3(NWV * 64), 0, st, s)
  if (many_tokens) {  // grouped GEMM kernels (ffn_gemm.hip); MOEINF_FFN_GEMM=0: the decode kernel looping 4 token tiles
    if constexpr (sizeof(T) == 2 && !std::is_same<T, uint16_t>::value) {
      // fp16 experts: the register ring first (long reductions, up to 340 rows per expert; its own translation unit), then
      // the same choice between the hybrid / LDS-staged / 256 x 256 kernels as bf16 (ffn_gemm_f16.hip, round 5)
      if (launch_ffn_gemm_ring2_f16(s, NMAT, grid, max_rows, st)) return;
      if (launch_ffn_gemm<T, NMAT>(s, grid, max_rows, st)) return;
    } else {
      if (launch_ffn_gemm<T, NMAT>(s, grid, max_rows, st)) return;
    }
    if (nw == 8) LAUNCH(8, 1, 4); else LAUNCH(4, 1, 4);
    return;
  }
  if (nw == 16) { LAUNCH(16, 4, 1); }
  else if (nw == 8) { if (u == 2) LAUNCH(8, 2, 1); else if (u == 8) LAUNCH(8, 8, 1); else LAUNCH(8, 4, 1); }
  else         { if (u == 2) LAUNCH(4, 2, 1); else if (u == 8) LAUNCH(4, 8, 1); else LAUNCH(4, 4, 1); }
#undef LAUNCH
}

hipError_t launch_ffn_stage(const FfnStage& s, int max_active, int max_rows_per_expert, hipStream_t st) {
  static const int env_nw = env_int("MOEINF_FFN_NW", 0), env_u = env_int("MOEINF_FFN_U", 0);
  const int rmax = s.R > s.R_sh ? s.R : s.R_sh;
  dim3 grid((rmax + 15) / 16, max_active);
  const bool gated = (s.epi == EPI_GATED_SILU || s.epi == EPI_GATED_GELU);
  // long reductions get 8 waves per block (more bytes in flight per CU), short ones 4
  const int kmax = s.K > s.K_sh ? s.K : s.K_sh;
  const size_t kbytes = (size_t)kmax * dt_bytes(s.dtype);
  // ... and a grid of at most one workgroup per CU (Switch-base-8 at batch 1: 192 / 48 workgroups for 256 CUs) SIXTEEN: a CU
  // that owns a single work item has nothing else to hide its load latency behind, so the whole item goes in flight at
  // once (round 4: stage 2 of Switch-base-8 streamed 9.45 MB in 16.8 us = 0.07 of HBM peak with 48 four-wave workgroups)
  const int kb_tiles = (int)(kbytes / 64);
  const bool few = (int64_t)grid.x * grid.y <= 256 && kb_tiles >= 32 && !s.fuse_combine;
  const int nw = env_nw ? env_nw : (few ? 16 : (kbytes >= 16384 ? 8 : 4));
  const int u = env_u ? env_u : 4;
  static const int env_nt = env_int("MOEINF_FFN_NT", 0);
  // the decode kernel re-streams an expert's weights for every 16 rows: from 17 rows on, the GEMM kernels (one pass
  // per 128/256 rows) win — Mixtral at 64 tokens: 761 -> 549 us per layer (profiles/r01_ffn_sweep_midsize.txt)
  static const int many_rows = env_int("MOEINF_FFN_MANY_ROWS", 16);
  const bool many = s.fuse_combine ? false : (env_nt ? env_nt > 1 : max_rows_per_expert > many_rows);
  if (s.dtype == DT_BF16) {
    if (gated) launch_ffn_t<uint16_t, 2>(s, grid, nw, u, many, max_rows_per_expert, st); else launch_ffn_t<uint16_t, 1>(s, grid, nw, u, many, max_rows_per_expert, st);
  } else if (s.dtype == DT_F16) {
    if (gated) launch_ffn_t<half_t, 2>(s, grid, nw, u, many, max_rows_per_expert, st); else launch_ffn_t<half_t, 1>(s, grid, nw, u, many, max_rows_per_expert, st);
  } else {
    if (gated) launch_ffn_t<float, 2>(s, grid, nw, u, many, max_rows_per_expert, st); else launch_ffn_t<float, 1>(s, grid, nw, u, many, max_rows_per_expert, st);
  }
  return hipGetLastError();
}

template <typename XT, typename WT, int TT>
__global__ __launch_bounds__(256) void gate_logits_kernel(const XT* __restrict__ x, const WT* __restrict__ wg,
                                                          float* __restrict__ logits, int T, int H, int E,
                                                          int round_bf16) {
  __shared__ double red[4][TT];
  gate_body<XT, WT, TT>(x, wg, logits, T, H, E, round_bf16, red, blockIdx.x, blockIdx.y * TT);
}

// Decode-sized DeepSeek forwards: the always-resident SHARED expert does not depend on the routing, so its FFN rides
// along with the router instead of sitting behind it — stage 1 in the gate launch (this kernel: blocks [0, n_gate) are
// gate blocks, the rest own 16 rows of the shared gate/up projections), stage 2 in the route/index launch
// (route_shared2_kernel).  The two router launches are latency-bound and leave HBM idle; the shared expert is a
// quarter of the layer's weight bytes (34.6 of 138 MB for DeepSeek-V2-Lite).
// NW: waves of a workgroup; the gate blocks always work with four (waves 4.. of a wider workgroup leave at once — a
// terminated wave does not count at the barriers of gate_body)
template <typename XT, typename WT, int TT, typename T, int U, int NW = 4>
__global__ __launch_bounds__(NW * 64) void gate_shared1_kernel(const XT* __restrict__ x, const WT* __restrict__ wg, float* __restrict__ logits,
                                                               int T_, int H, int E, int round_bf16, int n_gate, FfnStage s) {
  __shared__ double redg[4][TT];
  __shared__ float red[NW][2][256];
  const int b = blockIdx.x;
  if (b < n_gate) {
    if (NW > 4 && threadIdx.x >= 256) return;
    gate_body<XT, WT, TT>(x, wg, logits, T_, H, E, round_bf16, redg, b % E, (b / E) * TT);
  } else {
    const char* W = reinterpret_cast<const char*>(s.wptr[s.E]);
    ffn_rows_item<T, 2, NW, U, 1>(s, b - n_gate, W, true, T_, 0, red);
  }
}

// Mixtral's gate is an nn.Linear in the model dtype (mixtral.py:46): its output is rounded to
// that dtype.  The other routers compute fp32 logits from (exactly) up-cast inputs.
// (1: to bf16, 2: to fp16, 0: fp32 logits)
static inline int gate_rounds_bf16(const RouteArgs& a) { return a.kind != 0 /*MIXTRAL*/ ? 0 : (a.x_dtype == DT_BF16 ? 1 : (a.x_dtype == DT_F16 ? 2 : 0)); }

hipError_t launch_gate_shared1(const RouteArgs& a, const FfnStage& s, hipStream_t st) {
  constexpr int TT = 4;
  const int n_gate = a.E * ((a.T + TT - 1) / TT);
  dim3 grid(n_gate + (s.R_sh + 15) / 16);
  const int rb = gate_rounds_bf16(a);
  // bf16 model (DeepSeek): activations bf16, gate bf16 or fp32
  static const int u8 = env_int("MOEINF_SH1_U", 8) == 8;  // 8 tiles per wave and matrix per batch: 1.035 -> 1.007 ms/token (DeepSeek-V2-Lite)
  // MOEINF_SH1_NW=8: eight waves per workgroup (a shared-expert work item of 16 rows x 2 matrices x K = 128 KB for
  // DeepSeek-V2-Lite then goes in flight in ONE batch of loads per wave instead of two)
  static const int nw8 = env_int("MOEINF_SH1_NW", 8) == 8;  // round 4: 0.984 -> 0.968 ms/token (A/B/A in one run); 4 = the four-wave form
#define GS1(WT, UU) hipLaunchKernelGGL((gate_shared1_kernel<uint16_t, WT, TT, uint16_t, UU>), grid, dim3(256), 0, st, (const uint16_t*)a.x, (const WT*)a.gate_w, a.logits, a.T, a.H, a.E, rb, n_gate, s)
#define GS8(WT, UU) hipLaunchKernelGGL((gate_shared1_kernel<uint16_t, WT, TT, uint16_t, UU, 8>), grid, dim3(512), 0, st, (const uint16_t*)a.x, (const WT*)a.gate_w, a.logits, a.T, a.H, a.E, rb, n_gate, s)
  if (a.x_dtype == DT_F16) {  // fp16 model (round 5): the default eight-wave form only, gate fp16 or fp32
    if (a.gate_dtype == DT_F16) hipLaunchKernelGGL((gate_shared1_kernel<half_t, half_t, TT, half_t, 8, 8>), grid, dim3(512), 0, st, (const half_t*)a.x, (const half_t*)a.gate_w, a.logits, a.T, a.H, a.E, rb, n_gate, s);
    else if (a.gate_dtype == DT_F32) hipLaunchKernelGGL((gate_shared1_kernel<half_t, float, TT, half_t, 8, 8>), grid, dim3(512), 0, st, (const half_t*)a.x, (const float*)a.gate_w, a.logits, a.T, a.H, a.E, rb, n_gate, s);
    else return hipErrorInvalidValue;
  } else if (nw8) {
    if (a.gate_dtype == DT_BF16) GS8(uint16_t, 8); else GS8(float, 8);
  } else if (a.gate_dtype == DT_BF16) { if (u8) GS1(uint16_t, 8); else GS1(uint16_t, 4); }
  else { if (u8) GS1(float, 8); else GS1(float, 4); }
#undef GS1
#undef GS8
  return hipGetLastError();
}

// Prefill-sized router GEMM on the fp64 MATRIX instruction.  gate_logits_kernel is shaped for decode (one workgroup per
// expert and four tokens, every product converted and accumulated on the VALU): at 4 096 DeepSeek-V2-Lite tokens it is
// 65 536 tiny workgroups and 160 us — 11 % of the layer.  Here logits[T, E] = x[T, H] . Wg[E, H]^T runs on
// v_mfma_f64_16x16x4_f64 (same fp64 accumulation, so the routing stays independent of the summation order): a workgroup
// owns a 16-token x 16-expert tile, its four waves split K four ways (chunks of 64, round-robin) and wave 0 adds the
// partials in a fixed order.  No LDS staging of the operands: the sum over a 64-wide chunk of K may visit k in any order as
// long as both operands agree, so lane (r, q) covers k = 16 q + s for the chunk's 16 MFMA steps s — 32 contiguous bytes
// of its row per operand (bf16).
template <typename V>
__device__ __forceinline__ void load16(const V* p, float out[16]) { load8<V>(p, out); load8<V>(p + 8, out + 8); }

template <typename XT, typename WT>
__global__ __launch_bounds__(256) void gate_logits_mfma_kernel(const XT* __restrict__ x, const WT* __restrict__ wg, float* __restrict__ logits,
                                                               int T, int H, int E, int round_bf16) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  __shared__ double red[3][64][4];  // partials of waves 1..3
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int t0 = blockIdx.x * 16, e0 = blockIdx.y * 16;
  const XT* xrow = x + (size_t)min(t0 + r, T - 1) * H + q * 16;
  const WT* wrow = wg + (size_t)min(e0 + r, E - 1) * H + q * 16;
  d4 acc = d4{0.0, 0.0, 0.0, 0.0};
  const int NC = H >> 6;
  for (int c = wave; c < NC; c += 4) {
    float xf[16], wf[16];
    load16<XT>(xrow + c * 64, xf);
    load16<WT>(wrow + c * 64, wf);
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)xf[s2], (double)wf[s2], acc, 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int v = 0; v < 4; ++v) red[wave - 1][lane][v] = acc[v];
  }
  __syncthreads();
  if (wave > 0) return;
  // D layout of the 16x16 fp64 tile: register v of lane (r, q) is row q + 4 v (token), column r (expert) — NOT the
  // 4 q + v of the fp32 16x16 shapes (found by the golden-vector tests)
  const int e = e0 + r;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int t = t0 + q + 4 * v;
    if (t < T && e < E) {
      float f = (float)(((acc[v] + red[0][lane][v]) + red[1][lane][v]) + red[2][lane][v]);
      if (round_bf16 == 1) f = bf2f(f2bf(f)); else if (round_bf16 == 2) f = h2f(f2h(f));
      logits[(size_t)t * E + e] = f;
    }
  }
}

hipError_t launch_gate_logits(const RouteArgs& a, hipStream_t st) {
  constexpr int TT = 4;
  dim3 grid(a.E, (a.T + TT - 1) / TT);
  const int rb = gate_rounds_bf16(a);
  // the fp64-matrix form from 128 (16-token x 16-expert) tiles on — below that its few workgroups lose to the decode-shaped
  // kernel (measured: DeepSeek-V2-Lite 4096 / 512 / 128 tokens route 207 -> 75, 58 -> 47, 36 -> 39 us; NLLB 2048 tokens
  // 195 -> 71; Mixtral, one padded tile: 4096 tokens 46 -> 30, 512 tokens 27 -> 39).  MOEINF_GATE_MFMA_TILES=0: never
  static const int mfma_tiles = env_int("MOEINF_GATE_MFMA_TILES", 128);
  if (mfma_tiles > 0 && (int64_t)((a.T + 15) / 16) * ((a.E + 15) / 16) >= mfma_tiles && (a.H & 63) == 0) {
    const dim3 g16((a.T + 15) / 16, (a.E + 15) / 16);
#define GM(XT, WT) hipLaunchKernelGGL((gate_logits_mfma_kernel<XT, WT>), g16, dim3(256), 0, st, (const XT*)a.x, (const WT*)a.gate_w, a.logits, a.T, a.H, a.E, rb)
    if (a.x_dtype == DT_F16) { if (a.gate_dtype == DT_F16) GM(half_t, half_t); else GM(half_t, float); }
    else if (a.x_dtype == DT_BF16 && a.gate_dtype == DT_BF16) GM(uint16_t, uint16_t);
    else if (a.x_dtype == DT_BF16) GM(uint16_t, float);
    else if (a.gate_dtype == DT_BF16) GM(float, uint16_t);
    else GM(float, float);
#undef GM
    return hipGetLastError();
  }
  // one token (batch-1 decode, the gate launch in front of the self-routing stage 1): ONE cross-lane reduction per workgroup
  // instead of four (the fp64 shuffles of the three absent tokens were most of the kernel: round 5, seen in the timelines of
  // csrc/layer_fused.hip — gate done after 1.6 us instead of 2.6)
#define GL(XT, WT) do { if (a.T == 1) hipLaunchKernelGGL((gate_logits_kernel<XT, WT, 1>), grid, dim3(256), 0, st, (const XT*)a.x, (const WT*)a.gate_w, a.logits, a.T, a.H, a.E, rb); \
                        else hipLaunchKernelGGL((gate_logits_kernel<XT, WT, TT>), grid, dim3(256), 0, st, (const XT*)a.x, (const WT*)a.gate_w, a.logits, a.T, a.H, a.E, rb); } while (0)
  if (a.x_dtype == DT_F16) { if (a.gate_dtype == DT_F16) GL(half_t, half_t); else GL(half_t, float); }
  else if (a.x_dtype == DT_BF16 && a.gate_dtype == DT_BF16) GL(uint16_t, uint16_t);
  else if (a.x_dtype == DT_BF16) GL(uint16_t, float);
  else if (a.gate_dtype == DT_BF16) GL(float, uint16_t);
  else GL(float, float);
#undef GL
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void route_topk_kernel(RouteArgs a) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t < a.T) route_token(a, t, threadIdx.x & 63);
}

hipError_t launch_route_topk(const RouteArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(route_topk_kernel, dim3((a.T + 3) / 4), dim3(256), 0, st, a);
  return hipGetLastError();
}

__global__ __launch_bounds__(IDX_THREADS) void dispatch_index_kernel(IndexArgs a) {
  __shared__ int wave_cnt[IDX_WAVES * IDX_MAXE];
  __shared__ int running[IDX_MAXE];
  __shared__ int offs[IDX_MAXE + 1];
  __shared__ int scan_tmp[IDX_MAXE];
  if (a.T * a.K <= 64 && !(a.capacity > 0 && a.T > a.rows)) {
    if (threadIdx.x < 64) index_small(a, running, offs);  // decode-sized: one wave, no workgroup barriers
  } else {
    index_body(a, wave_cnt, running, offs, scan_tmp);
  }
}

// ---- many pairs (long prefill): the same stable counting sort over MANY workgroups --------------------------------
// One workgroup ranks 1024 pairs in ~5 us and walks the chunks serially: 36 us at 3 072 pairs, ~1 ms at T = 16 k x K = 6.
//   index_count   grid = chunks: rank of every pair inside (its chunk, its expert) + per-chunk expert counts
//   index_scan    1 workgroup:   per expert, exclusive scan of the chunk counts; totals -> counts / offsets / active /
//                                host mirror; chunk bases rebased to expert-sorted rows
//   index_scatter grid = chunks: slot = chunk base + rank; permutation arrays
// Same outputs, bit for bit, as index_body (ranks are stable in pair order).  Not for the Switch per-row capacity
// pass (a sequential cumsum per batch row), which stays on one workgroup.
__global__ __launch_bounds__(IDX_THREADS) void index_count_kernel(IndexArgs a, int32_t* __restrict__ chunk_cnt) {
  __shared__ int wave_cnt[IDX_WAVES * IDX_MAXE];
  __shared__ int running[IDX_MAXE];
  const int tid = threadIdx.x, E = a.E, npairs = a.T * a.K;
  for (int i = tid; i < IDX_MAXE; i += IDX_THREADS) running[i] = 0;
  __syncthreads();
  const int p = blockIdx.x * IDX_THREADS + tid;
  const bool in = p < npairs;
  const int key = in ? IDX_AT(a, p) : -1;
  const bool counted = in && key >= 0 && key < E && (a.pair_valid ? a.pair_valid[p] != 0 : true);
  const int pos = chunk_rank(counted ? key : 0, counted, wave_cnt, running, E);
  if (in) a.pair_slot[p] = pos;  // rank inside the chunk for now (-1: not dispatched)
  for (int e = tid; e < E; e += IDX_THREADS) chunk_cnt[(size_t)blockIdx.x * E + e] = running[e];
}
__global__ __launch_bounds__(IDX_THREADS) void index_scan_kernel(IndexArgs a, int32_t* __restrict__ chunk_cnt, int nchunks) {
  __shared__ int tot[IDX_MAXE];
  __shared__ int offs[IDX_MAXE + 1];
  const int tid = threadIdx.x, E = a.E, ne = E + 1, T = a.T;
  if (tid < E) {  // exclusive scan over the chunks, in place
    int base = 0;
    for (int c = 0; c < nchunks; ++c) {
      const int v = chunk_cnt[(size_t)c * E + tid];
      chunk_cnt[(size_t)c * E + tid] = base;
      base += v;
    }
    tot[tid] = base;
  }
  if (tid == E) tot[E] = a.shared ? T : 0;
  __syncthreads();
  if (tid == 0) {
    int acc = 0, na = 0;
    for (int e = 0; e < ne; ++e) {
      offs[e] = acc;
      acc += tot[e];
      if (tot[e] > 0) a.active[na++] = e;
    }
    offs[ne] = acc;
    *a.n_active = na;
    if (a.mirror) {
      a.mirror[0] = na;
      for (int i = 0; i < na; ++i) a.mirror[1 + ne + i] = a.active[i];
      for (int i = na; i < ne; ++i) a.mirror[1 + ne + i] = -1;
    }
  }
  __syncthreads();
  if (tid < ne) {
    a.counts[tid] = tot[tid];
    if (a.mirror) a.mirror[1 + tid] = tot[tid];
  }
  if (tid <= ne) a.offsets[tid] = offs[tid];
  if (tid < E) {
    const int o = offs[tid];
    for (int c = 0; c < nchunks; ++c) chunk_cnt[(size_t)c * E + tid] += o;
  }
}
__global__ __launch_bounds__(IDX_THREADS) void index_scatter_kernel(IndexArgs a, const int32_t* __restrict__ chunk_base) {
  const int npairs = a.T * a.K, E = a.E;
  const int p = blockIdx.x * IDX_THREADS + threadIdx.x;
  if (p < npairs) {
    const int rk = a.pair_slot[p];
    if (rk >= 0) {
      const int slot = chunk_base[(size_t)blockIdx.x * E + IDX_AT(a, p)] + rk;
      a.pair_slot[p] = slot;
      a.slot_token[slot] = p / a.K;
      a.slot_pair[slot] = p;
    }
  }
  if (a.shared) {
    const int base = a.offsets[E];  // written by index_scan (previous launch)
    for (int t = p; t < a.T; t += gridDim.x * IDX_THREADS) {
      a.slot_token[base + t] = t;
      a.slot_pair[base + t] = -1;
    }
  }
}
hipError_t launch_dispatch_index_wide(const IndexArgs& a, int32_t* chunk_scratch, hipStream_t st) {
  const int nchunks = (a.T * a.K + IDX_THREADS - 1) / IDX_THREADS;
  hipLaunchKernelGGL(index_count_kernel, dim3(nchunks), dim3(IDX_THREADS), 0, st, a, chunk_scratch);
  hipLaunchKernelGGL(index_scan_kernel, dim3(1), dim3(IDX_THREADS), 0, st, a, chunk_scratch, nchunks);
  hipLaunchKernelGGL(index_scatter_kernel, dim3(nchunks), dim3(IDX_THREADS), 0, st, a, chunk_scratch);
  return hipGetLastError();
}

hipError_t launch_dispatch_index(const IndexArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(dispatch_index_kernel, dim3(1), dim3(IDX_THREADS), 0, st, a);
  return hipGetLastError();
}

// decode-sized batches: softmax/top-k of every token (one wave each) and the dispatch index in ONE
// launch of one workgroup — saves a kernel boundary per layer where launches dominate the layer time
// pk.on: the expert-parallel send rows are written by this workgroup too (T*K <= 64; ep_pack_block, kdev.h)
__global__ __launch_bounds__(IDX_THREADS) void route_index_kernel(RouteArgs r, IndexArgs a, EpFuse pk) {
  __shared__ int wave_cnt[IDX_WAVES * IDX_MAXE];
  __shared__ int running[IDX_MAXE];
  __shared__ int offs[IDX_MAXE + 1];
  __shared__ int scan_tmp[IDX_MAXE];
  for (int t = threadIdx.x >> 6; t < r.T; t += IDX_WAVES) route_token(r, t, threadIdx.x & 63);
  __threadfence_block();
  __syncthreads();
  if (a.T * a.K <= 64 && !(a.capacity > 0 && a.T > a.rows)) {
    if (threadIdx.x < 64) index_small(a, running, offs);
  } else {
    index_body(a, wave_cnt, running, offs, scan_tmp);
  }
  if (pk.on) {  // block-uniform
    __threadfence_block();
    __syncthreads();  // pair_valid may have been edited by the index (Switch capacity)
    ep_pack_block_dt(pk, a.T * a.K, scan_tmp);
  }
}

// block 0: softmax/top-k of every token (4 waves) + the one-wave dispatch index (T*K <= 64); blocks 1..: 16 rows each of
// the shared expert's down projection over h_shared (written by gate_shared1_kernel)
template <typename T, int NW, int U>
__global__ __launch_bounds__(NW * 64) void route_shared2_kernel(RouteArgs r, IndexArgs a, FfnStage s, EpFuse pk) {
  __shared__ int running[IDX_MAXE];
  __shared__ int offs[IDX_MAXE + 1];
  __shared__ float red[NW][1][256];
  if (blockIdx.x == 0) {
    for (int t = threadIdx.x >> 6; t < r.T; t += NW) route_token(r, t, threadIdx.x & 63);
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x < 64) index_small(a, running, offs);
    if (pk.on) {  // expert-parallel: this workgroup writes the send rows as well (ep_pack_block, kdev.h)
      __syncthreads();
      ep_pack_block_dt(pk, a.T * a.K, reinterpret_cast<int*>(&red[0][0][0]));
    }
  } else {
    const char* W = reinterpret_cast<const char*>(s.wptr[s.E]);
    ffn_rows_item<T, 1, NW, U, 1>(s, (int)blockIdx.x - 1, W, true, r.T, 0, red);
  }
}
static EpFuse no_pack() {
  EpFuse f;
  memset(&f, 0, sizeof f);
  return f;
}
hipError_t launch_route_shared2(const RouteArgs& r, const IndexArgs& a, const FfnStage& s, hipStream_t st, const EpFuse* pack) {
  static const int nw = env_int("MOEINF_SH2_NW", 8), u = env_int("MOEINF_SH2_U", 4);
  const dim3 grid(1 + (s.R_sh + 15) / 16);
  const EpFuse pk = pack ? *pack : no_pack();
#define RS2(NWV, UU) hipLaunchKernelGGL((route_shared2_kernel<uint16_t, NWV, UU>), grid, dim3(NWV * 64), 0, st, r, a, s, pk)
  if (s.dtype == DT_F16) hipLaunchKernelGGL((route_shared2_kernel<half_t, 8, 4>), grid, dim3(512), 0, st, r, a, s, pk);  // fp16: the default form only
  else if (nw == 16) { if (u == 8) RS2(16, 8); else RS2(16, 4); }
  else if (nw == 4) { if (u == 8) RS2(4, 8); else RS2(4, 4); }
  else { if (u == 8) RS2(8, 8); else RS2(8, 4); }
#undef RS2
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// ffn1_selfroute: batch-1 decode (T == 1) of the gated families (Mixtral, DeepSeek), sync-free path.  FFN stage 1
// ROUTES FOR ITSELF: every wave of every block repeats the token's softmax/top-k from the E gate logits (a few
// hundred VALU instructions on registers, bit-identical by construction: the same route_core) and picks "its" expert
// = the blockIdx-th smallest chosen id — so the top-k/index launch and its kernel boundary leave the layer's critical
// path (Mixtral: gate 4.6 + route_index 5.9 us of a 125 us layer; DeepSeek: 7.9 of 47 us).  The logits and EVERY
// expert's blob pointer are fetched in one round of independent loads (lane j holds wptr[j]; the chosen pointer is a
// v_readlane away), so the weight stream starts two dependent round trips after the launch instead of four.
// Requires E <= 64 and a greedy softmax top-K router (route_set_lean).
// One extra block ("meta") writes what the later launches and the host read: top-k ids/weights/order, the dispatch
// index (counts, offsets, active list, slots) and the pinned routing mirror.  With a DeepSeek shared expert hidden
// under the router, the first n_sh2 blocks are its stage 2 (its stage 1 rode in the gate launch).
// grid = 1 (meta) + n_sh2 + K * ceil(R/16) blocks of NW waves.
// ------------------------------------------------------------------------------------------------
// amdgpu_num_sgpr: the meta block's generic router keeps its wave-uniform arrays in SGPRs and would push the kernel past
// 96, and 256-thread blocks are admitted per CU by floor(800 / (ceil(sgpr/16)*16 + 16)) (MI355X_MICROARCH.md): 106 SGPRs
// = 6 blocks per CU.  No VGPR cap: 76 VGPRs = 6 workgroups per CU is more than either model uses (a 72-VGPR cap for 7
// per CU spilled 12 bytes per thread = 2 MB of scratch writes per launch, for nothing once four per CU proved best).
// NMAT = 2: the gated families (Mixtral, DeepSeek); NMAT = 1 (round 4): Switch's plain ReLU experts, top-1 — the same three
// launches per layer (gate, this, ffn2_decode1) instead of five (gate, route_index, two FFN stages, combine).
template <typename T, int NMAT, int NW, int U>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_num_sgpr(80))) void ffn1_selfroute_kernel(RouteArgs r, IndexArgs a, FfnStage s, FfnStage sh2, int n_rg, int n_sh2) {
  __shared__ float red[NW][NMAT][256];
  __shared__ unsigned long long sh_w;
  __shared__ int sh_rank_ok;
  static_assert(sizeof(float) * NW * NMAT * 256 >= sizeof(int) * (2 * IDX_MAXE + 1), "index scratch aliases the reduction buffer");
  const int lane = threadIdx.x & 63;
  const int K = r.K, E = r.E;
  // block 0 is the meta block: it must be dispatched FIRST — when the grid exceeds one resident wave of blocks
  // (Mixtral: 1793 blocks, 6 per CU) a meta block at the end of the grid would start only when a slot frees up and
  // put its ~5 us of serial work (generic router, index, PCIe mirror writes) behind the weight stream's tail
  int b = (int)blockIdx.x - 1;
  // n_sh2 < 0 (MOEINF_SR_ORDER=1): the shared expert's (lighter) work items are dispatched LAST instead of first
  if (n_sh2 < 0) {
    n_sh2 = -n_sh2;
    if (b >= 0) { const int n_routed = (int)gridDim.x - 1 - n_sh2; b = b < n_routed ? b + n_sh2 : b - n_routed; }
  }
  if (b >= 0 && b < n_sh2) {  // shared expert, stage 2 (h_shared was written by the gate launch)
    const char* Wsh = reinterpret_cast<const char*>(sh2.wptr[sh2.E]);
    ffn_rows_item<T, 1, NW, U, 1>(sh2, b, Wsh, true, 1, 0, reinterpret_cast<float (*)[1][256]>(&red[0][0][0]));
    return;
  }
  b -= n_sh2;
  if (b < 0) {  // meta block: one wave
    if (threadIdx.x < 64) {
      int* scratch = reinterpret_cast<int*>(&red[0][0][0]);
      Routed o;
      route_core(r, 0, lane, o);
      int my_sel, rank;
      float my_w;
      route_store(r, 0, lane, o, &my_sel, &my_w, &rank);
      if (lane < K && s.dec_w) {  // batch-1 records for stage 2: blob pointer and combine weight by ascending expert id
        s.dec_w[rank] = my_sel >= 0 ? s.wptr[my_sel] : 0ull;
        s.dec_cw[rank] = my_w;
      }
      __threadfence_block();
      index_small(a, scratch, scratch + IDX_MAXE);
    }
    return;
  }
  const int u = b / n_rg, rg = b - u * n_rg;
  // ONE wave per block routes; the blob pointer travels through LDS.  (The generic route_core is > 1000 instructions
  // on this path — every router family, four experts per lane — and cost ~5 us in front of the weight stream;
  // route_set_lean is ~100-200.)
  if (threadIdx.x < 64) {
    // independent first round: the token's logits and all blob pointers
    uint64_t wp = 0;
    if (lane < E) wp = s.wptr[lane];
    uint64_t chosen = route_set_lean(r.logits, E, K, lane, r.kind == 2 /*SWITCH*/ ? r.x_dtype : DT_F32);
    for (int i = 0; i < u; ++i) chosen &= chosen - 1;  // drop the u smallest ids
    const int e = chosen ? (int)__builtin_ctzll(chosen) : -1;
    uint64_t wsel = 0;
    if (e >= 0) {
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wp, e);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(wp >> 32), e);
      wsel = ((uint64_t)hi << 32) | lo;
    }
    if (lane == 0) { sh_w = wsel; sh_rank_ok = e >= 0; }
  }
  __syncthreads();
  if (!sh_rank_ok) return;  // fewer than K experts selected (never with the supported routers)
  const char* W = reinterpret_cast<const char*>(sh_w);
  if (W == nullptr) {  // never on the sync-free path
    if (threadIdx.x == 0 && rg == 0) atomicExch(s.miss_flag, 1);
    return;
  }
  // T == 1: expert-sorted row of (token 0, expert e) = its rank u; the B operand is token 0
  ffn_rows_item<T, NMAT, NW, U, 1>(s, rg, W, false, 1, u, red, 0);
}

static thread_local hipEvent_t t_timer_start = nullptr, t_timer_stop = nullptr;
void arm_kernel_timer(hipEvent_t start, hipEvent_t stop) { t_timer_start = start; t_timer_stop = stop; }
bool take_kernel_timer(hipEvent_t* start, hipEvent_t* stop) {
  if (!t_timer_stop) return false;
  *start = t_timer_start; *stop = t_timer_stop;
  t_timer_start = t_timer_stop = nullptr;
  return true;
}
hipError_t launch_ffn1_selfroute(const RouteArgs& r, const IndexArgs& a, const FfnStage& s1, const FfnStage* sh2, hipStream_t st) {
  const int n_rg = (s1.R + 15) / 16;
  const int n_sh2 = sh2 ? (sh2->R_sh + 15) / 16 : 0;
  const dim3 grid(n_sh2 + r.K * n_rg + 1);
  // Extra dynamic LDS per workgroup = a cap on the workgroups resident per CU.  A grid of several workgroups per CU
  // (Mixtral: 1793) streams best with FOUR resident per CU (8 + 30 KB of LDS each), the rest dispatched as they retire:
  // 3.942 / 3.935 / 3.920 / 3.891 / 3.939 ms per token at 7 / 6 / 5 / 4 / 3 per CU — fewer concurrent DRAM streams,
  // staggered finishes.  Small grids (DeepSeek: 657 workgroups, all resident anyway) are left alone.
  static const int lds_env = env_int("MOEINF_SR_LDS_KB", -1);
  const size_t dyn = (size_t)(lds_env >= 0 ? lds_env : (grid.x > 4 * 256 ? 30 : 0)) * 1024;
  // tiles per wave and matrix fetched per batch: 8 for grids that are resident all at once (DeepSeek-V2-Lite: 657
  // workgroups, 1.035 -> 1.009 ms/token), 4 for multi-round grids (Mixtral: 1793 workgroups at four per CU)
  static const int sr_u_env = env_int("MOEINF_SR_U", 0);
  const int sr_u = sr_u_env ? sr_u_env : (grid.x > 4 * 256 ? 4 : 8);
  if (s1.epi != EPI_GATED_SILU) {
    // plain experts (Switch, top-1): a grid of at most one workgroup per CU gets sixteen waves per workgroup — the whole
    // work item in flight at once (see launch_ffn_stage)
    if (s1.dtype == DT_BF16) KL((ffn1_selfroute_kernel<uint16_t, 1, 16, 4>), grid, dim3(1024), 0, st, r, a, s1, s1, n_rg, 0);
    else if (s1.dtype == DT_F16) KL((ffn1_selfroute_kernel<half_t, 1, 16, 4>), grid, dim3(1024), 0, st, r, a, s1, s1, n_rg, 0);
    else KL((ffn1_selfroute_kernel<float, 1, 16, 4>), grid, dim3(1024), 0, st, r, a, s1, s1, n_rg, 0);
    return hipGetLastError();
  }
  if (s1.dtype == DT_F16) {  // fp16 gated families: the same kernel on the f16 matrix instruction
    if (sr_u == 8) KL((ffn1_selfroute_kernel<half_t, 2, 4, 8>), grid, dim3(256), dyn, st, r, a, s1, sh2 ? *sh2 : s1, n_rg, n_sh2);
    else KL((ffn1_selfroute_kernel<half_t, 2, 4, 4>), grid, dim3(256), dyn, st, r, a, s1, sh2 ? *sh2 : s1, n_rg, n_sh2);
    return hipGetLastError();
  }
  static const int sr_order = env_int("MOEINF_SR_ORDER", 0);
  const int n_sh2_arg = (sr_order && n_sh2 > 0) ? -n_sh2 : n_sh2;
  if (sr_u == 8) KL((ffn1_selfroute_kernel<uint16_t, 2, 4, 8>), grid, dim3(256), dyn, st, r, a, s1, sh2 ? *sh2 : s1, n_rg, n_sh2_arg);
  else KL((ffn1_selfroute_kernel<uint16_t, 2, 4, 4>), grid, dim3(256), dyn, st, r, a, s1, sh2 ? *sh2 : s1, n_rg, n_sh2_arg);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The same for SMALL decode batches (2..SR_MAX_T tokens; round 4): FFN stage 1 routes for itself for EVERY token.  The meta
// block (block 0) runs the generic router per token and the one-wave dispatch index — everything stage 2, the combine and the
// host read.  Every other workgroup repeats the tokens' top-k sets with route_set_lean (one call per token), takes the u-th
// smallest expert ANY token chose, and streams it over the rows of the tokens that chose it; its expert-sorted row offset is
// the number of (token, expert) pairs with smaller expert ids — the very order index_small produces (stable counting sort by
// expert, pairs in token order), so stage 2 (the generic kernel, combine fused) finds the rows where the index says they are.
// grid = 1 + n_sh2 + max_active * n_rg.  Saves the top-k/index launch of decode batches the batcher produces (serving).
constexpr int SR_MAX_T = 8;
template <typename T, int NW, int U>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_num_sgpr(96))) void ffn1_selfroute_multi_kernel(RouteArgs r, IndexArgs a, FfnStage s, FfnStage sh2, int n_rg, int n_sh2) {
  __shared__ float red[NW][2][256];
  __shared__ int s_in[SR_MAX_T];
  __shared__ unsigned long long s_w;
  __shared__ int s_cnt, s_off;
  static_assert(sizeof(float) * NW * 2 * 256 >= sizeof(int) * (2 * IDX_MAXE + 1), "index scratch aliases the reduction buffer");
  const int lane = threadIdx.x & 63;
  const int Tn = r.T;
  int blk = (int)blockIdx.x - 1;
  if (blk < 0) {  // meta block: dispatched first
    for (int t = threadIdx.x >> 6; t < Tn; t += NW) route_token(r, t, lane);
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x < 64) {
      int* scratch = reinterpret_cast<int*>(&red[0][0][0]);
      index_small(a, scratch, scratch + IDX_MAXE);
    }
    return;
  }
  if (blk < n_sh2) {  // hidden shared expert, stage 2, all tokens (h_shared was written by the gate launch)
    const char* Wsh = reinterpret_cast<const char*>(sh2.wptr[sh2.E]);
    ffn_rows_item<T, 1, NW, U, 1>(sh2, blk, Wsh, true, Tn, 0, reinterpret_cast<float (*)[1][256]>(&red[0][0][0]));
    return;
  }
  blk -= n_sh2;
  const int u = blk / n_rg, rg = blk - u * n_rg;
  if (threadIdx.x < 64) {
    uint64_t wp = 0;
    if (lane < r.E) wp = s.wptr[lane];
    uint64_t chosen[SR_MAX_T];
    uint64_t present = 0;
#pragma unroll
    for (int t = 0; t < SR_MAX_T; ++t) {
      chosen[t] = 0;
      if (t < Tn) chosen[t] = route_set_lean(r.logits + (size_t)t * r.E, r.E, r.K, lane);  // wave-uniform branch
      present |= chosen[t];
    }
    uint64_t m = present;
    for (int i = 0; i < u; ++i) m &= m - 1;  // drop the u smallest ids
    const int e = m ? (int)__builtin_ctzll(m) : -1;
    int cnt = 0, off = 0;
    if (e >= 0) {
      const uint64_t below = (1ull << e) - 1ull;
#pragma unroll
      for (int t = 0; t < SR_MAX_T; ++t) {
        off += __popcll(chosen[t] & below);
        if ((chosen[t] >> e) & 1ull) {
          if (lane == 0) s_in[cnt] = t;
          ++cnt;
        }
      }
    }
    uint64_t wsel = 0;
    if (e >= 0) {
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wp, e);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(wp >> 32), e);
      wsel = ((uint64_t)hi << 32) | lo;
    }
    if (lane == 0) { s_w = wsel; s_cnt = cnt; s_off = off; }
  }
  __syncthreads();
  const int cnt = s_cnt;
  if (cnt == 0) return;  // fewer than u+1 distinct experts chosen (block-uniform)
  const char* W = reinterpret_cast<const char*>(s_w);
  if (W == nullptr) {  // never on the sync-free path
    if (threadIdx.x == 0 && rg == 0) atomicExch(s.miss_flag, 1);
    return;
  }
  ffn_rows_item<T, 2, NW, U, 1>(s, rg, W, false, cnt, s_off, red, -1, s_in, nullptr);
}

hipError_t launch_ffn1_selfroute_multi(const RouteArgs& r, const IndexArgs& a, const FfnStage& s1, const FfnStage* sh2, int max_active, hipStream_t st) {
  const int n_rg = (s1.R + 15) / 16;
  const int n_sh2 = sh2 ? (sh2->R_sh + 15) / 16 : 0;
  const dim3 grid(1 + n_sh2 + max_active * n_rg);
  const size_t dyn = grid.x > 4 * 256 ? 30 * 1024 : 0;  // as launch_ffn1_selfroute
#define SRM(TT, UU) hipLaunchKernelGGL((ffn1_selfroute_multi_kernel<TT, 4, UU>), grid, dim3(256), dyn, st, r, a, s1, sh2 ? *sh2 : s1, n_rg, n_sh2)
  if (s1.dtype == DT_F16) { if (grid.x > 4 * 256) SRM(half_t, 4); else SRM(half_t, 8); }
  else { if (grid.x > 4 * 256) SRM(uint16_t, 4); else SRM(uint16_t, 8); }
#undef SRM
  return hipGetLastError();
}

// Stage 2 of a batch-1 self-routed forward, with the combine in its tail.  Differences from ffn_rows_kernel's fused
// form: one scalar round in the prologue (blob pointer dec_w[u] instead of active[u] -> {wptr, counts, offsets}); the
// combine weights (dec_cw, ascending expert id = rows 0..K-1 of y) are fetched at kernel START, so the last-arriving
// block's tail is one round of row loads instead of three dependent rounds (order -> slot/weight -> rows).
template <typename T, int NW, int U>
__global__ __launch_bounds__(NW * 64) void ffn2_decode1_kernel(FfnStage s) {
  __shared__ float red[NW][1][256];
  __shared__ int is_last;
  const int u = blockIdx.y, tid = threadIdx.x;
  const int K = s.comb.K;
  const char* W = reinterpret_cast<const char*>(s.dec_w[u]);
  CombineMeta m;
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) { m.slot[kk] = min(kk, K - 1); m.w[kk] = s.dec_cw[min(kk, K - 1)]; }
  const int r0 = blockIdx.x * 16;
  if (W == nullptr && tid == 0 && blockIdx.x == 0) atomicExch(s.miss_flag, 1);
  ffn_rows_item<T, 1, NW, U, 1>(s, blockIdx.x, W, false, W ? 1 : 0, u, red);
  wait_stores_acked();  // this thread's (write-through) y stores have reached device-coherent memory
  __syncthreads();
  // (top-1, Switch: this workgroup is the only one of its column tile — no counter to arrive at)
  if (tid == 0) is_last = K == 1 ? 1 : (__hip_atomic_fetch_add(&s.tile_done[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == K - 1);
  __syncthreads();
  if (is_last) {
    if (s.comb.kind == 2 /*SWITCH: out = Tr(router_prob * expert output), switch_transformers.py:99-109; K == 1*/) {
      if (tid < 4 && r0 + tid * 4 < s.R) {
        const int h0 = r0 + tid * 4;
        float v[4], o[4];
        DT<T>::unpack4(DT<T>::template fetch4<true>(reinterpret_cast<const T*>(s.comb.y) + h0), v);  // row 0 = (token 0, its expert)
        const float pr = s.comb.router_prob[0];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = DT<T>::round(pr * v[j]);
        DT<T>::store4(reinterpret_cast<T*>(s.comb.out) + h0, o);
      }
    } else if (tid < 4) {
      combine_apply<T, true>(s.comb, 0, r0 + tid * 4, m);
    }
    if (tid == 0 && K > 1) s.tile_done[blockIdx.x] = 0;
  }
}

// The same stage for K = 2 (Mixtral): ONE workgroup owns 16 output columns for BOTH chosen experts — wave group g
// streams expert slot g's 16 rows (k-tiles interleaved over its NWE waves), the partial sums meet in LDS, and the combine
// runs inside the workgroup: no write-through stores, no store-ack wait, no arrival counter, no coherent row loads
// (that tail costs 3.5 us per launch, tools/ffn_micro.hip).  H/16 = 256 workgroups for Mixtral = one per CU.
// Summation order inside an expert (waves 0..NWE-1, tiles in ascending k inside a wave) and the combine order
// (ascending expert id) are those of ffn2_decode1_kernel, so the two produce identical bits.  Requires K % 32 == 0.
template <typename T, int NWE, int U>  // T: uint16_t = bf16, half_t = fp16 (round 5)
__global__ __launch_bounds__(2 * NWE * 64) void ffn2_decode1_pair_kernel(FfnStage s) {
  constexpr int EPT = 32, EPV = 8;
  __shared__ float red[2][NWE][16];
  __shared__ float yv[2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = wave / NWE, wl = wave - g * NWE;
  const int n = lane & 15, q = lane >> 4;
  const int rg = blockIdx.x, r0 = rg * 16;
  const char* W = reinterpret_cast<const char*>(s.dec_w[g]);
  const float cw0 = s.dec_cw[0], cw1 = s.dec_cw[1];
  if (W == nullptr) {  // never on the sync-free path
    if (tid == 0 && rg == 0) atomicExch(s.miss_flag, 1);
    return;
  }
  const int KB = s.K / EPT;
  const char* a0 = W + s.off_a + (size_t)rg * KB * 1024 + lane * 16;
  const T* xr = reinterpret_cast<const T*>(s.in) + (size_t)g * s.ld_in + q * EPV;  // T == 1: h row of slot g
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int kb = wl; kb < KB; kb += U * NWE) {
    u32x4 av[U], xv[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
      if (kb + i * NWE < KB) {
        av[i] = ld16_nt(a0 + (size_t)(kb + i * NWE) * 1024);
        xv[i] = ld16(xr + (size_t)(kb + i * NWE) * EPT);
      }
    }
#pragma unroll
    for (int i = 0; i < U; ++i)
      if (kb + i * NWE < KB) mma16<T>(acc, av[i], xv[i]);
  }
  // every token column of the accumulator holds the same token: lanes n == 0 carry rows q*4 .. q*4+3
  if (n == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) red[g][wl][q * 4 + j] = acc[j];
  }
  __syncthreads();
  if (tid < 32) {  // (expert slot, row): sum the K split in wave order, round once, publish y
    const int gg = tid >> 4, row = tid & 15;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NWE; ++w) v += red[gg][w][row];
    v = DT<T>::round(v);
    yv[gg][row] = v;
    if (r0 + row < s.R) DT<T>::store(reinterpret_cast<T*>(s.out) + (size_t)gg * s.ld_out + r0 + row, v);
  }
  __syncthreads();
  if (tid < 16 && r0 + tid < s.R) {  // combine_apply for one column, ascending expert id (kinds 0 / 1 without a shared expert)
    float p0 = yv[0][tid] * cw0, p1 = yv[1][tid] * cw1;
    if (s.comb.kind != 1) { p0 = DT<T>::round(p0); p1 = DT<T>::round(p1); }
    float o = DT<T>::round(0.f + p0);
    o = DT<T>::round(o + p1);
    DT<T>::store(reinterpret_cast<T*>(s.comb.out) + r0 + tid, o);
  }
}

// (K = 3..8 — DeepSeek: six routed experts + the hidden shared expert — keeps the arrival-counter form.  ONE workgroup per column
// block with every chosen expert and the combine inside it was built twice: on half tiles, eight columns, 256 workgroups (round 3)
// and on whole tiles, 128 workgroups of twelve waves (round 6); both measured slower than the tail they remove — 33.3 / 33.0-33.5
// against 32.5 us per DeepSeek-V2-Lite layer, profiles/r06_deepseek_stage2_group_forms_rejected.txt — and both are deleted.)
hipError_t launch_ffn2_decode1(const FfnStage& s2, hipStream_t st) {
  static const int pair_env = env_int("MOEINF_DEC1_PAIR", 1);  // 0: always the arrival-counter form; 4 / 8: waves per expert
  if (pair_env && (s2.dtype == DT_BF16 || s2.dtype == DT_F16) && s2.comb.K == 2 && (s2.K % 32) == 0 && !(s2.comb.kind == 1 && s2.comb.y_shared) && s2.comb.kind <= 1) {
    const dim3 g1((s2.R + 15) / 16);
    // 4 waves per expert (8 per CU), batches of 4 tiles: 38.9 us per Mixtral launch; 8 waves per expert 40.5; the
    // arrival-counter form 41.9
    static const int pu = env_int("MOEINF_DEC1_PAIR_U", 4);
    if (s2.dtype == DT_F16) KL((ffn2_decode1_pair_kernel<half_t, 4, 4>), g1, dim3(512), 0, st, s2);  // fp16: the default form only
    else if (pair_env == 8) KL((ffn2_decode1_pair_kernel<uint16_t, 8, 4>), g1, dim3(1024), 0, st, s2);
    else if (pu == 8) KL((ffn2_decode1_pair_kernel<uint16_t, 4, 8>), g1, dim3(512), 0, st, s2);
    else if (pu == 2) KL((ffn2_decode1_pair_kernel<uint16_t, 4, 2>), g1, dim3(512), 0, st, s2);
    else KL((ffn2_decode1_pair_kernel<uint16_t, 4, 4>), g1, dim3(512), 0, st, s2);
    return hipGetLastError();
  }
  const dim3 grid((s2.R + 15) / 16, s2.comb.K);
  if (s2.comb.kind == 2) {  // Switch, top-1: H/16 workgroups (48 for Switch-base) of sixteen waves
    // twelve tiles per wave and batch: Switch-base's down projection is 192 tiles per row group = 16 waves x 12, i.e. the
    // workgroup's whole 197 KB in flight at once (MOEINF_DEC1_SWITCH_U=4: three batches of four, 10.6 us per launch)
    static const int su = env_int("MOEINF_DEC1_SWITCH_U", 12);
    if (su == 12) {
      if (s2.dtype == DT_BF16) KL((ffn2_decode1_kernel<uint16_t, 16, 12>), grid, dim3(1024), 0, st, s2);
      else if (s2.dtype == DT_F16) KL((ffn2_decode1_kernel<half_t, 16, 12>), grid, dim3(1024), 0, st, s2);
      else KL((ffn2_decode1_kernel<float, 16, 12>), grid, dim3(1024), 0, st, s2);
      return hipGetLastError();
    }
    if (s2.dtype == DT_BF16) KL((ffn2_decode1_kernel<uint16_t, 16, 4>), grid, dim3(1024), 0, st, s2);
    else if (s2.dtype == DT_F16) KL((ffn2_decode1_kernel<half_t, 16, 4>), grid, dim3(1024), 0, st, s2);
    else KL((ffn2_decode1_kernel<float, 16, 4>), grid, dim3(1024), 0, st, s2);
    return hipGetLastError();
  }
  if (s2.dtype == DT_F16) {  // fp16 gated families: the arrival-counter form
    if ((size_t)s2.K * 2 >= 16384) KL((ffn2_decode1_kernel<half_t, 8, 4>), grid, dim3(512), 0, st, s2);
    else KL((ffn2_decode1_kernel<half_t, 4, 4>), grid, dim3(256), 0, st, s2);
    return hipGetLastError();
  }
  const size_t kbytes = (size_t)s2.K * 2;
  static const int du = env_int("MOEINF_DEC1_U", 4);  // k-tiles per wave fetched per batch (short reductions)
  if (kbytes >= 16384) KL((ffn2_decode1_kernel<uint16_t, 8, 4>), grid, dim3(512), 0, st, s2);
  else if (du == 8) KL((ffn2_decode1_kernel<uint16_t, 4, 8>), grid, dim3(256), 0, st, s2);
  else if (du == 12) KL((ffn2_decode1_kernel<uint16_t, 4, 12>), grid, dim3(256), 0, st, s2);
  else KL((ffn2_decode1_kernel<uint16_t, 4, 4>), grid, dim3(256), 0, st, s2);
  return hipGetLastError();
}

hipError_t launch_route_index(const RouteArgs& r, const IndexArgs& a, hipStream_t st, const EpFuse* pack) {
  hipLaunchKernelGGL(route_index_kernel, dim3(1), dim3(IDX_THREADS), 0, st, r, a, pack ? *pack : no_pack());
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// mask_index: dispatch index from a dense router_mask[T,E] (what the reference's Python blocks hand to
// dispatch_local, expert_executor.py:32-58).  One workgroup; wave w owns experts w, w+16, ...; tokens
// are scanned 64 at a time with a ballot, so each expert's rows come out in ascending token order
// (= the boolean-mask gather order of expert_dispatcher.cpp:274-284).
// ------------------------------------------------------------------------------------------------
template <typename MT>
__global__ __launch_bounds__(IDX_THREADS) void mask_index_kernel(const MT* __restrict__ mask, int T, int E, IndexArgs a, const uint8_t* __restrict__ keep) {
  __shared__ int cnt[IDX_MAXE];
  __shared__ int offs[IDX_MAXE + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = wave; e < E; e += IDX_WAVES) {
    int c = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
      const int t = t0 + lane;
      const bool on = t < T && (!keep || keep[e]) && mask[(size_t)t * E + e] != (MT)0;
      c += __popcll(__ballot(on));
    }
    if (lane == 0) cnt[e] = c;
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0, na = 0;
    for (int e = 0; e < E; ++e) {
      offs[e] = acc;
      acc += cnt[e];
      if (cnt[e] > 0) a.active[na++] = e;
    }
    offs[E] = acc;
    offs[E + 1] = acc;  // no shared pseudo-expert on this path
    cnt[E] = 0;
    *a.n_active = na;
    if (a.mirror) {
      a.mirror[0] = na;
      for (int i = 0; i < na; ++i) a.mirror[1 + (E + 1) + i] = a.active[i];
      for (int i = na; i <= E; ++i) a.mirror[1 + (E + 1) + i] = -1;
    }
  }
  __syncthreads();
  if (tid <= E) {
    a.counts[tid] = cnt[tid];
    if (a.mirror) a.mirror[1 + tid] = cnt[tid];
  }
  if (tid <= E + 1) a.offsets[tid] = offs[tid];
  for (int e = wave; e < E; e += IDX_WAVES) {
    int base = offs[e];
    for (int t0 = 0; t0 < T; t0 += 64) {
      const int t = t0 + lane;
      const bool on = t < T && (!keep || keep[e]) && mask[(size_t)t * E + e] != (MT)0;
      const uint64_t b = __ballot(on);
      if (on) {
        const int slot = base + __popcll(b & lanes_below(lane));
        // a dense mask may route a token to more than K experts: rows past the workspace capacity are counted (the
        // host rejects the call from the counts) but never written
        if (a.slot_cap <= 0 || slot < a.slot_cap) {
          a.slot_token[slot] = t;
          a.slot_pair[slot] = t * E + e;
        }
      }
      base += __popcll(b);
    }
  }
}
hipError_t launch_mask_index(const void* mask, int mask_elem_bytes, int T, int E, const IndexArgs& a, hipStream_t st, const uint8_t* keep) {
  if (mask_elem_bytes == 1) hipLaunchKernelGGL(mask_index_kernel<uint8_t>, dim3(1), dim3(IDX_THREADS), 0, st, (const uint8_t*)mask, T, E, a, keep);
  else if (mask_elem_bytes == 4) hipLaunchKernelGGL(mask_index_kernel<int32_t>, dim3(1), dim3(IDX_THREADS), 0, st, (const int32_t*)mask, T, E, a, keep);
  else hipLaunchKernelGGL(mask_index_kernel<int64_t>, dim3(1), dim3(IDX_THREADS), 0, st, (const int64_t*)mask, T, E, a, keep);
  return hipGetLastError();
}

// topk_idx with capacity-dropped / unrouted pairs reported as -1 (what moeinf_get_routing returns on the host)
__global__ void masked_idx_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ valid, int32_t* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = valid[i] ? idx[i] : -1;
}
hipError_t launch_masked_idx(const int32_t* idx, const int32_t* valid, int32_t* out, int n, hipStream_t st) {
  hipLaunchKernelGGL(masked_idx_kernel, dim3((n + 255) / 256), dim3(256), 0, st, idx, valid, out, n);
  return hipGetLastError();
}

// standalone combine launch (see combine_cols)
template <typename T>
__global__ __launch_bounds__(256) void combine_kernel(CombineArgs a) {
  const int h0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (h0 < a.H) combine_cols<T>(a, blockIdx.y, h0);
}

__global__ void shared_only_index_kernel(IndexArgs a) {
  const int E = a.E, T = a.T;
  for (int e = threadIdx.x; e <= E; e += blockDim.x) { a.counts[e] = (e == E) ? T : 0; a.offsets[e] = 0; }
  if (threadIdx.x == 0) { a.offsets[E + 1] = T; a.active[0] = E; *a.n_active = 1; }
  for (int t = threadIdx.x; t < T; t += blockDim.x) { a.slot_token[t] = t; a.slot_pair[t] = -1; }
}
hipError_t launch_shared_only_index(const IndexArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(shared_only_index_kernel, dim3(1), dim3(256), 0, st, a);
  return hipGetLastError();
}

// peer-store exchange: a.y is this rank's return region, filled by the owners' kernels; every workgroup polls the
// owners' flags itself before it reads a row
template <typename T>
__global__ __launch_bounds__(256) void combine_wait_kernel(CombineArgs a, EpWait w) {
  if (threadIdx.x < 64) ep_poll(w.flags, w.n, w.epoch, w.timeout_ticks, w.err);
  __syncthreads();
  const int h0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (h0 < a.H) combine_cols<T>(a, blockIdx.y, h0);
}
hipError_t launch_combine(const CombineArgs& a, hipStream_t st, const EpWait* wait) {
  dim3 grid((a.H + 1023) / 1024, a.T);
  if (wait) {
    if (a.dtype == DT_BF16) hipLaunchKernelGGL(combine_wait_kernel<uint16_t>, grid, dim3(256), 0, st, a, *wait);
    else if (a.dtype == DT_F16) hipLaunchKernelGGL(combine_wait_kernel<half_t>, grid, dim3(256), 0, st, a, *wait);
    else hipLaunchKernelGGL(combine_wait_kernel<float>, grid, dim3(256), 0, st, a, *wait);
    return hipGetLastError();
  }
  if (a.dtype == DT_BF16) hipLaunchKernelGGL(combine_kernel<uint16_t>, grid, dim3(256), 0, st, a);
  else if (a.dtype == DT_F16) hipLaunchKernelGGL(combine_kernel<half_t>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(combine_kernel<float>, grid, dim3(256), 0, st, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
__global__ void poke_kernel(PokeArgs a) {
  const int i = threadIdx.x;
  if (i < a.n) a.table[a.idx[i]] = a.val[i];
}
hipError_t launch_poke(const PokeArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(poke_kernel, dim3(1), dim3(64), 0, st, a);
  return hipGetLastError();
}

// [T,K] routing of a caller that kept its own router (moeinf_combine) -> the engine's pair arrays: idx < 0 marks a
// dropped pair; pair_order = the token's k-indices by ascending expert id (stable), as route_store writes it
__global__ void prep_pairs_kernel(const int32_t* __restrict__ idx_in, const float* __restrict__ w_in, int T, int K, int32_t* __restrict__ topk_idx,
                                  float* __restrict__ topk_w, int32_t* __restrict__ pair_valid, int32_t* __restrict__ pair_order) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  int sel[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) sel[k] = k < K ? idx_in[(size_t)t * K + k] : 0x7fffffff;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k < K) {
      int rank = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < K) rank += (sel[j] < sel[k] || (sel[j] == sel[k] && j < k)) ? 1 : 0;
      topk_idx[(size_t)t * K + k] = sel[k];
      topk_w[(size_t)t * K + k] = w_in[(size_t)t * K + k];
      pair_valid[(size_t)t * K + k] = sel[k] >= 0 ? 1 : 0;
      pair_order[(size_t)t * K + rank] = k;
    }
  }
}
hipError_t launch_prep_pairs(const int32_t* idx_in, const float* w_in, int T, int K, int32_t* topk_idx, float* topk_w,
                             int32_t* pair_valid, int32_t* pair_order, hipStream_t st) {
  hipLaunchKernelGGL(prep_pairs_kernel, dim3((T + 127) / 128), dim3(128), 0, st, idx_in, w_in, T, K, topk_idx, topk_w, pair_valid, pair_order);
  return hipGetLastError();
}

}  // namespace moeinf
