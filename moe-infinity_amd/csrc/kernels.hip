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
#include "kernels.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

namespace moeinf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// ------------------------------------------------------------------------------------------------
// scalar helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(uint16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {  // round-to-nearest-even, NaN preserved
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
// Device-coherent accessors for data handed between workgroups INSIDE one launch (fused combine / fused router):
// relaxed agent-scope atomics compile to sc1 loads/stores that write through / miss the per-XCD L2 for lines it
// does not own, so no agent-scope fence (= a full L2 write-back + invalidate, tens of us on 8 XCDs) is needed;
// the producer only waits for its stores to be acknowledged (s_waitcnt) before it bumps the arrival counter.
template <typename V>
__device__ __forceinline__ V ld_coherent(const V* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename V>
__device__ __forceinline__ void st_coherent(V* p, V v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wait_stores_acked() { __builtin_amdgcn_s_waitcnt(0); }

template <typename T>
struct DT;
template <>
struct DT<uint16_t> {  // bf16 storage
  static constexpr int EPV = 8;  // elements per 16-byte vector
  __device__ static __forceinline__ float round(float f) { return bf2f(f2bf(f)); }
  __device__ static __forceinline__ float load(const uint16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void store(uint16_t* p, float f) { *p = f2bf(f); }
  // 4 consecutive elements, 8-byte aligned
  __device__ static __forceinline__ void load4(const uint16_t* p, float o[4]) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  }
  // raw 4-element fetch (issue many, unpack later: keeps independent loads back to back)
  struct Raw4 { unsigned long long v; };
  template <bool COH>
  __device__ static __forceinline__ Raw4 fetch4(const uint16_t* p) {
    Raw4 r;
    r.v = COH ? ld_coherent(reinterpret_cast<const unsigned long long*>(p)) : *reinterpret_cast<const unsigned long long*>(p);
    return r;
  }
  __device__ static __forceinline__ void unpack4(const Raw4& r, float o[4]) {
    const uint32_t x = (uint32_t)r.v, y = (uint32_t)(r.v >> 32);
    o[0] = __uint_as_float(x << 16); o[1] = __uint_as_float(x & 0xffff0000u);
    o[2] = __uint_as_float(y << 16); o[3] = __uint_as_float(y & 0xffff0000u);
  }
  __device__ static __forceinline__ void store_coherent(uint16_t* p, float f) { st_coherent(p, f2bf(f)); }
  __device__ static __forceinline__ void store4(uint16_t* p, const float f[4]) {
    uint2 v;
    v.x = (uint32_t)f2bf(f[0]) | ((uint32_t)f2bf(f[1]) << 16);
    v.y = (uint32_t)f2bf(f[2]) | ((uint32_t)f2bf(f[3]) << 16);
    *reinterpret_cast<uint2*>(p) = v;
  }
};
template <>
struct DT<float> {
  static constexpr int EPV = 4;
  __device__ static __forceinline__ float round(float f) { return f; }
  __device__ static __forceinline__ float load(const float* p) { return *p; }
  __device__ static __forceinline__ void store(float* p, float f) { *p = f; }
  __device__ static __forceinline__ void load4(const float* p, float o[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  struct Raw4 { unsigned long long a, b; };
  template <bool COH>
  __device__ static __forceinline__ Raw4 fetch4(const float* p) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    Raw4 r;
    r.a = COH ? ld_coherent(q) : q[0];
    r.b = COH ? ld_coherent(q + 1) : q[1];
    return r;
  }
  __device__ static __forceinline__ void unpack4(const Raw4& r, float o[4]) {
    o[0] = __uint_as_float((uint32_t)r.a); o[1] = __uint_as_float((uint32_t)(r.a >> 32));
    o[2] = __uint_as_float((uint32_t)r.b); o[3] = __uint_as_float((uint32_t)(r.b >> 32));
  }
  __device__ static __forceinline__ void store_coherent(float* p, float f) { st_coherent(p, f); }
  __device__ static __forceinline__ void store4(float* p, const float f[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
  }
};

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ u32x4 ld16_nt(const void* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
}

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
hipError_t launch_retile(const void* src, void* dst, int R, int K, int dtype, hipStream_t st) {
  const int ept = dtype == DT_BF16 ? 32 : 16;
  dim3 grid(((K + ept - 1) / ept + 3) / 4, (R + 15) / 16);
  if (dtype == DT_BF16) hipLaunchKernelGGL(retile_kernel<uint16_t>, grid, dim3(256), 0, st, (const uint16_t*)src, (char*)dst, R, K);
  else hipLaunchKernelGGL(retile_kernel<float>, grid, dim3(256), 0, st, (const float*)src, (char*)dst, R, K);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// ffn_rows: grouped expert FFN, one stage.  grid = (ceil(Rmax/16), n_active), block = NW waves.
//
// Block (rg, u) owns 16 consecutive output rows [16*rg, 16*rg+16) of expert active[u] (for the
// gated stage: the same 16 rows of BOTH the gate and the up matrix).  The reduction dimension is
// split over the block's NW waves, which take interleaved k-tiles: every weight byte is read exactly
// once from HBM, contiguous 1 KiB per wave-instruction, non-temporal, straight into VGPRs (no LDS
// round trip: the stream is not shared between waves).
// MFMA operands: A = one weight tile (lane: row r = lane&15, quad q = lane>>4), B = activations of
// up to 16 tokens (lane: token n = lane&15, quad q, same k elements as A's quad).
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void mma16(f32x4& acc, const u32x4& a, const u32x4& b);
template <>
__device__ __forceinline__ void mma16<uint16_t>(f32x4& acc, const u32x4& a, const u32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma16<float>(f32x4& acc, const u32x4& a, const u32x4& b) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[j]), __uint_as_float(b[j]), acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// combine: out[t] = sum over the token's experts in ASCENDING expert id of w * y, with the
// reference block's dtype rounding points (mixtral.py:96-101, deepseek.py:123-136,
// switch_transformers.py:99-109, nllb_moe.py:84-104).  Used by combine_kernel (grid =
// (ceil(H/(256*4)), T)) and by the fused epilogue of the decode-sized FFN stage 2.
// ------------------------------------------------------------------------------------------------
// columns [h0, h0+4) of token t; H % 4 == 0 (moeinf_create checks), rows 8/16-byte aligned
struct CombineMeta {  // a token's combine order resolved to row slots and weights
  int slot[8];
  float w[8];
};
__device__ __forceinline__ void combine_meta(const CombineArgs& a, const int t, CombineMeta& m) {
  // two dependent rounds (order -> slot/weight), each issued back to back: entries kk >= K repeat entry K-1 (ignored
  // by the caller) so the rounds stay branch-free
  const int K = a.K;
  const size_t p0 = (size_t)t * K;
  int ko[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) ko[kk] = a.pair_order[p0 + min(kk, K - 1)];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    m.slot[kk] = a.pair_slot[p0 + ko[kk]];
    m.w[kk] = a.topk_w[p0 + ko[kk]];
  }
}
template <typename T, bool COH = false>  // COH: y / y_shared were written by other workgroups of THIS launch
__device__ __forceinline__ void combine_apply(const CombineArgs& a, const int t, const int h0, const CombineMeta& m) {
  const int K = a.K;
  const T* y = reinterpret_cast<const T*>(a.y);
  T* out = reinterpret_cast<T*>(a.out) + (size_t)t * a.H + h0;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const bool has_sh = (a.kind == 1 && a.y_shared);
  const int row0 = (has_sh && a.shared_offsets) ? a.shared_offsets[a.shared_E] : 0;
  typename DT<T>::Raw4 rsh, ry[8];
  // absent slots fetch row 0 (ignored below) so the round stays branch-free
  rsh = DT<T>::template fetch4<COH>(reinterpret_cast<const T*>(has_sh ? a.y_shared : a.y) + (size_t)(row0 + (has_sh ? t : 0)) * a.H + h0);
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) ry[kk] = DT<T>::template fetch4<COH>(y + (size_t)max(m.slot[kk], 0) * a.H + h0);
  float sh[4] = {0.f, 0.f, 0.f, 0.f};
  if (has_sh) DT<T>::unpack4(rsh, sh);
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    if (kk < K && m.slot[kk] >= 0) {
      float yv[4];
      DT<T>::unpack4(ry[kk], yv);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float prod = yv[j] * m.w[kk];
        // Mixtral/NLLB multiply in the model dtype (weights were cast to it); DeepSeek keeps the
        // product in fp32 (fp32 gate weights promote the bf16 expert output)
        if (a.kind != 1) prod = DT<T>::round(prod);
        acc[j] = DT<T>::round(acc[j] + prod);
      }
    }
  }
  if (has_sh) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = DT<T>::round(acc[j] + sh[j]);
  }
  if (a.kind == 3 /*NLLB: next_states[next_states == 0] = hidden_states[...] */) {
    float xv[4];
    DT<T>::load4(reinterpret_cast<const T*>(a.x) + (size_t)t * a.H + h0, xv);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (acc[j] == 0.f) acc[j] = xv[j];
  }
  DT<T>::store4(out, acc);
}
template <typename T, bool COH = false>
__device__ __forceinline__ void combine_cols(const CombineArgs& a, const int t, const int h0) {
  if (a.kind == 2 /*SWITCH*/) {
    const T* y = reinterpret_cast<const T*>(a.y);
    T* out = reinterpret_cast<T*>(a.out) + (size_t)t * a.H + h0;
    const int slot = a.pair_slot[t];
    const T* src = (slot >= 0) ? y + (size_t)slot * a.H : reinterpret_cast<const T*>(a.x) + (size_t)t * a.H;
    const float pr = a.router_prob[t];
    float v[4], acc[4];
    DT<T>::load4(src + h0, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = DT<T>::round(pr * v[j]);
    DT<T>::store4(out, acc);
    return;
  }
  CombineMeta m;
  combine_meta(a, t, m);
  combine_apply<T, COH>(a, t, h0, m);
}

// One work item of ffn_rows: 16 output rows [16*bx, 16*bx+16) of one expert (blob W, rows off..off+cnt of the
// expert-sorted activations).  Shared by ffn_rows_kernel and by the router kernels that carry the always-resident
// shared expert's FFN along (gate_shared1_kernel / route_shared2_kernel).
template <typename T, int NMAT, int NW, int U, int NT>
__device__ __forceinline__ void ffn_rows_item(const FfnStage& s, const int bx, const char* W, const bool sh, const int cnt, const int off,
                                              float (*red)[NMAT][256], const int xrow_fixed = -1) {
  constexpr int EPV = DT<T>::EPV;
  constexpr int EPT = 4 * EPV;  // k elements per tile (64 bytes per row)
  const int K = sh ? s.K_sh : s.K;
  const int R = sh ? s.R_sh : s.R;
  const int r0 = bx * 16;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int KB = (K + EPT - 1) / EPT;  // tiles per row group (last one zero-padded)
  const int KBfull = K / EPT;
  const char* a0 = W + (sh ? s.off_a_sh : s.off_a) + (size_t)bx * KB * 1024 + lane * 16;
  const char* a1 = NMAT == 2 ? W + (sh ? s.off_b_sh : s.off_b) + (size_t)bx * KB * 1024 + lane * 16 : nullptr;
  const int kq = q * EPV;  // this lane's k offset inside a tile

  // NT token tiles (16 tokens each) share one pass over the weights: experts with many tokens
  // (prefill, big batches) re-stream their weights every 16*NT tokens instead of every 16
  for (int tile0 = 0; tile0 * 16 < cnt; tile0 += NT) {
    const int ntl = min(NT, (cnt - tile0 * 16 + 15) / 16);  // live token tiles in this pass (block-uniform)
    const T* xr[NT];
    f32x4 acc0[NT], acc1[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      const int srow = off + min((tile0 + tt) * 16 + n, cnt - 1);
      // xrow_fixed >= 0: every row of this item is token xrow_fixed (the self-routing decode kernel: the row map is
      // being written by another block of the same launch)
      const int64_t xrow = xrow_fixed >= 0 ? (int64_t)xrow_fixed : (s.row_map ? (int64_t)s.row_map[srow] : (int64_t)srow);
      xr[tt] = reinterpret_cast<const T*>(s.in) + xrow * s.ld_in + kq;
      acc0[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc1[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // k-tiles wave, wave+NW, ... in batches of U: every load of a batch is issued before its first MFMA; the last,
    // shorter batch is predicated (wave-uniform), not peeled into one-tile round trips (DeepSeek stage 2, 11 tiles per
    // wave: 3 round trips instead of 5, 13.7 -> 11.8 us per launch; issuing the first weight batch ahead of the
    // row_map round was measured too and bought nothing)
    for (int kb = wave; kb < KBfull; kb += U * NW) {
      u32x4 av[U], bv[U], xv[U][NT];
#pragma unroll
      for (int i = 0; i < U; ++i) {
        if (kb + i * NW < KBfull) {
          av[i] = ld16_nt(a0 + (size_t)(kb + i * NW) * 1024);
          if (NMAT == 2) bv[i] = ld16_nt(a1 + (size_t)(kb + i * NW) * 1024);
#pragma unroll
          for (int tt = 0; tt < NT; ++tt)
            if (tt < ntl) xv[i][tt] = ld16(xr[tt] + (size_t)(kb + i * NW) * EPT);
        }
      }
#pragma unroll
      for (int i = 0; i < U; ++i) {
        if (kb + i * NW < KBfull) {
#pragma unroll
          for (int tt = 0; tt < NT; ++tt) {
            if (tt < ntl) {
              mma16<T>(acc0[tt], av[i], xv[i][tt]);
              if (NMAT == 2) mma16<T>(acc1[tt], bv[i], xv[i][tt]);
            }
          }
        }
      }
    }
    if (KB != KBfull && wave == (KBfull % NW)) {  // zero-padded last tile: guard only the activation read
      const u32x4 z = {0u, 0u, 0u, 0u};
      const u32x4 w0 = ld16_nt(a0 + (size_t)KBfull * 1024);
      u32x4 w1 = w0;
      if (NMAT == 2) w1 = ld16_nt(a1 + (size_t)KBfull * 1024);
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        if (tt < ntl) {
          const u32x4 x0 = (KBfull * EPT + kq < K) ? ld16(xr[tt] + (size_t)KBfull * EPT) : z;
          mma16<T>(acc0[tt], w0, x0);
          if (NMAT == 2) mma16<T>(acc1[tt], w1, x0);
        }
      }
    }
    // cross-wave reduction of the K split + epilogue, one token tile at a time
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      if (tt >= ntl) break;
      const int tile = tile0 + tt;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        red[wave][0][lane * 4 + j] = acc0[tt][j];
        if (NMAT == 2) red[wave][1][lane * 4 + j] = acc1[tt][j];
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
        const int tn = l & 15;                    // token column
        const int orow = r0 + (l >> 4) * 4 + j;  // output row
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
          T* op = reinterpret_cast<T*>(s.out) + (size_t)(s.out_map ? s.out_map[srow] : srow) * s.ld_out + orow;
          if (NMAT == 1 && NT == 1 && s.fuse_combine) DT<T>::store_coherent(op, v); else DT<T>::store(op, v);
        }
      }
      __syncthreads();
    }
  }
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
#pragma unroll
    for (int a = 0; a < RGW; ++a) {
      const int r0 = (rg0 + wr * RGW + a) * 16 + q * 4;
#pragma unroll
      for (int b = 0; b < NTW; ++b) {
        const int tok = (tile0 + wc * NTW + b) * 16 + n;
        if (tok < cnt && rg0 + wr * RGW + a < nrg_total) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int orow = r0 + j;
            if (orow < R) {
              float v = DT<T>::round(acc[a][b][0][j]);
              if (s.epi == EPI_GATED_SILU) {
                const float bb = DT<T>::round(acc[a][b][NMAT - 1][j]);
                const float sl = DT<T>::round(v / (1.0f + expf(-v)));
                v = DT<T>::round(sl * bb);
              } else {
                if (s.epi == EPI_BIAS || s.epi == EPI_BIAS_RELU)
                  v = DT<T>::round(v + DT<T>::load(reinterpret_cast<const T*>(W + s.off_bias) + orow));
                if (s.epi == EPI_RELU || s.epi == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
              }
              DT<T>::store(reinterpret_cast<T*>(s.out) + (size_t)(s.out_map ? s.out_map[off + tok] : off + tok) * s.ld_out + orow, v);
            }
          }
        }
      }
    }
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
#pragma unroll
    for (int a = 0; a < RW; ++a) {
      const int r0 = (rgw0 + a) * 16 + q * 4;
#pragma unroll
      for (int b = 0; b < NTB; ++b) {
        const int tok = (tile0 + b) * 16 + n;
        if (tok < cnt && rgw0 + a < nrg_total) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int orow = r0 + j;
            if (orow < R) {
              float v = DT<T>::round(acc[a][b][0][j]);
              if (s.epi == EPI_GATED_SILU) {
                const float bb = DT<T>::round(acc[a][b][NMAT - 1][j]);
                const float sl = DT<T>::round(v / (1.0f + expf(-v)));
                v = DT<T>::round(sl * bb);
              } else {
                if (s.epi == EPI_BIAS || s.epi == EPI_BIAS_RELU)
                  v = DT<T>::round(v + DT<T>::load(reinterpret_cast<const T*>(W + s.off_bias) + orow));
                if (s.epi == EPI_RELU || s.epi == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
              }
              DT<T>::store(reinterpret_cast<T*>(s.out) + (size_t)(s.out_map ? s.out_map[off + tok] : off + tok) * s.ld_out + orow, v);
            }
          }
        }
      }
    }
    __syncthreads();  // the next pass re-uses LDS buffer 0
  }
}

// ------------------------------------------------------------------------------------------------
// ffn_gemm_ring: the GATED stage (x -> silu(x W1^T) * (x W3^T)) for experts with ~64..1000 tokens (prefill), bf16,
// long reductions (K >= 4096).
// What bounds ffn_gemm_lds / ffn_gemm_hyb at ~128 tokens per expert is WEIGHT BYTES IN FLIGHT: one stage ahead,
// drained by `vmcnt(0)` + `__syncthreads()` every k-step, leaves 16-32 KiB of weights outstanding per CU against a
// ~2 us HBM round trip = ~4 TB/s chip-wide (measured 3.4-3.9).  Here
//   * block = 8 waves, ONE block per CU; wave w owns the same 16 rows of BOTH matrices against NTB token groups
//     (128 or 256 tokens): 16 / 32 accumulator tiles, SiLU*mul in registers;
//   * weights go HBM -> VGPRs directly (the tiled layout IS the MFMA A fragment) through a ring of D register stages
//     (a stage = 2 k-tiles = 4 one-KiB tiles per wave); D = 4: 12 KiB per wave = 96 KiB per CU in flight (D = 3 for
//     the 256-token variant, which needs the registers for accumulators).  The loads are inline asm so that the
//     compiler's vmcnt bookkeeping cannot drain the ring;
//   * activations go L2 -> LDS by global_load_lds in full 128-byte lines (source-side XOR swizzle, as ffn_gemm_lds
//     XL) through a ring of 3 LDS stages;
//   * ONE raw s_barrier per stage and a COUNTED s_waitcnt: every wave issues the same VM ops in the same order
//     (... W(k) X(k) W(k+1) X(k+1) ..., 4 weight loads and XPW activation DMAs per stage; pieces of absent token
//     groups are still issued, clamped, so the count never varies).
// Mixtral stage 1 at 512 tokens (128 rows per expert): 565 -> 420-450 us per layer (3.4 -> 4.4 TB/s); at 2048 tokens
// 1490 -> 1250-1420 us.  The same structure for the PLAIN stage (two row groups per wave, or one row group with the
// k-tiles of a 4-tile stage split over two partial accumulators) was built and measured too: Mixtral's down
// projection 250 -> 300-380 us at 512 tokens, 805 -> 840-1180 us at 2048 — slower than ffn_gemm_lds there (a matrix
// with H = 4096 rows gives 128-256 eight-wave blocks for 256 CUs), so the plain stage stays on ffn_gemm_lds.
// Requires K % 64 == 0.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ring_load(u32x4& dst, const char* p) {
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void ring_wait(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
  // the counted wait carries the stage's registers as in/out operands: no MFMA that reads them can be scheduled above it
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}

template <int NTB, int D>
__global__ __launch_bounds__(512) void ffn_gemm_ring_kernel(FfnStage s) {
  static_assert(D == 3 || D == 4, "register ring of 3 or 4 stages");
  typedef uint16_t T;
  constexpr int EPT = 32, EPV = 8;
  constexpr int NWV = 8;
  constexpr int XPW = 2 * NTB / NWV;        // activation DMA pieces (8 rows x 128 B) per wave and stage
  constexpr int XSTAGE = 2 * NTB * 1024;    // activation bytes per stage (2 k-tiles)
  constexpr int NX = 3;                     // LDS ring
  __shared__ __attribute__((aligned(16))) char smem[NX * XSTAGE];

  const int u = blockIdx.y;
  if (u >= (s.n_active_host >= 0 ? s.n_active_host : *s.n_active)) return;
  const int e = s.active[u];
  const bool sh = (e == s.E);
  const int K = sh ? s.K_sh : s.K;
  const int R = sh ? s.R_sh : s.R;
  const int nrg_total = (R + 15) / 16;
  if ((int)blockIdx.x * NWV >= nrg_total) return;
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
  const int KB = K / EPT;
  const int KS = KB / 2;
  const size_t rg_stride = (size_t)KB * 1024;
  // this wave's two weight-tile streams (a row group past the end re-reads the last one; its results are dropped)
  const int rg = min((int)blockIdx.x * NWV + wave, nrg_total - 1);
  const bool rg_live = (int)blockIdx.x * NWV + wave < nrg_total;
  const char* ap[2];
  ap[0] = W + (sh ? s.off_a_sh : s.off_a) + (size_t)rg * rg_stride + lane * 16;
  ap[1] = W + (sh ? s.off_b_sh : s.off_b) + (size_t)rg * rg_stride + lane * 16;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  for (int tile0 = 0; tile0 * 16 < cnt; tile0 += NTB) {
    const int ntl = min(NTB, (cnt - tile0 * 16 + 15) / 16);  // token groups present in this pass (block-uniform)
    const T* xrp[XPW];
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
      const int trow = tile0 * 16 + (wave + NWV * i) * 8 + (lane >> 3);
      const int srow = off + min(trow, cnt - 1);
      const int64_t xrow = s.row_map ? (int64_t)s.row_map[srow] : (int64_t)srow;
      xrp[i] = reinterpret_cast<const T*>(s.in) + xrow * s.ld_in + (((lane & 7) ^ (lane >> 3)) * EPV);
    }
    f32x4 acc[NTB][2];
#pragma unroll
    for (int b = 0; b < NTB; ++b) { acc[b][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[b][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    u32x4 wr[D][4];  // [ring stage][k-tile * 2 + matrix]
    auto issue_w = [&](int ks, u32x4 (&dst)[4]) {
      const int kb = min(ks, KS - 1) * 2;  // past the end: re-read the last stage (never multiplied), the count stays fixed
      ring_load(dst[0], ap[0] + (size_t)kb * 1024);
      ring_load(dst[1], ap[1] + (size_t)kb * 1024);
      ring_load(dst[2], ap[0] + (size_t)(kb + 1) * 1024);
      ring_load(dst[3], ap[1] + (size_t)(kb + 1) * 1024);
    };
    auto issue_x = [&](int ks) {
      char* base = smem + (ks % NX) * XSTAGE;
      const int kc = min(ks, KS - 1);
#pragma unroll
      for (int i = 0; i < XPW; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(xrp[i] + (size_t)kc * 2 * EPT), (lptr_t)(base + (wave + NWV * i) * 1024), 16, 0, 0);
    };
    // token groups are multiplied in chunks of 4 (absent groups of a partly filled chunk hold clamped copies of the last
    // row and are dropped by the epilogue): one wave-uniform branch per chunk instead of one per group, so the LDS
    // fragment reads of a chunk are issued together and its 8 MFMAs run back to back (a branch per group serialised
    // ds_read -> wait -> 2 MFMAs)
    auto compute = [&](int ks, const u32x4 (&w)[4]) {
      const char* base = smem + (ks % NX) * XSTAGE;
      const int r = n & 7;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int ch = kk * 4 + q;
#pragma unroll
        for (int c = 0; c < NTB / 4; ++c) {
          if (c * 4 < ntl) {
            u32x4 bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) bf[i] = *reinterpret_cast<const u32x4*>(base + ((c * 4 + i) * 2 + (n >> 3)) * 1024 + r * 128 + ((ch ^ r) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int b = c * 4 + i;
              acc[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[kk * 2 + 0]), __builtin_bit_cast(bf16x8, bf[i]), acc[b][0], 0, 0, 0);
              acc[b][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[kk * 2 + 1]), __builtin_bit_cast(bf16x8, bf[i]), acc[b][1], 0, 0, 0);
            }
          }
        }
      }
    };
    // Issue order of every wave: ... W(k) X(k) W(k+1) X(k+1) ...
    //   D == 4: prologue W0 X0 W1 X1 W2, step S issues X(S+2) W(S+3); before stage S is consumed W(S+1) X(S+1) W(S+2) may
    //           be outstanding: vmcnt(8 + XPW);
    //   D == 3: prologue W0 X0 W1 X1,    step S issues W(S+2) X(S+2); outstanding W(S+1) X(S+1): vmcnt(4 + XPW).
    // Unrolled by D so that the register ring is indexed statically.  Past the end the issues are clamped re-reads
    // (count-preserving); the final wait below drains them.
    issue_w(0, wr[0]); issue_x(0);
    issue_w(1, wr[1]); issue_x(1);
    if (D == 4) issue_w(2, wr[2]);
#define RING_STEP(S, CUR, NXT)                                                                   \
    if ((S) < KS) {                                                                              \
      ring_wait<(D - 2) * 4 + XPW>(wr[CUR][0], wr[CUR][1], wr[CUR][2], wr[CUR][3]); /* this wave's W(S), X(S) landed */ \
      __builtin_amdgcn_s_barrier();              /* everybody's X(S) landed; LDS buffer (S+2)%3 is free */        \
      if (D == 4) { issue_x((S) + 2); issue_w((S) + 3, wr[NXT]); }                                \
      else { issue_w((S) + 2, wr[NXT]); issue_x((S) + 2); }                                      \
      compute((S), wr[CUR]);                                                                     \
    }
    if (D == 4) {
      for (int ks = 0; ks < KS; ks += 4) {
        RING_STEP(ks, 0, 3)
        RING_STEP(ks + 1, 1, 0)
        RING_STEP(ks + 2, 2, 1)
        RING_STEP(ks + 3, 3, 2)
      }
    } else {
      for (int ks = 0; ks < KS; ks += 3) {
        RING_STEP(ks, 0, 2)
        RING_STEP(ks + 1, 1, 0)
        RING_STEP(ks + 2, 2, 1)
      }
    }
#undef RING_STEP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // clamped tail issues
    // epilogue straight from the accumulators: lane holds 4 consecutive rows of one token
#pragma unroll
    for (int b = 0; b < NTB; ++b) {
      const int tok = (tile0 + b) * 16 + n;
      if (tok < cnt && rg_live) {
        const int srow = s.out_map ? s.out_map[off + tok] : off + tok;
        const int r0 = rg * 16 + q * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (r0 + j < R) {
            float v = DT<T>::round(acc[b][0][j]);
            const float bb = DT<T>::round(acc[b][1][j]);
            const float sl = DT<T>::round(v / (1.0f + expf(-v)));
            v = DT<T>::round(sl * bb);
            DT<T>::store(reinterpret_cast<T*>(s.out) + (size_t)srow * s.ld_out + r0 + j, v);
          }
        }
      }
    }
    __syncthreads();  // the next pass re-uses the LDS ring from stage 0
  }
}

// tuning knobs (overridable for sweeps: MOEINF_FFN_NW=4|8, MOEINF_FFN_U=2|4|8)
static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

template <typename T, int NMAT>
static void launch_ffn_t(const FfnStage& s, dim3 grid, int nw, int u, bool many_tokens, int max_rows, hipStream_t st) {
#define LAUNCH(NWV, UU, NTT) hipLaunchKernelGGL((ffn_rows_kernel<T, NMAT, NWV, UU, NTT>), grid, dim3(NWV * 64), 0, st, s)
  if (many_tokens) {  // register-tiled grouped GEMM; 16 accumulator tiles per wave in every shape
    static const int use_gemm = env_int("MOEINF_FFN_GEMM", 2);
    static const int force_nt = env_int("MOEINF_FFN_GEMM_NT", 0);
    const int ept = sizeof(T) == 2 ? 32 : 16;
    const bool k_ok = (s.K % ept) == 0 && (s.K_sh % ept) == 0;
    // 17-64 rows per expert (e.g. NLLB's 128 experts at a 2048-token batch): too many for the decode kernel, too few to
    // amortise staging the weights in LDS -> the hybrid kernel (measured -15 % on that shape, sweep in profiles/)
    static const int hyb_rows = env_int("MOEINF_GEMM_HYB_ROWS", 64);
    static const int ring_env = env_int("MOEINF_GEMM_RING", 1);
    if constexpr (sizeof(T) == 2 && NMAT == 2) {
      // bf16 gated stage, more than hyb_rows rows per expert, long reduction: the register-ring kernel.  With a short K
      // (DeepSeek: 32 stages) filling and draining the ring costs more than it hides (297 vs 287 us at 512 tokens).
      static const int ring_min_k = env_int("MOEINF_RING_MIN_K", 4096);
      const bool ring_ok = (s.K % 64) == 0 && (s.K_sh % 64) == 0 && s.K >= ring_min_k && (s.K_sh == 0 || s.K_sh >= ring_min_k);
      if ((use_gemm == 4 || (use_gemm == 2 && ring_env && max_rows > hyb_rows)) && ring_ok) {
        static const int ring_wide = env_int("MOEINF_RING_WIDE", -1);
        const bool wide = ring_wide >= 0 ? ring_wide != 0 : max_rows > 128;
        const dim3 g2((grid.x + 7) / 8, grid.y);
        if (wide) hipLaunchKernelGGL((ffn_gemm_ring_kernel<16, 3>), g2, dim3(512), 0, st, s);
        else hipLaunchKernelGGL((ffn_gemm_ring_kernel<8, 4>), g2, dim3(512), 0, st, s);
        return;
      }
    }
    if ((use_gemm == 3 || (use_gemm == 2 && max_rows <= hyb_rows)) && k_ok) {  // weights -> registers, activations -> LDS
      static const int kk = env_int("MOEINF_GEMM_HYB_KK", 4);
      static const int hxl_env = env_int("MOEINF_GEMM_XL", 1);
      const bool hxl = hxl_env && (s.K % (2 * ept)) == 0 && (s.K_sh % (2 * ept)) == 0;
#define HYB(NM, RWV, KKV, XLV) hipLaunchKernelGGL((ffn_gemm_hyb_kernel<T, NM, RWV, KKV, XLV>), dim3((grid.x + 4 * RWV - 1) / (4 * RWV), grid.y), dim3(256), 0, st, s)
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
        hipLaunchKernelGGL(kern, dim3((grid.x + rgb - 1) / rgb, grid.y), dim3(nwv * 64), 0, st, s);
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
        if (nt <= 4) hipLaunchKernelGGL((ffn_gemm_kernel<T, 2, 2, 4, 4>), dim3((grid.x + 1) / 2, grid.y), dim3(256), 0, st, s);
        else hipLaunchKernelGGL((ffn_gemm_kernel<T, 2, 1, 8, 4>), grid, dim3(256), 0, st, s);
      } else {                    // plain: (4,4) or (2,8)
        if (nt <= 4) hipLaunchKernelGGL((ffn_gemm_kernel<T, 1, 4, 4, 4>), dim3((grid.x + 3) / 4, grid.y), dim3(256), 0, st, s);
        else hipLaunchKernelGGL((ffn_gemm_kernel<T, 1, 2, 8, 4>), dim3((grid.x + 1) / 2, grid.y), dim3(256), 0, st, s);
      }
    } else {
      if (nw == 8) LAUNCH(8, 1, 4); else LAUNCH(4, 1, 4);
    }
    return;
  }
  if (nw == 8) { if (u == 2) LAUNCH(8, 2, 1); else if (u == 8) LAUNCH(8, 8, 1); else LAUNCH(8, 4, 1); }
  else         { if (u == 2) LAUNCH(4, 2, 1); else if (u == 8) LAUNCH(4, 8, 1); else LAUNCH(4, 4, 1); }
#undef LAUNCH
}

hipError_t launch_ffn_stage(const FfnStage& s, int max_active, int max_rows_per_expert, hipStream_t st) {
  static const int env_nw = env_int("MOEINF_FFN_NW", 0), env_u = env_int("MOEINF_FFN_U", 0);
  const int rmax = s.R > s.R_sh ? s.R : s.R_sh;
  dim3 grid((rmax + 15) / 16, max_active);
  const bool gated = (s.epi == EPI_GATED_SILU);
  // long reductions get 8 waves per block (more bytes in flight per CU), short ones 4
  const int kmax = s.K > s.K_sh ? s.K : s.K_sh;
  const size_t kbytes = (size_t)kmax * (s.dtype == DT_BF16 ? 2 : 4);
  const int nw = env_nw ? env_nw : (kbytes >= 16384 ? 8 : 4);
  const int u = env_u ? env_u : 4;
  static const int env_nt = env_int("MOEINF_FFN_NT", 0);
  // the decode kernel re-streams an expert's weights for every 16 rows: from 17 rows on, the GEMM kernels (one pass
  // per 128/256 rows) win — Mixtral at 64 tokens: 761 -> 549 us per layer (profiles/r01_ffn_sweep_midsize.txt)
  static const int many_rows = env_int("MOEINF_FFN_MANY_ROWS", 16);
  const bool many = s.fuse_combine ? false : (env_nt ? env_nt > 1 : max_rows_per_expert > many_rows);
  if (s.dtype == DT_BF16) {
    if (gated) launch_ffn_t<uint16_t, 2>(s, grid, nw, u, many, max_rows_per_expert, st); else launch_ffn_t<uint16_t, 1>(s, grid, nw, u, many, max_rows_per_expert, st);
  } else {
    if (gated) launch_ffn_t<float, 2>(s, grid, nw, u, many, max_rows_per_expert, st); else launch_ffn_t<float, 1>(s, grid, nw, u, many, max_rows_per_expert, st);
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// gate_logits: logits[t][e] = round_once( sum_h x[t][h] * wg[e][h] ), fp64 accumulation.
// grid = (E, ceil(T/TT)), block = 256.  fp64 makes the result independent of summation order to
// ~1e-16, so the bf16/fp32 rounding (and with it the top-k choice) matches the oracle bit for bit.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void load8(const T* p, float out[8]);
template <>
__device__ __forceinline__ void load8<uint16_t>(const uint16_t* p, float out[8]) {
  const u32x4 v = ld16(p);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    out[2 * j] = __uint_as_float(v[j] << 16);
    out[2 * j + 1] = __uint_as_float(v[j] & 0xffff0000u);
  }
}
template <>
__device__ __forceinline__ void load8<float>(const float* p, float out[8]) {
  const u32x4 a = ld16(p), b = ld16(p + 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) { out[j] = __uint_as_float(a[j]); out[4 + j] = __uint_as_float(b[j]); }
}

template <typename XT, typename WT, int TT>
__device__ __forceinline__ void gate_body(const XT* __restrict__ x, const WT* __restrict__ wg, float* __restrict__ logits,
                                          int T, int H, int E, int round_bf16, double (*red)[TT], const int e, const int t0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double acc[TT];
#pragma unroll
  for (int i = 0; i < TT; ++i) acc[i] = 0.0;
  const WT* wrow = wg + (size_t)e * H;
  const int nt = min(TT, T - t0);
  for (int h = tid * 8; h < H; h += 256 * 8) {  // H % 8 == 0
    float wv[8];
    load8<WT>(wrow + h, wv);
#pragma unroll
    for (int i = 0; i < TT; ++i) {
      if (i < nt) {
        float xv[8];
        load8<XT>(x + (size_t)(t0 + i) * H + h, xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i] = fma((double)wv[j], (double)xv[j], acc[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < TT; ++i) {
    double v = acc[i];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (tid < TT && t0 + tid < T) {
    const double v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    float f = (float)v;
    if (round_bf16) f = bf2f(f2bf(f));
    logits[(size_t)(t0 + tid) * E + e] = f;
  }
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
template <typename XT, typename WT, int TT, typename T, int U>
__global__ __launch_bounds__(256) void gate_shared1_kernel(const XT* __restrict__ x, const WT* __restrict__ wg, float* __restrict__ logits,
                                                           int T_, int H, int E, int round_bf16, int n_gate, FfnStage s) {
  __shared__ double redg[4][TT];
  __shared__ float red[4][2][256];
  const int b = blockIdx.x;
  if (b < n_gate) {
    gate_body<XT, WT, TT>(x, wg, logits, T_, H, E, round_bf16, redg, b % E, (b / E) * TT);
  } else {
    const char* W = reinterpret_cast<const char*>(s.wptr[s.E]);
    ffn_rows_item<T, 2, 4, U, 1>(s, b - n_gate, W, true, T_, 0, red);
  }
}

// Mixtral's gate is an nn.Linear in the model dtype (mixtral.py:46): its output is rounded to
// that dtype.  The other routers compute fp32 logits from (exactly) up-cast inputs.
static inline int gate_rounds_bf16(const RouteArgs& a) { return (a.kind == 0 /*MIXTRAL*/ && a.x_dtype == DT_BF16) ? 1 : 0; }

hipError_t launch_gate_shared1(const RouteArgs& a, const FfnStage& s, hipStream_t st) {
  constexpr int TT = 4;
  const int n_gate = a.E * ((a.T + TT - 1) / TT);
  dim3 grid(n_gate + (s.R_sh + 15) / 16);
  const int rb = gate_rounds_bf16(a);
  // bf16 model (DeepSeek): activations bf16, gate bf16 or fp32
  static const int u8 = env_int("MOEINF_SH1_U", 4) == 8;
#define GS1(WT, UU) hipLaunchKernelGGL((gate_shared1_kernel<uint16_t, WT, TT, uint16_t, UU>), grid, dim3(256), 0, st, (const uint16_t*)a.x, (const WT*)a.gate_w, a.logits, a.T, a.H, a.E, rb, n_gate, s)
  if (a.gate_dtype == DT_BF16) { if (u8) GS1(uint16_t, 8); else GS1(uint16_t, 4); }
  else { if (u8) GS1(float, 8); else GS1(float, 4); }
#undef GS1
  return hipGetLastError();
}

hipError_t launch_gate_logits(const RouteArgs& a, hipStream_t st) {
  constexpr int TT = 4;
  dim3 grid(a.E, (a.T + TT - 1) / TT);
  const int rb = gate_rounds_bf16(a);
#define GL(XT, WT) hipLaunchKernelGGL((gate_logits_kernel<XT, WT, TT>), grid, dim3(256), 0, st, (const XT*)a.x, (const WT*)a.gate_w, a.logits, a.T, a.H, a.E, rb)
  if (a.x_dtype == DT_BF16 && a.gate_dtype == DT_BF16) GL(uint16_t, uint16_t);
  else if (a.x_dtype == DT_BF16) GL(uint16_t, float);
  else if (a.gate_dtype == DT_BF16) GL(float, uint16_t);
  else GL(float, float);
#undef GL
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// route_topk: one wave per token.  E <= 256 (<= 4 experts per lane, expert id = lane + 64*j).
// ------------------------------------------------------------------------------------------------
// Wave-wide reductions on DPP (data-parallel primitives: register-to-register lane permutes on the
// VALU) instead of __shfl_xor, which lowers to ds_bpermute through the LDS crossbar (~100+ cycles of
// latency per step, and these chains are serial on the single wave that routes a token).
// Inside a row of 16 lanes: swap neighbours, swap pairs, half-row mirror, row mirror -> every lane holds
// the row result; the four row results are then read with v_readlane and combined on the scalar unit.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
#define MOEINF_ROW_REDUCE(OP)      \
  OP(0xB1)  /* quad_perm [1,0,3,2] */ \
  OP(0x4E)  /* quad_perm [2,3,0,1] */ \
  OP(0x141) /* row_half_mirror */     \
  OP(0x140) /* row_mirror */
__device__ __forceinline__ float wave_max(float v) {
#define STEP(C) v = fmaxf(v, __uint_as_float(dpp_mov<C>(__float_as_uint(v))));
  MOEINF_ROW_REDUCE(STEP)
#undef STEP
  const int b = __float_as_int(v);
  float r = __int_as_float(__builtin_amdgcn_readlane(b, 0));
  r = fmaxf(r, __int_as_float(__builtin_amdgcn_readlane(b, 16)));
  r = fmaxf(r, __int_as_float(__builtin_amdgcn_readlane(b, 32)));
  r = fmaxf(r, __int_as_float(__builtin_amdgcn_readlane(b, 48)));
  return r;
}
__device__ __forceinline__ float wave_sum(float v) {
#define STEP(C) v += __uint_as_float(dpp_mov<C>(__float_as_uint(v)));
  MOEINF_ROW_REDUCE(STEP)
#undef STEP
  const int b = __float_as_int(v);
  return ((__int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16))) +
          __int_as_float(__builtin_amdgcn_readlane(b, 32))) + __int_as_float(__builtin_amdgcn_readlane(b, 48));
}
// arg-max over the wave of (value desc, index asc); entries with idx < 0 never win.
// (value, index) is packed into one order-preserving 64-bit key so a single max-reduction decides.
__device__ __forceinline__ void wave_argmax(float& v, int& idx) {
  uint32_t ub = __float_as_uint(v);
  ub ^= (ub >> 31) ? 0xFFFFFFFFu : 0x80000000u;  // monotone map float -> uint32 (handles negatives, -inf)
  uint32_t hi = idx >= 0 ? ub : 0u;
  uint32_t lo = idx >= 0 ? (0xFFFFFFFFu - (uint32_t)idx) : 0u;  // larger lo = smaller index
#define STEP(C)                                                        \
  {                                                                    \
    const uint32_t oh = dpp_mov<C>(hi), ol = dpp_mov<C>(lo);           \
    const bool take = (oh > hi) || (oh == hi && ol > lo);              \
    hi = take ? oh : hi;                                               \
    lo = take ? ol : lo;                                               \
  }
  MOEINF_ROW_REDUCE(STEP)
#undef STEP
  uint32_t bh = 0, bl = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t oh = (uint32_t)__builtin_amdgcn_readlane((int)hi, r * 16), ol = (uint32_t)__builtin_amdgcn_readlane((int)lo, r * 16);
    const bool take = (oh > bh) || (oh == bh && ol > bl);
    bh = take ? oh : bh;
    bl = take ? ol : bl;
  }
  if (bh == 0 && bl == 0) { idx = -1; v = 0.f; return; }
  idx = (int)(0xFFFFFFFFu - bl);
  bh ^= (bh >> 31) ? 0x80000000u : 0xFFFFFFFFu;  // inverse map
  v = __uint_as_float(bh);
}
// pick the best not-yet-taken entry among this lane's 4 and reduce
__device__ __forceinline__ void pick_best(const float key[4], uint32_t taken, int lane, int E, float& bv, int& bi) {
  bv = 0.f; bi = -1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (64 * j >= E) break;  // wave-uniform: E <= 64 scans one entry per lane
    const int e = lane + 64 * j;
    if (e < E && !((taken >> j) & 1u)) {
      if (bi < 0 || key[j] > bv) { bv = key[j]; bi = e; }  // ascending e inside a lane: strict > keeps lowest
    }
  }
  wave_argmax(bv, bi);
}

// what the router decides for one token, in registers (every entry wave-uniform)
struct Routed {
  int sel[8];     // chosen experts in the router's own order (-1 = nothing selected)
  float w[8];     // combine weights
  int valid[8];   // 0: the pair is dropped (NLLB zero weight)
  float val0;     // Switch: probability of the top-1 expert
};

__device__ __forceinline__ void route_core(const RouteArgs& a, const int t, const int lane, Routed& o) {
  const int E = a.E, K = a.K;
  const float* lg = a.logits + (size_t)t * E;
  float l[4], p[4];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = lane + 64 * j;
    l[j] = (e < E) ? lg[e] : -INFINITY;
    m = fmaxf(m, l[j]);
  }
  m = wave_max(m);
  float ssum = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = lane + 64 * j;
    p[j] = 0.f;
    if (64 * j < E) {  // wave-uniform skip of the expf for absent columns
      p[j] = (e < E) ? expf(l[j] - m) : 0.f;
      ssum += p[j];
    }
  }
  ssum = wave_sum(ssum);
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = p[j] / ssum;

  const bool x_bf16 = (a.x_dtype == DT_BF16);
  int sel[8];
  float val[8];
  int valid[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { sel[k] = -1; val[k] = 0.f; valid[k] = 1; }
  uint32_t taken = 0;

  if (a.kind == 0 /*MIXTRAL*/ || (a.kind == 1 /*DEEPSEEK*/ && a.n_group <= 1)) {
    for (int k = 0; k < K; ++k) {
      float bv; int bi;
      pick_best(p, taken, lane, E, bv, bi);
      sel[k] = bi; val[k] = bv;
      if (bi >= 0 && (bi & 63) == lane) taken |= 1u << (bi >> 6);
    }
  } else if (a.kind == 1) {  // group_limited_greedy (modeling_deepseek.py:484-503)
    const int gs = E / a.n_group;
    // group scores: lane g (< n_group) ends up holding max over group g
    float gscore = -INFINITY;
    for (int g = 0; g < a.n_group; ++g) {
      float gm = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = lane + 64 * j;
        if (e < E && e / gs == g) gm = fmaxf(gm, p[j]);
      }
      gm = wave_max(gm);
      if (lane == g) gscore = gm;
    }
    uint64_t gmask = 0;  // selected groups (n_group <= 64)
    bool gtaken = false;
    for (int k = 0; k < a.topk_group; ++k) {
      float bv = gscore; int bi = (lane < a.n_group && !gtaken) ? lane : -1;
      wave_argmax(bv, bi);
      if (bi == lane) gtaken = true;
      if (bi >= 0) gmask |= 1ull << bi;
    }
    float pm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = lane + 64 * j;
      pm[j] = (e < E && ((gmask >> (e / gs)) & 1ull)) ? p[j] : 0.f;
    }
    for (int k = 0; k < K; ++k) {
      float bv; int bi;
      pick_best(pm, taken, lane, E, bv, bi);
      sel[k] = bi; val[k] = bv;
      if (bi >= 0 && (bi & 63) == lane) taken |= 1u << (bi >> 6);
    }
  } else if (a.kind == 2 /*SWITCH*/) {
    float pin[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pin[j] = x_bf16 ? bf2f(f2bf(p[j])) : p[j];
    float bv; int bi;
    pick_best(pin, 0u, lane, E, bv, bi);
    sel[0] = bi; val[0] = bv;
  } else {  /*NLLB*/
    float pin[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pin[j] = x_bf16 ? bf2f(f2bf(p[j])) : p[j];
    float bv; int bi;
    pick_best(pin, 0u, lane, E, bv, bi);  // top-1 over probabilities cast to the input dtype
    sel[0] = bi; val[0] = bv;
    if ((bi & 63) == lane) taken |= 1u << (bi >> 6);
    float lv; int li;
    pick_best(l, taken, lane, E, lv, li);  // top-2 over fp32 logits with top-1 masked out
    sel[1] = li;
    // probability (input dtype) of the top-2 expert
    float p2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (lane + 64 * j == li) p2 = pin[j];
    p2 = wave_sum(p2);
    val[1] = p2;
  }

  // weights
  float w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = 0.f;
  if (a.kind == 0) {
    float den = 0.f;
    for (int k = 0; k < K; ++k) den += val[k];
    for (int k = 0; k < K; ++k) { w[k] = val[k] / den; if (x_bf16) w[k] = bf2f(f2bf(w[k])); }
  } else if (a.kind == 1) {
    if (K > 1 && a.norm_topk_prob) {
      float den = 0.f;
      for (int k = 0; k < K; ++k) den += val[k];
      den += 1e-20f;
      for (int k = 0; k < K; ++k) w[k] = val[k] / den;
    } else {
      for (int k = 0; k < K; ++k) w[k] = val[k] * a.scale;
    }
  } else if (a.kind == 2) {
    w[0] = val[0];
  } else {
    // normalize_router_probabilities in the input dtype (nllb router, eval: capacity never drops)
    const float eps = x_bf16 ? 0.0078125f : 1.1920928955078125e-07f;
    float den = val[0] + val[1];
    if (x_bf16) den = bf2f(f2bf(den));
    den = fmaxf(den, eps);
    w[0] = val[0] / den; w[1] = val[1] / den;
    if (x_bf16) { w[0] = bf2f(f2bf(w[0])); w[1] = bf2f(f2bf(w[1])); }
    valid[0] = (w[0] != 0.f); valid[1] = (w[1] != 0.f);  // router_mask = combining_weights.bool()
  }

#pragma unroll
  for (int k = 0; k < 8; ++k) { o.sel[k] = sel[k]; o.w[k] = w[k]; o.valid[k] = valid[k]; }
  o.val0 = val[0];
}

__device__ __forceinline__ void route_store(const RouteArgs& a, const int t, const int lane, const Routed& o, int* sel_out = nullptr,
                                            float* w_out = nullptr, int* rank_out = nullptr) {
  const int K = a.K;
  // lane k (< K) owns entry k: its rank among the token's experts by ascending id is its place in the
  // (deterministic) combine order.  Stable for repeated ids (-1 = nothing selected).
  int my_sel = -1, my_valid = 0, rank = 0;
  float my_w = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (lane == k) { my_sel = o.sel[k]; my_w = o.w[k]; my_valid = o.valid[k]; }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (j < K) rank += (o.sel[j] < my_sel || (o.sel[j] == my_sel && j < lane)) ? 1 : 0;
  if (lane < K) {
    a.topk_idx[(size_t)t * K + lane] = my_sel;
    a.topk_w[(size_t)t * K + lane] = my_w;
    a.pair_valid[(size_t)t * K + lane] = (my_sel >= 0) ? my_valid : 0;
    a.pair_order[(size_t)t * K + rank] = lane;
  }
  if (lane == 0 && a.router_prob) a.router_prob[t] = o.val0;
  if (sel_out) { *sel_out = my_sel; *w_out = my_w; *rank_out = rank; }
}

__device__ __forceinline__ void route_token(const RouteArgs& a, const int t, const int lane) {
  Routed o;
  route_core(a, t, lane, o);
  route_store(a, t, lane, o);
}

__global__ __launch_bounds__(256) void route_topk_kernel(RouteArgs a) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t < a.T) route_token(a, t, threadIdx.x & 63);
}

hipError_t launch_route_topk(const RouteArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(route_topk_kernel, dim3((a.T + 3) / 4), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// dispatch_index: single workgroup (1024 threads = 16 waves), chunks of 1024 pairs in pair order
// (token-major).  Stable counting sort by expert id built from wave ballots:
//   rank inside the wave   = popc(ballot(same expert) & lanes-below)
//   rank across waves      = per-wave counts scanned by one thread per expert
//   rank across chunks     = running per-expert counters in LDS
// Outputs replace the dense router_mask[T,E] of the reference (mixtral.py:56-65) and the
// tokens-per-expert D2H sum of dispatch_local (expert_executor.py:34-43).
// ------------------------------------------------------------------------------------------------
constexpr int IDX_THREADS = 1024;
constexpr int IDX_WAVES = IDX_THREADS / 64;
constexpr int IDX_MAXE = 257;  // E + shared pseudo-expert
#define IDX_AT(a, p) ((a).topk_idx[(size_t)(p) * ((a).idx_stride > 1 ? (a).idx_stride : 1)])

__device__ __forceinline__ uint64_t lanes_below(int lane) { return (lane == 0) ? 0ull : (~0ull >> (64 - lane)); }

// rank of each counted lane among earlier counted pairs with the same key; updates running[]
__device__ __forceinline__ int chunk_rank(int key, bool counted, int* wave_cnt /*[IDX_WAVES][IDX_MAXE]*/,
                                          int* running /*[IDX_MAXE]*/, int nkeys) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < IDX_WAVES * nkeys; i += IDX_THREADS) wave_cnt[(i / nkeys) * IDX_MAXE + (i % nkeys)] = 0;
  __syncthreads();
  int rank_in_wave = 0;
  uint64_t todo = __ballot(counted);
  while (todo) {
    const int leader = __ffsll((unsigned long long)todo) - 1;
    const int k = __shfl(key, leader);
    const uint64_t same = __ballot(counted && key == k);
    if (counted && key == k) rank_in_wave = __popcll(same & lanes_below(lane));
    if (lane == leader) wave_cnt[wave * IDX_MAXE + k] = __popcll(same);
    todo &= ~same;
  }
  __syncthreads();
  if (tid < nkeys) {  // exclusive scan over waves, seeded with the running count
    int base = running[tid];
    for (int w = 0; w < IDX_WAVES; ++w) {
      const int c = wave_cnt[w * IDX_MAXE + tid];
      wave_cnt[w * IDX_MAXE + tid] = base;
      base += c;
    }
    running[tid] = base;
  }
  __syncthreads();
  const int pos = counted ? wave_cnt[wave * IDX_MAXE + key] + rank_in_wave : -1;
  __syncthreads();
  return pos;
}

__device__ __forceinline__ void index_small(const IndexArgs& a, int* cnt, int* offs);

__device__ __forceinline__ void index_body(const IndexArgs& a, int* wave_cnt, int* running, int* offs, int* scan_tmp) {
  const int tid = threadIdx.x;
  const int E = a.E, K = a.K, T = a.T;
  const int nkeys = E;
  const int npairs = T * K;

  // pass A (Switch): per-batch-row capacity.  token_priority = cumsum over the sequence dim of the
  // un-masked one-hot; tokens with priority > capacity are dropped (HF SwitchTransformersTop1Router).
  if (a.capacity > 0 && a.pair_valid) {
    const int S = T / a.rows;  // K == 1
    for (int b = 0; b < a.rows; ++b) {
      for (int i = tid; i < nkeys; i += IDX_THREADS) running[i] = 0;
      __syncthreads();
      for (int c0 = 0; c0 < S; c0 += IDX_THREADS) {
        const int s = c0 + tid;
        const bool in = s < S;
        const int p = b * S + s;
        const int key = in ? IDX_AT(a, p) : -1;
        const bool counted = in && key >= 0 && key < E;
        const int pos = chunk_rank(counted ? key : 0, counted, wave_cnt, running, nkeys);
        if (counted && pos + 1 > a.capacity) a.pair_valid[p] = 0;
      }
    }
    __threadfence_block();
    __syncthreads();
  }

  // pass B: stable rank of every dispatched pair inside its expert
  for (int i = tid; i < IDX_MAXE; i += IDX_THREADS) running[i] = 0;
  __syncthreads();
  for (int c0 = 0; c0 < npairs; c0 += IDX_THREADS) {
    const int p = c0 + tid;
    const bool in = p < npairs;
    const int key = in ? IDX_AT(a, p) : -1;
    const bool counted = in && key >= 0 && key < E && (a.pair_valid ? a.pair_valid[p] != 0 : true);
    const int pos = chunk_rank(counted ? key : 0, counted, wave_cnt, running, nkeys);
    if (in) a.pair_slot[p] = pos;  // rank for now; rebased below
  }
  // counts (+ shared pseudo-expert), exclusive scan, active list
  const int ne = E + 1;
  if (tid == 0) running[E] = a.shared ? T : 0;
  __syncthreads();
  if (tid < ne) scan_tmp[tid] = running[tid];
  __syncthreads();
  if (tid == 0) {  // ne <= 257: a serial scan costs < 1 us and keeps the order obvious
    int acc = 0, na = 0;
    for (int e = 0; e < ne; ++e) {
      offs[e] = acc;
      acc += scan_tmp[e];
      if (scan_tmp[e] > 0) a.active[na++] = e;
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
    a.counts[tid] = scan_tmp[tid];
    if (a.mirror) a.mirror[1 + tid] = scan_tmp[tid];
  }
  if (tid <= ne) a.offsets[tid] = offs[tid];
  // rebase ranks to expert-sorted rows
  for (int p = tid; p < npairs; p += IDX_THREADS) {
    const int rk = a.pair_slot[p];
    if (rk >= 0) {
      const int slot = offs[IDX_AT(a, p)] + rk;
      a.pair_slot[p] = slot;
      a.slot_token[slot] = p / K;
      a.slot_pair[slot] = p;
    }
  }
  if (a.shared) {
    for (int t = tid; t < T; t += IDX_THREADS) {
      a.slot_token[offs[E] + t] = t;
      a.slot_pair[offs[E] + t] = -1;
    }
  }
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

// <= 64 (token,k) pairs: the whole dispatch index in ONE wave, no workgroup barriers.  Same outputs as
// index_body (stable ranks by ballot/popc, prefix sums by wave shuffles).  Not for per-row capacity.
__device__ __forceinline__ void index_small(const IndexArgs& a, int* cnt /*LDS [IDX_MAXE]*/, int* offs /*LDS [IDX_MAXE+1]*/) {
  const int lane = threadIdx.x & 63;
  const int E = a.E, K = a.K, T = a.T, npairs = T * K, ne = E + 1;
  for (int e = lane; e < ne; e += 64) cnt[e] = 0;
  const bool in = lane < npairs;
  const int key = in ? IDX_AT(a, lane) : -1;
  const bool counted = in && key >= 0 && key < E && (a.pair_valid ? a.pair_valid[lane] != 0 : true);
  int rank = 0;
  uint64_t todo = __ballot(counted);
  while (todo) {
    const int leader = __ffsll((unsigned long long)todo) - 1;
    const int k = __shfl(key, leader);
    const uint64_t same = __ballot(counted && key == k);
    if (counted && key == k) rank = __popcll(same & lanes_below(lane));
    if (lane == leader) cnt[k] = __popcll(same);
    todo &= ~same;
  }
  if (lane == 0) cnt[E] = a.shared ? T : 0;
  __builtin_amdgcn_wave_barrier();
  // exclusive scan over ne <= 257 experts: 5 consecutive experts per lane
  int c[5], loc = 0, nz = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int e = lane * 5 + j;
    c[j] = (e < ne) ? cnt[e] : 0;
    loc += c[j];
    nz += c[j] > 0;
  }
  int pre = loc, pnz = nz;
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(pre, o), w = __shfl_up(pnz, o);
    if (lane >= o) { pre += v; pnz += w; }
  }
  const int total_nz = __shfl(pnz, 63), total = __shfl(pre, 63);
  pre -= loc; pnz -= nz;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int e = lane * 5 + j;
    if (e < ne) {
      offs[e] = pre;
      a.offsets[e] = pre;
      a.counts[e] = c[j];
      // the mirror is pinned HOST memory (every store is a PCIe write): only the active experts are reported,
      // the host zeroes the counts before it hands the buffer out and reads active[] up to n_active only
      if (c[j] > 0) {
        a.active[pnz] = e;
        if (a.mirror) { a.mirror[1 + e] = c[j]; a.mirror[1 + ne + pnz] = e; }
        ++pnz;
      }
      pre += c[j];
    }
  }
  if (lane == 0) {
    a.offsets[ne] = total;
    *a.n_active = total_nz;
    if (a.mirror) a.mirror[0] = total_nz;
  }
  __builtin_amdgcn_wave_barrier();
  if (in) {
    int slot = -1;
    if (counted) {
      slot = offs[key] + rank;
      a.slot_token[slot] = lane / K;
      a.slot_pair[slot] = lane;
    }
    a.pair_slot[lane] = slot;
  }
  if (a.shared) {
    const int base = offs[E];
    for (int t = lane; t < T; t += 64) { a.slot_token[base + t] = t; a.slot_pair[base + t] = -1; }
  }
}

// decode-sized batches: softmax/top-k of every token (one wave each) and the dispatch index in ONE
// launch of one workgroup — saves a kernel boundary per layer where launches dominate the layer time
__global__ __launch_bounds__(IDX_THREADS) void route_index_kernel(RouteArgs r, IndexArgs a) {
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
}

// block 0: softmax/top-k of every token (4 waves) + the one-wave dispatch index (T*K <= 64); blocks 1..: 16 rows each of
// the shared expert's down projection over h_shared (written by gate_shared1_kernel)
template <typename T, int NW, int U>
__global__ __launch_bounds__(NW * 64) void route_shared2_kernel(RouteArgs r, IndexArgs a, FfnStage s) {
  __shared__ int running[IDX_MAXE];
  __shared__ int offs[IDX_MAXE + 1];
  __shared__ float red[NW][1][256];
  if (blockIdx.x == 0) {
    for (int t = threadIdx.x >> 6; t < r.T; t += NW) route_token(r, t, threadIdx.x & 63);
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x < 64) index_small(a, running, offs);
  } else {
    const char* W = reinterpret_cast<const char*>(s.wptr[s.E]);
    ffn_rows_item<T, 1, NW, U, 1>(s, (int)blockIdx.x - 1, W, true, r.T, 0, red);
  }
}
hipError_t launch_route_shared2(const RouteArgs& r, const IndexArgs& a, const FfnStage& s, hipStream_t st) {
  static const int nw = env_int("MOEINF_SH2_NW", 8), u = env_int("MOEINF_SH2_U", 4);
  const dim3 grid(1 + (s.R_sh + 15) / 16);
#define RS2(NWV, UU) hipLaunchKernelGGL((route_shared2_kernel<uint16_t, NWV, UU>), grid, dim3(NWV * 64), 0, st, r, a, s)
  if (nw == 16) { if (u == 8) RS2(16, 8); else RS2(16, 4); }
  else if (nw == 4) { if (u == 8) RS2(4, 8); else RS2(4, 4); }
  else { if (u == 8) RS2(8, 8); else RS2(8, 4); }
#undef RS2
  return hipGetLastError();
}

// The SET of experts route_core picks for one token of a greedy softmax top-K router (Mixtral; DeepSeek with
// n_group <= 1), E <= 64 (one logit per lane), as a bit mask.  Same arithmetic as route_core on that path — p =
// expf(l - max) / sum with the same wave reductions, K rounds of (largest p, ties -> lowest id) — in ~100-200
// instructions: p >= 0, so its bit pattern orders like an unsigned integer; a round is one DPP max-reduction, one
// ballot of the lanes that hold the maximum, and the lowest such lane wins.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#define STEP(C) v = max(v, dpp_mov<C>(v));
  MOEINF_ROW_REDUCE(STEP)
#undef STEP
  uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
  r = max(r, (uint32_t)__builtin_amdgcn_readlane((int)v, 16));
  r = max(r, (uint32_t)__builtin_amdgcn_readlane((int)v, 32));
  r = max(r, (uint32_t)__builtin_amdgcn_readlane((int)v, 48));
  return r;
}
__device__ __forceinline__ uint64_t route_set_lean(const float* __restrict__ logits, const int E, const int K, const int lane) {
  const bool in = lane < E;
  const float l = in ? logits[lane] : -INFINITY;
  const float m = wave_max(l);
  float p = in ? expf(l - m) : 0.f;
  const float ssum = wave_sum(p);
  p = p / ssum;
  const uint32_t key = in ? __float_as_uint(p) : 0u;
  uint64_t chosen = 0, avail = __ballot(in);
  for (int k = 0; k < K && avail; ++k) {
    const uint64_t mine = 1ull << lane;
    const uint32_t best = wave_max_u32((avail & mine) ? key : 0u);
    const uint64_t at = __ballot(key == best) & avail;
    if (!at) break;
    const uint64_t win = at & (~at + 1);  // lowest lane holding the maximum
    chosen |= win;
    avail &= ~win;
  }
  return chosen;
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
template <typename T, int NW, int U>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_num_sgpr(80))) void ffn1_selfroute_kernel(RouteArgs r, IndexArgs a, FfnStage s, FfnStage sh2, int n_rg, int n_sh2) {
  __shared__ float red[NW][2][256];
  __shared__ unsigned long long sh_w;
  __shared__ int sh_rank_ok;
  static_assert(sizeof(float) * NW * 2 * 256 >= sizeof(int) * (2 * IDX_MAXE + 1), "index scratch aliases the reduction buffer");
  const int lane = threadIdx.x & 63;
  const int K = r.K, E = r.E;
  // block 0 is the meta block: it must be dispatched FIRST — when the grid exceeds one resident wave of blocks
  // (Mixtral: 1793 blocks, 6 per CU) a meta block at the end of the grid would start only when a slot frees up and
  // put its ~5 us of serial work (generic router, index, PCIe mirror writes) behind the weight stream's tail
  int b = (int)blockIdx.x - 1;
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
    uint64_t chosen = route_set_lean(r.logits, E, K, lane);
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
  ffn_rows_item<T, 2, NW, U, 1>(s, rg, W, false, 1, u, red, 0);
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
  hipLaunchKernelGGL((ffn1_selfroute_kernel<uint16_t, 4, 4>), grid, dim3(256), dyn, st, r, a, s1, sh2 ? *sh2 : s1, n_rg, n_sh2);
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
  if (tid == 0) is_last = (__hip_atomic_fetch_add(&s.tile_done[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == K - 1);
  __syncthreads();
  if (is_last) {
    if (tid < 4) combine_apply<T, true>(s.comb, 0, r0 + tid * 4, m);
    if (tid == 0) s.tile_done[blockIdx.x] = 0;
  }
}

// The same stage for K = 2 (Mixtral): ONE workgroup owns 16 output columns for BOTH chosen experts — wave group g
// streams expert slot g's 16 rows (k-tiles interleaved over its NWE waves), the partial sums meet in LDS, and the combine
// runs inside the workgroup: no write-through stores, no store-ack wait, no arrival counter, no coherent row loads
// (that tail costs 3.5 us per launch, tools/ffn_micro.hip).  H/16 = 256 workgroups for Mixtral = one per CU.
// Summation order inside an expert (waves 0..NWE-1, tiles in ascending k inside a wave) and the combine order
// (ascending expert id) are those of ffn2_decode1_kernel, so the two produce identical bits.  Requires K % 32 == 0.
template <int NWE, int U>
__global__ __launch_bounds__(2 * NWE * 64) void ffn2_decode1_pair_kernel(FfnStage s) {
  typedef uint16_t T;
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

hipError_t launch_ffn2_decode1(const FfnStage& s2, hipStream_t st) {
  static const int pair_env = env_int("MOEINF_DEC1_PAIR", 1);  // 0: always the arrival-counter form; 4 / 8: waves per expert
  if (pair_env && s2.comb.K == 2 && (s2.K % 32) == 0 && !(s2.comb.kind == 1 && s2.comb.y_shared) && s2.comb.kind <= 1) {
    const dim3 g1((s2.R + 15) / 16);
    // 4 waves per expert (8 per CU), batches of 4 tiles: 38.9 us per Mixtral launch; 8 waves per expert 40.5; the
    // arrival-counter form 41.9
    static const int pu = env_int("MOEINF_DEC1_PAIR_U", 4);
    if (pair_env == 8) hipLaunchKernelGGL((ffn2_decode1_pair_kernel<8, 4>), g1, dim3(1024), 0, st, s2);
    else if (pu == 8) hipLaunchKernelGGL((ffn2_decode1_pair_kernel<4, 8>), g1, dim3(512), 0, st, s2);
    else if (pu == 2) hipLaunchKernelGGL((ffn2_decode1_pair_kernel<4, 2>), g1, dim3(512), 0, st, s2);
    else hipLaunchKernelGGL((ffn2_decode1_pair_kernel<4, 4>), g1, dim3(512), 0, st, s2);
    return hipGetLastError();
  }
  const dim3 grid((s2.R + 15) / 16, s2.comb.K);
  const size_t kbytes = (size_t)s2.K * 2;
  if (kbytes >= 16384) hipLaunchKernelGGL((ffn2_decode1_kernel<uint16_t, 8, 4>), grid, dim3(512), 0, st, s2);
  else hipLaunchKernelGGL((ffn2_decode1_kernel<uint16_t, 4, 4>), grid, dim3(256), 0, st, s2);
  return hipGetLastError();
}

hipError_t launch_route_index(const RouteArgs& r, const IndexArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(route_index_kernel, dim3(1), dim3(IDX_THREADS), 0, st, r, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// mask_index: dispatch index from a dense router_mask[T,E] (what the reference's Python blocks hand to
// dispatch_local, expert_executor.py:32-58).  One workgroup; wave w owns experts w, w+16, ...; tokens
// are scanned 64 at a time with a ballot, so each expert's rows come out in ascending token order
// (= the boolean-mask gather order of expert_dispatcher.cpp:274-284).
// ------------------------------------------------------------------------------------------------
template <typename MT>
__global__ __launch_bounds__(IDX_THREADS) void mask_index_kernel(const MT* __restrict__ mask, int T, int E, IndexArgs a) {
  __shared__ int cnt[IDX_MAXE];
  __shared__ int offs[IDX_MAXE + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = wave; e < E; e += IDX_WAVES) {
    int c = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
      const int t = t0 + lane;
      const bool on = t < T && mask[(size_t)t * E + e] != (MT)0;
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
      const bool on = t < T && mask[(size_t)t * E + e] != (MT)0;
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
hipError_t launch_mask_index(const void* mask, int mask_elem_bytes, int T, int E, const IndexArgs& a, hipStream_t st) {
  if (mask_elem_bytes == 1) hipLaunchKernelGGL(mask_index_kernel<uint8_t>, dim3(1), dim3(IDX_THREADS), 0, st, (const uint8_t*)mask, T, E, a);
  else if (mask_elem_bytes == 4) hipLaunchKernelGGL(mask_index_kernel<int32_t>, dim3(1), dim3(IDX_THREADS), 0, st, (const int32_t*)mask, T, E, a);
  else hipLaunchKernelGGL(mask_index_kernel<int64_t>, dim3(1), dim3(IDX_THREADS), 0, st, (const int64_t*)mask, T, E, a);
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

hipError_t launch_combine(const CombineArgs& a, hipStream_t st) {
  dim3 grid((a.H + 1023) / 1024, a.T);
  if (a.dtype == DT_BF16) hipLaunchKernelGGL(combine_kernel<uint16_t>, grid, dim3(256), 0, st, a);
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

// ------------------------------------------------------------------------------------------------
// expert-parallel helpers
// ------------------------------------------------------------------------------------------------
__global__ void ep_dest_key_kernel(const int32_t* topk_idx, const int32_t* pair_valid, int32_t* key, int32_t* pair_pos, int n, int ep) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int e = topk_idx[p];
  key[p] = (e >= 0 && (!pair_valid || pair_valid[p])) ? (e % ep) : -1;
  if (pair_pos) pair_pos[p] = -1;  // ep_pack fills the dispatched ones
}
hipError_t launch_ep_dest_key(const int32_t* topk_idx, const int32_t* pair_valid, int32_t* key, int32_t* pair_pos,
                              int n_pairs, int ep_size, hipStream_t st) {
  hipLaunchKernelGGL(ep_dest_key_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, st, topk_idx, pair_valid, key,
                     pair_pos, n_pairs, ep_size);
  return hipGetLastError();
}

// grid = (ep_size*cap_rows), block = 256: one send row per block
template <typename T>
__global__ __launch_bounds__(256) void ep_pack_kernel(EpPackArgs a) {
  const int row = blockIdx.x;
  const int d = row / a.cap_rows, pos = row % a.cap_rows;
  const int cnt = a.counts[d];
  T* dst = reinterpret_cast<T*>(a.send) + (size_t)row * a.ld_send;
  int32_t* tail = reinterpret_cast<int32_t*>(dst + a.H);
  if (pos >= cnt) {
    if (threadIdx.x == 0) tail[0] = -1;
    return;
  }
  const int pair = a.slot_pair[a.offsets[d] + pos];
  const int t = pair / a.K;
  if (threadIdx.x == 0) {
    tail[0] = a.topk_idx[pair];
    a.pair_pos[pair] = row;
  }
  const T* src = reinterpret_cast<const T*>(a.x) + (size_t)t * a.H;
  constexpr int EPV = DT<T>::EPV;
  for (int h = threadIdx.x * EPV; h < a.H; h += 256 * EPV) *reinterpret_cast<u32x4*>(dst + h) = ld16(src + h);
}
// <= 64 (token,k) pairs (decode): destination keys, stable ranks and the row copy in ONE launch.  Every block
// (= one send row (d, pos)) re-derives "which pair is the pos-th one bound for rank d" with two ballots over the
// pairs — the same stable order the dest-key + dispatch_index + pack sequence produces.
template <typename T>
__global__ __launch_bounds__(256) void ep_pack_small_kernel(EpPackArgs a, const int32_t* pair_valid, int n_pairs, int32_t* send_counts) {
  __shared__ int s_pair, s_cnt;
  const int row = blockIdx.x;
  const int d = row / a.cap_rows, pos = row % a.cap_rows;
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    int key = -1;
    if (lane < n_pairs) {
      const int e = a.topk_idx[lane];
      if (e >= 0 && (!pair_valid || pair_valid[lane])) key = e % a.ep_size;
    }
    const uint64_t mine = __ballot(key == d);
    const int rank = __popcll(mine & lanes_below(lane));
    const uint64_t hit = __ballot(key == d && rank == pos);
    if (lane == 0) { s_pair = hit ? (__ffsll((unsigned long long)hit) - 1) : -1; s_cnt = __popcll(mine); }
    if (row == 0 && lane < n_pairs && key < 0) a.pair_pos[lane] = -1;  // never dispatched
  }
  __syncthreads();
  const int pair = s_pair;
  T* dst = reinterpret_cast<T*>(a.send) + (size_t)row * a.ld_send;
  int32_t* tail = reinterpret_cast<int32_t*>(dst + a.H);
  if (threadIdx.x == 0 && pos == 0 && send_counts) send_counts[d] = s_cnt;
  if (pair < 0) {
    if (threadIdx.x == 0) tail[0] = -1;
    return;
  }
  if (threadIdx.x == 0) {
    tail[0] = a.topk_idx[pair];
    a.pair_pos[pair] = row;
  }
  const T* src = reinterpret_cast<const T*>(a.x) + (size_t)(pair / a.K) * a.H;
  constexpr int EPV = DT<T>::EPV;
  for (int h = threadIdx.x * EPV; h < a.H; h += 256 * EPV) *reinterpret_cast<u32x4*>(dst + h) = ld16(src + h);
}
// Variable-split exchange (prefill-sized batches): send rows are COMPACT and sorted by destination rank — row r of
// `send` is the r-th pair in destination order (slot_pair from dispatch_index over the destination keys), so the
// all-to-all moves exactly the routed rows (split sizes = counts per destination) instead of a fixed capacity per
// peer.  grid = n_pairs blocks; blocks past the number of dispatched pairs exit.
template <typename T>
__global__ __launch_bounds__(256) void ep_pack_compact_kernel(EpPackArgs a, int n_pairs) {
  const int row = blockIdx.x;
  const int total = a.offsets[a.ep_size];
  if (row >= total) return;
  const int pair = a.slot_pair[row];
  T* dst = reinterpret_cast<T*>(a.send) + (size_t)row * a.ld_send;
  if (threadIdx.x == 0) {
    reinterpret_cast<int32_t*>(dst + a.H)[0] = a.topk_idx[pair];
    a.pair_pos[pair] = row;
  }
  const T* src = reinterpret_cast<const T*>(a.x) + (size_t)(pair / a.K) * a.H;
  constexpr int EPV = DT<T>::EPV;
  for (int h = threadIdx.x * EPV; h < a.H; h += 256 * EPV) *reinterpret_cast<u32x4*>(dst + h) = ld16(src + h);
}
hipError_t launch_ep_pack_compact(const EpPackArgs& a, int n_pairs, hipStream_t st) {
  if (a.dtype == DT_BF16) hipLaunchKernelGGL(ep_pack_compact_kernel<uint16_t>, dim3(n_pairs), dim3(256), 0, st, a, n_pairs);
  else hipLaunchKernelGGL(ep_pack_compact_kernel<float>, dim3(n_pairs), dim3(256), 0, st, a, n_pairs);
  return hipGetLastError();
}

hipError_t launch_ep_pack_small(const EpPackArgs& a, const int32_t* pair_valid, int n_pairs, int32_t* send_counts, hipStream_t st) {
  if (a.dtype == DT_BF16) hipLaunchKernelGGL(ep_pack_small_kernel<uint16_t>, dim3(a.ep_size * a.cap_rows), dim3(256), 0, st, a, pair_valid, n_pairs, send_counts);
  else hipLaunchKernelGGL(ep_pack_small_kernel<float>, dim3(a.ep_size * a.cap_rows), dim3(256), 0, st, a, pair_valid, n_pairs, send_counts);
  return hipGetLastError();
}

hipError_t launch_ep_pack(const EpPackArgs& a, hipStream_t st) {
  if (a.dtype == DT_BF16) hipLaunchKernelGGL(ep_pack_kernel<uint16_t>, dim3(a.ep_size * a.cap_rows), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(ep_pack_kernel<float>, dim3(a.ep_size * a.cap_rows), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace moeinf
