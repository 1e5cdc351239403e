// kdev.h — device-side building blocks shared by the kernel translation units (kernels.hip, ffn_gemm.hip, ...):
// dtype helpers, the MFMA wrapper, the combine, the weight-streaming work item of the decode FFN, the router and the
// dispatch-index device functions.  Everything here is __device__ __forceinline__ / constexpr / static.
#pragma once
#include "kernels.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <type_traits>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

// a launch that carries the armed profiling timer, if there is one (kernels.h: arm_kernel_timer)
#define KL(kern, grid, block, shm, st, ...) do { hipEvent_t ta_, tb_; \
    if (moeinf::take_kernel_timer(&ta_, &tb_)) hipExtLaunchKernelGGL(kern, grid, block, shm, st, ta_, tb_, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kern, grid, block, shm, st, __VA_ARGS__); } while (0)

namespace moeinf {

// ffn_gemm_big.hip: the compute-bound grouped GEMM (false: shape not supported, the caller picks another kernel)
bool launch_ffn_gemm_big(const FfnStage& s, int nmat, dim3 grid, int max_rows, hipStream_t st);

// tuning knobs are read once per process from the environment (sweeps only)
static inline int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}


typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// ------------------------------------------------------------------------------------------------
// scalar helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(uint16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
// fp32 -> bf16, round-to-nearest-even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, two values per
// instruction; the integer form — add 0x7fff + lsb, NaN test — is 6 VALU instructions and a compare per value, and the
// GEMM epilogues round three or four times per output element)
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// fp16 storage (the reference's expert dtype id 2, core/parallel/expert_module.h:20-23): _Float16 is a distinct 2-byte type,
// so the kernel templates tell it from bf16 (uint16_t); conversions are the hardware's (v_cvt_f16_f32 / v_cvt_f32_f16, RNE)
typedef _Float16 half_t;
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float h2f(half_t h) { return (float)h; }
__device__ __forceinline__ half_t f2h(float f) { return (half_t)f; }
// round a router quantity to the MODEL dtype (DT_BF16 / DT_F16; DT_F32: unchanged)
__device__ __forceinline__ float round_model(int dtype, float f) {
  return dtype == 0 /*DT_BF16*/ ? bf2f(f2bf(f)) : (dtype == 2 /*DT_F16*/ ? h2f(f2h(f)) : f);
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
// 16-byte write-through (sc1) store: a relaxed agent-scope __hip_atomic_store lowers to an sc1 store only up to 8 bytes, and
// narrow sc1 stores are one fabric write each (a 2-byte one costs ~12x a 16-byte one per byte, MI355X_MICROARCH.md).  hipcc does
// not count an asm store: the caller drains it with wait_stores_acked() (the s_nop keeps the data registers alive until read).
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_;
__device__ __forceinline__ void st16_coherent(void* p, u32x4_ v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// ---- peer-store exchange (kernels.h: EpPeers): rows and flags written into ANOTHER rank's window -------------------
// System scope = sc0 sc1: the store is written through every cache level of the writer and acknowledged by the memory it
// lands in (this device's or, over xGMI, a peer's), so "drain (s_waitcnt vmcnt(0)), then publish" orders payload before flag.
__device__ __forceinline__ void st16_system(void* p, u32x4_ v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
template <typename V>
__device__ __forceinline__ void st_system(V* p, V v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// row `row` = (destination rank d, position) of the sender's numbering -> its place in rank d's receive region
__device__ __forceinline__ char* ep_peer_recv_row(const EpPeers& pv, const int row, const int64_t row_bytes) {
  const int d = row / pv.cap_rows, pos = row - d * pv.cap_rows;
  return reinterpret_cast<char*>(pv.base[d]) + pv.recv_off + ((int64_t)pv.rank * pv.cap_rows + pos) * row_bytes;
}
// arrival row `row` = (source rank p, position) on the owner -> its place in rank p's return region
__device__ __forceinline__ char* ep_peer_ret_row(const EpPeers& pv, const int row, const int64_t row_bytes) {
  const int p = row / pv.cap_rows, pos = row - p * pv.cap_rows;
  return reinterpret_cast<char*>(pv.base[p]) + pv.ret_off + ((int64_t)pv.rank * pv.cap_rows + pos) * row_bytes;
}
// publish exchange `epoch` in flag word `rank` of every peer's flag set (flags_off = 0: rows, EP_RET_FLAGS_OFF: outputs);
// call with ONE thread after every producer's stores were drained
__device__ __forceinline__ void ep_publish(const EpPeers& pv, const int64_t flags_off) {
  for (int p = 0; p < pv.size; ++p)
    st_system(reinterpret_cast<uint32_t*>(pv.base[p] + flags_off) + pv.rank * EP_FLAG_WORDS, pv.epoch);
}
// many-workgroup producers: every workgroup drains its stores and arrives; the last one publishes (and re-arms the counter)
__device__ __forceinline__ void ep_arrive_publish(const EpPeers& pv, const int expected, const int64_t flags_off) {
  wait_stores_acked();
  __syncthreads();
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(pv.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == expected) {
      __hip_atomic_store(pv.done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ep_publish(pv, flags_off);
    }
  }
}
// ... in two levels for grids of many hundreds of workgroups (one hot counter serialises them: 768 adds on one address cost
// the owner's stage 2 several microseconds): `per_tile` workgroups share a tile counter, the last of a tile arrives at the
// launch-wide counter, the last of those publishes
__device__ __forceinline__ void ep_arrive_publish2(const EpPeers& pv, int32_t* tile_counter, const int per_tile, const int tiles, const int64_t flags_off) {
  wait_stores_acked();
  __syncthreads();
  if (threadIdx.x == 0) {
    if (__hip_atomic_fetch_add(tile_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == per_tile) {
      __hip_atomic_store(tile_counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_fetch_add(pv.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == tiles) {
        __hip_atomic_store(pv.done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ep_publish(pv, flags_off);
      }
    }
  }
}
// consumer: lanes 0..n-1 of ONE wave poll flag words 0..n-1 until each has reached `epoch` (bounded by wall clock)
__device__ __forceinline__ void ep_poll(const uint32_t* flags, const int n, const uint32_t epoch, const int64_t timeout_ticks, int32_t* err) {
  const int lane = threadIdx.x & 63;
  const long long t0 = wall_clock64();
  for (;;) {
    uint32_t v = epoch;
    if (lane < n) v = __hip_atomic_load(flags + lane * EP_FLAG_WORDS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (__all((int32_t)(v - epoch) >= 0)) {
      // a producer AHEAD of this exchange: one of the two ranks lost a call (a failed moeinf_ep_moe_forward); the rows read
      // next belong to another exchange.  Not waited for — reported (3), like a timeout (2).
      if (__any((int32_t)(v - epoch) > 0) && lane == 0) atomicExch(err, 3);
      break;
    }
    if (wall_clock64() - t0 > timeout_ticks) {
      if (lane == 0) atomicExch(err, 2);
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");  // later loads stay behind the poll
}

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
  // two elements in one 32-bit word (low half first), and one element from its bit pattern
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) { return f2bf2(lo, hi); }
  __device__ static __forceinline__ void unpack2(uint32_t w, float& lo, float& hi) { lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u); }
  __device__ static __forceinline__ uint16_t from_bits(uint16_t b) { return b; }
  __device__ static __forceinline__ void store_coherent(uint16_t* p, float f) { st_coherent(p, f2bf(f)); }
  __device__ static __forceinline__ void store_system(uint16_t* p, float f) { st_system(p, f2bf(f)); }
  __device__ static __forceinline__ void store4(uint16_t* p, const float f[4]) {
    uint2 v;
    v.x = f2bf2(f[0], f[1]);
    v.y = f2bf2(f[2], f[3]);
    *reinterpret_cast<uint2*>(p) = v;
  }
};
template <>
struct DT<half_t> {  // fp16 storage
  static constexpr int EPV = 8;
  __device__ static __forceinline__ float round(float f) { return h2f(f2h(f)); }
  __device__ static __forceinline__ float load(const half_t* p) { return h2f(*p); }
  __device__ static __forceinline__ void store(half_t* p, float f) { *p = f2h(f); }
  __device__ static __forceinline__ void load4(const half_t* p, float o[4]) {
    const half4_t v = *reinterpret_cast<const half4_t*>(p);
    o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3];
  }
  struct Raw4 { unsigned long long v; };
  template <bool COH>
  __device__ static __forceinline__ Raw4 fetch4(const half_t* p) {
    Raw4 r;
    r.v = COH ? ld_coherent(reinterpret_cast<const unsigned long long*>(p)) : *reinterpret_cast<const unsigned long long*>(p);
    return r;
  }
  __device__ static __forceinline__ void unpack4(const Raw4& r, float o[4]) {
    const half4_t v = __builtin_bit_cast(half4_t, r.v);
    o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3];
  }
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) {
    return (uint32_t)__builtin_bit_cast(uint16_t, f2h(lo)) | ((uint32_t)__builtin_bit_cast(uint16_t, f2h(hi)) << 16);
  }
  __device__ static __forceinline__ void unpack2(uint32_t w, float& lo, float& hi) {
    lo = h2f(__builtin_bit_cast(half_t, (uint16_t)(w & 0xffffu))); hi = h2f(__builtin_bit_cast(half_t, (uint16_t)(w >> 16)));
  }
  __device__ static __forceinline__ half_t from_bits(uint16_t b) { return __builtin_bit_cast(half_t, b); }
  __device__ static __forceinline__ void store_coherent(half_t* p, float f) { st_coherent(reinterpret_cast<uint16_t*>(p), __builtin_bit_cast(uint16_t, f2h(f))); }
  __device__ static __forceinline__ void store_system(half_t* p, float f) { st_system(reinterpret_cast<uint16_t*>(p), __builtin_bit_cast(uint16_t, f2h(f))); }
  __device__ static __forceinline__ void store4(half_t* p, const float f[4]) {
    const half4_t v = {f2h(f[0]), f2h(f[1]), f2h(f[2]), f2h(f[3])};
    *reinterpret_cast<half4_t*>(p) = v;
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
  __device__ static __forceinline__ void store_system(float* p, float f) { st_system(p, f); }
  __device__ static __forceinline__ void store4(float* p, const float f[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
  }
};

// GEMM epilogues: four consecutive output rows r0..r0+3 of ONE token (an accumulator quad of the 16x16 MFMA) -> `orow_p`
// (the token's output row).  The epilogue kind is a compile-time constant (a run-time switch on s.epi per element costs
// more than the arithmetic), the quad leaves as one 8-byte (fp32: 16-byte) store when it is whole and aligned.
template <typename T, int EPI>
__device__ __forceinline__ void epi_quad(const f32x4& a0, const f32x4& a1, const T* bias, int r0, int R, bool aligned, T* orow_p) {
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float x = a0[j];
    if constexpr (EPI == EPI_GATED_SILU) {
      x = DT<T>::round(x);
      const float bb = DT<T>::round(a1[j]);
      const float sl = DT<T>::round(x / (1.0f + expf(-x)));
      x = sl * bb;  // rounded by the store
    } else {
      if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) x = DT<T>::round(DT<T>::round(x) + DT<T>::load(bias + min(r0 + j, R - 1)));
      if constexpr (EPI == EPI_RELU || EPI == EPI_BIAS_RELU) x = fmaxf(DT<T>::round(x), 0.f);
    }
    v[j] = x;
  }
  if (aligned && r0 + 3 < R) {
    DT<T>::store4(orow_p + r0, v);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (r0 + j < R) DT<T>::store(orow_p + r0 + j, v[j]);
  }
}
// f(std::integral_constant<int, EPI>) for the stage's epilogue kind (two matrices = the gated stage, launch_ffn_stage)
template <int NMAT, typename F>
__device__ __forceinline__ void epi_switch(int epi, F&& f) {
  if constexpr (NMAT == 2) {
    f(std::integral_constant<int, EPI_GATED_SILU>{});
  } else {
    if (epi == EPI_NONE) f(std::integral_constant<int, EPI_NONE>{});
    else if (epi == EPI_BIAS) f(std::integral_constant<int, EPI_BIAS>{});
    else if (epi == EPI_RELU) f(std::integral_constant<int, EPI_RELU>{});
    else f(std::integral_constant<int, EPI_BIAS_RELU>{});
  }
}

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
// 16 bytes written by ANOTHER workgroup of this launch (write-through stores + a counter, layer_fused.hip): two 8-byte
// agent-scope loads (a relaxed __hip_atomic_load lowers to an sc1 load up to 8 bytes; the compiler counts them like any load)
__device__ __forceinline__ u32x4 ld16_coherent(const void* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long a = ld_coherent(q), b = ld_coherent(q + 1);
  return u32x4{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
}
__device__ __forceinline__ u32x4 ld16_nt(const void* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
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
__device__ __forceinline__ void mma16<half_t>(f32x4& acc, const u32x4& a, const u32x4& b) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
// 32x32x16 (the compute-bound GEMM, ffn_gemm_big.hip): 16 accumulator floats per lane
typedef float f32x16_ __attribute__((ext_vector_type(16)));
template <typename T>
__device__ __forceinline__ f32x16_ mma32(const u32x4& a, const u32x4& b, const f32x16_& acc);
template <>
__device__ __forceinline__ f32x16_ mma32<uint16_t>(const u32x4& a, const u32x4& b, const f32x16_& acc) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16_ mma32<half_t>(const u32x4& a, const u32x4& b, const f32x16_& acc) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
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
  typename DT<T>::Raw4 rsh = {}, ry[8];
  // (a `has_sh ? a.y_shared : a.y` pointer select here was compiled into a two-entry pointer table in SCRATCH, 24 bytes per
  // thread; a wave-uniform branch around the one load costs nothing)
  if (has_sh) rsh = DT<T>::template fetch4<COH>(reinterpret_cast<const T*>(a.y_shared) + (size_t)(row0 + t) * a.H + h0);
  // absent slots fetch row 0 (ignored below) so the round stays branch-free
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
// COHI / COHO (layer_fused.hip): the activation rows were written / the output rows are read by OTHER workgroups of the same
// launch — agent-scope loads, write-through stores
template <typename T, int NMAT, int NW, int U, int NT, bool COHI = false, bool COHO = false>
__device__ __forceinline__ void ffn_rows_item(const FfnStage& s, const int bx, const char* W, const bool sh, const int cnt, const int off,
                                              float (*red)[NMAT][256], const int xrow_fixed = -1,
                                              const int* in_rows = nullptr, const int* out_rows = nullptr, const EpPeers* pvp = nullptr,
                                              const bool to_peers = false) {
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
  // fused-combine hand-off rows as 16-byte stores: needs all 16 rows of the group inside the matrix and no in/out row lists
  // (also how the peer-store exchange's owner writes its output rows into their home ranks' windows: pv != nullptr)
  // (a pointer that is SELECTED between the kernel-argument struct and nullptr makes the compiler copy the struct to scratch:
  // the caller always passes the address and says separately whether it is used)
  const bool wide_out = NMAT == 1 && NT == 1 && r0 + 16 <= R && (to_peers ? true : (s.fuse_combine == 2 && !out_rows));  // fuse_combine 1: narrow stores (A/B)

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
      // in_rows / out_rows (workgroup-local arrays, indexed by the row's position INSIDE this item): the expert-parallel
      // owner-side kernel derives an expert's rows itself and keeps their ids in LDS
      const int64_t xrow = xrow_fixed >= 0 ? (int64_t)xrow_fixed
                           : in_rows     ? (int64_t)in_rows[min((tile0 + tt) * 16 + n, cnt - 1)]
                                         : (s.row_map ? (int64_t)s.row_map[srow] : (int64_t)srow);
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
            if (tt < ntl) xv[i][tt] = COHI ? ld16_coherent(xr[tt] + (size_t)(kb + i * NW) * EPT) : ld16(xr[tt] + (size_t)(kb + i * NW) * EPT);
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
          const u32x4 x0 = (KBfull * EPT + kq < K) ? (COHI ? ld16_coherent(xr[tt] + (size_t)KBfull * EPT) : ld16(xr[tt] + (size_t)KBfull * EPT)) : z;
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
          } else if (s.epi == EPI_GATED_GELU) {
            const float b = DT<T>::round(s1);
            const float gl = DT<T>::round(0.5f * v * (1.0f + erff(v * 0.70710678118654752f)));
            v = DT<T>::round(gl * b);
          } else {
            if (s.epi == EPI_BIAS || s.epi == EPI_BIAS_RELU)
              v = DT<T>::round(v + DT<T>::load(reinterpret_cast<const T*>(W + s.off_bias) + orow));
            if (s.epi == EPI_RELU || s.epi == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
          }
          const int srow = off + tile * 16 + tn;
          const int drow = out_rows ? out_rows[tile * 16 + tn] : (s.out_map ? s.out_map[srow] : srow);
          if (wide_out) {
            red[0][0][i] = v;  // gathered below into 16-byte stores
          } else if (to_peers) {
            DT<T>::store_system(reinterpret_cast<T*>(ep_peer_ret_row(*pvp, drow, (int64_t)s.ld_out * sizeof(T))) + orow, v);
          } else {
            T* op = reinterpret_cast<T*>(s.out) + (size_t)drow * s.ld_out + orow;
            if (COHO || (NMAT == 1 && NT == 1 && s.fuse_combine)) DT<T>::store_coherent(op, v);
            else DT<T>::store(op, v);
          }
        }
      }
      if constexpr (NMAT == 1 && NT == 1) {
        if (wide_out) {
          // the hand-off rows of the fused combine leave as 16-byte write-through stores: thread t gathers the 16 output rows
          // of token t (sixteen 2-byte sc1 stores were sixteen fabric writes)
          __syncthreads();
          const int tn = tid;
          if (tn < 16 && tile * 16 + tn < cnt) {
            float v16[16];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
              for (int j = 0; j < 4; ++j) v16[qq * 4 + j] = red[0][0][(qq * 16 + tn) * 4 + j];
            const int srow = off + tile * 16 + tn;
            const int drow = out_rows ? out_rows[tile * 16 + tn] : (s.out_map ? s.out_map[srow] : srow);
            T* op = to_peers ? reinterpret_cast<T*>(ep_peer_ret_row(*pvp, drow, (int64_t)s.ld_out * sizeof(T))) + r0
                             : reinterpret_cast<T*>(s.out) + (size_t)drow * s.ld_out + r0;
            if constexpr (sizeof(T) == 2) {
              u32x4 w0, w1;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                w0[j] = (uint32_t)f2bf(v16[2 * j]) | ((uint32_t)f2bf(v16[2 * j + 1]) << 16);
                w1[j] = (uint32_t)f2bf(v16[8 + 2 * j]) | ((uint32_t)f2bf(v16[8 + 2 * j + 1]) << 16);
              }
              if (to_peers) { st16_system(op, w0); st16_system(op + 8, w1); }
              else { st16_coherent(op, w0); st16_coherent(op + 8, w1); }
            } else {
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                u32x4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = __float_as_uint(v16[c * 4 + j]);
                if (to_peers) st16_system(op + c * 4, w); else st16_coherent(op + c * 4, w);
              }
            }
          }
        }
      }
      __syncthreads();
    }
  }
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
__device__ __forceinline__ void load8<half_t>(const half_t* p, float out[8]) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  const f16x8 v = __builtin_bit_cast(f16x8, ld16(p));
#pragma unroll
  for (int j = 0; j < 8; ++j) out[j] = (float)v[j];
}
template <>
__device__ __forceinline__ void load8<float>(const float* p, float out[8]) {
  const u32x4 a = ld16(p), b = ld16(p + 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) { out[j] = __uint_as_float(a[j]); out[4 + j] = __uint_as_float(b[j]); }
}

template <typename XT, typename WT, int TT, bool COHO = false>  // COHO: the logits are read by other workgroups of this launch
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
    if (round_bf16 == 1) f = bf2f(f2bf(f)); else if (round_bf16 == 2) f = h2f(f2h(f));  // (1: bf16, 2: fp16 — the model dtype of Mixtral's gate)
    if (COHO) st_coherent(&logits[(size_t)(t0 + tid) * E + e], f);
    else logits[(size_t)(t0 + tid) * E + e] = f;
  }
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

template <bool COHL = false>  // COHL: the logits were written by other workgroups of this launch (layer_fused.hip)
__device__ __forceinline__ void route_core(const RouteArgs& a, const int t, const int lane, Routed& o) {
  const int E = a.E, K = a.K;
  const float* lg = a.logits + (size_t)t * E;
  float l[4], p[4];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = lane + 64 * j;
    l[j] = (e < E) ? (COHL ? ld_coherent(lg + e) : lg[e]) : -INFINITY;
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
  if (a.kind == 1 && a.v3) {  // DeepSeek-V3 (modeling_deepseek_v3 MoEGate, :478-479): scores = sigmoid(logits), no softmax
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int e = lane + 64 * j; p[j] = (e < E) ? 1.0f / (1.0f + expf(-l[j])) : 0.f; }
  }

  const int xdt = a.x_dtype;  // the model dtype: quantities the reference keeps in it are rounded to it
  int sel[8];
  float val[8];
  int valid[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { sel[k] = -1; val[k] = 0.f; valid[k] = 1; }
  uint32_t taken = 0;

  if (a.kind == 0 /*MIXTRAL*/ || (a.kind == 1 /*DEEPSEEK*/ && a.n_group <= 1 && !a.v3)) {
    for (int k = 0; k < K; ++k) {
      float bv; int bi;
      pick_best(p, taken, lane, E, bv, bi);
      sel[k] = bi; val[k] = bv;
      if (bi >= 0 && (bi & 63) == lane) taken |= 1u << (bi >> 6);
    }
  } else if (a.kind == 1 && a.v3) {  // noaux_tc (modeling_deepseek_v3/modeling_deepseek.py:484-512)
    const int gs = E / a.n_group;
    float sfc[4];  // scores_for_choice = scores + e_score_correction_bias
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int e = lane + 64 * j; sfc[j] = (e < E) ? p[j] + (a.e_bias ? a.e_bias[e] : 0.f) : -INFINITY; }
    // group score = the sum of the group's two best scores_for_choice; lane g (< n_group) ends up holding group g's
    float gscore = -INFINITY;
    for (int g = 0; g < a.n_group; ++g) {
      float s2 = 0.f;
      uint32_t tk = 0;  // (per lane: which of its four columns are taken inside this group's top-2 search)
      for (int r = 0; r < 2; ++r) {
        float bv = -INFINITY; int bi = -1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = lane + 64 * j;
          if (e < E && e / gs == g && !((tk >> j) & 1u) && (bi < 0 || sfc[j] > bv)) { bv = sfc[j]; bi = e; }
        }
        wave_argmax(bv, bi);
        if (bi >= 0) { s2 += bv; if ((bi & 63) == lane) tk |= 1u << (bi >> 6); }
      }
      if (lane == g) gscore = s2;
    }
    uint64_t gmask = 0;
    bool gtaken = false;
    for (int k = 0; k < a.topk_group; ++k) {
      float bv = gscore; int bi = (lane < a.n_group && !gtaken) ? lane : -1;
      wave_argmax(bv, bi);
      if (bi == lane) gtaken = true;
      if (bi >= 0) gmask |= 1ull << bi;
    }
    float pm[4];  // masked_fill(~score_mask, 0.0) of scores_for_choice
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = lane + 64 * j;
      pm[j] = (e < E && ((gmask >> (e / gs)) & 1ull)) ? sfc[j] : 0.f;
    }
    for (int k = 0; k < K; ++k) {
      float bv; int bi;
      pick_best(pm, taken, lane, E, bv, bi);
      sel[k] = bi;
      // the WEIGHT is the un-biased score of the chosen expert (scores.gather, :513)
      float sc = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) if (lane + 64 * j == bi) sc = p[j];
      val[k] = wave_sum(sc);
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
    for (int j = 0; j < 4; ++j) pin[j] = round_model(xdt, p[j]);
    float bv; int bi;
    pick_best(pin, 0u, lane, E, bv, bi);
    sel[0] = bi; val[0] = bv;
  } else {  /*NLLB*/
    float pin[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pin[j] = round_model(xdt, p[j]);
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
    if (a.no_renorm) den = 1.f;  // Grok / Arctic: the softmax probabilities themselves, cast to the model dtype (x / 1.0f is exact)
    for (int k = 0; k < K; ++k) w[k] = round_model(xdt, val[k] / den);
  } else if (a.kind == 1) {
    if (a.v3) {  // V3: normalise (if asked), then ALWAYS the scaling factor (:519-524)
      float den = 1.f;
      if (K > 1 && a.norm_topk_prob) { den = 0.f; for (int k = 0; k < K; ++k) den += val[k]; den += 1e-20f; }
      for (int k = 0; k < K; ++k) w[k] = (K > 1 && a.norm_topk_prob ? val[k] / den : val[k]) * a.scale;
    } else if (K > 1 && a.norm_topk_prob) {
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
    const float eps = xdt == DT_BF16 ? 0.0078125f : (xdt == DT_F16 ? 0.0009765625f : 1.1920928955078125e-07f);  // torch.finfo(dtype).eps
    float den = round_model(xdt, val[0] + val[1]);
    den = fmaxf(den, eps);
    w[0] = val[0] / den; w[1] = val[1] / den;
    w[0] = round_model(xdt, w[0]); w[1] = round_model(xdt, w[1]);
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
// round_p_dtype: Switch's top-1 runs on probabilities cast to the model's dtype (route_core, kind 2)
template <bool COHL = false>
__device__ __forceinline__ uint64_t route_set_lean(const float* __restrict__ logits, const int E, const int K, const int lane, const int round_p_dtype = 1 /*DT_F32: none*/) {
  const bool in = lane < E;
  const float l = in ? (COHL ? ld_coherent(logits + lane) : logits[lane]) : -INFINITY;
  const float m = wave_max(l);
  float p = in ? expf(l - m) : 0.f;
  const float ssum = wave_sum(p);
  p = p / ssum;
  p = round_model(round_p_dtype, p);
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
// Expert-parallel exchange, sender side, decode-sized forwards (<= 64 (token,k) pairs): the send rows of the
// all-to-all written by the SAME workgroup that just routed and indexed the tokens (no pack launch).  Row (d, pos) of
// `send` = the pos-th pair (stable, pair order) whose expert lives on rank d = e % ep_size; its 16-byte tail carries the
// expert id, tails of unused rows are -1.  Call with every thread of the workgroup after a __syncthreads() that made
// topk_idx / pair_valid visible.  Same layout, bit for bit, as ep_pack_small_kernel (ep_kernels.hip).
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void ep_pack_block(const EpPackArgs& a, const EpPeers& pv, const int32_t* pair_valid, const int n_pairs, int32_t* send_counts,
                                              int* s_row /*LDS [64]*/) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int nrows = a.ep_size * a.cap_rows;
  const int64_t row_bytes = a.ld_send * (int64_t)sizeof(T);
  const bool peer = pv.on != 0;  // block-uniform: rows go straight into the destination ranks' windows
  if (tid < 64) {
    const int lane = tid;
    int key = -1;
    if (lane < n_pairs) {
      const int e = a.topk_idx[lane];
      if (e >= 0 && (!pair_valid || pair_valid[lane])) key = e % a.ep_size;
    }
    int row = -1;
    for (int d = 0; d < a.ep_size; ++d) {
      const uint64_t mine = __ballot(key == d);
      if (key == d) row = d * a.cap_rows + __popcll(mine & lanes_below(lane));  // < cap_rows: the host checks cap_rows >= tokens * min(K, experts per rank)
      if (send_counts && lane == 0) send_counts[d] = __popcll(mine);
    }
    if (lane < n_pairs) a.pair_pos[lane] = row;
    s_row[lane] = row;
  }
  __syncthreads();
  // tails: the expert id of the pair that occupies the row, -1 for an unused row (one store per tail)
  for (int r = tid; r < nrows; r += nthr) {
    int id = -1;
    for (int p = 0; p < n_pairs; ++p)
      if (s_row[p] == r) id = a.topk_idx[p];
    char* dst = peer ? ep_peer_recv_row(pv, r, row_bytes) : reinterpret_cast<char*>(a.send) + (size_t)r * row_bytes;
    int32_t* tail = reinterpret_cast<int32_t*>(dst + (size_t)a.H * sizeof(T));
    if (peer) st_system(tail, (int32_t)id); else *tail = id;
  }
  constexpr int EPV = DT<T>::EPV;
  const int cpr = a.H / EPV;  // 16-byte chunks per row
  for (int i = tid; i < n_pairs * cpr; i += nthr) {
    const int p = i / cpr, c = i - p * cpr;
    const int row = s_row[p];
    if (row < 0) continue;
    const u32x4 v = ld16(reinterpret_cast<const T*>(a.x) + (size_t)(p / a.K) * a.H + c * EPV);
    if (peer) st16_system(ep_peer_recv_row(pv, row, row_bytes) + (size_t)c * 16, v);
    else *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.send) + (size_t)row * row_bytes + (size_t)c * 16) = v;
  }
  if (peer) {  // drain this workgroup's stores, then tell every destination that exchange `epoch`'s rows from this rank are there
    wait_stores_acked();
    __syncthreads();
    if (tid == 0) ep_publish(pv, 0);
  }
}
__device__ __forceinline__ void ep_pack_block_dt(const EpFuse& f, const int n_pairs, int* s_row) {
  if (f.a.dtype != DT_F32) ep_pack_block<uint16_t>(f.a, f.peers, f.pair_valid, n_pairs, f.send_counts, s_row);  // (a row copy: any 2-byte dtype)
  else ep_pack_block<float>(f.a, f.peers, f.pair_valid, n_pairs, f.send_counts, s_row);
}

}  // namespace moeinf
