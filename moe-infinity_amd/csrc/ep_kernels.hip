// ep_kernels.hip — expert-parallel exchange: pack kernels of the sender side and the
// self-indexing FFN stage of the owner side.  The reference serves multi-GPU from ONE process with implicit P2P
// `tensor.to(device)` copies (core/parallel/expert_dispatcher.cpp:284,405); here one process per GPU exchanges routed
// rows with one all-to-all each way and these kernels sit on either side of it.
#include "kdev.h"

#include <string.h>

namespace moeinf {

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


// ------------------------------------------------------------------------------------------------
// ffn_ep: owner-side FFN stage that indexes for itself (decode-sized exchange, <= 64 received row slots, every owned
// expert of the layer resident).  grid = (ceil(R/16) [+1 meta block in stage 1], max_active); workgroup (rg, u):
//   wave 0 reads the expert id in the tail of every received row (one load per lane) and — in the same round — the blob
//   pointer of every owned expert (lane j holds wptr[j*ep_size + ep_rank]); a scalar loop over the distinct ids builds
//   the bit mask of owned experts present; "its" expert is the u-th smallest, its rows = one ballot, its expert-sorted
//   offset = popcount of the rows with smaller ids; the row ids go to LDS and the blob pointer is a v_readlane away.
// Then the ordinary weight stream (ffn_rows_item): stage 1 gathers the rows from the receive buffer, stage 2 scatters
// its output rows straight to their ARRIVAL positions in the reply buffer.  Every workgroup of both launches derives
// the same sets from the same tails, so the two stages agree without any index launch between them.
// The stage-1 meta block (blockIdx.x == gridDim.x - 1, u == 0) writes the pinned routing mirror the host applies to its
// hit counters lazily.
// ------------------------------------------------------------------------------------------------
template <typename T, int NMAT, int NW, int U>
__global__ __launch_bounds__(NW * 64) void ffn_ep_kernel(FfnStage s, EpOwnArgs o) {
  __shared__ float red[NW][NMAT][256];
  __shared__ int s_rows[64];
  __shared__ unsigned long long s_w;
  __shared__ int s_cnt, s_off;
  const int u = blockIdx.y;
  const bool meta = o.stage == 1 && o.mirror && blockIdx.x == gridDim.x - 1;
  if (meta && u != 0) return;
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    int key = -1;
    if (lane < o.nrows)
      key = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(o.recv) + ((size_t)lane * o.ld_recv + o.H) * sizeof(T));
    const int own_e = lane * o.ep_size + o.ep_rank;  // owned expert with local id `lane`
    uint64_t wp = 0;
    if (own_e < s.E) wp = s.wptr[own_e];
    const bool valid = key >= 0 && key < s.E;
    uint64_t todo = __ballot(valid), mask = 0;
    while (todo) {  // wave-uniform: one iteration per distinct expert id
      const int leader = __ffsll((unsigned long long)todo) - 1;
      const int k = __builtin_amdgcn_readlane(key, leader);
      const uint64_t same = __ballot(valid && key == k);
      mask |= 1ull << (k / o.ep_size);
      todo &= ~same;
    }
    if (meta) {  // the routing mirror: {n_active, counts[E+1], active[E+1]} — only active experts' counts (the host zeroed the rest)
      const int E1 = s.E + 1;
      int na = 0;
      for (uint64_t m = mask; m; m &= m - 1, ++na) {
        const int e = (int)__builtin_ctzll(m) * o.ep_size + o.ep_rank;
        const int c = __popcll(__ballot(valid && key == e));
        if (lane == 0) { o.mirror[1 + e] = c; o.mirror[1 + E1 + na] = e; }
      }
      if (lane == 0) o.mirror[0] = na;
    } else {
      uint64_t m = mask;
      for (int i = 0; i < u; ++i) m &= m - 1;  // drop the u smallest ids
      const int j = m ? (int)__builtin_ctzll(m) : -1;
      const int e = j >= 0 ? j * o.ep_size + o.ep_rank : -1;
      const uint64_t rows = __ballot(valid && key == e);
      const int off = __popcll(__ballot(valid && key < e));
      if (valid && key == e) s_rows[__popcll(rows & lanes_below(lane))] = lane;
      uint64_t wsel = 0;
      if (j >= 0) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wp, j);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(wp >> 32), j);
        wsel = ((uint64_t)hi << 32) | lo;
      }
      if (lane == 0) { s_w = wsel; s_cnt = j >= 0 ? __popcll(rows) : 0; s_off = off; }
    }
  }
  if (meta) return;
  __syncthreads();
  const int cnt = s_cnt;
  if (cnt == 0) return;  // fewer than u+1 experts present (block-uniform)
  const int rg = blockIdx.x;
  if (rg * 16 >= s.R) return;
  const char* W = reinterpret_cast<const char*>(s_w);
  if (W == nullptr) {  // never on the sync-free path
    if (threadIdx.x == 0 && rg == 0) atomicExch(s.miss_flag, 1);
    return;
  }
  ffn_rows_item<T, NMAT, NW, U, 1>(s, rg, W, false, cnt, s_off, red, -1, o.stage == 1 ? s_rows : nullptr, o.stage == 2 ? s_rows : nullptr);
}

hipError_t launch_ffn_ep_stage(const FfnStage& s, const EpOwnArgs& o, hipStream_t st) {
  const dim3 grid((s.R + 15) / 16 + ((o.stage == 1 && o.mirror) ? 1 : 0), o.max_active);
  const bool gated = (s.epi == EPI_GATED_SILU);
  const size_t kbytes = (size_t)s.K * (s.dtype == DT_BF16 ? 2 : 4);
  const bool nw8 = kbytes >= 16384;  // long reductions: 8 waves per workgroup (as launch_ffn_stage)
#define EPK(TT, NM, NWV) hipLaunchKernelGGL((ffn_ep_kernel<TT, NM, NWV, 4>), grid, dim3(NWV * 64), 0, st, s, o)
  if (s.dtype == DT_BF16) {
    if (gated) { if (nw8) EPK(uint16_t, 2, 8); else EPK(uint16_t, 2, 4); }
    else       { if (nw8) EPK(uint16_t, 1, 8); else EPK(uint16_t, 1, 4); }
  } else {
    if (gated) { if (nw8) EPK(float, 2, 8); else EPK(float, 2, 4); }
    else       { if (nw8) EPK(float, 1, 8); else EPK(float, 1, 4); }
  }
#undef EPK
  return hipGetLastError();
}

}  // namespace moeinf
