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
// one send row: tail (expert id, -1 = unused) + activations, into the send buffer or (peer-store exchange) straight into
// the destination rank's window; in the peer form the last workgroup of the launch publishes the exchange
template <typename T>
__device__ __forceinline__ void ep_write_row(const EpPackArgs& a, const EpPeers& pv, const int row, const int pair) {
  const bool peer = pv.on != 0;
  const int64_t row_bytes = a.ld_send * (int64_t)sizeof(T);
  T* dst = peer ? reinterpret_cast<T*>(ep_peer_recv_row(pv, row, row_bytes)) : reinterpret_cast<T*>(a.send) + (size_t)row * a.ld_send;
  if (threadIdx.x == 0) {
    const int32_t id = pair >= 0 ? a.topk_idx[pair] : -1;
    int32_t* tail = reinterpret_cast<int32_t*>(dst + a.H);
    if (peer) st_system(tail, id); else tail[0] = id;
    if (pair >= 0) a.pair_pos[pair] = row;
  }
  if (pair >= 0) {
    const T* src = reinterpret_cast<const T*>(a.x) + (size_t)(pair / a.K) * a.H;
    constexpr int EPV = DT<T>::EPV;
    for (int h = threadIdx.x * EPV; h < a.H; h += 256 * EPV) {
      const u32x4 v = ld16(src + h);
      if (peer) st16_system(dst + h, v); else *reinterpret_cast<u32x4*>(dst + h) = v;
    }
  }
  if (peer) ep_arrive_publish(pv, (int)gridDim.x, 0);
}
template <typename T>
__global__ __launch_bounds__(256) void ep_pack_kernel(EpPackArgs a, EpPeers pv) {
  const int row = blockIdx.x;
  const int d = row / a.cap_rows, pos = row % a.cap_rows;
  const int cnt = a.counts[d];
  const int pair = pos < cnt ? a.slot_pair[a.offsets[d] + pos] : -1;
  ep_write_row<T>(a, pv, row, pair);
}
// <= 64 (token,k) pairs (decode): destination keys, stable ranks and the row copy in ONE launch.  Every block
// (= one send row (d, pos)) re-derives "which pair is the pos-th one bound for rank d" with two ballots over the
// pairs — the same stable order the dest-key + dispatch_index + pack sequence produces.
template <typename T>
__global__ __launch_bounds__(256) void ep_pack_small_kernel(EpPackArgs a, const int32_t* pair_valid, int n_pairs, int32_t* send_counts, EpPeers pv) {
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
  if (threadIdx.x == 0 && pos == 0 && send_counts) send_counts[d] = s_cnt;
  ep_write_row<T>(a, pv, row, s_pair);
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
  if (a.dtype != DT_F32) hipLaunchKernelGGL(ep_pack_compact_kernel<uint16_t>, dim3(n_pairs), dim3(256), 0, st, a, n_pairs);
  else hipLaunchKernelGGL(ep_pack_compact_kernel<float>, dim3(n_pairs), dim3(256), 0, st, a, n_pairs);
  return hipGetLastError();
}

static EpPeers no_peers() {
  EpPeers p;
  memset(&p, 0, sizeof p);
  return p;
}
hipError_t launch_ep_pack_small(const EpPackArgs& a, const int32_t* pair_valid, int n_pairs, int32_t* send_counts, hipStream_t st, const EpPeers* peers) {
  const EpPeers pv = peers ? *peers : no_peers();
  if (a.dtype != DT_F32) hipLaunchKernelGGL(ep_pack_small_kernel<uint16_t>, dim3(a.ep_size * a.cap_rows), dim3(256), 0, st, a, pair_valid, n_pairs, send_counts, pv);
  else hipLaunchKernelGGL(ep_pack_small_kernel<float>, dim3(a.ep_size * a.cap_rows), dim3(256), 0, st, a, pair_valid, n_pairs, send_counts, pv);
  return hipGetLastError();
}

hipError_t launch_ep_pack(const EpPackArgs& a, hipStream_t st, const EpPeers* peers) {
  const EpPeers pv = peers ? *peers : no_peers();
  if (a.dtype != DT_F32) hipLaunchKernelGGL(ep_pack_kernel<uint16_t>, dim3(a.ep_size * a.cap_rows), dim3(256), 0, st, a, pv);
  else hipLaunchKernelGGL(ep_pack_kernel<float>, dim3(a.ep_size * a.cap_rows), dim3(256), 0, st, a, pv);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// peer-store exchange: the pieces that are kernels of their own
// ------------------------------------------------------------------------------------------------
// one wave polls a flag set (ranks that SHARE a GPU must not spin inside wide kernels: the peer they wait for needs CUs)
__global__ __launch_bounds__(64) void ep_wait_kernel(EpWait w) { ep_poll(w.flags, w.n, w.epoch, w.timeout_ticks, w.err); }
hipError_t launch_ep_wait(const EpWait& w, hipStream_t st) {
  hipLaunchKernelGGL(ep_wait_kernel, dim3(1), dim3(64), 0, st, w);
  return hipGetLastError();
}
// owner side, generic path: y row r (arrival order) -> its home rank's return region; grid = ep_size*cap_rows rows
template <typename T>
__global__ __launch_bounds__(256) void ep_push_kernel(const void* y, const void* recv, int64_t ld_recv, int H, EpPeers pv) {
  const int row = blockIdx.x;
  const int32_t id = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(recv) + ((size_t)row * ld_recv + H) * sizeof(T));
  if (id >= 0) {
    const T* src = reinterpret_cast<const T*>(y) + (size_t)row * H;
    T* dst = reinterpret_cast<T*>(ep_peer_ret_row(pv, row, (int64_t)H * sizeof(T)));
    constexpr int EPV = DT<T>::EPV;
    for (int h = threadIdx.x * EPV; h < H; h += 256 * EPV) st16_system(dst + h, ld16(src + h));
  }
  ep_arrive_publish(pv, (int)gridDim.x, EP_RET_FLAGS_OFF);
}
hipError_t launch_ep_push(const void* y, const void* recv, int64_t ld_recv, int H, int dtype, const EpPeers& peers, hipStream_t st) {
  const dim3 grid(peers.size * peers.cap_rows);
  if (dtype != DT_F32) hipLaunchKernelGGL(ep_push_kernel<uint16_t>, grid, dim3(256), 0, st, y, recv, ld_recv, H, peers);
  else hipLaunchKernelGGL(ep_push_kernel<float>, grid, dim3(256), 0, st, y, recv, ld_recv, H, peers);
  return hipGetLastError();
}
// transport self-test (moeinf_ep_peer_selftest): tag of (writer w, reader r, word i, region g)
__device__ __forceinline__ uint32_t ep_tag(int w, int r, int i, int g) { return 0x5e000000u + ((uint32_t)g << 20) + ((uint32_t)w << 16) + ((uint32_t)r << 12) + (uint32_t)(i & 0xfff); }
__global__ __launch_bounds__(256) void ep_selftest_send_kernel(EpPeers pv, int words, int64_t recv_seg_bytes, int64_t ret_seg_bytes) {
  for (int p = 0; p < pv.size; ++p) {
    uint32_t* r0 = reinterpret_cast<uint32_t*>(pv.base[p] + pv.recv_off + pv.rank * recv_seg_bytes);
    uint32_t* r1 = reinterpret_cast<uint32_t*>(pv.base[p] + pv.ret_off + pv.rank * ret_seg_bytes);
    for (int i = threadIdx.x; i < words; i += blockDim.x) {
      st_system(r0 + i, ep_tag(pv.rank, p, i, 0));
      st_system(r1 + i, ep_tag(pv.rank, p, i, 1));
    }
  }
  wait_stores_acked();
  __syncthreads();
  if (threadIdx.x == 0) { ep_publish(pv, 0); ep_publish(pv, EP_RET_FLAGS_OFF); }
}
__global__ __launch_bounds__(256) void ep_selftest_check_kernel(EpPeers pv, int words, int64_t recv_seg_bytes, int64_t ret_seg_bytes, int32_t* ok) {
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  if (threadIdx.x < 64) {
    ep_poll(reinterpret_cast<const uint32_t*>(pv.base[pv.rank]), pv.size, pv.epoch, pv.timeout_ticks, pv.err);
    ep_poll(reinterpret_cast<const uint32_t*>(pv.base[pv.rank] + EP_RET_FLAGS_OFF), pv.size, pv.epoch, pv.timeout_ticks, pv.err);
  }
  __syncthreads();
  for (int p = 0; p < pv.size; ++p) {
    const uint32_t* r0 = reinterpret_cast<const uint32_t*>(pv.base[pv.rank] + pv.recv_off + p * recv_seg_bytes);
    const uint32_t* r1 = reinterpret_cast<const uint32_t*>(pv.base[pv.rank] + pv.ret_off + p * ret_seg_bytes);
    for (int i = threadIdx.x; i < words; i += blockDim.x)  // PLAIN loads, as the consumers of the exchange use
      if (r0[i] != ep_tag(p, pv.rank, i, 0) || r1[i] != ep_tag(p, pv.rank, i, 1)) atomicAdd(&bad, 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) *ok = (bad == 0 && *pv.err == 0) ? 1 : 0;
}
hipError_t launch_ep_selftest_send(const EpPeers& peers, int words, hipStream_t st) {
  hipLaunchKernelGGL(ep_selftest_send_kernel, dim3(1), dim3(256), 0, st, peers, words, peers.recv_row_bytes * peers.cap_rows, peers.ret_row_bytes * peers.cap_rows);
  return hipGetLastError();
}
hipError_t launch_ep_selftest_check(const EpPeers& peers, int words, int32_t* ok_dev, hipStream_t st) {
  hipLaunchKernelGGL(ep_selftest_check_kernel, dim3(1), dim3(256), 0, st, peers, words, peers.recv_row_bytes * peers.cap_rows, peers.ret_row_bytes * peers.cap_rows, ok_dev);
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
  __shared__ int s_cnt, s_off, s_present;
  const int u = blockIdx.y;
  const bool meta = o.stage == 1 && o.mirror && blockIdx.x == gridDim.x - 1;
  if (meta && u != 0) return;
  const bool peer = o.peers.on != 0;  // peer-store exchange: recv is this rank's window, written by the other ranks' kernels
  const bool from_rec = o.stage == 2 && o.rec != nullptr;  // stage 2 reads what stage 1 derived
  if (from_rec) {
    if (threadIdx.x < 64) {
      const EpOwnArgs::Rec* rc = o.rec + u;
      const int c = rc->cnt;
      if (threadIdx.x < c) s_rows[threadIdx.x] = rc->rows[threadIdx.x];
      if (threadIdx.x == 0) { s_w = rc->w; s_cnt = c; s_off = rc->off; s_present = rc->present; }
    }
  } else
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    // the rows of exchange `epoch` must have landed before their tails are read (stage 2 runs behind stage 1: they have)
    if (peer && o.peers.poll && o.stage == 1)
      ep_poll(reinterpret_cast<const uint32_t*>(o.peers.base[o.peers.rank]), o.peers.size, o.peers.epoch, o.peers.timeout_ticks, o.peers.err);
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
      if (lane == 0) { s_w = wsel; s_cnt = j >= 0 ? __popcll(rows) : 0; s_off = off; s_present = __popcll(mask); }
      if (o.stage == 1 && o.rec && blockIdx.x == 0) {  // one workgroup per expert slot leaves the record for stage 2
        EpOwnArgs::Rec* rc = o.rec + u;
        if (valid && key == e) rc->rows[__popcll(rows & lanes_below(lane))] = lane;
        if (lane == 0) { rc->w = wsel; rc->cnt = j >= 0 ? __popcll(rows) : 0; rc->off = off; rc->present = __popcll(mask); }
      }
    }
  }
  if (meta) return;
  __syncthreads();
  const int cnt = s_cnt;
  const int rg = blockIdx.x;
  const bool push = peer && o.stage == 2;  // stage 2 of the peer-store exchange: output rows go straight to their home ranks
  if (cnt == 0) {  // fewer than u+1 experts present (block-uniform)
    // no row arrived at all: nobody streams, workgroup (0, 0) alone tells the peers that this owner has nothing for them
    if (push && s_present == 0 && u == 0 && rg == 0 && threadIdx.x == 0) ep_publish(o.peers, EP_RET_FLAGS_OFF);
    return;
  }
  if (rg * 16 >= s.R) return;  // (the grid has exactly ceil(R/16) row groups: only the meta block gets here)
  const char* W = reinterpret_cast<const char*>(s_w);
  if (W == nullptr) {  // never on the sync-free path
    if (threadIdx.x == 0 && rg == 0) atomicExch(s.miss_flag, 1);
  } else {
    ffn_rows_item<T, NMAT, NW, U, 1>(s, rg, W, false, cnt, s_off, red, -1, o.stage == 1 ? s_rows : nullptr, o.stage == 2 ? s_rows : nullptr,
                                      &o.peers, push);
  }
  // every workgroup that owns (expert present, row group) arrives; the last one publishes this owner's outputs
  if (push) {
    if (o.tile_done) ep_arrive_publish2(o.peers, o.tile_done + rg, s_present, (int)gridDim.x, EP_RET_FLAGS_OFF);
    else ep_arrive_publish(o.peers, s_present * (int)gridDim.x, EP_RET_FLAGS_OFF);
  }
}

hipError_t launch_ffn_ep_stage(const FfnStage& s, const EpOwnArgs& o, hipStream_t st) {
  const dim3 grid((s.R + 15) / 16 + ((o.stage == 1 && o.mirror) ? 1 : 0), o.max_active);
  const bool gated = (s.epi == EPI_GATED_SILU);
  const size_t kbytes = (size_t)s.K * dt_bytes(s.dtype);
  const bool nw8 = kbytes >= 16384;  // long reductions: 8 waves per workgroup (as launch_ffn_stage)
#define EPK(TT, NM, NWV) hipLaunchKernelGGL((ffn_ep_kernel<TT, NM, NWV, 4>), grid, dim3(NWV * 64), 0, st, s, o)
  if (s.dtype == DT_BF16) {
    if (gated) { if (nw8) EPK(uint16_t, 2, 8); else EPK(uint16_t, 2, 4); }
    else       { if (nw8) EPK(uint16_t, 1, 8); else EPK(uint16_t, 1, 4); }
  } else if (s.dtype == DT_F16) {
    if (gated) { if (nw8) EPK(half_t, 2, 8); else EPK(half_t, 2, 4); }
    else       { if (nw8) EPK(half_t, 1, 8); else EPK(half_t, 1, 4); }
  } else {
    if (gated) { if (nw8) EPK(float, 2, 8); else EPK(float, 2, 4); }
    else       { if (nw8) EPK(float, 1, 8); else EPK(float, 1, 4); }
  }
#undef EPK
  return hipGetLastError();
}

}  // namespace moeinf
