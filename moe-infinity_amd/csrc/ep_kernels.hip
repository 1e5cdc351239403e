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
  const bool gated = (s.epi == EPI_GATED_SILU || s.epi == EPI_GATED_GELU);
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


// ------------------------------------------------------------------------------------------------
// Batch-1 decode over the peer-store exchange, BROADCAST form (kernels.h: EpBcastArgs).
// ------------------------------------------------------------------------------------------------
// Block 0's work (every thread of a workgroup of >= 64 threads): (a) this rank's token row and its E gate logits go to
// EVERY rank (row 0 of this rank's segment of the destination's recv region; the logits into its broadcast region), drained,
// then published in the destinations' recv flags; (b) the home token's routing — top-k ids, weights, combine order (the
// generic router, as the meta block of ffn1_selfroute) and, for the combine, the ret row of every pair: the owner e % G puts
// the output for the token's j-th chosen expert among those it owns (ascending id) at position j of its segment.
template <typename T>
__device__ __forceinline__ void epb_broadcast_and_route_home(const RouteArgs& r, const EpBcastArgs& b) {
  const EpPeers& pv = b.peers;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
  const int64_t row_bytes = pv.recv_row_bytes;
  constexpr int EPV = DT<T>::EPV;
  const int cpr = r.H / EPV;
  for (int i = tid; i < pv.size * cpr; i += nthr) {
    const int p = i / cpr, c = i - p * cpr;
    const u32x4 v = ld16(reinterpret_cast<const T*>(b.x) + (size_t)c * EPV);
    st16_system(ep_peer_recv_row(pv, p * pv.cap_rows, row_bytes) + (size_t)c * 16, v);
  }
  for (int i = tid; i < pv.size * r.E; i += nthr) {
    const int p = i / r.E, e = i - p * r.E;
    st_system(reinterpret_cast<float*>(pv.base[p] + pv.bcast_off + (int64_t)pv.rank * pv.bcast_stride) + e, r.logits[e]);
  }
  wait_stores_acked();
  __syncthreads();
  if (tid == 0) ep_publish(pv, 0);
  if (tid < 64) {
    Routed o;
    route_core(r, 0, lane, o);
    int my_sel, rank;
    float my_w;
    route_store(r, 0, lane, o, &my_sel, &my_w, &rank);
    if (lane < r.K) {
      int pos = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < r.K && o.sel[j] >= 0 && o.sel[j] < my_sel && (o.sel[j] % pv.size) == (my_sel % pv.size)) ++pos;
      b.pair_pos[lane] = my_sel >= 0 ? (my_sel % pv.size) * pv.cap_rows + pos : -1;
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void ep_bcast_kernel(RouteArgs r, EpBcastArgs b) { epb_broadcast_and_route_home<T>(r, b); }
hipError_t launch_ep_bcast(const RouteArgs& r, const EpBcastArgs& b, hipStream_t st) {
  if (r.x_dtype == DT_F16) hipLaunchKernelGGL(ep_bcast_kernel<half_t>, dim3(1), dim3(256), 0, st, r, b);
  else hipLaunchKernelGGL(ep_bcast_kernel<uint16_t>, dim3(1), dim3(256), 0, st, r, b);
  return hipGetLastError();
}

// One wave: wait for every rank's broadcast, then the chosen-expert SET of every rank's token restricted to the experts this
// rank owns (bit e of own[s]); G <= EP_MAX_PEERS.  Every workgroup of every owner derives the same sets from the same logits
// with the same instructions (route_set_lean: the arithmetic of the generic router on this path).
__device__ __forceinline__ void epb_owned_sets(const RouteArgs& r, const EpPeers& pv, const int lane, const bool poll, uint64_t own[EP_MAX_PEERS]) {
  if (poll) ep_poll(reinterpret_cast<const uint32_t*>(pv.base[pv.rank]), pv.size, pv.epoch, pv.timeout_ticks, pv.err);
  uint64_t mine = 0;  // experts e with e % G == rank
  for (int e = pv.rank; e < r.E; e += pv.size) mine |= 1ull << e;
  for (int s = 0; s < EP_MAX_PEERS; ++s) {
    own[s] = 0;
    if (s < pv.size) {  // wave-uniform
      const float* lg = reinterpret_cast<const float*>(pv.base[pv.rank] + pv.bcast_off + (int64_t)s * pv.bcast_stride);
      own[s] = route_set_lean(lg, r.E, r.K, lane) & mine;
    }
  }
}

// FFN stage 1 of the broadcast form.  grid = 1 (block 0) + n_sh2 (the hidden shared expert's stage 2 for the home token, as in
// ffn1_selfroute) + max_active * n_rg (+ 1 mirror block).  Workgroup (u, rg): the u-th smallest owned expert that any token
// chose; its rows = the tokens (source ranks) that chose it; x rows are row 0 of the sources' segments of the recv region.
template <typename T, int NW, int U>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_num_sgpr(96))) void ffn_epb_kernel(RouteArgs r, FfnStage s, FfnStage sh2, EpBcastArgs b,
                                                                                                EpOwnArgs::Rec* rec, int n_rg, int n_sh2, int max_active, int with_bcast) {
  __shared__ float red[NW][2][256];
  __shared__ int s_in[EP_MAX_PEERS], s_out[EP_MAX_PEERS];
  __shared__ unsigned long long s_w;
  __shared__ int s_cnt, s_off;
  const EpPeers& pv = b.peers;
  const int lane = threadIdx.x & 63;
  int blk = (int)blockIdx.x - 1;
  if (blk < 0) {  // block 0: dispatched first, waits for nothing
    if (with_bcast) epb_broadcast_and_route_home<T>(r, b);
    return;
  }
  if (blk < n_sh2) {  // shared expert, stage 2, home token (h_shared was written by the gate launch)
    const char* Wsh = reinterpret_cast<const char*>(sh2.wptr[sh2.E]);
    ffn_rows_item<T, 1, NW, U, 1>(sh2, blk, Wsh, true, 1, 0, reinterpret_cast<float (*)[1][256]>(&red[0][0][0]));
    return;
  }
  blk -= n_sh2;
  const bool mirror_blk = blk >= max_active * n_rg;
  const int u = mirror_blk ? 0 : blk / n_rg, rg = mirror_blk ? 0 : blk - u * n_rg;
  if (threadIdx.x < 64) {
    uint64_t wp = 0;
    if (lane < r.E) wp = s.wptr[lane];
    uint64_t own[EP_MAX_PEERS];
    epb_owned_sets(r, pv, lane, pv.poll != 0, own);
    uint64_t present = 0;
#pragma unroll
    for (int q = 0; q < EP_MAX_PEERS; ++q) present |= own[q];
    if (mirror_blk) {  // the routing mirror: {n_active, counts[E+1], active[E+1]} (the host zeroed the counts)
      const int E1 = s.E + 1;
      int na = 0;
      for (uint64_t m = present; m; m &= m - 1, ++na) {
        const int e = (int)__builtin_ctzll(m);
        int c = 0;
#pragma unroll
        for (int q = 0; q < EP_MAX_PEERS; ++q) c += (int)((own[q] >> e) & 1ull);
        if (lane == 0) { b.mirror[1 + e] = c; b.mirror[1 + E1 + na] = e; }
      }
      if (lane == 0) b.mirror[0] = na;
    } else {
      uint64_t m = present;
      for (int i = 0; i < u; ++i) m &= m - 1;  // drop the u smallest ids
      const int e = m ? (int)__builtin_ctzll(m) : -1;
      int cnt = 0, off = 0;
      if (e >= 0) {
        const uint64_t below = (1ull << e) - 1ull;
#pragma unroll
        for (int q = 0; q < EP_MAX_PEERS; ++q) {
          off += __popcll(own[q] & below & present);  // rows of the experts in front of this one (expert-sorted h rows)
          if ((own[q] >> e) & 1ull) {
            if (lane == 0) { s_in[cnt] = q * pv.cap_rows; s_out[cnt] = q * pv.cap_rows + __popcll(own[q] & below); }
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
      if (lane == 0) {
        s_w = wsel; s_cnt = cnt; s_off = off;
        if (rg == 0) {  // one workgroup per expert slot leaves the record for stage 2 (ffn_ep_kernel, from_rec)
          EpOwnArgs::Rec* rc = rec + u;
          rc->w = wsel; rc->cnt = cnt; rc->off = off; rc->present = __popcll(present);
          for (int i = 0; i < cnt; ++i) rc->rows[i] = s_out[i];
        }
      }
    }
  }
  if (mirror_blk) return;
  __syncthreads();
  const int cnt = s_cnt;
  if (cnt == 0) return;
  const char* W = reinterpret_cast<const char*>(s_w);
  if (W == nullptr) {  // never on the sync-free path
    if (threadIdx.x == 0 && rg == 0) atomicExch(s.miss_flag, 1);
    return;
  }
  ffn_rows_item<T, 2, NW, U, 1>(s, rg, W, false, cnt, s_off, red, -1, s_in, nullptr);
}

hipError_t launch_ffn_epb_stage1(const RouteArgs& r, const FfnStage& s1, const FfnStage* sh2, const EpBcastArgs& b, EpOwnArgs::Rec* rec,
                                 int max_active, int with_bcast, hipStream_t st) {
  const int n_rg = (s1.R + 15) / 16;
  const int n_sh2 = sh2 ? (sh2->R_sh + 15) / 16 : 0;
  const dim3 grid(1 + n_sh2 + max_active * n_rg + (b.mirror ? 1 : 0));
  // as launch_ffn1_selfroute: multi-round grids (Mixtral) stream best with four workgroups per CU and four tiles per batch
  const size_t dyn = grid.x > 4 * 256 ? 30 * 1024 : 0;
#define EPB(TT, UU) hipLaunchKernelGGL((ffn_epb_kernel<TT, 4, UU>), grid, dim3(256), dyn, st, r, s1, sh2 ? *sh2 : s1, b, rec, n_rg, n_sh2, max_active, with_bcast)
  if (s1.dtype == DT_F16) { if (grid.x > 4 * 256) EPB(half_t, 4); else EPB(half_t, 8); }
  else { if (grid.x > 4 * 256) EPB(uint16_t, 4); else EPB(uint16_t, 8); }
#undef EPB
  return hipGetLastError();
}

// slow path of the owner (kernels.h: launch_ep_bcast_unpack): one workgroup
template <typename T>
__global__ __launch_bounds__(256) void ep_bcast_unpack_kernel(RouteArgs r, EpBcastArgs b, void* recv_like, int64_t ld) {
  __shared__ unsigned long long s_own[EP_MAX_PEERS];
  const EpPeers& pv = b.peers;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid < 64) {
    uint64_t own[EP_MAX_PEERS];
    epb_owned_sets(r, pv, lane, true, own);  // (behind a wait kernel the poll succeeds at once)
#pragma unroll
    for (int q = 0; q < EP_MAX_PEERS; ++q)
      if (lane == 0) s_own[q] = own[q];
  }
  __syncthreads();
  constexpr int EPV = DT<T>::EPV;
  const int cpr = r.H / EPV;
  const int nrows = pv.size * pv.cap_rows;
  T* out = reinterpret_cast<T*>(recv_like);
  for (int row = tid; row < nrows; row += blockDim.x) {  // tails: expert id of the row, -1 = unused
    const int q = row / pv.cap_rows, pos = row - q * pv.cap_rows;
    uint64_t m = s_own[q];
    for (int i = 0; i < pos && m; ++i) m &= m - 1;
    reinterpret_cast<int32_t*>(out + (size_t)row * ld + r.H)[0] = m ? (int)__builtin_ctzll(m) : -1;
  }
  for (int i = tid; i < pv.size * pv.cap_rows * cpr; i += blockDim.x) {  // the token's row, once per owned chosen expert
    const int row = i / cpr, c = i - row * cpr;
    const int q = row / pv.cap_rows, pos = row - q * pv.cap_rows;
    if (pos < __popcll(s_own[q])) {
      const T* src = reinterpret_cast<const T*>(pv.base[pv.rank] + pv.recv_off + (int64_t)q * pv.cap_rows * pv.recv_row_bytes);
      *reinterpret_cast<u32x4*>(out + (size_t)row * ld + (size_t)c * EPV) = ld16(src + (size_t)c * EPV);
    }
  }
}
hipError_t launch_ep_bcast_unpack(const RouteArgs& r, const EpBcastArgs& b, void* recv_like, int64_t ld, int dtype, hipStream_t st) {
  if (dtype == DT_F16) hipLaunchKernelGGL(ep_bcast_unpack_kernel<half_t>, dim3(1), dim3(256), 0, st, r, b, recv_like, ld);
  else hipLaunchKernelGGL(ep_bcast_unpack_kernel<uint16_t>, dim3(1), dim3(256), 0, st, r, b, recv_like, ld);
  return hipGetLastError();
}

}  // namespace moeinf
