// ep_peer.h — direct peer-store transport of the expert-parallel exchange (host side; the device side is EpPeers in
// kernels.h and the kernels in ep_kernels.hip / kdev.h).
//
// What this replaces in the reference: cudaDeviceEnablePeerAccess for every device pair at start-up
// (core/prefetch/archer_prefetch_handle.cpp:37-61) and the implicit P2P copies `tensor.to(device)` that carry an expert's
// input rows to the GPU holding the expert and its output rows back (core/parallel/expert_dispatcher.cpp:284,405) — one
// process, N GPUs.  Here: one process per GPU; every rank allocates ONE exchange window in uncached device memory,
// exports it (hipIpcGetMemHandle) and maps every peer's window (hipIpcOpenMemHandle; ranks inside one process use the
// pointer as it is, after hipDeviceEnablePeerAccess when they sit on different devices).  The router's pack step and the
// owner's FFN stage 2 then STORE rows straight into the destination's window and publish an exchange number in its flag
// words; consumers poll their own flags.  No collective, no send/recv kernel (RCCL: ~11 us per all-to-all of a few KB,
// even to itself), and — unlike RCCL, which refuses two ranks on one GPU — it runs between processes that share a
// device, which is how the multi-rank path is tested on a one-GPU box.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "kernels.h"

namespace moeinf {

constexpr uint32_t kEpPeerMagic = 0x4d504531u;  // "MPE1"
constexpr int kEpPeerBlobBytes = 192;           // MOEINF_EP_PEER_BLOB_BYTES

// what the ranks hand each other (any channel: torch.distributed all_gather, a file, MPI)
struct EpPeerBlob {
  uint32_t magic, bytes_of_blob;
  int32_t rank, size;
  int32_t pid, device;      // exporting process, its HIP device ordinal
  char bus_id[32];          // PCI bus id of that device: equal ids + different pids = ranks sharing one GPU
  uint64_t ptr;             // window address in the exporting process
  uint64_t window_bytes;
  int32_t cap_rows;
  int32_t wishes;           // this rank's environment overrides, so the decisions below are taken from the SAME data on every rank:
                            // bits 0-1: MOEINF_EP_PEER_POLL (0 unset, 1 "0", 2 "1"); bit 2: broadcast form allowed (MOEINF_EP_BCAST != 0)
  int64_t recv_row_bytes, ret_row_bytes;
  hipIpcMemHandle_t handle;  // 64 bytes
};
static_assert(sizeof(EpPeerBlob) <= kEpPeerBlobBytes, "blob layout");

struct EpPeerWindow {
  void* base = nullptr;  // this rank's window
  size_t bytes = 0;
  int cap_rows = 0;
  int64_t recv_off = 0, ret_off = 0, bcast_off = 0, recv_row_bytes = 0, ret_row_bytes = 0;
  int bcast_stride = 0;  // bytes per rank in the broadcast region (E gate logits, batch-1 broadcast form)
  const char* mem_kind = "";
  std::vector<void*> peer;       // window of every rank as mapped here
  std::vector<char> opened;      // 1: mapped with hipIpcOpenMemHandle (close it)
  bool attached = false;
  bool shared_device = false;    // some other rank runs on THIS GPU (another process): consumers must not spin in wide kernels
  // decided in attach() from ALL ranks' blobs — identical on every rank, because the exchange FORM must be (a rank in the
  // broadcast form and a rank in the routed form read regions of each other's windows that nobody wrote):
  bool shared_anywhere = false;  // ANY two ranks of the group share a GPU
  bool poll_agreed = true;       // consumer kernels poll for themselves (false: a one-wave wait kernel in front) — on every rank
  bool bcast_agreed = true;      // the batch-1 broadcast form may be taken — on every rank
  uint32_t epoch = 0;
  int32_t* done = nullptr;       // arrival counter (ordinary device memory)
  std::string bus_id;

  static int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

  // allocate + zero the window; "" or an error text
  std::string create(int size, int cap, int64_t recv_row, int64_t ret_row, int num_experts) {
    if (size > EP_MAX_PEERS) return "peer-store exchange supports up to " + std::to_string(EP_MAX_PEERS) + " ranks";
    cap_rows = cap; recv_row_bytes = recv_row; ret_row_bytes = ret_row;
    recv_off = EP_WINDOW_HDR;
    ret_off = align_up(recv_off + (int64_t)size * cap * recv_row, 256);
    bcast_off = align_up(ret_off + (int64_t)size * cap * ret_row, 256);
    bcast_stride = (int)align_up((int64_t)num_experts * 4, 256);
    bytes = (size_t)align_up(bcast_off + (int64_t)size * bcast_stride, 4096);
    // uncached (MTYPE UC): no cache level of this GPU keeps a line of the window, so a consumer's plain loads see what a
    // peer's kernel stored a moment ago; measured alternatives: tools/ipc_probe.hip
    const char* want = getenv("MOEINF_EP_PEER_MEM");
    hipError_t e = hipErrorUnknown;
    if (!want || !strcmp(want, "uncached")) { e = hipExtMallocWithFlags(&base, bytes, hipDeviceMallocUncached); mem_kind = "uncached"; }
    else if (!strcmp(want, "finegrained")) { e = hipExtMallocWithFlags(&base, bytes, hipDeviceMallocFinegrained); mem_kind = "finegrained"; }
    else if (!strcmp(want, "coarse")) { e = hipMalloc(&base, bytes); mem_kind = "coarse"; }
    if (e != hipSuccess) { base = nullptr; return std::string("exchange window (") + mem_kind + "): " + hipGetErrorString(e); }
    if ((e = hipMemset(base, 0, bytes)) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) return std::string("zeroing the window: ") + hipGetErrorString(e);
    if ((e = hipMalloc((void**)&done, 64)) != hipSuccess || (e = hipMemset(done, 0, 64)) != hipSuccess) return std::string("arrival counter: ") + hipGetErrorString(e);
    return "";
  }

  static int32_t env_wishes() {
    int32_t w = 0;
    if (const char* e = getenv("MOEINF_EP_PEER_POLL")) w |= atoi(e) != 0 ? 2 : 1;
    const char* b = getenv("MOEINF_EP_BCAST");
    if (!b || atoi(b) != 0) w |= 4;
    return w;
  }

  std::string export_blob(int rank, int size, int device, EpPeerBlob* b) const {
    memset(b, 0, sizeof *b);
    b->wishes = env_wishes();
    b->magic = kEpPeerMagic; b->bytes_of_blob = sizeof *b;
    b->rank = rank; b->size = size; b->pid = (int32_t)getpid(); b->device = device;
    hipError_t e = hipDeviceGetPCIBusId(b->bus_id, sizeof b->bus_id, device);
    if (e != hipSuccess) return std::string("hipDeviceGetPCIBusId: ") + hipGetErrorString(e);
    b->ptr = (uint64_t)base; b->window_bytes = bytes; b->cap_rows = cap_rows;
    b->recv_row_bytes = recv_row_bytes; b->ret_row_bytes = ret_row_bytes;
    e = hipIpcGetMemHandle(&b->handle, base);
    if (e != hipSuccess) return std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e);
    return "";
  }

  // map every peer's window; blobs in rank order
  std::string attach(const EpPeerBlob* blobs, int rank, int size, int device) {
    if (attached) return "already attached";
    peer.assign(size, nullptr); opened.assign(size, 0);
    char mine[32] = {0};
    hipDeviceGetPCIBusId(mine, sizeof mine, device);
    bus_id = mine;
    // the reference's start-up loop: peer access between this device and every other visible one
    // (archer_prefetch_handle.cpp:37-61); "already enabled" and "not supported" are not errors here — the mapping below decides
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) == hipSuccess)
      for (int d = 0; d < ndev; ++d) {
        int can = 0;
        if (d != device && hipDeviceCanAccessPeer(&can, device, d) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(d, 0);
      }
    (void)hipGetLastError();
    for (int p = 0; p < size; ++p) {
      const EpPeerBlob& b = blobs[p];
      if (b.magic != kEpPeerMagic || b.rank != p || b.size != size) { detach(); return "blob " + std::to_string(p) + " is not rank " + std::to_string(p) + "'s export"; }
      if (b.window_bytes != bytes || b.cap_rows != cap_rows || b.recv_row_bytes != recv_row_bytes || b.ret_row_bytes != ret_row_bytes) {
        detach();
        return "rank " + std::to_string(p) + " built a different window (cap_tokens / model shape must be the same on every rank)";
      }
      if (p == rank) { peer[p] = base; continue; }
      if (b.pid == (int32_t)getpid()) {
        peer[p] = (void*)b.ptr;  // same process (several engines in one process): the address is valid here
      } else {
        void* q = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&q, b.handle, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { detach(); return "hipIpcOpenMemHandle(rank " + std::to_string(p) + "): " + hipGetErrorString(e); }
        peer[p] = q; opened[p] = 1;
        if (!strncmp(b.bus_id, mine, sizeof mine)) shared_device = true;
      }
    }
    // group-wide decisions, from data every rank holds identically (the blobs): who shares a GPU with whom, and every
    // rank's environment overrides
    shared_anywhere = false;
    for (int p = 0; p < size; ++p)
      for (int q = p + 1; q < size; ++q)
        if (blobs[p].pid != blobs[q].pid && !strncmp(blobs[p].bus_id, blobs[q].bus_id, sizeof blobs[p].bus_id)) shared_anywhere = true;
    poll_agreed = true; bcast_agreed = true;
    for (int p = 0; p < size; ++p) {
      const int pw = blobs[p].wishes & 3;
      const bool vote = pw ? pw == 2 : !shared_anywhere;  // a rank's wish, else the default: never spin where a peer needs the CUs
      poll_agreed = poll_agreed && vote;
      bcast_agreed = bcast_agreed && (blobs[p].wishes & 4) != 0;
    }
    attached = true;
    return "";
  }

  // unmap the peers (the window itself stays: it can be exported and attached again)
  void detach() {
    for (size_t p = 0; p < peer.size(); ++p) if (opened[p] && peer[p]) (void)hipIpcCloseMemHandle(peer[p]);
    peer.clear(); opened.clear();
    attached = false; shared_device = false;
  }

  void view(EpPeers* v, int rank, int size, int32_t* err, int64_t timeout_ticks, bool poll) const {
    memset(v, 0, sizeof *v);
    for (int p = 0; p < size; ++p) v->base[p] = (uint64_t)peer[p];
    v->recv_off = recv_off; v->ret_off = ret_off; v->recv_row_bytes = recv_row_bytes; v->ret_row_bytes = ret_row_bytes;
    v->bcast_off = bcast_off; v->bcast_stride = bcast_stride;
    v->timeout_ticks = timeout_ticks; v->done = done; v->err = err; v->epoch = epoch;
    v->rank = rank; v->size = size; v->cap_rows = cap_rows; v->on = 1; v->poll = poll ? 1 : 0;
  }
  void* recv_region() const { return (char*)base + recv_off; }
  void* ret_region() const { return (char*)base + ret_off; }
  const uint32_t* recv_flags() const { return (const uint32_t*)base; }
  const uint32_t* ret_flags() const { return (const uint32_t*)((const char*)base + EP_RET_FLAGS_OFF); }

  void destroy() {
    detach();
    if (base) (void)hipFree(base);
    if (done) (void)hipFree(done);
    base = nullptr; done = nullptr; attached = false; bytes = 0;
    // a released window starts over: ranks that fell out of step (one of them failed before it took its exchange number)
    // must meet again at exchange 0 after the next export / attach, not at their old, different epochs
    epoch = 0; shared_anywhere = false; poll_agreed = true; bcast_agreed = true;
  }
};

}  // namespace moeinf
