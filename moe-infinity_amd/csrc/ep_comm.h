// ep_comm.h — native collective transport of the expert-parallel exchange (host-only).
//
// RCCL is used directly (ncclSend / ncclRecv inside one group = an all-to-all over xGMI, one kernel per collective), so
// that a whole expert-parallel MoE layer is ONE host call into the engine instead of five calls through
// torch.distributed.  The library is bound at RUN TIME with dlopen: a process that already carries RCCL (PyTorch-ROCm
// loads its own librccl.so.1) is joined to that copy (RTLD_NOLOAD first), a plain C++ host gets ROCm's.  Nothing here
// depends on torch.
//
// What this replaces in the reference: nothing collective — the reference moves rows between GPUs from ONE process with
// `tensor.to(device)` after cudaDeviceEnablePeerAccess (core/parallel/expert_dispatcher.cpp:284,405,
// core/prefetch/archer_prefetch_handle.cpp:37-61).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <string.h>

#include <string>

namespace moeinf {

struct RcclUniqueId { char internal[128]; };  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* RcclComm;                        // ncclComm_t

class RcclApi {
 public:
  // nullptr + err on failure; the library stays loaded for the life of the process
  static const RcclApi* get(std::string* err) {
    static RcclApi api;
    static std::string load_err;
    static bool tried = false;
    if (!tried) {
      tried = true;
      load_err = api.load();
    }
    if (!load_err.empty()) { if (err) *err = load_err; return nullptr; }
    return &api;
  }
  int (*GetUniqueId)(RcclUniqueId*) = nullptr;
  int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
  int (*CommDestroy)(RcclComm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string where;  // which library was bound

 private:
  std::string load() {
    void* h = nullptr;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) { where = std::string(n) + " (already loaded)"; break; } }
    if (!h) for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) { where = n; break; } }
    if (!h) for (const char* n : {"/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) { where = n; break; } }
    if (!h) return std::string("librccl not found: ") + (dlerror() ? dlerror() : "?");
#define MOEINF_SYM(field, name)                                               \
    *(void**)(&field) = dlsym(h, name);                                       \
    if (!field) return std::string("librccl has no symbol ") + name;
    MOEINF_SYM(GetUniqueId, "ncclGetUniqueId")
    MOEINF_SYM(CommInitRank, "ncclCommInitRank")
    MOEINF_SYM(CommDestroy, "ncclCommDestroy")
    MOEINF_SYM(GroupStart, "ncclGroupStart")
    MOEINF_SYM(GroupEnd, "ncclGroupEnd")
    MOEINF_SYM(Send, "ncclSend")
    MOEINF_SYM(Recv, "ncclRecv")
    MOEINF_SYM(GetErrorString, "ncclGetErrorString")
#undef MOEINF_SYM
    return "";
  }
};

// One equal-split all-to-all: segment p of `send` (bytes_per_peer bytes) goes to rank p, segment p of `recv` comes from
// rank p.  Returns "" or an error text.
inline std::string rccl_all_to_all(const RcclApi* api, RcclComm comm, int nranks, const void* send, void* recv, size_t bytes_per_peer,
                                   hipStream_t st) {
  constexpr int kUint8 = 1;  // ncclUint8
  int rc = api->GroupStart();
  if (rc) return std::string("ncclGroupStart: ") + api->GetErrorString(rc);
  for (int p = 0; p < nranks && rc == 0; ++p) {
    rc = api->Send((const char*)send + (size_t)p * bytes_per_peer, bytes_per_peer, kUint8, p, comm, st);
    if (rc == 0) rc = api->Recv((char*)recv + (size_t)p * bytes_per_peer, bytes_per_peer, kUint8, p, comm, st);
  }
  const int rc2 = api->GroupEnd();
  if (rc) return std::string("ncclSend/ncclRecv: ") + api->GetErrorString(rc);
  if (rc2) return std::string("ncclGroupEnd: ") + api->GetErrorString(rc2);
  return "";
}

}  // namespace moeinf
