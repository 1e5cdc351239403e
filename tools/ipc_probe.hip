// ipc_probe.hip — which device-memory kinds can be shared between PROCESSES with hipIpc* on this box, and is a
// store -> flag -> load hand-off between two concurrently running kernels of two processes coherent and how fast?
// (Design probe for the peer-store expert-parallel exchange, csrc/ep_peer.h.)
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ipc_probe tools/ipc_probe.hip && /tmp/ipc_probe
//
// Parent: for kind in {uncached, finegrained, coarse}: allocate a window, export it, start a child process (exec of this
// binary) that maps it.  Then both sides run ONE persistent kernel each and play ping-pong N times:
//   parent  writes 64 payload words = i (system-scope write-through stores), drains them, sets flagA = i, polls flagB;
//   child   8 workgroups (one per XCD) poll flagA, read the payload with PLAIN loads (stale words are counted: the same
//           lines were read one round earlier, so a cacheable mapping would serve them from that XCD's L2), workgroup 0
//           writes its own payload and sets flagB = i.
// Every poll is bounded (wall clock), so a kind that does not work is reported, never a hang.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#include <string>

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) { printf("  %s -> %s\n", #x, hipGetErrorString(e_)); return 1; }      \
  } while (0)

constexpr int kFlagA = 0, kFlagB = 32, kStaleBase = 64 /* 3 x 8 words: plain, sc1, sys */, kRounds = 96, kTimeoutW = 97, kPayA = 1024, kPayB = 2048;  // word offsets
constexpr long long kTicksPerSec = 100000000ll;  // wall_clock64: 100 MHz

__device__ __forceinline__ uint32_t ld_sys(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

__device__ bool wait_ge(const uint32_t* flag, uint32_t want, long long budget) {
  const long long t0 = wall_clock64();
  while ((int32_t)(ld_sys(flag) - want) < 0) {
    if (wall_clock64() - t0 > budget) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
  return true;
}

// parent side: one wave
__global__ void ping_kernel(uint32_t* w, int n, long long budget, long long* ticks_out) {
  const int lane = threadIdx.x;
  const long long t0 = wall_clock64();
  int done = 0;
  for (int i = 1; i <= n; ++i) {
    for (int j = 0; j < 3; ++j) st_sys(w + kPayA + 64 * j + lane, (uint32_t)i);
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0) st_sys(w + kFlagA, (uint32_t)i);
    if (!wait_ge(w + kFlagB, (uint32_t)i, budget)) { if (lane == 0) st_sys(w + kTimeoutW, 1u); break; }
    const uint32_t v = w[kPayB + lane];  // plain load of the child's payload
    if (v < (uint32_t)i && lane == 0) atomicAdd(w + kStaleBase + 24, 1u);
    done = i;
  }
  if (lane == 0) { *ticks_out = wall_clock64() - t0; st_sys(w + kRounds, (uint32_t)done); }
}

// child side: 8 workgroups of one wave.  first/last: the rounds this launch plays (persistent: 1..n in one launch; or one
// launch per round, as the exchange's consumer kernels are: the L1 is invalidated at kernel start, the L2 is not)
__global__ void pong_kernel(uint32_t* w, int first, int last, long long budget) {
  const int lane = threadIdx.x, b = blockIdx.x;
  for (int i = first; i <= last; ++i) {
    if (ld_sys(w + kTimeoutW)) return;  // an earlier wait gave up: every later launch leaves at once
    if (!wait_ge(w + kFlagA, (uint32_t)i, budget)) { if (lane == 0) st_sys(w + kTimeoutW, 2u); return; }
    const uint32_t v0 = w[kPayA + lane];  // PLAIN load
    const uint32_t v1 = __hip_atomic_load(w + kPayA + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1
    const uint32_t v2 = ld_sys(w + kPayA + 128 + lane);                                                        // sc0 sc1
    if (v0 < (uint32_t)i) atomicAdd(w + kStaleBase + b, 1u);
    if (v1 < (uint32_t)i) atomicAdd(w + kStaleBase + 8 + b, 1u);
    if (v2 < (uint32_t)i) atomicAdd(w + kStaleBase + 16 + b, 1u);
    if (b == 0) {
      st_sys(w + kPayB + lane, (uint32_t)i);
      __builtin_amdgcn_s_waitcnt(0);
      if (lane == 0) st_sys(w + kFlagB, (uint32_t)i);
    }
  }
}

static const char* kind_name(int k) { return k == 0 ? "uncached" : k == 1 ? "finegrained" : "coarse (hipMalloc)"; }

static int run_pong(uint32_t* p, int n, int per_round, hipStream_t st) {
  if (!per_round) hipLaunchKernelGGL(pong_kernel, dim3(8), dim3(64), 0, st, p, 1, n, 3 * kTicksPerSec);
  else for (int i = 1; i <= n; ++i) hipLaunchKernelGGL(pong_kernel, dim3(8), dim3(64), 0, st, p, i, i, 3 * kTicksPerSec);
  return 0;
}
static int child_main(const char* path, int n, int per_round) {
  FILE* f = fopen(path, "rb");
  if (!f) { printf("  child: cannot read %s\n", path); return 1; }
  hipIpcMemHandle_t h;
  if (fread(&h, sizeof h, 1, f) != 1) return 1;
  fclose(f);
  CK(hipSetDevice(0));
  void* p = nullptr;
  CK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  run_pong((uint32_t*)p, n, per_round, 0);
  CK(hipDeviceSynchronize());
  CK(hipIpcCloseMemHandle(p));
  return 0;
}

// mode: 0 = peer is another PROCESS, one persistent kernel; 1 = another process, one kernel launch per round;
//       2 = peer is a second STREAM of this process (control), persistent; 3 = second stream, one launch per round
int main(int argc, char** argv) {
  if (argc >= 5 && !strcmp(argv[1], "child")) return child_main(argv[2], atoi(argv[3]), atoi(argv[4]));
  CK(hipSetDevice(0));
  for (int mode = 0; mode < 4; ++mode)
  for (int kind = 0; kind < 3; ++kind) {
    const int n = (mode & 1) ? 400 : 2000;
    printf("== %s, peer = %s, %s\n", kind_name(kind), mode < 2 ? "another process" : "a second stream of this process",
           (mode & 1) ? "one consumer launch per round" : "persistent consumer kernel");
    fflush(stdout);
    void* p = nullptr;
    const size_t bytes = 1 << 20;
    hipError_t e = kind == 0   ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached)
                   : kind == 1 ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained)
                               : hipMalloc(&p, bytes);
    if (e != hipSuccess) { printf("  allocation: %s\n", hipGetErrorString(e)); continue; }
    if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { printf("  memset failed\n"); continue; }
    pid_t pid = -1;
    hipStream_t st2 = nullptr;
    char path[64] = "";
    if (mode < 2) {
      hipIpcMemHandle_t h;
      e = hipIpcGetMemHandle(&h, p);
      if (e != hipSuccess) { printf("  hipIpcGetMemHandle: %s\n", hipGetErrorString(e)); hipFree(p); continue; }
      snprintf(path, sizeof path, "/tmp/ipc_probe_%d_%d_%d.bin", (int)getpid(), kind, mode);
      FILE* f = fopen(path, "wb");
      fwrite(&h, sizeof h, 1, f);
      fclose(f);
      char nb[16], mb[16];
      snprintf(nb, sizeof nb, "%d", n);
      snprintf(mb, sizeof mb, "%d", mode & 1);
      pid = fork();  // exec at once: the child never touches this process's HIP state
      if (pid == 0) { execl(argv[0], argv[0], "child", path, nb, mb, (char*)nullptr); _exit(127); }
    }
    long long* ticks = nullptr;
    CK(hipHostMalloc((void**)&ticks, 8));
    *ticks = 0;
    hipStream_t st1;
    CK(hipStreamCreateWithFlags(&st1, hipStreamNonBlocking));
    if (mode >= 2) {
      CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
      run_pong((uint32_t*)p, n, mode & 1, st2);
    }
    hipLaunchKernelGGL(ping_kernel, dim3(1), dim3(64), 0, st1, (uint32_t*)p, n, 3 * kTicksPerSec, ticks);
    e = hipDeviceSynchronize();
    int status = 0;
    if (pid > 0) waitpid(pid, &status, 0);
    uint32_t host[128];
    CK(hipMemcpy(host, p, sizeof host, hipMemcpyDeviceToHost));
    unsigned stale[3] = {0, 0, 0};
    for (int j = 0; j < 3; ++j) for (int b = 0; b < 8; ++b) stale[j] += host[kStaleBase + 8 * j + b];
    printf("  sync: %s; child exit %d; rounds completed %u of %d%s; %.2f us per round trip (two hops)\n", hipGetErrorString(e),
           pid > 0 ? (WIFEXITED(status) ? WEXITSTATUS(status) : -1) : 0, host[kRounds], n, host[kTimeoutW] ? " (TIMEOUT)" : "",
           host[kRounds] ? (double)*ticks / kTicksPerSec * 1e6 / host[kRounds] : 0.0);
    printf("  stale payload reads on the consumer (of %d): plain %u, sc1 %u, sc0 sc1 %u; on the producer side (plain, of %d): %u\n",
           8 * 64 * n, stale[0], stale[1], stale[2], n, host[kStaleBase + 24]);
    fflush(stdout);
    if (path[0]) unlink(path);
    hipFree(p);
    hipHostFree(ticks);
    hipStreamDestroy(st1);
    if (st2) hipStreamDestroy(st2);
  }
  return 0;
}
