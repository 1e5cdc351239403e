// small_stream.hip — microbenchmark for SMALL weight-streaming launches (DeepSeek-V2-Lite decode: 35-92 MB per launch,
// ~10 us kernels): where do the microseconds beyond bytes / HBM bandwidth go?  Build: hipcc --offload-arch=gfx950 -O3.
// Work item = 16 weight rows x rowbytes (tiled layout: contiguous 1-KiB tiles), like ffn_rows_kernel.
// Launches rotate over enough distinct buffers (> 256 MB) that the Infinity Cache cannot serve them.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ u32x4 ldnt(const void* p) { return __builtin_nontemporal_load((const u32x4*)p); }

struct Tab { const char* base[8]; int cnt[8]; int off[8]; };

__global__ void k_empty(unsigned* sink) { if (threadIdx.x == 9999) sink[0] = 1; }

// MODE bit0: dependent prologue (active[u] -> {wptr, counts, offsets} -> row_map -> x row), bit1: LDS cross-wave
// reduction + 16 stores, bit2: ALL loads up front (else batches of U=4)
template <int NW, int MAXT, int MODE>
__global__ __launch_bounds__(NW * 64) void k_item(const char* src, int rowbytes, int rgs, const int* active, const unsigned long long* wptr,
                                                 const int* counts, const int* offsets, const int* row_map, const float* x, float* out, unsigned* sink) {
  __shared__ float red[NW][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u = blockIdx.y;
  const char* W = src;
  float xv = 1.f;
  if (MODE & 1) {
    const int e = active[u];
    const int off = offsets[e];
    const int cnt = counts[e];
    W = (const char*)wptr[e];
    const int tok = row_map[off + min(lane & 15, cnt - 1)];
    xv = x[tok * 64 + (lane >> 4)];
  } else {
    W = src + (size_t)u * rgs * 16 * rowbytes;
  }
  const char* base = W + (size_t)blockIdx.x * 16 * rowbytes + lane * 16;
  const int ntile = 16 * rowbytes / 1024;
  u32x4 acc = {0, 0, 0, 0};
  if (MODE & 4) {
    u32x4 v[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) { const int t = wave + i * NW; if (t < ntile) v[i] = ldnt(base + (size_t)t * 1024); else v[i] = u32x4{0, 0, 0, 0}; }
#pragma unroll
    for (int i = 0; i < MAXT; ++i) acc ^= v[i];
  } else {
    int t = wave;
    for (; t + 3 * NW < ntile; t += 4 * NW) {
      u32x4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = ldnt(base + (size_t)(t + i * NW) * 1024);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc ^= v[i];
    }
    for (; t < ntile; t += NW) acc ^= ldnt(base + (size_t)t * 1024);
  }
  float r = __uint_as_float((acc.x ^ acc.y ^ acc.z ^ acc.w) & 0x3fffffffu) * xv;
  if (MODE & 2) {
    red[wave][lane * 4 + 0] = r; red[wave][lane * 4 + 1] = r + 1; red[wave][lane * 4 + 2] = r + 2; red[wave][lane * 4 + 3] = r + 3;
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += NW * 64) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[w][i];
      if ((i & 15) == 0) out[((size_t)u * rgs + blockIdx.x) * 16 + (i >> 4)] = s;
    }
  } else if (r == 123.456f) sink[0] = 1;
}

// persistent: grid = nblk blocks, each walks items blockIdx.x, +nblk, ...; the next item's loads are issued before the
// current one is reduced (all loads of an item up front)
template <int NW, int MAXT>
__global__ __launch_bounds__(NW * 64) void k_persist(const char* src, int rowbytes, int nitems, float* out, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntile = 16 * rowbytes / 1024;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
    const char* base = src + (size_t)it * 16 * rowbytes + lane * 16;
    u32x4 v[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) { const int t = wave + i * NW; if (t < ntile) v[i] = ldnt(base + (size_t)t * 1024); else v[i] = u32x4{0, 0, 0, 0}; }
#pragma unroll
    for (int i = 0; i < MAXT; ++i) acc ^= v[i];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main(int argc, char** argv) {
  const int rowbytes = argc > 1 ? atoi(argv[1]) : 2816;  // stage 2 of DeepSeek-V2-Lite: K = 1408 bf16
  const int rgs = argc > 2 ? atoi(argv[2]) : 128;        // row groups per expert (H / 16)
  const int nexp = argc > 3 ? atoi(argv[3]) : 7;
  const size_t per = (size_t)nexp * rgs * 16 * rowbytes;
  const int nbuf = (int)((600ull << 20) / per) + 1;
  printf("item = 16 rows x %d B, %d row groups x %d experts = %.1f MB per launch, %d rotating buffers\n", rowbytes, rgs, nexp, per / 1e6, nbuf);
  std::vector<char*> bufs(nbuf);
  for (auto& b : bufs) { CK(hipMalloc(&b, per)); CK(hipMemset(b, 1, per)); }
  unsigned* sink; float *out, *x; int *active, *counts, *offsets, *row_map; unsigned long long* wptr;
  CK(hipMalloc(&sink, 4)); CK(hipMalloc(&out, (size_t)nexp * rgs * 16 * 4)); CK(hipMalloc(&x, 64 * 64 * 4)); CK(hipMemset(x, 0, 64 * 64 * 4));
  CK(hipMalloc(&active, 64 * 4)); CK(hipMalloc(&counts, 64 * 4)); CK(hipMalloc(&offsets, 64 * 4)); CK(hipMalloc(&row_map, 64 * 4)); CK(hipMalloc(&wptr, 64 * 8 * nbuf));
  std::vector<int> ha(64), hc(64, 1), ho(64), hr(64, 0);
  for (int i = 0; i < 64; ++i) { ha[i] = i; ho[i] = i; }
  CK(hipMemcpy(active, ha.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(counts, hc.data(), 256, hipMemcpyHostToDevice));
  CK(hipMemcpy(offsets, ho.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(row_map, hr.data(), 256, hipMemcpyHostToDevice));
  std::vector<unsigned long long> hw(64 * nbuf);
  for (int b = 0; b < nbuf; ++b) for (int e = 0; e < 64; ++e) hw[b * 64 + e] = (unsigned long long)(bufs[b] + (size_t)(e % nexp) * rgs * 16 * rowbytes);
  CK(hipMemcpy(wptr, hw.data(), hw.size() * 8, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 400;
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 20; ++i) launch(i % nbuf);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) launch(i % nbuf);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / iters;
    printf("%-64s %7.2f us  %7.1f GB/s\n", name, us, per / us / 1e3);
  };
  dim3 grid(rgs, nexp);
  run("empty kernel, same grid (launch floor, back to back)", [&](int) { k_empty<<<grid, 256>>>(sink); });
#define ITEM(NW, MAXT, MODE) [&](int i) { k_item<NW, MAXT, MODE><<<grid, NW * 64>>>(bufs[i], rowbytes, rgs, active, wptr + i * 64, counts, offsets, row_map, x, out, sink); }
  run("NW4 batches of 4                         ", ITEM(4, 12, 0));
  run("NW4 all loads up front                   ", ITEM(4, 12, 4));
  run("NW4 batches of 4 + dependent prologue    ", ITEM(4, 12, 1));
  run("NW4 up front     + dependent prologue    ", ITEM(4, 12, 5));
  run("NW4 batches of 4 + prologue + LDS reduce ", ITEM(4, 12, 3));
  run("NW4 up front     + prologue + LDS reduce ", ITEM(4, 12, 7));
  run("NW8 up front                             ", ITEM(8, 6, 4));
  run("NW8 up front     + prologue + LDS reduce ", ITEM(8, 6, 7));
  run("NW2 up front                             ", ITEM(2, 24, 4));
  run("NW1 up front (one wave per item)         ", ITEM(1, 48, 4));
  const int nitems = rgs * nexp;
#define PERS(NW, MAXT, NB) [&](int i) { k_persist<NW, MAXT><<<NB, NW * 64>>>(bufs[i], rowbytes, nitems, out, sink); }
  run("persistent NW4, 256 blocks               ", PERS(4, 12, 256));
  run("persistent NW4, 512 blocks               ", PERS(4, 12, 512));
  run("persistent NW4, 768 blocks               ", PERS(4, 12, 768));
  run("persistent NW8, 256 blocks               ", PERS(8, 6, 256));
  run("persistent NW8, 512 blocks               ", PERS(8, 6, 512));
  return 0;
}
