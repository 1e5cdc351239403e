// stream_patterns.hip — microbenchmark: which HBM read patterns reach what bandwidth on MI355X.
// Decides the weight layout / lane mapping of ffn_rows_kernel.  Build: hipcc --offload-arch=gfx950 -O3.
//   A  contiguous: every wave-instruction reads 1 KiB contiguous (copy-like upper bound)
//   B  16 rows x 64 B per wave-instruction, row stride = rowbytes (the MFMA A-operand pattern)
//   C  like B but the 4/8 waves of a block take interleaved 128-B windows (current kernel)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int NT>
__device__ __forceinline__ u32x4 ld(const void* p) {
  if (NT) return __builtin_nontemporal_load((const u32x4*)p);
  return *(const u32x4*)p;
}

// A: grid-stride contiguous
template <int NT, int U>
__global__ __launch_bounds__(256) void k_contig(const char* src, size_t bytes, unsigned* sink) {
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
  const size_t stride = (size_t)gridDim.x * 256 * 16;
  u32x4 acc = {0, 0, 0, 0};
  for (; i + (U - 1) * stride < bytes; i += U * stride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<NT>(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

// B/C: block = NW waves owns 16 rows of `rowbytes`; grid.x = nrows/16.  MODE 0: waves take interleaved
// 128-B windows, two 64-B halves per lane pair of loads (current kernel); MODE 1: each wave a contiguous
// K range; MODE 2: lane loads 32 contiguous bytes (old mapping)
template <int NT, int U, int NW, int MODE>
__global__ __launch_bounds__(NW * 64) void k_rows(const char* src, int rowbytes, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const char* row = src + ((size_t)blockIdx.x * 16 + r) * rowbytes;
  const int nwin = rowbytes / 128;
  u32x4 acc = {0, 0, 0, 0};
  if (MODE == 1) {
    const int per = nwin / NW;
    for (int w = wave * per; w < (wave + 1) * per; w += U) {
      u32x4 v[U][2];
#pragma unroll
      for (int u = 0; u < U; ++u) { v[u][0] = ld<NT>(row + (size_t)(w + u) * 128 + q * 16); v[u][1] = ld<NT>(row + (size_t)(w + u) * 128 + 64 + q * 16); }
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u][0] ^ v[u][1];
    }
  } else {
    for (int w = wave; w + (U - 1) * NW < nwin; w += U * NW) {
      u32x4 v[U][2];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t o = (size_t)(w + u * NW) * 128;
        if (MODE == 0) { v[u][0] = ld<NT>(row + o + q * 16); v[u][1] = ld<NT>(row + o + 64 + q * 16); }
        else           { v[u][0] = ld<NT>(row + o + q * 32); v[u][1] = ld<NT>(row + o + q * 32 + 16); }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u][0] ^ v[u][1];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

// E: pre-tiled layout: a block owns one contiguous region of 16*rowbytes; every wave-instruction reads
// 1 KiB contiguous (one 16x32 bf16 MFMA A-tile); waves interleave tiles
template <int NT, int U, int NW>
__global__ __launch_bounds__(NW * 64) void k_tiles(const char* src, int rowbytes, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = src + (size_t)blockIdx.x * 16 * rowbytes + lane * 16;
  const int ntile = 16 * rowbytes / 1024;
  u32x4 acc = {0, 0, 0, 0};
  for (int t = wave; t + (U - 1) * NW < ntile; t += U * NW) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<NT>(base + (size_t)(t + u * NW) * 1024);
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <typename F>
static double timeit(F f, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / iters;
}

int main() {
  const int rowbytes = 8192;            // Mixtral stage 1: H=4096 bf16
  const int nrows = 2 * 2 * 14336;      // 2 experts x (w1+w3) x F rows  = 470 MB
  const size_t bytes = (size_t)nrows * rowbytes;
  char* src; unsigned* sink;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(src, 1, bytes));
  // flush-ish buffer to defeat the 256 MB infinity cache between iterations: bytes (470 MB) > 256 MB already
  printf("pattern, bytes=%.1f MB, rowbytes=%d\n", bytes / 1e6, rowbytes);
#define REPORT(name, call) { double ms = timeit([&] { call; }, 20); printf("%-44s %8.1f us  %7.1f GB/s\n", name, ms * 1e3, bytes / ms / 1e6); }
  REPORT("A contiguous, grid 2048, U4", (k_contig<0, 4><<<dim3(2048), dim3(256)>>>(src, bytes, sink)));
  REPORT("A contiguous nt, grid 2048, U4", (k_contig<1, 4><<<dim3(2048), dim3(256)>>>(src, bytes, sink)));
  REPORT("A contiguous nt, grid 8192, U4", (k_contig<1, 4><<<dim3(8192), dim3(256)>>>(src, bytes, sink)));
  REPORT("A contiguous nt, grid 1024, U8", (k_contig<1, 8><<<dim3(1024), dim3(256)>>>(src, bytes, sink)));
  REPORT("C rows nt NW4 U2 interleaved 64B-halves", (k_rows<1, 2, 4, 0><<<dim3(nrows / 16), dim3(256)>>>(src, rowbytes, sink)));
  REPORT("C rows nt NW4 U4 interleaved 64B-halves", (k_rows<1, 4, 4, 0><<<dim3(nrows / 16), dim3(256)>>>(src, rowbytes, sink)));
  REPORT("C rows    NW4 U4 interleaved 64B-halves", (k_rows<0, 4, 4, 0><<<dim3(nrows / 16), dim3(256)>>>(src, rowbytes, sink)));
  REPORT("C rows nt NW8 U2 interleaved 64B-halves", (k_rows<1, 2, 8, 0><<<dim3(nrows / 16), dim3(512)>>>(src, rowbytes, sink)));
  REPORT("C rows nt NW8 U4 interleaved 64B-halves", (k_rows<1, 4, 8, 0><<<dim3(nrows / 16), dim3(512)>>>(src, rowbytes, sink)));
  REPORT("B rows nt NW4 U4 contiguous K range/wave", (k_rows<1, 4, 4, 1><<<dim3(nrows / 16), dim3(256)>>>(src, rowbytes, sink)));
  REPORT("D rows nt NW4 U4 32B-per-lane (old)", (k_rows<1, 4, 4, 2><<<dim3(nrows / 16), dim3(256)>>>(src, rowbytes, sink)));
  REPORT("D rows nt NW4 U2 32B-per-lane (old)", (k_rows<1, 2, 4, 2><<<dim3(nrows / 16), dim3(256)>>>(src, rowbytes, sink)));
  REPORT("C rows nt NW2 U4 interleaved 64B-halves", (k_rows<1, 4, 2, 0><<<dim3(nrows / 16), dim3(128)>>>(src, rowbytes, sink)));
  REPORT("C rows nt NW1 U4 (one wave per 16 rows)", (k_rows<1, 4, 1, 0><<<dim3(nrows / 16), dim3(64)>>>(src, rowbytes, sink)));
  REPORT("C rows nt NW1 U8 (one wave per 16 rows)", (k_rows<1, 8, 1, 0><<<dim3(nrows / 16), dim3(64)>>>(src, rowbytes, sink)));
  REPORT("E tiles nt NW4 U4 (1 KiB contiguous / instr)", (k_tiles<1, 4, 4><<<dim3(nrows / 16), dim3(256)>>>(src, rowbytes, sink)));
  REPORT("E tiles nt NW4 U8", (k_tiles<1, 8, 4><<<dim3(nrows / 16), dim3(256)>>>(src, rowbytes, sink)));
  REPORT("E tiles nt NW8 U4", (k_tiles<1, 4, 8><<<dim3(nrows / 16), dim3(512)>>>(src, rowbytes, sink)));
  REPORT("E tiles nt NW8 U8", (k_tiles<1, 8, 8><<<dim3(nrows / 16), dim3(512)>>>(src, rowbytes, sink)));
  REPORT("E tiles    NW4 U8 (default policy)", (k_tiles<0, 8, 4><<<dim3(nrows / 16), dim3(256)>>>(src, rowbytes, sink)));
  REPORT("E tiles nt NW4 U8, 32 rows/block", (k_tiles<1, 8, 4><<<dim3(nrows / 32), dim3(256)>>>(src, rowbytes * 2, sink)));
  REPORT("E tiles nt NW4 U16", (k_tiles<1, 16, 4><<<dim3(nrows / 16), dim3(256)>>>(src, rowbytes, sink)));
  return 0;
}
