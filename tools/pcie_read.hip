// pcie_read.hip — can a KERNEL pull expert blobs from pinned host memory as fast as the SDMA engines copy them?
// (round 6: every hipMemcpyAsync of the tier mover costs ~60 us on top of its bytes — 16.5 MiB DeepSeek experts reach 46-52 GB/s
// where 112 MiB Mixtral pieces reach 54.6; a fetch kernel that reads the host blob directly and writes the tiled slot would have
// no per-copy cost, no staging buffer and no re-tile launch.)  Measures, for a 16.5 MiB and a 336 MiB blob: hipMemcpyAsync,
// and a copy kernel reading host memory with W workgroups of 256 threads, 16 bytes per lane, contiguous per wave.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pcie_read tools/pcie_read.hip && /tmp/pcie_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void pull(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n16; i += stride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + (size_t)u * 256 < n16) v[u] = __builtin_nontemporal_load(src + i + (size_t)u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + (size_t)u * 256 < n16) dst[i + (size_t)u * 256] = v[u];
  }
}

int main() {
  const size_t sizes[] = {(size_t)17301504, (size_t)352321536};
  hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  for (size_t bytes : sizes) {
    const int nblob = bytes < (64u << 20) ? 32 : 4;
    char* h; CHK(hipHostMalloc((void**)&h, bytes * nblob, hipHostMallocDefault));
    for (size_t i = 0; i < bytes * nblob; i += 4096) h[i] = (char)i;
    char* d; CHK(hipMalloc((void**)&d, bytes * nblob));
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
      CHK(hipEventRecord(a, st));
      for (int k = 0; k < nblob; ++k) CHK(hipMemcpyAsync(d + k * bytes, h + k * bytes, bytes, hipMemcpyHostToDevice, st));
      CHK(hipEventRecord(b, st)); CHK(hipEventSynchronize(b)); CHK(hipEventElapsedTime(&ms, a, b));
    }
    printf("blob %6.1f MiB x %2d  hipMemcpyAsync back to back      : %7.1f us per blob, %6.2f GB/s\n", bytes / 1048576.0, nblob, ms * 1e3 / nblob, bytes * nblob / ms / 1e6);
    for (int wgs : {8, 16, 32, 64, 128, 256, 512}) {
      for (int u : {4, 8}) {
        for (int rep = 0; rep < 2; ++rep) {
          CHK(hipEventRecord(a, st));
          for (int k = 0; k < nblob; ++k) {
            if (u == 4) hipLaunchKernelGGL(pull<4>, dim3(wgs), dim3(256), 0, st, (const u32x4*)(h + k * bytes), (u32x4*)(d + k * bytes), bytes / 16);
            else hipLaunchKernelGGL(pull<8>, dim3(wgs), dim3(256), 0, st, (const u32x4*)(h + k * bytes), (u32x4*)(d + k * bytes), bytes / 16);
          }
          CHK(hipEventRecord(b, st)); CHK(hipEventSynchronize(b)); CHK(hipEventElapsedTime(&ms, a, b));
        }
        printf("blob %6.1f MiB x %2d  pull kernel %3d workgroups, %d x 16 B per lane in flight: %7.1f us per blob, %6.2f GB/s\n", bytes / 1048576.0, nblob, wgs, u, ms * 1e3 / nblob, bytes * nblob / ms / 1e6);
      }
    }
    CHK(hipFree(d)); CHK(hipHostFree(h));
  }
  return 0;
}
