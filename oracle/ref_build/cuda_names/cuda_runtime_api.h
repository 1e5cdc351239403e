// TEST INFRASTRUCTURE ONLY — never on the product's include path.
// The reference's headers (core/model/model_topology.h, core/utils/cuda_utils.h) name a handful of CUDA runtime
// types and functions in signatures and inline helpers.  oracle/build_ref.py compiles TWO host-only reference
// sources (core/parallel/expert_module.cpp, core/aio/archer_tensor_index.cpp — pure ATen / iostream code that never
// calls the CUDA runtime) where they lie under /root/reference; these declarations let those headers PARSE here.
// Nothing below is called by the code that gets built.
#pragma once
#include <hip/hip_runtime_api.h>
typedef hipStream_t cudaStream_t;
typedef hipError_t cudaError_t;
typedef hipMemcpyKind cudaMemcpyKind;
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaMalloc hipMalloc
#define cudaFree hipFree
#define cudaHostAlloc hipHostMalloc
#define cudaHostAllocDefault hipHostMallocDefault
#define cudaFreeHost hipHostFree
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaMemcpyDeviceToDevice hipMemcpyDeviceToDevice
