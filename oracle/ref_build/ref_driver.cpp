// ref_driver.cpp — TEST INFRASTRUCTURE ONLY.  C entry points around the REFERENCE'S OWN code, compiled from where it
// lies under /root/reference (oracle/build_ref.py; output oracle/_ref/libmoeinf_ref.so, git-ignored):
//   core/parallel/expert_module.cpp   the expert FFN modules (the ATen op sequences of R6) + SetTensorsFromBlob
//   core/aio/archer_tensor_index.cpp  the archer_index serializer (disk-tier format)
// Used to pin oracle/moe_ref.py:expert_ffn and oracle/offload_format_ref.py against real reference code, and to
// generate tests/golden/ffn_ref_*.npz (oracle/gen_golden_ref.py).  Never linked into or called by the product.
#include <stdint.h>
#include <string.h>

#include <memory>
#include <vector>

#include "aio/archer_tensor_index.h"
#include "parallel/expert_module.h"

// the globals the reference's headers declare extern and its (unbuilt) other translation units define
std::unique_ptr<ArcherTopologyHandle> kTopologyHandle(nullptr);

static torch::Tensor blob(const void* p, std::vector<int64_t> shape, torch::ScalarType st) {
  return torch::from_blob(const_cast<void*>(p), shape, torch::TensorOptions().dtype(st).device(torch::kCPU));
}

extern "C" {

// y[T,H] = <reference expert module of `expert_type`>.forward(x[T,H]); tensors in the reference's blob order
// (tensor_ids order of SetTensorsFromBlob).  dtype: expert_module.h DTYPE_* ids.  Returns 0 on success.
int ref_expert_ffn(int expert_type, int dtype, const void* x, int64_t T, int64_t H, int64_t F, const void* const* tensors,
                   int n_tensors, void* y) {
  try {
    if (!kTensorIndex) kTensorIndex = std::make_unique<ArcherTensorIndex>();
    const auto st = dtype_to_torch(dtype);
    std::vector<std::vector<int64_t>> shapes;
    switch (expert_type) {
      case MIXTRAL_MOE_DENSE_ACT_DENSE: shapes = {{F, H}, {H, F}, {F, H}}; break;
      case DEEPSEEK_MOE_DENSE_ACT_DENSE: case SWITCH_TRANSFORMERS_DENSE_GATED_ACT_DENSE: shapes = {{F, H}, {F, H}, {H, F}}; break;
      case NLLB_MOE_DENSE_ACT_DENSE: case FSGPT_MOE_DENSE_ACT_DENSE: shapes = {{F, H}, {F}, {H, F}, {H}}; break;
      case SWITCH_TRANSFORMERS_DENSE_ACT_DENSE: shapes = {{F, H}, {H, F}}; break;
      default: return 2;
    }
    if ((int)shapes.size() != n_tensors) return 3;
    std::vector<uint32_t> ids;
    for (int i = 0; i < n_tensors; ++i) {
      TensorStorageMeta m;
      m.tensor = blob(tensors[i], shapes[i], st);
      m.shape = shapes[i];
      (*kTensorIndex)[900000u + (uint32_t)i] = m;
      ids.push_back(900000u + (uint32_t)i);
    }
    torch::NoGradGuard ng;
    torch::Tensor xin = blob(x, {T, H}, st), out;
    const torch::Device cpu(torch::kCPU);
    switch (expert_type) {
      case MIXTRAL_MOE_DENSE_ACT_DENSE: { MixtralMoEDenseActDense m(dtype); m.SetTensorsFromBlob(nullptr, ids, cpu); out = m.forward(xin); break; }
      case DEEPSEEK_MOE_DENSE_ACT_DENSE: { DeepSeekMoEDenseActDense m(dtype); m.SetTensorsFromBlob(nullptr, ids, cpu); out = m.forward(xin); break; }
      case NLLB_MOE_DENSE_ACT_DENSE: { NllbMoeDenseActDense m(dtype); m.SetTensorsFromBlob(nullptr, ids, cpu); out = m.forward(xin); break; }
      case FSGPT_MOE_DENSE_ACT_DENSE: { FSGPTMoEDenseActDense m(dtype); m.SetTensorsFromBlob(nullptr, ids, cpu); out = m.forward(xin); break; }
      case SWITCH_TRANSFORMERS_DENSE_GATED_ACT_DENSE: { SwitchTransformersDenseGatedActDense m(dtype); m.SetTensorsFromBlob(nullptr, ids, cpu); out = m.forward(xin); break; }
      default: { SwitchTransformersDenseActDense m(dtype); m.SetTensorsFromBlob(nullptr, ids, cpu); out = m.forward(xin); break; }
    }
    out = out.contiguous();
    if (out.numel() != T * H) return 4;
    memcpy(y, out.data_ptr(), (size_t)out.numel() * out.element_size());
    for (auto id : ids) kTensorIndex->erase(id);
    return 0;
  } catch (...) {
    return 1;
  }
}

// ArcherTensorIndex::Serialize over n entries (dims: n x 8, row-major).  options = dtype on CPU, defaults otherwise
// (what prefetch_handle.offload stores for a CPU tensor, archer_tensor_handle.cpp:53-86).
int ref_index_write(const char* path, int n, const uint32_t* ids, const uint32_t* file_ids, const int64_t* offsets,
                    const uint64_t* sizes, const int32_t* ndims, const int64_t* dims, const int32_t* scalar_types) {
  try {
    ArcherTensorIndex idx;
    for (int i = 0; i < n; ++i) {
      TensorStorageMeta m;
      m.file_id = file_ids[i]; m.offset = offsets[i]; m.size = sizes[i];
      m.shape.assign(dims + (size_t)i * 8, dims + (size_t)i * 8 + ndims[i]);
      m.options = torch::TensorOptions().dtype(static_cast<c10::ScalarType>(scalar_types[i])).device(torch::kCPU);
      m.id = ids[i];
      idx.insert({ids[i], m});
    }
    idx.Serialize(path);
    return 0;
  } catch (...) {
    return 1;
  }
}

// ArcherTensorIndex::Deserialize; fills up to `capacity` entries, *n = entries in the file
int ref_index_read(const char* path, int capacity, int32_t* n, uint32_t* ids, uint32_t* file_ids, int64_t* offsets, uint64_t* sizes,
                   int32_t* ndims, int64_t* dims, int32_t* scalar_types) {
  try {
    ArcherTensorIndex idx;
    idx.Deserialize(path);
    *n = (int32_t)idx.size();
    int i = 0;
    for (auto& kv : idx) {
      if (i >= capacity) break;
      ids[i] = kv.first; file_ids[i] = kv.second.file_id; offsets[i] = kv.second.offset; sizes[i] = kv.second.size;
      ndims[i] = (int32_t)kv.second.shape.size();
      for (size_t d = 0; d < kv.second.shape.size() && d < 8; ++d) dims[(size_t)i * 8 + d] = kv.second.shape[d];
      scalar_types[i] = (int32_t)kv.second.options.dtype().toScalarType();
      ++i;
    }
    return 0;
  } catch (...) {
    return 1;
  }
}
}
