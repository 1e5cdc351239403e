// ffn_gemm.hip — bf16 and fp32 entry points of the grouped-GEMM kernels for experts with many rows (kernels and the
// dispatch between them: ffn_gemm_kernels.h).  Called by launch_ffn_stage (kernels.hip).
#include "ffn_gemm_kernels.h"

namespace moeinf {

template bool launch_ffn_gemm<uint16_t, 1>(const FfnStage&, dim3, int, hipStream_t);
template bool launch_ffn_gemm<uint16_t, 2>(const FfnStage&, dim3, int, hipStream_t);
template bool launch_ffn_gemm<float, 1>(const FfnStage&, dim3, int, hipStream_t);
template bool launch_ffn_gemm<float, 2>(const FfnStage&, dim3, int, hipStream_t);

}  // namespace moeinf
