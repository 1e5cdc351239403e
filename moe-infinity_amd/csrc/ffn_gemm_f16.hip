// ffn_gemm_f16.hip — fp16 entry points (the reference's expert dtype id 2, core/parallel/expert_module.h:20-23) of the
// grouped-GEMM kernels for experts with many rows: the hybrid (17 .. 64 / 128 rows per expert) and the LDS-staged kernel
// (.. 256 rows) on the f16 matrix instruction, so that the short-reduction families (DeepSeek-V2-Lite, NLLB's first stage)
// run fp16 at every size without falling back to the decode kernel looping over token tiles (round 5; until then only the
// register ring and the 256 x 256 kernel were built for fp16).  Kernels: ffn_gemm_kernels.h.
#include "ffn_gemm_kernels.h"

namespace moeinf {

template bool launch_ffn_gemm<half_t, 1>(const FfnStage&, dim3, int, hipStream_t);
template bool launch_ffn_gemm<half_t, 2>(const FfnStage&, dim3, int, hipStream_t);

}  // namespace moeinf
