// ffn_gemm_ring2.hip — bf16 entry point of the register-ring GEMM (kernel: ffn_ring2_kernel.h).  Which form a stage takes:
// ring2_form (kernels.h).  Called by launch_ffn_gemm (ffn_gemm.hip); false: not handled.
#include "ffn_ring2_kernel.h"

namespace moeinf {

bool launch_ffn_gemm_ring2_bf16(const FfnStage& s, int nmat, dim3 grid, int max_rows, hipStream_t st) {
  static const Ring2Knobs knobs = Ring2Knobs::from_env();
  if ((nmat == 2) != (s.epi == EPI_GATED_SILU)) return false;
  if (s.rows_bound > 0 && s.rows_bound * s.ld_in >= (int64_t(1) << 32)) return false;  // 32-bit element offsets into the activations (xoff)
  const Ring2Form f = ring2_form(2, false, nmat, s.K, s.K_sh, (int)grid.x, (int)grid.y, max_rows, ring2_num_cus(), knobs);
  if (!f.ntb) return false;
  if (nmat == 2) launch_ring2<uint16_t, 2>(s, grid, f, st);
  else launch_ring2<uint16_t, 1>(s, grid, f, st);
  return true;
}

}  // namespace moeinf
