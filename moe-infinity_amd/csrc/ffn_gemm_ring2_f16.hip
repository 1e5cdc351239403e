// ffn_gemm_ring2_f16.hip — fp16 entry point of the register-ring GEMM (kernel: ffn_ring2_kernel.h).  Called by launch_ffn_t
// (kernels.hip) for fp16 experts.
#include "ffn_ring2_kernel.h"

namespace moeinf {

// fp16 experts: of the mid-sized GEMM kernels only ring2 is built for the f16 matrix instruction (the hybrid / LDS kernels are
// bf16 and fp32); same conditions as above, from 65 rows per expert on (gated) / 17 (plain).  false: not handled
bool launch_ffn_gemm_ring2_f16(const FfnStage& s, int nmat, dim3 grid, int max_rows, hipStream_t st) {
  static const Ring2Knobs knobs = Ring2Knobs::from_env();
  if ((nmat == 2) != (s.epi == EPI_GATED_SILU)) return false;
  if (s.rows_bound > 0 && s.rows_bound * s.ld_in >= (int64_t(1) << 32)) return false;  // 32-bit element offsets into the activations (xoff)
  const Ring2Form f = ring2_form(2, true, nmat, s.K, s.K_sh, (int)grid.x, (int)grid.y, max_rows, ring2_num_cus(), knobs);
  if (!f.ntb) return false;
  if (nmat == 2) launch_ring2<half_t, 2>(s, grid, f, st);
  else launch_ring2<half_t, 1>(s, grid, f, st);
  return true;
}

}  // namespace moeinf
