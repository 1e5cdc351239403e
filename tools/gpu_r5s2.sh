#!/bin/bash
# round 5: short last passes of ffn_gemm_big split along the weight rows (MOEINF_GEMM_BIG_SPLIT): parity, then A/B at ragged token counts
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5s2}; mkdir -p "$OUT"
timeout 500 python -m pytest tests/test_gpu_parity.py -q -rf -x -k "short_last_pass or compute_bound_grouped or long_prefill" > "$OUT/pytest_split.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_split.log"
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit" "$OUT/pytest_split.log" | tail -8
if grep -q "pytest exit 0" "$OUT/pytest_split.log"; then
  SWEEP_ENVS="A=1;MOEINF_GEMM_BIG_SPLIT=0;MOEINF_GEMM_BIG_SPLIT=1;MOEINF_GEMM_BIG_SPLIT=2" timeout 500 python tools/ffn_sweep.py mixtral_8x7b:4096:2 mixtral_8x7b:4224:2 mixtral_8x7b:2048:2 mixtral_8x7b:3840:2 deepseek_v2_lite:4096:4 2>&1 | tee "$OUT/big_split_ab.txt" | tail -24
fi
