#!/bin/bash
# round 5: every short last pass of ffn_gemm_big behind the full passes (split when it holds <= 64 tokens): parity, then A/B
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5s5}; mkdir -p "$OUT"
timeout 400 python -m pytest tests/test_gpu_parity.py -q -rf -x -k "short_last_pass or compute_bound_grouped or long_prefill" > "$OUT/pytest_split.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_split.log"
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit" "$OUT/pytest_split.log" | tail -8
if grep -q "pytest exit 0" "$OUT/pytest_split.log"; then
  SWEEP_ENVS="A=1;MOEINF_GEMM_BIG_SPLIT=0;A=2;MOEINF_GEMM_BIG_SPLIT=0" timeout 400 python tools/ffn_sweep.py mixtral_8x7b:2048:2 mixtral_8x7b:3072:2 mixtral_8x7b:4096:2 mixtral_8x7b:4224:2 2>&1 | tee "$OUT/big_split_ab.txt" | tail -24
fi
