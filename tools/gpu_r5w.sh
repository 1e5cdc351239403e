#!/bin/bash
# round 5: fp16 Mixtral batch-1 decode with the K = 2 pair form of stage 2 (ffn2_decode1_pair_kernel<half_t>)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5w}; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -rf -k "fp16_experts_all_families and mixtral" > "$OUT/pytest_fp16.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_fp16.log"
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -rf -s -k "mixtral_8x7b_layer_fp16" >> "$OUT/pytest_fp16.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_fp16.log"
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit|mixtral fp16" "$OUT/pytest_fp16.log" | tail -20
for pr in 1 0 1 0; do
  MOEINF_DEC1_PAIR=$pr timeout 300 python tools/fp16_mixtral_ab.py fp16 2>&1 | tail -1 | tee -a "$OUT/fp16_mixtral_decode_pair_ab.txt"
done
timeout 300 python tools/fp16_mixtral_ab.py bf16 2>&1 | tail -1 | tee -a "$OUT/fp16_mixtral_decode_pair_ab.txt"
