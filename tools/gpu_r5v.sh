#!/bin/bash
# round 5: fp16 DeepSeek decode with the shared expert hidden under the router (gate_shared1 / route_shared2 / moe_front1 on half_t)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5v}; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -rf -k "fp16_experts_all_families" > "$OUT/pytest_fp16.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_fp16.log"
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -rf -s -k "deepseek_v2_lite_layer_fp16 and decode_b1" >> "$OUT/pytest_fp16.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_fp16.log"
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit|deepseek fp16" "$OUT/pytest_fp16.log" | tail -20
for hs in 1 0 1 0; do
  MOEINF_HIDE_SHARED=$hs timeout 300 python tools/fp16_deepseek_ab.py fp16 2>&1 | tail -1 | tee -a "$OUT/fp16_deepseek_decode_ab.txt"
done
timeout 300 python tools/fp16_deepseek_ab.py bf16 2>&1 | tail -1 | tee -a "$OUT/fp16_deepseek_decode_ab.txt"
