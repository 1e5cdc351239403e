#!/bin/bash
# round 5: the PERSISTENT one-launch decode layer — parity, timelines, timing
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5g}; mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -k "deepseek or DeepSeek" > "$OUT/pytest_deepseek.log" 2>&1; echo "pytest exit $?"
tail -4 "$OUT/pytest_deepseek.log"
for v in ${VARIANTS:-"X=0" "MOEINF_LAYER1_POLL=vector" "MOEINF_LAYER1_SLEEP=8" "MOEINF_LAYER1_PRE=4" "MOEINF_LAYER1_WPC=2"}; do
  echo "== $v"; env ${v//,/ } timeout 120 python tools/layer1_trace.py "$OUT/trace_${v//,/_}.txt"
done
SWEEP_ENVS="MOEINF_LAYER1=0;MOEINF_LAYER1=1;MOEINF_LAYER1_SLEEP=8;MOEINF_LAYER1_WPC=2" timeout 300 python tools/ffn_sweep.py deepseek_v2_lite:1:26 2>&1 | tee "$OUT/ffn_sweep_layer1.txt"
