#!/bin/bash
# round 5: timeline of the one-launch decode layer under different poll intervals
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5e}; mkdir -p "$OUT"
for v in ${VARIANTS:-"X=0" "MOEINF_LAYER1_SLEEP=8" "MOEINF_LAYER1_SLEEP=32" "MOEINF_LAYER1_SLEEP=128"}; do
  echo "== $v"; env ${v//,/ } timeout 120 python tools/layer1_trace.py "$OUT/trace_${v//,/_}.txt"
done
