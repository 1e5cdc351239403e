#!/bin/bash
# the real FFN kernels under a back-to-back harness (tools/ffn_micro.hip), DeepSeek-V2-Lite decode shapes
set -u
OUT=gpurun_out/${1:-ffnmicro}; mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I moe-infinity_amd/csrc -o /tmp/ffn_micro tools/ffn_micro.hip 2> "$OUT/build.err" || { tail -20 "$OUT/build.err"; exit 1; }
{ echo "== with shared expert as 7th active expert"; /tmp/ffn_micro 1; echo "== routed only"; /tmp/ffn_micro 0; } 2>&1 | tee "$OUT/ffn_micro.txt"
