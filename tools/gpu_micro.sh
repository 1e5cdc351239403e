#!/bin/bash
# small-launch streaming microbenchmark (tools/small_stream.hip) on the shapes of DeepSeek-V2-Lite's decode launches
set -u
OUT=gpurun_out/${1:-micro}; mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/small_stream tools/small_stream.hip 2> "$OUT/build.err" || { cat "$OUT/build.err"; exit 1; }
{
echo "== stage 2 routed (6 experts x 128 row groups x 2816 B)"; /tmp/small_stream 2816 128 6
echo "== stage 1 routed, both matrices as one 8 KiB row (6 x 88 x 8192 B)"; /tmp/small_stream 8192 88 6
echo "== Mixtral stage 2 (2 x 256 x 28672 B)"; /tmp/small_stream 28672 256 2
} 2>&1 | tee "$OUT/small_stream.txt"
