#!/bin/bash
# round 5, after the closing run: the one stale test of that run re-run after its fix, and rocprofv3 kernel stats at HEAD for the two
# workloads whose launches changed this round (DeepSeek-V2-Lite: moe_front1 + ffn2_decode1; Switch-base-8: moe_layer1_switch)
set -u
export TMPDIR=/tmp
R=$(pwd)
OUT=gpurun_out/${1:-r5x}; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -rf -k "error_paths or golden or batch1" > "$OUT/pytest_fix.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_fix.log"
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit" "$OUT/pytest_fix.log" | tail -5
LEAN="--no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --windows 1 --no-traffic"
for wl in deepseek-v2-lite switch-base-8; do
  tag=${wl//-/}; tag=${tag//./}
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/kt_$tag" -o m -- \
      python "$R/bench.py" --workload $wl --steps 10 --warmup 2 $LEAN > "$R/$OUT/kt_bench_$tag.json" 2> "$R/$OUT/kt_$tag.err")
  python tools/rocprof_summary.py "$OUT/kt_$tag/m_kernel_stats.csv" "$OUT/kernel_stats_$tag.csv"
  head -8 "$OUT/kernel_stats_$tag.csv"
done
rm -rf "$OUT"/kt_*/*kernel_trace.csv "$OUT"/kt_*/*.csv 2>/dev/null
