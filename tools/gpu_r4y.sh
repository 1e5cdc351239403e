#!/bin/bash
# rocprofv3 kernel stats of the lean bench line at HEAD (ring2 with the split tail)
set -u
export TMPDIR=/tmp
R=$(pwd)
OUT=gpurun_out/${1:-r4y}; mkdir -p "$OUT"
LEAN="--no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --windows 1 --no-traffic"
(cd /tmp && timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/kt" -o m -- \
    python "$R/bench.py" --workload mixtral-8x7b --steps 10 --warmup 2 $LEAN > "$R/$OUT/kt_bench.json" 2> "$R/$OUT/kt.err")
python tools/rocprof_summary.py "$OUT/kt/m_kernel_stats.csv" "$OUT/kernel_stats_mixtral8x7b.csv"
rm -rf "$OUT/kt"
head -8 "$OUT/kernel_stats_mixtral8x7b.csv"
