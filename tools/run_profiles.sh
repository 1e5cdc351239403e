#!/bin/bash
# Reproduce the evidence under profiles/ on an MI355X box (run from the repo root).
# Every profiler invocation is wrapped in `timeout`: a hung python exit under rocprofv3 once burned
# 20 GPU-minutes.
set -u
export TMPDIR=/tmp
R=$(pwd)
OUT=${1:-gpurun_out/profiles_run}
mkdir -p "$OUT"
timeout 600 python bench.py > "$OUT/bench_mixtral.json" 2> "$OUT/bench_mixtral.err"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/kt" -o m -- \
    python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$R/$OUT/kt_bench.json" 2> "$R/$OUT/kt.err")
python tools/rocprof_summary.py "$OUT/kt/m_kernel_stats.csv" "$OUT/kernel_stats_mixtral.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/$OUT/pmc_$c" -o m -- \
      python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --layers 8 --prompt 0 > /dev/null 2> "$R/$OUT/pmc_$c.err")
done
python tools/pmc_summary.py "$OUT/pmc_FETCH_SIZE/m_counter_collection.csv" "$OUT/pmc_WRITE_SIZE/m_counter_collection.csv" "$OUT/pmc_traffic_mixtral.json" > /dev/null
timeout 300 python bench.py --workload deepseek-v2-lite > "$OUT/bench_deepseek.json" 2> /dev/null
timeout 300 python bench.py --workload nllb-moe-54b --batch 32 --prompt 0 --cpu-sample-layers 1 --cpu-sample-steps 1 > "$OUT/bench_nllb_b32.json" 2> /dev/null
ls -la "$OUT"
