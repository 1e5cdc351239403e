#!/bin/bash
# round-2 GPU call A: full pytest -m gpu, default bench line, baseline kernel trace of DeepSeek-V2-Lite
set -u
export TMPDIR=/tmp
R=$(pwd)
OUT=gpurun_out/r02a
mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 1100 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
tail -15 "$OUT/pytest_gpu.log"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench exit $?"; tail -5 "$OUT/bench_default.err"; head -c 3000 "$OUT/bench_default.json"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/kt_ds" -o m -- \
    python "$R/bench.py" --workload deepseek-v2-lite --steps 10 --warmup 2 --no-cpu-baseline --windows 1 --miss-heavy-frac 0 > "$R/$OUT/kt_ds_bench.json" 2> "$R/$OUT/kt_ds.err")
python tools/rocprof_summary.py "$OUT/kt_ds/m_kernel_stats.csv" "$OUT/kernel_stats_deepseek.csv" && cat "$OUT/kernel_stats_deepseek.csv"
rm -rf "$OUT"/kt_ds/*kernel_trace.csv 2>/dev/null
