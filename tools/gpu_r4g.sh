#!/bin/bash
# round 4 evidence in one GPU call: full pytest -m gpu, smoke, the default bench line, rocprofv3 kernel stats (Mixtral, DeepSeek,
# Switch), PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs), MFMA-busy of the prefill kernels.
# Every profiler invocation is wrapped in `timeout`; PMC passes use --kernel-trace only (no sys/hip/hsa trace domains).
set -u
export TMPDIR=/tmp
R=$(pwd)
OUT=gpurun_out/${1:-r4g}
mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 1500 python -m pytest tests -m gpu -q -rf > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit" "$OUT/pytest_gpu.log" | tail -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench exit $?"; tail -3 "$OUT/bench_default.err"
LEAN="--no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --windows 1 --no-traffic"
for wl in mixtral-8x7b deepseek-v2-lite switch-base-8; do
  tag=${wl//-/}; tag=${tag//./}
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/kt_$tag" -o m -- \
      python "$R/bench.py" --workload $wl --steps 10 --warmup 2 $LEAN > "$R/$OUT/kt_bench_$tag.json" 2> "$R/$OUT/kt_$tag.err")
  python tools/rocprof_summary.py "$OUT/kt_$tag/m_kernel_stats.csv" "$OUT/kernel_stats_$tag.csv"
done
for wl in mixtral-8x7b deepseek-v2-lite; do
  tag=${wl//-/}; tag=${tag//./}
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/$OUT/pmc_${c}_$tag" -o m -- \
        python "$R/bench.py" --workload $wl --steps 3 --warmup 1 $LEAN --layers 8 --prompt 0 > /dev/null 2> "$R/$OUT/pmc_${c}_$tag.err")
  done
  python tools/pmc_summary.py "$OUT/pmc_FETCH_SIZE_$tag/m_counter_collection.csv" "$OUT/pmc_WRITE_SIZE_$tag/m_counter_collection.csv" "$OUT/pmc_traffic_$tag.json" > /dev/null
done
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$R/$OUT/pmc_mfma" -o m -- \
    python "$R/bench.py" --steps 2 --warmup 1 $LEAN --layers 8 --prompt 512 > /dev/null 2> "$R/$OUT/pmc_mfma.err")
python tools/mfma_summary.py "$OUT/pmc_mfma/m_counter_collection.csv" "$OUT/pmc_mfma/m_kernel_trace.csv" "$OUT/pmc_mfma_prefill512_mixtral8x7b.json" > /dev/null 2> "$OUT/mfma_summary.err"
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$R/$OUT/pmc_mfma4096" -o m -- \
    python "$R/tools/prefill_once.py" mixtral_8x7b 4096 1 6 > /dev/null 2> "$R/$OUT/pmc_mfma4096.err")
python tools/mfma_summary.py "$OUT/pmc_mfma4096/m_counter_collection.csv" "$OUT/pmc_mfma4096/m_kernel_trace.csv" "$OUT/pmc_mfma_prefill4096_mixtral8x7b.json" > /dev/null 2> "$OUT/mfma4096_summary.err"
rm -rf "$OUT"/kt_*/*kernel_trace.csv "$OUT"/pmc_*/m_kernel_trace.csv "$OUT"/pmc_*/m_counter_collection.csv "$OUT"/kt_*/*.csv 2>/dev/null
head -8 "$OUT/kernel_stats_mixtral8x7b.csv"; head -8 "$OUT/kernel_stats_deepseekv2lite.csv"; head -6 "$OUT/kernel_stats_switchbase8.csv"
cat "$OUT"/pmc_traffic_*.json | head -40
ls "$OUT"
