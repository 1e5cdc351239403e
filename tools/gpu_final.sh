#!/bin/bash
# Round evidence in one GPU call: full pytest -m gpu, the default bench line, rocprofv3 kernel stats (Mixtral, DeepSeek),
# PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs) and the MFMA-busy pass of the prefill kernels.
# Every profiler invocation is wrapped in `timeout`; PMC passes use --kernel-trace only (no sys/hip/hsa trace domains).
set -u
export TMPDIR=/tmp
R=$(pwd)
OUT=gpurun_out/${1:-final}
mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 1100 python -m pytest tests -m gpu -q -rf > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit" "$OUT/pytest_gpu.log" | tail -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench exit $?"; tail -3 "$OUT/bench_default.err"
LEAN="--no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --windows 1 --no-traffic"
for wl in mixtral-8x7b deepseek-v2-lite; do
  tag=${wl//-/}; tag=${tag//./}
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/kt_$tag" -o m -- \
      python "$R/bench.py" --workload $wl --steps 10 --warmup 2 $LEAN > "$R/$OUT/kt_bench_$tag.json" 2> "$R/$OUT/kt_$tag.err")
  python tools/rocprof_summary.py "$OUT/kt_$tag/m_kernel_stats.csv" "$OUT/kernel_stats_$tag.csv"
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/$OUT/pmc_${c}_$tag" -o m -- \
        python "$R/bench.py" --workload $wl --steps 3 --warmup 1 $LEAN --layers 8 --prompt 0 > /dev/null 2> "$R/$OUT/pmc_${c}_$tag.err")
  done
  python tools/pmc_summary.py "$OUT/pmc_FETCH_SIZE_$tag/m_counter_collection.csv" "$OUT/pmc_WRITE_SIZE_$tag/m_counter_collection.csv" "$OUT/pmc_traffic_$tag.json" > /dev/null
done
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$R/$OUT/pmc_mfma" -o m -- \
    python "$R/bench.py" --steps 2 --warmup 1 $LEAN --layers 8 --prompt 512 > /dev/null 2> "$R/$OUT/pmc_mfma.err")
python tools/mfma_summary.py "$OUT/pmc_mfma/m_counter_collection.csv" "$OUT/pmc_mfma/m_kernel_trace.csv" "$OUT/pmc_mfma_prefill_mixtral8x7b.json" > /dev/null 2> "$OUT/mfma_summary.err"
# compute-bound regime: 4096 tokens through one Mixtral layer (ffn_gemm_big), MFMA-busy + kernel durations
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$R/$OUT/pmc_mfma4096" -o m -- \
    python "$R/tools/prefill_once.py" mixtral_8x7b 4096 1 6 > /dev/null 2> "$R/$OUT/pmc_mfma4096.err")
python tools/mfma_summary.py "$OUT/pmc_mfma4096/m_counter_collection.csv" "$OUT/pmc_mfma4096/m_kernel_trace.csv" "$OUT/pmc_mfma_prefill4096_mixtral8x7b.json" > /dev/null 2> "$OUT/mfma4096_summary.err"
SWEEP_ENVS="A=1;MOEINF_GEMM_BIG_MODE=1;MOEINF_GEMM_BIG=0" timeout 900 python tools/ffn_sweep.py mixtral_8x7b:512:2 mixtral_8x7b:2048:2 mixtral_8x7b:4096:2 deepseek_v2_lite:4096:4 > "$OUT/ffn_sweep_prefill_final.txt" 2>&1
for wl in mixtral-8x7b deepseek-v2-lite; do
  timeout 400 python bench.py --workload $wl --force-ep --no-other-configs --miss-heavy-frac 0 --prompt 0 > "$OUT/bench_ep1_$wl.json" 2> "$OUT/bench_ep1_$wl.err"
done
rm -rf "$OUT"/kt_*/*kernel_trace.csv "$OUT"/pmc_*/m_kernel_trace.csv "$OUT"/pmc_*/m_counter_collection.csv "$OUT"/kt_*/*.csv 2>/dev/null
head -12 "$OUT/kernel_stats_mixtral8x7b.csv"; head -12 "$OUT/kernel_stats_deepseekv2lite.csv"
ls "$OUT"
