#!/bin/bash
# Reproduce the evidence under profiles/ on an MI355X box (run from the repo root).
# Every profiler invocation is wrapped in `timeout`: a hung python exit under rocprofv3 once burned
# 20 GPU-minutes.  PMC passes are separate runs with --kernel-trace only (no sys/hip/hsa trace domains).
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
# MFMA-pipe utilisation of the prefill GEMM kernels (512-token prompt, 8 layers)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$R/$OUT/pmc_mfma" -o m -- \
    python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --layers 8 --prompt 512 > /dev/null 2> "$R/$OUT/pmc_mfma.err")
python tools/mfma_summary.py "$OUT/pmc_mfma/m_counter_collection.csv" "$OUT/pmc_mfma/m_kernel_trace.csv" "$OUT/pmc_mfma_prefill_mixtral.json" > /dev/null 2> "$OUT/mfma_summary.err"
timeout 300 python bench.py --workload deepseek-v2-lite > "$OUT/bench_deepseek.json" 2> /dev/null
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/kt_ds" -o m -- \
    python "$R/bench.py" --workload deepseek-v2-lite --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> "$R/$OUT/kt_ds.err")
python tools/rocprof_summary.py "$OUT/kt_ds/m_kernel_stats.csv" "$OUT/kernel_stats_deepseek.csv"
timeout 300 python bench.py --workload nllb-moe-54b --batch 32 --prompt 0 --cpu-sample-layers 1 --cpu-sample-steps 1 > "$OUT/bench_nllb_b32.json" 2> /dev/null
timeout 200 python bench.py --workload switch-base-8 --cpu-sample-layers 2 --cpu-sample-steps 2 > "$OUT/bench_switch.json" 2> /dev/null
rm -rf "$OUT"/kt/*kernel_trace.csv "$OUT"/kt_ds/*kernel_trace.csv "$OUT"/pmc_*/m_kernel_trace.csv "$OUT"/pmc_*/m_counter_collection.csv 2>/dev/null
ls -la "$OUT"
