#!/bin/bash
# Reproduce the evidence under profiles/ on an MI355X box (run from the repo root):  tools/run_profiles.sh [outdir]
# Every profiler invocation is wrapped in `timeout`.  PMC passes are separate runs with --kernel-trace only (no
# sys/hip/hsa trace domains), one counter set per run, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
set -u
export TMPDIR=/tmp
R=$(pwd)
OUT=${1:-gpurun_out/profiles_run}
mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
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
# MFMA-pipe utilisation of the prefill GEMM kernels (512-token prompt, 8 layers)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$R/$OUT/pmc_mfma" -o m -- \
    python "$R/bench.py" --steps 2 --warmup 1 $LEAN --layers 8 --prompt 512 > /dev/null 2> "$R/$OUT/pmc_mfma.err")
python tools/mfma_summary.py "$OUT/pmc_mfma/m_counter_collection.csv" "$OUT/pmc_mfma/m_kernel_trace.csv" "$OUT/pmc_mfma_prefill_mixtral8x7b.json" > /dev/null 2> "$OUT/mfma_summary.err"
rm -rf "$OUT"/kt_*/*kernel_trace.csv "$OUT"/pmc_*/m_kernel_trace.csv "$OUT"/pmc_*/m_counter_collection.csv 2>/dev/null
ls -la "$OUT"
