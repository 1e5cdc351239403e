#!/bin/bash
# round-4 closing evidence at HEAD: full pytest -m gpu, smoke, the default bench line, rocprofv3 kernel stats of the bench (Mixtral),
# MFMA-busy of the 512-token prefill kernels (ring2).  Every step under its own timeout.
set -u
export TMPDIR=/tmp
R=$(pwd)
OUT=gpurun_out/${1:-r4v}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 480 python -m pytest tests -m gpu -q -rf > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit" "$OUT/pytest_gpu.log" | tail -12
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
timeout 300 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench exit $?"; tail -2 "$OUT/bench_default.err"
LEAN="--no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --windows 1 --no-traffic"
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/kt_mixtral8x7b" -o m -- \
    python "$R/bench.py" --workload mixtral-8x7b --steps 10 --warmup 2 $LEAN > "$R/$OUT/kt_bench_mixtral8x7b.json" 2> "$R/$OUT/kt_mixtral8x7b.err")
python tools/rocprof_summary.py "$OUT/kt_mixtral8x7b/m_kernel_stats.csv" "$OUT/kernel_stats_mixtral8x7b.csv"
(cd /tmp && timeout 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$R/$OUT/pmc_mfma" -o m -- \
    python "$R/bench.py" --steps 2 --warmup 1 $LEAN --layers 8 --prompt 512 > /dev/null 2> "$R/$OUT/pmc_mfma.err")
python tools/mfma_summary.py "$OUT/pmc_mfma/m_counter_collection.csv" "$OUT/pmc_mfma/m_kernel_trace.csv" "$OUT/pmc_mfma_prefill512_mixtral8x7b.json" > /dev/null 2> "$OUT/mfma_summary.err"
rm -rf "$OUT"/kt_*/*kernel_trace.csv "$OUT"/pmc_*/m_kernel_trace.csv "$OUT"/pmc_*/m_counter_collection.csv "$OUT"/kt_*/*.csv 2>/dev/null
head -12 "$OUT/kernel_stats_mixtral8x7b.csv"
python - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("metric","value","ms_per_step")}, d.get("roofline"), d.get("prefill"))
PY
