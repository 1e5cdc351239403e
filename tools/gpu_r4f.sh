#!/bin/bash
# round 4, call F: the batch-1 broadcast form of the exchange — tests (world 1, real processes) and the forced-EP lines
set -u
export TMPDIR=/tmp
R=$(pwd)
OUT=gpurun_out/r4f; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 1200 python -m pytest tests/test_gpu_ep_peer.py tests/test_gpu_ep_processes.py "tests/test_gpu_interface.py::test_expert_parallel_module_through_rccl_world_size_1" -m gpu -q -rf -s > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit|^E  " "$OUT/pytest_gpu.log" | cut -c1-300 | tail -20
for wl in mixtral-8x7b deepseek-v2-lite; do
  timeout 300 python bench.py --workload $wl --force-ep --ep-transport peer-store --no-other-configs --miss-heavy-frac 0 --prompt 0 --cpu-sample-layers 4 --cpu-sample-steps 3 > "$OUT/bench_ep1_${wl}_peer-store.json" 2> "$OUT/bench_ep1_${wl}.err"
  python - "$OUT/bench_ep1_${wl}_peer-store.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("forced EP", d["ms_per_step"], d["ep_transport"]["chosen"], d["parity"]["ok"], d["parity"]["fp32_exact_arm"]["ratio_over_the_sample"], d.get("ep_phases_us_per_layer"))
except Exception as ex: print("no line", ex)
PY
  tail -2 "$OUT/bench_ep1_${wl}.err"
done
LEAN="--no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --windows 1 --prompt 0"
for wl in mixtral-8x7b deepseek-v2-lite; do
  tag=${wl//-/}; tag=${tag//./}
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/kt_ep_$tag" -o m -- \
      python "$R/bench.py" --workload $wl --force-ep --ep-transport peer-store --steps 10 --warmup 2 $LEAN > "$R/$OUT/kt_ep_bench_$tag.json" 2> "$R/$OUT/kt_ep_$tag.err")
  python tools/rocprof_summary.py "$OUT/kt_ep_$tag/m_kernel_stats.csv" "$OUT/kernel_stats_ep_peer_$tag.csv" 2>/dev/null || cp "$OUT/kt_ep_$tag/m_kernel_stats.csv" "$OUT/kernel_stats_ep_peer_$tag.csv"
  head -9 "$OUT/kernel_stats_ep_peer_$tag.csv" | cut -c1-160
done
rm -rf "$OUT"/kt_*/*kernel_trace.csv "$OUT"/kt_*/*.csv 2>/dev/null
