#!/bin/bash
# round 4, call D: fp16 experts + Switch self-routing + fp32 exact-arm factor on the GPU; default bench line; kernel trace of the
# forced-EP lines (peer-store); DeepSeek speculation with lower request thresholds
set -u
export TMPDIR=/tmp
R=$(pwd)
OUT=gpurun_out/r4d; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_ep_peer.py tests/test_gpu_interface.py tests/test_gpu_chained.py -m gpu -q -rf -s > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit|fp16 t=|switch " "$OUT/pytest_gpu.log" | tail -30
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench exit $?"; tail -2 "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("main", d["ms_per_step"], d["roofline"]["frac"], d["parity"]["ok"], d["parity"].get("fp32_exact_arm"))
for o in d.get("other_configs", []):
    print(o["config"]["workload"][:40], o["ms_per_step"], o.get("parity",{}).get("ok"), o.get("parity",{}).get("fp32_exact_arm",{}).get("ratio_over_the_sample"), o.get("parity",{}).get("pairs_checked"))
PY
LEAN="--no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --windows 1 --prompt 0"
for wl in mixtral-8x7b deepseek-v2-lite; do
  tag=${wl//-/}; tag=${tag//./}
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/kt_ep_$tag" -o m -- \
      python "$R/bench.py" --workload $wl --force-ep --ep-transport peer-store --steps 10 --warmup 2 $LEAN > "$R/$OUT/kt_ep_bench_$tag.json" 2> "$R/$OUT/kt_ep_$tag.err")
  python tools/rocprof_summary.py "$OUT/kt_ep_$tag/m_kernel_stats.csv" "$OUT/kernel_stats_ep_peer_$tag.csv" 2>/dev/null || cp "$OUT/kt_ep_$tag/m_kernel_stats.csv" "$OUT/kernel_stats_ep_peer_$tag.csv"
  head -9 "$OUT/kernel_stats_ep_peer_$tag.csv" | cut -c1-200
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/kt_switch" -o m -- \
    python "$R/bench.py" --workload switch-base-8 --steps 20 --warmup 5 $LEAN > "$R/$OUT/kt_bench_switch.json" 2> "$R/$OUT/kt_switch.err")
python tools/rocprof_summary.py "$OUT/kt_switch/m_kernel_stats.csv" "$OUT/kernel_stats_switchbase8.csv" 2>/dev/null || cp "$OUT/kt_switch/m_kernel_stats.csv" "$OUT/kernel_stats_switchbase8.csv"
head -8 "$OUT/kernel_stats_switchbase8.csv" | cut -c1-200
P="lfu,lfu+engine_predictor_la1,lfu+engine_predictor_la2+governor"
for ms in 0.08 0.04; do
timeout 300 python tools/prefetch_study.py --workload deepseek_v2_lite --layers 26 --cache-frac 0.25 --attn-us 270 --min-share $ms --max-prefetch 8 --policies $P --note "min_share $ms, attention 270 us (batch 1, 2k context)" > "$OUT/ps_deepseek_b1_ctx2k_share$ms.jsonl" 2> "$OUT/ps_$ms.err"
python - "$OUT/ps_deepseek_b1_ctx2k_share$ms.jsonl" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(d["policy"], d["ms_per_token"], "hit", d["hit_rate"], "pf", d["prefetch_issued"], d["prefetch_useful"], "exposed", d["exposed_wait_ms"])
PY
done
rm -rf "$OUT"/kt_*/*kernel_trace.csv "$OUT"/kt_*/*.csv 2>/dev/null
