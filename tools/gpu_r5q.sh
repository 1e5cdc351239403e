#!/bin/bash
# round 5: gate + self-routing stage 1 (+ hidden shared expert) as ONE launch (moe_front1_kernel) — parity, timelines, A/B
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5q}; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chained.py tests/test_gpu_interface.py tests/test_gpu_fullsize.py -q -x -k "mixtral or Mixtral or deepseek or DeepSeek or selfrout or batch1 or chained or fp16" --deselect tests/test_gpu_fullsize.py::test_one_launch_decode_layer_is_parity_green > "$OUT/pytest.log" 2>&1; echo "pytest exit $?"; tail -4 "$OUT/pytest.log"
echo "== mixtral"; timeout 200 python tools/layer1_trace.py --mixtral "$OUT/trace_front1_mixtral.txt"
echo "== deepseek"; timeout 200 python tools/layer1_trace.py "$OUT/trace_front1_deepseek.txt"
for wl in mixtral-8x7b deepseek-v2-lite; do
 for v in 0 1 0 1; do
  MOEINF_FRONT1=$v timeout 200 python bench.py --workload $wl --no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --prompt 0 --windows 5 --no-traffic > "$OUT/bench_${wl}_front1_$v.json" 2> "$OUT/bench_${wl}_front1_$v.err"; echo "bench $wl FRONT1=$v exit $?"
  python - <<PY
import json
d=json.loads(open("$OUT/bench_${wl}_front1_$v.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$wl FRONT1=$v", d["ms_per_step"], d["windows_ms"], "stage1", k["ffn_stage1"]["avg_launch_us"], "stage2", k["ffn_stage2"]["avg_launch_us"], "route", k.get("route(gate+topk+index)",{}).get("avg_launch_us"))
PY
 done
done
