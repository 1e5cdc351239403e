#!/bin/bash
# round 5: one-token gate reduction (TT = 1) in the front / Switch launches — parity + timelines + A/B of DeepSeek against the previous build is across calls
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5r}; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or selfrout or one_launch or ties" > "$OUT/pytest.log" 2>&1; echo "pytest exit $?"; tail -3 "$OUT/pytest.log"
echo "== deepseek front"; timeout 200 python tools/layer1_trace.py "$OUT/trace_front1_deepseek.txt"
echo "== switch"; timeout 200 python tools/layer1_trace.py --switch "$OUT/trace_switch.txt"
for wl in deepseek-v2-lite switch-base-8; do
  timeout 200 python bench.py --workload $wl --no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --prompt 0 --windows 5 --no-traffic > "$OUT/bench_$wl.json" 2> "$OUT/bench_$wl.err"; echo "bench $wl exit $?"
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$wl.json").read().strip().splitlines()[-1])
print("$wl", d["ms_per_step"], d["windows_ms"])
PY
done
