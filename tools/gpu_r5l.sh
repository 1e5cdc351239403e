#!/bin/bash
# round 5: Switch-base-8 batch-1 decode as ONE launch per layer — parity, timeline, A/B against the three launches
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5l}; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_chained.py tests/test_gpu_interface.py -q -x -k "switch or Switch" > "$OUT/pytest_switch.log" 2>&1; echo "pytest exit $?"; tail -4 "$OUT/pytest_switch.log"
timeout 120 python tools/layer1_trace.py --switch "$OUT/trace_switch.txt"
for v in 0 1 0 1; do
  MOEINF_LAYER1_SWITCH=$v timeout 200 python bench.py --workload switch-base-8 --no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --prompt 0 --windows 5 --no-traffic > "$OUT/bench_switch_$v.json" 2> "$OUT/bench_switch_$v.err"; echo "bench LAYER1_SWITCH=$v exit $?"
  python - <<PY
import json
d=json.loads(open("$OUT/bench_switch_$v.json").read().strip().splitlines()[-1])
print("LAYER1_SWITCH=$v", d["ms_per_step"], d["windows_ms"])
PY
done
