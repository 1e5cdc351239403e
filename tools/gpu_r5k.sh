#!/bin/bash
# round 5: batch-1 Mixtral decode with the gate inside FFN stage 1 (two launches per layer) — parity, then A/B/A/B in one run
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5k}; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chained.py tests/test_gpu_interface.py -q -x -k "mixtral or Mixtral or selfrout or batch1 or chained" > "$OUT/pytest_mixtral.log" 2>&1; echo "pytest exit $?"; tail -4 "$OUT/pytest_mixtral.log"
for v in 0 1 0 1; do
  MOEINF_SR_GATE_IN=$v timeout 200 python bench.py --no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --prompt 0 --windows 5 --no-traffic > "$OUT/bench_gate_in_$v.json" 2> "$OUT/bench_gate_in_$v.err"; echo "bench GATE_IN=$v exit $?"
  python - <<PY
import json
d=json.loads(open("$OUT/bench_gate_in_$v.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("GATE_IN=$v", d["ms_per_step"], d["windows_ms"], "stage1", k["ffn_stage1"]["avg_launch_us"], "stage2", k["ffn_stage2"]["avg_launch_us"], "route", k.get("route(gate+topk+index)",{}).get("avg_launch_us"))
PY
done
