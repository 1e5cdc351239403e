#!/bin/bash
# round 5: Mixtral, front launch vs three launches once more (after the one-token gate reduction), A/B/A/B + timeline
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5t}; mkdir -p "$OUT"
MOEINF_FRONT1=1 timeout 200 python tools/layer1_trace.py --mixtral "$OUT/trace_front1_mixtral.txt"
for v in 0 1 0 1; do
  MOEINF_FRONT1=$v timeout 200 python bench.py --no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --prompt 0 --windows 5 --no-traffic > "$OUT/bench_front1_$v.json" 2> "$OUT/bench_front1_$v.err"; echo "bench FRONT1=$v exit $?"
  python - <<PY
import json
d=json.loads(open("$OUT/bench_front1_$v.json").read().strip().splitlines()[-1])
print("mixtral FRONT1=$v", d["ms_per_step"], d["windows_ms"])
PY
done
