#!/bin/bash
# round 5: one-token gate reduction in the plain gate launch (Mixtral batch 1) — parity + bench
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5s}; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or selfrout or ties" > "$OUT/pytest.log" 2>&1; echo "pytest exit $?"; tail -3 "$OUT/pytest.log"
for i in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --prompt 0 --windows 5 --no-traffic > "$OUT/bench_mixtral_$i.json" 2> "$OUT/bench_mixtral_$i.err"; echo "bench exit $?"
  python - <<PY
import json
d=json.loads(open("$OUT/bench_mixtral_$i.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("mixtral", d["ms_per_step"], d["windows_ms"], "route", k["route(gate+topk+index)"]["avg_launch_us"], "stage1", k["ffn_stage1"]["avg_launch_us"])
PY
done
