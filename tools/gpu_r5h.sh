#!/bin/bash
# round 5: persistent one-launch layer, two routed stage-2 items at a time: timelines + timing of the register budgets
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5h}; mkdir -p "$OUT"
for v in "X=0" "MOEINF_LAYER1_WPE=2" "MOEINF_LAYER1_PRE=4"; do
  echo "== $v"; env $v timeout 120 python tools/layer1_trace.py "$OUT/trace_$v.txt"
done
SWEEP_ENVS="MOEINF_LAYER1=0;MOEINF_LAYER1=1;MOEINF_LAYER1_WPE=2;MOEINF_LAYER1_PRE=4" timeout 300 python tools/ffn_sweep.py deepseek_v2_lite:1:26 2>&1 | tee "$OUT/ffn_sweep_layer1.txt"
for v in "MOEINF_LAYER1=0" "MOEINF_LAYER1=1" "MOEINF_LAYER1_WPE=2"; do
  env $v timeout 200 python bench.py --workload deepseek-v2-lite --no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --prompt 0 --windows 3 > "$OUT/bench_ds_$v.json" 2> "$OUT/bench_ds_$v.err"; echo "bench $v exit $?"
  python - <<PY
import json
d=json.loads(open("$OUT/bench_ds_$v.json").read().strip().splitlines()[-1])
print("$v", d["ms_per_step"], d["windows_ms"])
PY
done
