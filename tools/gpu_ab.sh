#!/bin/bash
# A/B sweeps of the batch-1 decode path on the headline workloads (env knobs), one lean bench line each
set -u
OUT=gpurun_out/${1:-ab}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
LEAN="--no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --windows 3 --prompt 0"
run() {
  local wl=$1; local label=$2; shift 2
  env "$@" timeout 200 python bench.py --workload $wl $LEAN > "$OUT/b.json" 2> "$OUT/b.err" || tail -3 "$OUT/b.err"
  python - "$OUT/b.json" "$wl $label" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["kernels"]
print(f"{sys.argv[2]:52s} ms/step {d['ms_per_step']:.4f}", {n:v["avg_launch_us"] for n,v in k.items() if isinstance(v,dict)})
PY
}
run mixtral-8x7b "selfroute" MOEINF_SR_DEBUG=0
run mixtral-8x7b "selfroute, routing skipped (experts 0..K-1)" MOEINF_SR_DEBUG=1
run mixtral-8x7b "old path" MOEINF_SELFROUTE=0
run deepseek-v2-lite "selfroute" MOEINF_SR_DEBUG=0
run deepseek-v2-lite "selfroute, routing skipped" MOEINF_SR_DEBUG=1
