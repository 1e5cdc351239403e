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
timeout 300 python -m pytest tests -m gpu -q -x -k "batch1 or fused or decode_b1 or golden" > "$OUT/pytest.log" 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E  " "$OUT/pytest.log" | tail -5
# the shipped path against its switches (one line each; edit freely for a sweep)
run mixtral-8x7b "shipped" MOEINF_SELFROUTE=1
run mixtral-8x7b "stage 2: arrival-counter form instead of the pair kernel" MOEINF_DEC1_PAIR=0
run mixtral-8x7b "round-1 path (separate top-k/index launch)" MOEINF_SELFROUTE=0
run deepseek-v2-lite "shipped" MOEINF_SELFROUTE=1
run deepseek-v2-lite "round-1 path" MOEINF_SELFROUTE=0
