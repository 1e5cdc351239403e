#!/bin/bash
# full GPU suite + prefill sweep + lean decode lines (after a change that touches every kernel)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3n}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 1100 python -m pytest tests -m gpu -q -rf > "$OUT/pytest_gpu.log" 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E  " "$OUT/pytest_gpu.log" | tail -14
SWEEP_ENVS="${SWEEP_ENVS:-A=1}" timeout 900 python tools/ffn_sweep.py ${2:-mixtral_8x7b:4096:2 deepseek_v2_lite:4096:4 mixtral_8x7b:2048:2 mixtral_8x7b:512:2} 2>&1 | tee "$OUT/ffn_sweep.txt" | tail -20
LEAN="--no-cpu-baseline --no-other-configs --miss-heavy-frac 0"
for wl in mixtral-8x7b deepseek-v2-lite; do
  timeout 300 python bench.py --workload $wl $LEAN > "$OUT/bench_$wl.json" 2> "$OUT/bench_$wl.err"
  python - "$OUT/bench_$wl.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["workload"], d["ms_per_step"], "parity", (d.get("parity") or {}).get("ok"), "prefill", (d.get("prefill") or {}))
PY
done
