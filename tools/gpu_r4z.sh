#!/bin/bash
# lean bench line at HEAD: checks the profiled prefill pass (prefill.kernels)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4z}; mkdir -p "$OUT"
timeout 90 python bench.py --no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --windows 1 > "$OUT/bench_lean.json" 2> "$OUT/bench_lean.err"; echo "bench exit $?"
tail -3 "$OUT/bench_lean.err"
python - <<PY
import json
d=json.loads(open("$OUT/bench_lean.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], json.dumps(d.get("prefill")))
PY
