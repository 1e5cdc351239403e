#!/bin/bash
# HEAD check after the split tail + 32-bit activation offsets: every full-size parity test, smoke, a lean bench line (prefill number)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4x}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 200 python -m pytest tests/test_gpu_fullsize.py -q 2>&1 | tail -3 | tee "$OUT/fullsize.txt"
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee "$OUT/smoke.txt"
timeout 120 python bench.py --no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --windows 2 > "$OUT/bench_lean.json" 2> "$OUT/bench_lean.err"; echo "bench exit $?"
python - <<PY
import json
d=json.loads(open("$OUT/bench_lean.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d.get("prefill"), d.get("parity",{}).get("ok"))
PY
