#!/bin/bash
# round 4, call H: small-batch self-routing (2..8 tokens) — the tests that decode such batches, then A/B at batch 4 and 8
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4h; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_serving.py tests/test_gpu_chained.py tests/test_gpu_interface.py tests/test_gpu_fullsize.py -m gpu -q -rf > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit|^E  " "$OUT/pytest_gpu.log" | cut -c1-300 | tail -15
LEAN="--no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --prompt 0"
for wl in deepseek-v2-lite mixtral-8x7b; do for b in 2 4 8; do for m in 8 0; do
  MOEINF_SELFROUTE_MULTI=$m timeout 200 python bench.py --workload $wl --batch $b $LEAN > "$OUT/b_${wl}_${b}_$m.json" 2> "$OUT/b_${wl}_${b}_$m.err"
  python - "$OUT/b_${wl}_${b}_$m.json" "$wl batch $b selfroute_multi=$m" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d["ms_per_step"], d["windows_ms"])
except Exception as ex: print(sys.argv[2], "no line", ex)
PY
done; done; done
timeout 200 python bench.py --workload switch-base-8 $LEAN > "$OUT/switch.json" 2> "$OUT/switch.err"; python - "$OUT/switch.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("switch", d["ms_per_step"], d["windows_ms"])
PY
