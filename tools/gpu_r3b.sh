#!/bin/bash
# full pytest -m gpu + forced-EP world-1 lines with the native transport and with torch.distributed
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3b}
mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 1500 python -m pytest tests -m gpu -q -rf -s > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit|^chained" "$OUT/pytest_gpu.log" | tail -30
for wl in mixtral-8x7b deepseek-v2-lite; do
  for tr in auto torch; do
    timeout 400 python bench.py --workload $wl --force-ep --ep-transport $tr --no-other-configs --miss-heavy-frac 0 --prompt 0 > "$OUT/bench_ep1_${wl}_$tr.json" 2> "$OUT/bench_ep1_${wl}_$tr.err"
    echo "ep $wl $tr exit $?"; grep "transport" "$OUT/bench_ep1_${wl}_$tr.err" | tail -1
  done
done
python - "$OUT" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.load(open(f))
        print(f, d["ms_per_step"], d["windows_ms"], (d.get("parity") or {}).get("ok"), d.get("ep_phases_us_per_layer"))
    except Exception as e: print(f, "ERR", e)
PY
