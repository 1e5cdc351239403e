#!/bin/bash
# round-3 first GPU call: full pytest -m gpu, default bench line, forced-EP world-1 lines (Mixtral, DeepSeek)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3a}
mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 1500 python -m pytest tests -m gpu -q -rf -x > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit|chained" "$OUT/pytest_gpu.log" | tail -20
timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench exit $?"; tail -3 "$OUT/bench_default.err"
for wl in mixtral-8x7b deepseek-v2-lite; do
  timeout 400 python bench.py --workload $wl --force-ep --no-other-configs --miss-heavy-frac 0 --prompt 0 > "$OUT/bench_ep1_$wl.json" 2> "$OUT/bench_ep1_$wl.err"
  echo "ep $wl exit $?"; tail -2 "$OUT/bench_ep1_$wl.err"
done
python - <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r3a/bench_*.json")):
    try:
        d=json.load(open(f))
        print(f, d["ms_per_step"], d.get("parity"), d.get("ep_phases_us_per_layer"))
        for o in d.get("other_configs") or []: print("   ", o.get("workload","")[:40], o.get("ms_per_step"), o.get("parity",{}) and o["parity"].get("ok"), o.get("error"))
    except Exception as e: print(f, "ERR", e)
PY
